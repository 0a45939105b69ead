"""ctypes binding of libgb25hip.so (the C ABI in include/gb25.h).

This is the only place the shared library is loaded.  There is no fallback: if the
library is missing or no HIP device is visible, construction raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GB25_LIB") or os.path.join(_HERE, "libgb25hip.so")   # GB25_LIB: A/B builds while tuning
# one library per Oceananigans float type (src/arg_parsing.jl:12-16): same source, same symbols
LIB_PATHS = {"Float32": LIB_PATH, "Float64": os.path.join(_HERE, "libgb25hip_f64.so")}
DTYPES = {"Float32": np.float32, "Float64": np.float64}

FIELD_IDS = {
    "u": 0, "v": 1, "w": 2, "T": 3, "S": 4, "pHY": 5,
    "Gn.u": 6, "Gn.v": 7, "Gn.T": 8, "Gn.S": 9,
    "Gm.u": 10, "Gm.v": 11, "Gm.T": 12, "Gm.S": 13,
    "eta": 14, "U": 15, "V": 16,
    "eta_bar": 17, "U_bar": 18, "V_bar": 19,
    "Gn.U": 20, "Gn.V": 21,
    # closure = CATKEVerticalDiffusivity() only
    "e": 22, "Gn.e": 23, "Gm.e": 24, "kappa_u": 25, "kappa_c": 26, "kappa_e": 27, "Le": 28, "Jb": 29,
    "previous_u": 30, "previous_v": 31,   # diffusivity_fields.previous_velocities (CATKE)
}
METRIC2_IDS = ["dxfc", "dxcc", "dxcf", "dxff", "dyfc", "dycc", "dycf", "dyff", "azcc", "azfc", "azcf", "azff", "fff", "phicc"]
METRIC_IDS = {"phif": 0, "phic": 1, "dxc": 2, "dxf": 3, "azc": 4, "azf": 5, "fcor": 6,
              "zf": 7, "zc": 8, "dzc": 9, "dzf": 10}
ATMOSPHERE_IDS = {"u": 0, "v": 1, "T": 2, "q": 3, "p": 4, "shortwave": 5, "longwave": 6}
KERNEL_IDS = {"fill_halos": 0, "compute_w": 1, "compute_p": 2, "gu": 3, "gv": 4, "tracers": 5,
              "ab2_velocities": 6, "ab2_tracers": 7, "barotropic": 8, "corrector": 9, "implicit": 10, "closure": 11,
              "fluxes": 12}

# every symbol include/gb25.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "gb25_default_config", "gb25_create", "gb25_destroy", "gb25_last_error_string", "gb25_version",
    "gb25_real_bytes", "gb25_config_bytes", "gb25_catke_parameters_bytes",
    "gb25_set_stream", "gb25_use_own_stream", "gb25_synchronize", "gb25_field_dims", "gb25_set_field", "gb25_get_field",
    "gb25_field_device_ptr", "gb25_get_metric", "gb25_get_metric2", "gb25_get_substepping", "gb25_set_vertical_diffusivity",
    "gb25_get_vertical_diffusivity", "gb25_set_closure_catke", "gb25_set_prescribed_atmosphere",
    "gb25_compute_atmosphere_ocean_fluxes", "gb25_get_top_flux", "gb25_default_catke_parameters",
    "gb25_set_catke_parameters", "gb25_get_catke_parameters", "gb25_set_bottom_drag", "gb25_get_bottom_drag", "gb25_set_tracer_advection_order",
    "gb25_get_tracer_advection_order",
    "gb25_set_baroclinic_instability",
    "gb25_get_clock", "gb25_set_dt", "gb25_initialize", "gb25_mask_immersed_fields",
    "gb25_fill_halo_regions", "gb25_compute_auxiliaries", "gb25_fill_diffusivity_halos",
    "gb25_compute_momentum_tendencies", "gb25_compute_tracer_tendencies", "gb25_compute_boundary_tendencies",
    "gb25_compute_tendencies", "gb25_ab2_step", "gb25_correct_velocities_and_cache_previous_tendencies",
    "gb25_update_state", "gb25_first_time_step", "gb25_time_step", "gb25_loop",
    "gb25_set_option", "gb25_get_option", "gb25_set_bottom_height", "gb25_set_curvilinear_grid", "gb25_set_vertical_faces", "gb25_get_bottom_info", "gb25_set_top_flux",
    "gb25_comm_unique_id", "gb25_comm_init_rccl", "gb25_comm_init_local", "gb25_comm_init_callback", "gb25_comm_finalize",
    "gb25_comm_info", "gb25_debug_exchange_plan",
    "gb25_lookahead_state", "gb25_debug_sequence", "gb25_save_state",
    "gb25_profile_enable", "gb25_profile_reset", "gb25_profile_get",
]
# gb25_option (include/gb25.h)
OPTION_IDS = {"kernels": 0, "ab2_lookahead": 1, "subcycle_lookahead": 2, "subcycle_block": 3, "fill_fused": 4,
              "two_streams": 5, "store_pressure": 6, "split_tendencies": 7, "pressure_precision": 8, "immersed_kernels": 9, "fold_fills": 10,
              "lazy_corrector": 11, "momentum_chunk_levels": 12, "tracer_chunk_levels": 13, "tracers_first": 14, "w_on_the_fly": 15, "sub_stream_priority": 16, "subcycle_whole": 17, "early_strips": 18,
              "catke_stale_e_halos": 19, "comm_timeout_seconds": 20, "roctx_ranges": 21, "substep_order": 22, "fold_pivot_slaved": 23}
UNIQUE_ID_BYTES = 128
# int32 fn(void *user, int32 buffer_set, const void *send_w, const void *send_e, void *recv_w, void *recv_e, int64 nbytes)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)


class CatkeParameters(C.Structure):
    """gb25_catke_parameters (include/gb25.h); the arrays are psi = u, c, e, D."""
    _fields_ = [("Cs", C.c_double), ("Cb", C.c_double), ("Csp", C.c_double), ("CRid", C.c_double), ("CRi0", C.c_double),
                ("Chi", C.c_double * 4), ("Clo", C.c_double * 4), ("Cun", C.c_double * 4), ("Cc", C.c_double * 4),
                ("Ce", C.c_double * 4), ("CWu", C.c_double), ("CWw", C.c_double), ("minimum_tke", C.c_double),
                ("minimum_convective_buoyancy_flux", C.c_double), ("negative_tke_damping_time_scale", C.c_double),
                ("CWeps", C.c_double)]

    def as_list(self):
        out = [self.Cs, self.Cb, self.Csp, self.CRid, self.CRi0]
        for a in (self.Chi, self.Clo, self.Cun, self.Cc, self.Ce):
            out += list(a)
        return out + [self.CWu, self.CWw, self.minimum_tke, self.minimum_convective_buoyancy_flux, self.negative_tke_damping_time_scale,
                      self.CWeps]


class Config(C.Structure):
    """gb25_config (include/gb25.h)."""
    _fields_ = [
        ("Nx", C.c_int32), ("Ny", C.c_int32), ("Nz", C.c_int32), ("halo", C.c_int32),
        ("substeps", C.c_int32), ("rank", C.c_int32), ("nranks", C.c_int32), ("device", C.c_int32),
        ("dt", C.c_double), ("chi", C.c_double),
        ("lat_south", C.c_double), ("lat_north", C.c_double), ("lon_west", C.c_double), ("lon_east", C.c_double),
        ("depth", C.c_double), ("zexp_h", C.c_double),
        ("g", C.c_double), ("Omega", C.c_double), ("radius", C.c_double), ("rho0", C.c_double),
        ("slab_mode", C.c_int32), ("grid_type", C.c_int32), ("ranks_y", C.c_int32),
    ]


class GB25Error(RuntimeError):
    pass


_libs = {}
_stale_loaded = set()   # float types whose library was loaded although its sources are newer


def library_stale(float_type="Float32"):
    """True when load_library fell back to a binary older than its sources (no compiler on the host, or GB25_ALLOW_STALE=1)."""
    return float_type in _stale_loaded


def load_library(float_type="Float32"):
    """Load libgb25hip.so (or its Float64 build); raises if it has not been built (no fallback path exists)."""
    if float_type not in LIB_PATHS:
        raise GB25Error(f"float type must be one of {sorted(LIB_PATHS)}, got {float_type!r}")
    if float_type in _libs:
        return _libs[float_type]
    path = LIB_PATHS[float_type]
    if not os.environ.get("GB25_LIB"):
        # a fresh checkout, or sources newer than the binary: compile (hipcc cross-compiles gfx950 anywhere).  The build
        # writes a temporary file and renames it under a lock, so concurrent ranks never load a half-written library.
        from .build import build_library, _stale
        if _stale(path):
            try:
                print(f"gb25_amd: building {os.path.basename(path)} with hipcc ...", flush=True)
                build_library(float_types=(float_type,))
            except Exception as e:
                if not os.path.exists(path):
                    raise GB25Error(
                        f"{path} not found and building it failed ({e}): run `python -c 'import __graft_entry__ as g; "
                        "g.build()'` (hipcc --offload-arch=gfx950).  gb25_amd has no CPU fallback.") from e
                # A loadable library exists but is older than its sources.  Loading it would let tests and bench.py report
                # numbers of a build that is not HEAD, so that is an error -- unless there is no compiler at all on this
                # host (FileNotFoundError: a box that only runs what was built elsewhere) or the caller opts in with
                # GB25_ALLOW_STALE=1.  Either way the fact is recorded (library_stale(), bench.py's "library_stale").
                no_compiler = isinstance(e, FileNotFoundError)
                if not (no_compiler or os.environ.get("GB25_ALLOW_STALE") == "1"):
                    raise GB25Error(
                        f"{path} is older than its sources and rebuilding it failed ({e}).  Fix the build, or set "
                        "GB25_ALLOW_STALE=1 (or GB25_LIB=1 to skip the check) to load the existing binary.") from e
                import warnings
                warnings.warn(f"gb25_amd: {path} is older than its sources and rebuilding it failed ({e}); "
                              "loading the existing binary", RuntimeWarning)
                _stale_loaded.add(float_type)
    if not os.path.exists(path):
        raise GB25Error(f"{path} not found.  gb25_amd has no CPU fallback.")
    # One HIP runtime per process.  PyTorch ships its own copies of libamdhip64 / libhsa-runtime64 / librccl; were this
    # library loaded first it would bind /opt/rocm's copies, a later `import torch` (gb25_amd.distributed, bench.py) would
    # bring a second runtime into the process, and the RCCL the exchanges find by soname -- torch's -- would sit on the
    # runtime that does not own the device (ncclCommInitRank: "no ROCm-capable device is detected").  With torch imported
    # first every HIP user of the process shares torch's copies.  (A host without PyTorch -- the Julia binding -- gets
    # /opt/rocm's throughout.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    P = C.c_void_p
    lib.gb25_version.restype = C.c_char_p
    lib.gb25_last_error_string.restype = C.c_char_p
    lib.gb25_last_error_string.argtypes = [P]
    lib.gb25_default_config.argtypes = [C.POINTER(Config), C.c_int32, C.c_int32, C.c_int32]
    lib.gb25_default_config.restype = None
    lib.gb25_create.argtypes = [C.POINTER(Config), C.POINTER(P)]
    lib.gb25_destroy.argtypes = [P]
    lib.gb25_destroy.restype = None
    lib.gb25_set_stream.argtypes = [P, P]
    lib.gb25_field_dims.argtypes = [P, C.c_int, C.c_int, C.POINTER(C.c_int32)]
    lib.gb25_set_field.argtypes = [P, C.c_int, P, C.c_int]
    lib.gb25_get_field.argtypes = [P, C.c_int, P, C.c_int]
    lib.gb25_field_device_ptr.argtypes = [P, C.c_int, C.POINTER(P)]
    lib.gb25_get_metric.argtypes = [P, C.c_int, C.c_int32, C.POINTER(C.c_double)]
    lib.gb25_get_metric2.argtypes = [P, C.c_int, C.POINTER(C.c_double), C.c_int64]
    lib.gb25_set_vertical_diffusivity.argtypes = [P, C.c_double, C.c_double]
    lib.gb25_set_closure_catke.argtypes = [P, C.c_int32]
    lib.gb25_set_tracer_advection_order.argtypes = [P, C.c_int32]
    lib.gb25_get_tracer_advection_order.argtypes = [P, C.POINTER(C.c_int32)]
    lib.gb25_set_bottom_drag.argtypes = [P, C.c_double]
    lib.gb25_get_bottom_drag.argtypes = [P, C.POINTER(C.c_double)]
    lib.gb25_default_catke_parameters.argtypes = [C.POINTER(CatkeParameters)]
    lib.gb25_default_catke_parameters.restype = None
    lib.gb25_set_catke_parameters.argtypes = [P, C.POINTER(CatkeParameters)]
    lib.gb25_get_catke_parameters.argtypes = [P, C.POINTER(CatkeParameters)]
    lib.gb25_set_prescribed_atmosphere.argtypes = [P, C.c_int, C.c_void_p]
    lib.gb25_get_vertical_diffusivity.argtypes = [P, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.gb25_get_substepping.argtypes = [P, C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.gb25_get_clock.argtypes = [P, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    lib.gb25_set_dt.argtypes = [P, C.c_double]
    lib.gb25_ab2_step.argtypes = [P, C.c_double, C.c_int]
    lib.gb25_correct_velocities_and_cache_previous_tendencies.argtypes = [P, C.c_double]
    lib.gb25_loop.argtypes = [P, C.c_int32]
    lib.gb25_save_state.argtypes = [P, C.c_char_p, C.c_char_p]
    lib.gb25_set_top_flux.argtypes = [P, C.c_int, P]
    lib.gb25_set_bottom_height.argtypes = [P, P]
    lib.gb25_set_curvilinear_grid.argtypes = [P, P, C.c_int32, C.c_int32]
    lib.gb25_set_vertical_faces.argtypes = [P, C.POINTER(C.c_double), C.c_int32]
    lib.gb25_get_bottom_info.argtypes = [P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double)]
    lib.gb25_set_option.argtypes = [P, C.c_int, C.c_int32]
    lib.gb25_get_option.argtypes = [P, C.c_int, C.POINTER(C.c_int32)]
    lib.gb25_comm_unique_id.argtypes = [P]
    lib.gb25_comm_info.argtypes = [P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.gb25_debug_exchange_plan.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_int64]
    lib.gb25_debug_exchange_plan.restype = C.c_int64
    lib.gb25_comm_init_rccl.argtypes = [P, P]
    lib.gb25_comm_init_local.argtypes = [C.POINTER(P), C.c_int32]
    lib.gb25_comm_init_callback.argtypes = [P, EXCHANGE_FN, P]
    lib.gb25_comm_finalize.argtypes = [P]
    lib.gb25_debug_sequence.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_int64]
    lib.gb25_debug_sequence.restype = C.c_int64
    lib.gb25_lookahead_state.argtypes = [P, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.gb25_profile_enable.argtypes = [P, C.c_int]
    lib.gb25_profile_get.argtypes = [P, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double)]
    for name in ["gb25_use_own_stream", "gb25_synchronize", "gb25_set_baroclinic_instability", "gb25_initialize",
                 "gb25_mask_immersed_fields", "gb25_fill_halo_regions", "gb25_compute_auxiliaries",
                 "gb25_fill_diffusivity_halos", "gb25_compute_momentum_tendencies",
                 "gb25_compute_tracer_tendencies", "gb25_compute_boundary_tendencies",
                 "gb25_compute_tendencies", "gb25_update_state", "gb25_first_time_step", "gb25_time_step",
                 "gb25_profile_reset"]:
        getattr(lib, name).argtypes = [P]
    lib.gb25_real_bytes.restype = C.c_int32
    if lib.gb25_real_bytes() != np.dtype(DTYPES[float_type]).itemsize:
        raise GB25Error(f"{path} holds {lib.gb25_real_bytes()}-byte elements, expected {float_type}")
    lib.gb25_config_bytes.restype = lib.gb25_catke_parameters_bytes.restype = C.c_int32
    if lib.gb25_config_bytes() != C.sizeof(Config) or lib.gb25_catke_parameters_bytes() != C.sizeof(CatkeParameters):
        raise GB25Error(f"{path}: gb25_config is {lib.gb25_config_bytes()} bytes there and {C.sizeof(Config)} in this binding "
                        f"(gb25_catke_parameters: {lib.gb25_catke_parameters_bytes()} / {C.sizeof(CatkeParameters)}): "
                        "the library and gb25_amd/binding.py are of different versions")
    _libs[float_type] = lib
    return lib


class HipBackend:
    """One gb25_model handle.  Method names follow the phase list of
    GB-25 src/precompile.jl:31-42 and the entry points of src/timestepping_utils.jl:21-45."""

    def __init__(self, Nx, Ny, Nz, *, dt, halo=8, substeps=30, device=0, rank=0, nranks=1, float_type="Float32",
                 options=None, **overrides):
        float_type = getattr(float_type, "__name__", float_type)   # accepts "Float64", np.float64, ...
        float_type = {"float32": "Float32", "float64": "Float64"}.get(float_type, float_type)
        self.lib = load_library(float_type)
        self.float_type = float_type
        self.dtype = DTYPES[float_type]
        cfg = Config()
        self.lib.gb25_default_config(C.byref(cfg), Nx, Ny, Nz)
        cfg.halo, cfg.substeps, cfg.dt, cfg.device, cfg.rank, cfg.nranks = halo, substeps, dt, device, rank, nranks
        for k, v in overrides.items():
            if not hasattr(cfg, k):
                raise TypeError(f"unknown configuration field {k!r}")
            setattr(cfg, k, v)
        self.cfg = cfg
        self.h = C.c_void_p()
        st = self.lib.gb25_create(C.byref(cfg), C.byref(self.h))
        if st != 0:
            msg = self.lib.gb25_last_error_string(self.h).decode() if self.h else "gb25_create failed"
            if self.h:
                self.lib.gb25_destroy(self.h)
                self.h = None
            raise GB25Error(f"gb25_create: status {st}: {msg}")
        # Partition(Rx, Ry, 1): rank = ry Rx + rx owns Nx / Rx columns and Ny / Ry rows
        self.Ry = max(1, int(cfg.ranks_y))
        self.Rx = cfg.nranks // self.Ry
        self.rx, self.ry = cfg.rank % self.Rx, cfg.rank // self.Rx
        self.Nx_local, self.Ny_local = cfg.Nx // self.Rx, cfg.Ny // self.Ry
        self._keep = []            # ctypes callbacks handed to the library
        for k, v in (options or {}).items():
            self.set_option(k, v)

    def close(self):
        if getattr(self, "h", None):
            self.lib.gb25_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st, what):
        if st != 0:
            raise GB25Error(f"{what}: status {st}: {self.lib.gb25_last_error_string(self.h).decode()}")

    def _call(self, name, *args):
        self._chk(getattr(self.lib, name)(self.h, *args), name)

    # ---- fields
    def field_dims(self, name, include_halos=True):
        d = (C.c_int32 * 3)()
        self._call("gb25_field_dims", FIELD_IDS[name], int(include_halos), d)
        return tuple(d)

    def get_field(self, name, include_halos=True):
        """numpy array shaped like parent(field) / interior(field), index order [i, j, k]."""
        d = self.field_dims(name, include_halos)
        out = np.empty(d[::-1], dtype=self.dtype)  # memory order is i fastest
        self._call("gb25_get_field", FIELD_IDS[name], out.ctypes.data_as(C.c_void_p), int(include_halos))
        return out.transpose(2, 1, 0)

    def set_field(self, name, array, include_halos=True):
        d = self.field_dims(name, include_halos)
        a = np.asarray(array, dtype=self.dtype)
        if a.ndim == 2:
            a = a[:, :, None]
        if a.shape != d:
            raise ValueError(f"{name}: expected shape {d}, got {a.shape}")
        buf = np.ascontiguousarray(a.transpose(2, 1, 0))
        self._call("gb25_set_field", FIELD_IDS[name], buf.ctypes.data_as(C.c_void_p), int(include_halos))

    def field_device_ptr(self, name):
        p = C.c_void_p()
        self._call("gb25_field_device_ptr", FIELD_IDS[name], C.byref(p))
        return p.value

    def metric(self, name, index):
        v = C.c_double()
        self._call("gb25_get_metric", METRIC_IDS[name], index, C.byref(v))
        return v.value

    def set_catke(self, on=True):
        self._call("gb25_set_closure_catke", int(on))

    def catke_parameters(self):
        p = CatkeParameters()
        self._call("gb25_get_catke_parameters", C.byref(p))
        return p

    def set_catke_parameters(self, **changes):
        """Changes some of CATKE's parameters (names of gb25_catke_parameters; arrays as 4-sequences psi = u, c, e, D)."""
        p = self.catke_parameters()
        for k, v in changes.items():
            if not hasattr(p, k):
                raise TypeError(f"unknown CATKE parameter {k!r}")
            if isinstance(getattr(p, k), float):
                setattr(p, k, float(v))
            else:
                for q in range(4):
                    getattr(p, k)[q] = float(v[q])
        self._call("gb25_set_catke_parameters", C.byref(p))

    def set_vertical_diffusivity(self, nu, kappa):
        self._call("gb25_set_vertical_diffusivity", float(nu), float(kappa))

    def set_tracer_advection_order(self, order):
        """tracer_advection = WENO(order = 5) (the default) or WENO(order = 7) (ClimaOcean's ocean_simulation)."""
        self._call("gb25_set_tracer_advection_order", int(order))

    def tracer_advection_order(self):
        v = C.c_int32()
        self._call("gb25_get_tracer_advection_order", C.byref(v))
        return v.value

    def set_bottom_drag(self, Cd):
        """Quadratic bottom drag coefficient (ClimaOcean's ocean_simulation: 0.003); 0: none."""
        self._call("gb25_set_bottom_drag", float(Cd))

    def bottom_drag(self):
        v = C.c_double()
        self._call("gb25_get_bottom_drag", C.byref(v))
        return v.value

    def set_prescribed_atmosphere(self, name, values):
        """One field of the PrescribedAtmosphere at the cell centres, halo cells included: (Nx + 2H, Ny + 2H) [i, j]; None
        clears it.  name: u | v | T | q | p | shortwave | longwave."""
        f = ATMOSPHERE_IDS[name]
        if values is None:
            self._call("gb25_set_prescribed_atmosphere", f, None)
            return
        H = self.cfg.halo
        a = np.ascontiguousarray(np.asarray(values, np.float64).reshape(self.Nx_local + 2 * H, self.Ny_local + 2 * H).T)
        self._call("gb25_set_prescribed_atmosphere", f, a.ctypes.data_as(C.c_void_p))

    def compute_atmosphere_ocean_fluxes(self):
        self._call("gb25_compute_atmosphere_ocean_fluxes")

    def top_flux(self, name):
        """The top flux boundary condition of u | v | T | S as the device holds it (interior points of the field)."""
        d = self.field_dims(name, False)
        a = np.empty((d[1], d[0]), self.dtype)
        self._call("gb25_get_top_flux", FIELD_IDS[name], a.ctypes.data_as(C.c_void_p))
        return a.T

    def vertical_diffusivity(self):
        nu, kappa = C.c_double(), C.c_double()
        self._call("gb25_get_vertical_diffusivity", C.byref(nu), C.byref(kappa))
        return nu.value, kappa.value

    def metric2(self, name):
        """One horizontal metric of a curvilinear grid (grid_type >= 2): (Nx + 2H, Ny + 2H + 1) float64, [i, j]."""
        H = self.cfg.halo
        shape = (self.Ny_local + 2 * H + 1, self.Nx_local + 2 * H)
        out = np.empty(shape, np.float64)
        self._call("gb25_get_metric2", METRIC2_IDS.index(name), out.ctypes.data_as(C.POINTER(C.c_double)), out.size)
        return out.T

    def substepping(self):
        n, frac = C.c_int32(), C.c_double()
        w = (C.c_double * 4096)()
        self._call("gb25_get_substepping", C.byref(n), C.byref(frac), w)
        return n.value, frac.value, np.array(w[: n.value])

    def clock(self):
        t, it, dt = C.c_double(), C.c_int64(), C.c_double()
        self._call("gb25_get_clock", C.byref(t), C.byref(it), C.byref(dt))
        return t.value, it.value, dt.value

    def set_dt(self, dt):
        self._call("gb25_set_dt", float(dt))

    def set_stream(self, stream_ptr):
        """stream_ptr: a hipStream_t as an integer (0 / None = HIP's default stream)."""
        self._call("gb25_set_stream", C.c_void_p(stream_ptr or None))

    def use_own_stream(self):
        self._call("gb25_use_own_stream")

    # ---- phases / composites (one ABI call each)
    def synchronize(self): self._call("gb25_synchronize")
    def set_baroclinic_instability(self): self._call("gb25_set_baroclinic_instability")
    def initialize(self): self._call("gb25_initialize")
    def mask_immersed_fields(self): self._call("gb25_mask_immersed_fields")
    def fill_halo_regions(self): self._call("gb25_fill_halo_regions")
    def compute_auxiliaries(self): self._call("gb25_compute_auxiliaries")
    def fill_diffusivity_halos(self): self._call("gb25_fill_diffusivity_halos")
    def compute_momentum_tendencies(self): self._call("gb25_compute_momentum_tendencies")
    def compute_tracer_tendencies(self): self._call("gb25_compute_tracer_tendencies")
    def compute_boundary_tendencies(self): self._call("gb25_compute_boundary_tendencies")
    def compute_tendencies(self): self._call("gb25_compute_tendencies")
    def ab2_step(self, dt, euler=False): self._call("gb25_ab2_step", float(dt), int(euler))
    def correct_velocities_and_cache_previous_tendencies(self, dt=0.0):
        self._call("gb25_correct_velocities_and_cache_previous_tendencies", float(dt))
    def update_state(self): self._call("gb25_update_state")
    def first_time_step(self): self._call("gb25_first_time_step")
    def time_step(self): self._call("gb25_time_step")
    def loop(self, n): self._call("gb25_loop", int(n))

    def save_state(self, directory, label="checkpoint"):
        """save_model_state: this rank's slab -> <directory>/<label>/fields_rank<R>.npz; returns the path."""
        self._call("gb25_save_state", str(directory).encode(), str(label).encode())
        return os.path.join(str(directory), label, f"fields_rank{self.cfg.rank}.npz")

    def set_top_flux(self, name, J):
        """FluxBoundaryCondition at the top of u | v | T | S: interior-shaped array (None: back to no-flux)."""
        if J is None:
            self._call("gb25_set_top_flux", FIELD_IDS[name], None)
            return
        d = self.field_dims(name, False)
        a = np.ascontiguousarray(np.asarray(J, dtype=self.dtype).reshape(d[0], d[1]).T)
        self._call("gb25_set_top_flux", FIELD_IDS[name], a.ctypes.data_as(C.c_void_p))

    # ---- immersed boundary
    def set_bottom_height(self, zb):
        """GridFittedBottom(zb): bottom height at the GLOBAL cell centres, shape (Nx_global, Ny) (a slab takes its columns,
        its neighbours' and its fold partner's from it: every rank passes the same array)."""
        a = np.ascontiguousarray(np.asarray(zb, dtype=np.float64).T)      # i fastest
        if a.shape != (self.cfg.Ny, self.cfg.Nx):
            raise ValueError(f"bottom height: expected shape ({self.cfg.Nx}, {self.cfg.Ny})")
        self._call("gb25_set_bottom_height", a.ctypes.data_as(C.c_void_p))

    # ---- the host's grid
    def set_curvilinear_grid(self, metrics):
        """metrics: {name: array} for the 14 names of METRIC2_IDS (dxfc ... azff, fff, phicc), each the parent array over the
        GLOBAL grid, shape (Nx_global + 2H, Ny + 2H) or (Nx_global + 2H, Ny + 2H + 1) -- grid.Δxᶠᶜᵃ etc. of the host's grid."""
        H = self.cfg.halo
        arrs = [np.ascontiguousarray(np.asarray(metrics[n], dtype=np.float64).T) for n in METRIC2_IDS]
        ny, nx = arrs[0].shape
        if nx != self.cfg.Nx + 2 * H or any(a.shape != (ny, nx) for a in arrs):
            raise ValueError(f"metrics: {len(METRIC2_IDS)} arrays of shape ({self.cfg.Nx + 2 * H}, {self.cfg.Ny + 2 * H} [+ 1])")
        ptrs = (C.POINTER(C.c_double) * len(arrs))(*[a.ctypes.data_as(C.POINTER(C.c_double)) for a in arrs])
        self._call("gb25_set_curvilinear_grid", ptrs, nx, ny)

    def set_vertical_faces(self, zf):
        """grid.z faces, bottom to top: Nz + 1 values."""
        a = np.ascontiguousarray(np.asarray(zf, dtype=np.float64))
        self._call("gb25_set_vertical_faces", a.ctypes.data_as(C.POINTER(C.c_double)), int(a.size))

    def bottom_info(self, which, i, j):
        """1-based (i, j) like the Julia sources; which: kbot | Hfc | Hcf."""
        v = C.c_double()
        self._call("gb25_get_bottom_info", {"kbot": 0, "Hfc": 1, "Hcf": 2}[which], i - 1, j - 1, C.byref(v))
        return v.value

    # ---- options (gb25_option)
    def set_option(self, name, value):
        self._call("gb25_set_option", OPTION_IDS[name], int(value))

    def get_option(self, name):
        v = C.c_int32()
        self._call("gb25_get_option", OPTION_IDS[name], C.byref(v))
        return v.value

    def lookahead_state(self):
        """(velocity look-ahead of the next step exists, stage 0 of this step adopted the sub-cycle look-ahead)"""
        a, b = C.c_int32(), C.c_int32()
        self._call("gb25_lookahead_state", C.byref(a), C.byref(b))
        return bool(a.value), bool(b.value)

    # ---- exchange context of a slab (x decomposition)
    def comm_unique_id(self):
        """128 bytes from ncclGetUniqueId (rank 0 calls this and hands them to every rank)."""
        buf = C.create_string_buffer(UNIQUE_ID_BYTES)
        st = self.lib.gb25_comm_unique_id(buf)
        if st != 0:
            raise GB25Error(f"gb25_comm_unique_id: status {st} (librccl could not be loaded?)")
        return buf.raw

    def comm_init_rccl(self, unique_id):
        if len(unique_id) != UNIQUE_ID_BYTES:
            raise ValueError("unique id must be 128 bytes")
        self._call("gb25_comm_init_rccl", C.c_char_p(bytes(unique_id)))

    def comm_info(self):
        """(transport, comm_ranks): "none" | "rccl" | "local" | "callback", and the communicator's size as RCCL reports it."""
        t, n = C.c_int32(0), C.c_int32(0)
        self._call("gb25_comm_info", C.byref(t), C.byref(n))
        return ("none", "rccl", "local", "callback")[t.value], n.value

    def comm_init_callback(self, fn):
        """fn(buffer_set, send_west, send_east, recv_west, recv_east, nbytes) with device pointers as integers."""
        def trampoline(user, b, sw, se, rw, re, nbytes):
            try:
                fn(b, sw, se, rw, re, nbytes)
                return 0
            except Exception as e:     # never let an exception cross the C frames
                print(f"gb25_amd: exchange callback failed: {e!r}", flush=True)
                return 1
        cb = EXCHANGE_FN(trampoline)
        self._keep.append(cb)
        self._call("gb25_comm_init_callback", cb, None)

    def comm_finalize(self):
        self._call("gb25_comm_finalize")

    @staticmethod
    def comm_init_local(backends):
        """All slabs of one decomposition in this process: ring of device copies; the composites of any member step all."""
        b0 = backends[0]
        arr = (C.c_void_p * len(backends))(*[b.h for b in backends])
        b0._chk(b0.lib.gb25_comm_init_local(arr, len(backends)), "gb25_comm_init_local")

    # ---- timers
    def profile_enable(self, on=True, only=None):
        """on: time every kernel; only="momentum"|"gu"|...: time that kernel alone (least perturbation)."""
        self._call("gb25_profile_enable", 2 + KERNEL_IDS[only] if only else int(on))
    def profile_reset(self): self._call("gb25_profile_reset")

    def profile_get(self, kernel):
        n, ms = C.c_int64(), C.c_double()
        self._call("gb25_profile_get", KERNEL_IDS[kernel], C.byref(n), C.byref(ms))
        return n.value, ms.value
