"""Test-side adapter: drives the CPU oracle (oracle/gb25_oracle.c) through the same backend
interface the product's HipBackend offers, so parity tests read like the reference's
correctness script (two models, same calls, compare_states).

TEST INFRASTRUCTURE ONLY.  Nothing under gb-25_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

# The oracle is OpenMP code.  On the GPU box os.cpu_count() reports the whole host (256) while the job owns 16 cores:
# an oversubscribed, spinning OpenMP team turns a 6 s test into minutes.  Size the team to the cores we may use,
# BEFORE libgomp is loaded.
try:
    _cores = len(os.sched_getaffinity(0))
except AttributeError:
    _cores = os.cpu_count() or 1
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(_cores, 16))))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

FIELD_IDS = {
    "u": 0, "v": 1, "w": 2, "T": 3, "S": 4, "pHY": 5,
    "Gn.u": 6, "Gn.v": 7, "Gn.T": 8, "Gn.S": 9,
    "Gm.u": 10, "Gm.v": 11, "Gm.T": 12, "Gm.S": 13,
    "eta": 14, "U": 15, "V": 16, "eta_bar": 17, "U_bar": 18, "V_bar": 19, "Gn.U": 20, "Gn.V": 21,
    # closure = CATKEVerticalDiffusivity(): the TKE tracer and the diffusivity fields (src/correctness.jl:60-67)
    "e": 22, "Gn.e": 23, "Gm.e": 24, "kappa_u": 25, "kappa_c": 26, "kappa_e": 27, "Le": 28, "Jb": 29,
    "previous_u": 30, "previous_v": 31,
}
METRIC_IDS = {"phif": 0, "phic": 1, "dxc": 2, "dxf": 3, "azc": 4, "azf": 5, "fcor": 6,
              "zf": 7, "zc": 8, "dzc": 9, "dzf": 10}


class OConfig(C.Structure):
    _fields_ = [("Nx", C.c_int), ("Ny", C.c_int), ("Nz", C.c_int), ("H", C.c_int), ("substeps", C.c_int),
                ("dt", C.c_double), ("chi", C.c_double),
                ("lat_south", C.c_double), ("lat_north", C.c_double), ("lon_west", C.c_double),
                ("lon_east", C.c_double), ("depth", C.c_double), ("zexp_h", C.c_double),
                ("g", C.c_double), ("Omega", C.c_double), ("radius", C.c_double), ("rho0", C.c_double),
                ("grid_type", C.c_int)]


def build_oracle():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR], check=True)


_libs = {}


def oracle_lib(precision):
    if precision not in _libs:
        path = os.path.join(ORACLE_DIR, "_build", f"libgb25_oracle_{precision}.so")
        src = os.path.join(ORACLE_DIR, "gb25_oracle.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            build_oracle()
        _libs[precision] = C.CDLL(path)
    return _libs[precision]


class OracleBackend:
    """Same duck-typed interface as gb25_amd.binding.HipBackend."""

    def __init__(self, Nx, Ny, Nz, *, dt, halo=8, substeps=30, precision="f64", **overrides):
        self.sfx = "_" + precision
        self.lib = oracle_lib(precision)
        # "f32p64": fp32 state, equation of state + hydrostatic integral in fp64 (accuracy study)
        self.dtype = np.float64 if precision == "f64" else np.float32
        self.ctype = C.c_double if precision == "f64" else C.c_float
        cfg = OConfig(Nx, Ny, Nz, halo, substeps, dt, 0.1, -80, 80, 0, 360, 4000, 30, 9.80665, 7.292115e-5, 6371e3,
                      1020.0, 0)
        for k, v in overrides.items():
            setattr(cfg, k, v)
        self.cfg = cfg
        f = self._fn("create")
        f.restype = C.c_void_p
        f.argtypes = [C.POINTER(OConfig)]
        self.h = f(C.byref(cfg))
        if not self.h:
            raise RuntimeError("oracle create failed")
        self._fn("field_ptr").restype = C.POINTER(self.ctype)
        self._fn("field_ptr").argtypes = [C.c_void_p, C.c_int]
        self._fn("field_dims").argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        self._fn("metric").restype = C.c_double
        self._fn("metric").argtypes = [C.c_void_p, C.c_int, C.c_int]
        self._fn("ab2_step").argtypes = [C.c_void_p, C.c_double, C.c_int]
        self._fn("loop").argtypes = [C.c_void_p, C.c_int]
        self._fn("set_dt").argtypes = [C.c_void_p, C.c_double]
        self._fn("get_time").restype = C.c_double
        self._fn("get_iteration").restype = C.c_long
        self._fn("time_step_euler").argtypes = [C.c_void_p, C.c_int]
        self._dt = dt
        self.H = halo

    def _fn(self, name):
        return getattr(self.lib, "gb25o_" + name + self.sfx)

    def _call(self, name, *a):
        f = self._fn(name)
        if f.argtypes is None:
            f.argtypes = [C.c_void_p]
        f(self.h, *a)

    def close(self):
        if getattr(self, "h", None):
            f = self._fn("destroy")
            f.argtypes = [C.c_void_p]
            f(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- fields
    def _view(self, name):
        d = (C.c_int * 4)()
        self._fn("field_dims")(self.h, FIELD_IDS[name], d)
        p = self._fn("field_ptr")(self.h, FIELD_IDS[name])
        a = np.ctypeslib.as_array(p, shape=(d[2], d[3], d[0]))[:, :d[1], :]   # (a y-face field of a folded grid: Ny + 2H of its rows)
        return a.transpose(2, 1, 0)  # [i, j, k] view of the oracle's memory

    def field_dims(self, name, include_halos=True):
        v = self._view(name)
        if include_halos:
            return v.shape
        H = self.H
        return (v.shape[0] - 2 * H, v.shape[1] - 2 * H, 1 if v.shape[2] == 1 else v.shape[2] - 2 * H)

    def _interior(self, v):
        H = self.H
        if v.shape[2] == 1:
            return v[H:-H, H:-H, :]
        return v[H:-H, H:-H, H:-H]

    def get_field(self, name, include_halos=True):
        v = self._view(name)
        return np.array(v if include_halos else self._interior(v))

    def set_field(self, name, array, include_halos=True):
        v = self._view(name)
        a = np.asarray(array)
        if a.ndim == 2:
            a = a[:, :, None]
        tgt = v if include_halos else self._interior(v)
        if a.shape != tgt.shape:
            raise ValueError(f"{name}: expected {tgt.shape}, got {a.shape}")
        tgt[...] = a
        # set!(model, ...) on an immersed grid masks what it has set (flat bottom: nothing to do, as in the product)
        if name in ("u", "v", "T", "S", "U", "V") and self._fn("is_immersed")(C.c_void_p(self.h)):
            self._call("mask_immersed_fields")

    def metric(self, name, index):
        return self._fn("metric")(self.h, METRIC_IDS[name], index)

    def set_catke(self, on=True):
        f = self._fn("set_catke")
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int]
        f(self.h, int(on))

    def set_catke_parameters(self, **changes):
        """Same call as HipBackend.set_catke_parameters: the defaults of the library with `changes` applied."""
        from gb25_amd.binding import CatkeParameters, load_library
        p = getattr(self, "_catke_par", None)
        if p is None:
            p = CatkeParameters()
            load_library("Float32").gb25_default_catke_parameters(C.byref(p))
        for k, v in changes.items():
            if isinstance(getattr(p, k), float):
                setattr(p, k, float(v))
            else:
                for q in range(4):
                    getattr(p, k)[q] = float(v[q])
        self._catke_par = p
        a = (C.c_double * 31)(*p.as_list())
        f = self._fn("set_catke_parameters")
        f.restype = None
        f.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        f(self.h, a)

    def set_option(self, name, value):
        """The restatement choices the library exposes as options (same names); its schedule options mean nothing here."""
        fn = {"catke_stale_e_halos": "set_catke_stale_e_halos", "substep_order": "set_substep_order",
              "fold_pivot_slaved": "set_fold_pivot_slaved"}.get(name)
        if fn is None:
            raise KeyError(f"the oracle has no option {name!r}")
        f = self._fn(fn)
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int]
        f(self.h, int(value))

    def teos10_sensitivities(self, T, S, Z):
        """(-d rho / d Theta, d rho / d S_A) of the oracle's TEOS-10 polynomial."""
        f = self._fn("teos10_sensitivities")
        f.restype = None
        f.argtypes = [C.c_double, C.c_double, C.c_double, C.POINTER(C.c_double)]
        out = (C.c_double * 2)()
        f(float(T), float(S), float(Z), out)
        return out[0], out[1]

    def set_vertical_diffusivity(self, nu, kappa):
        f = self._fn("set_vertical_diffusivity")
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_double, C.c_double]
        f(self.h, float(nu), float(kappa))

    def metric2(self, name, i, j):
        """2-D metric at the 1-based logical (i, j): dxfc dxcc dxcf dxff dyfc dycc dycf dyff azcc azfc azcf azff fff phicc."""
        ids = ["dxfc", "dxcc", "dxcf", "dxff", "dyfc", "dycc", "dycf", "dyff", "azcc", "azfc", "azcf", "azff", "fff", "phicc"]
        f = self._fn("metric2")
        f.restype = C.c_double
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        return f(self.h, ids.index(name), i, j)

    def substepping(self):
        w = (C.c_double * 4096)()
        frac = C.c_double()
        f = self._fn("substep_info")
        f.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        n = f(self.h, C.byref(frac), w)
        return n, frac.value, np.array(w[:n])

    def clock(self):
        return self._fn("get_time")(C.c_void_p(self.h)), self._fn("get_iteration")(C.c_void_p(self.h)), self._dt

    def set_dt(self, dt):
        self._dt = dt
        self._fn("set_dt")(self.h, dt)

    # ---- phases
    def synchronize(self): pass
    def set_baroclinic_instability(self): self._call("set_baroclinic_instability")
    def initialize(self): self._call("initialize")
    def mask_immersed_fields(self): self._call("mask_immersed_fields")

    def set_bottom_height(self, zb):
        """GridFittedBottom(zb): bottom height at the interior cell centres, shape (Nx, Ny)."""
        a = np.ascontiguousarray(np.asarray(zb, dtype=np.float64).T)      # i fastest
        f = self._fn("set_bottom_height")
        f.argtypes = [C.c_void_p, C.c_void_p]
        f(self.h, a.ctypes.data_as(C.c_void_p))
        if self._fn("is_immersed")(C.c_void_p(self.h)):
            self._call("mask_immersed_fields")

    def set_curvilinear_grid(self, metrics):
        """Same call as HipBackend.set_curvilinear_grid (the oracle is a single domain: global = local)."""
        from gb25_amd.binding import METRIC2_IDS
        arrs = [np.ascontiguousarray(np.asarray(metrics[n], dtype=np.float64).T) for n in METRIC2_IDS]
        ptrs = (C.POINTER(C.c_double) * len(arrs))(*[a.ctypes.data_as(C.POINTER(C.c_double)) for a in arrs])
        f = self._fn("set_curvilinear_grid")
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        if f(self.h, ptrs, arrs[0].shape[0]) != 0:
            raise ValueError("set_curvilinear_grid: not a curvilinear grid, or arrays of the wrong shape")

    def set_vertical_faces(self, zf):
        a = np.ascontiguousarray(np.asarray(zf, dtype=np.float64))
        f = self._fn("set_vertical_faces")
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        if f(self.h, a.ctypes.data_as(C.c_void_p), int(a.size)) != 0:
            raise ValueError("set_vertical_faces: Nz + 1 faces")
        if self._fn("is_immersed")(C.c_void_p(self.h)):
            self._call("mask_immersed_fields")

    def metric2_array(self, name):
        """One horizontal metric as the parent array (Nx + 2H, Ny + 2H + 1), like HipBackend.metric2(name)."""
        H = self.H
        return np.array([[self.metric2(name, i, j) for j in range(1 - H, self.cfg.Ny + H + 2)] for i in range(1 - H, self.cfg.Nx + H + 1)])

    def bottom_info(self, which, i, j):
        f = self._fn("bottom_info")
        f.restype = C.c_double
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        return f(self.h, {"kbot": 0, "Hcc": 1, "Hfc": 2, "Hcf": 3}[which], i, j)
    def fill_halo_regions(self): self._call("fill_halos")
    def compute_auxiliaries(self): self._call("compute_auxiliaries")
    def fill_diffusivity_halos(self): pass
    def compute_momentum_tendencies(self): self._call("compute_momentum_tendencies")
    def compute_tracer_tendencies(self): self._call("compute_tracer_tendencies")
    def compute_boundary_tendencies(self): self._call("compute_boundary_tendencies")

    def set_top_flux(self, name, J):
        """FluxBoundaryCondition at the top of u | v | T | S: interior-shaped array (None: back to no-flux)."""
        f = self._fn("set_top_flux")
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        q = {"u": 0, "v": 1, "T": 2, "S": 3}[name]
        if J is None:
            f(self.h, q, None)
            return
        a = np.ascontiguousarray(np.asarray(J, dtype=np.float64).reshape(self.field_dims(name, False)[:2]).T)
        f(self.h, q, a.ctypes.data_as(C.c_void_p))

    def get_top_flux(self, name):
        f = self._fn("get_top_flux")
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        nx, ny = self.field_dims(name, False)[:2]
        a = np.zeros((ny, nx), dtype=np.float64)
        f(self.h, {"u": 0, "v": 1, "T": 2, "S": 3}[name], a.ctypes.data_as(C.c_void_p))
        return np.ascontiguousarray(a.T)

    def set_tracer_advection_order(self, order):
        f = self._fn("set_tracer_advection_order")
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int]
        f(self.h, int(order))

    def set_bottom_drag(self, Cd):
        f = self._fn("set_bottom_drag")
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_double]
        f(self.h, float(Cd))

    def set_prescribed_atmosphere(self, name, values):
        f = self._fn("set_prescribed_atmosphere")
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        q = {"u": 0, "v": 1, "T": 2, "q": 3, "p": 4, "shortwave": 5, "longwave": 6}[name]
        if values is None:
            f(self.h, q, None)
            return
        H = self.H
        a = np.ascontiguousarray(np.asarray(values, np.float64).reshape(self.cfg.Nx + 2 * H, self.cfg.Ny + 2 * H).T)
        f(self.h, q, a.ctypes.data_as(C.c_void_p))

    def compute_atmosphere_ocean_fluxes(self): self._call("compute_atmosphere_ocean_fluxes")

    def top_flux(self, name):
        f = self._fn("get_top_flux")
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        d = self.field_dims(name, False)
        a = np.empty((d[1], d[0]), np.float64)
        f(self.h, {"u": 0, "v": 1, "T": 2, "S": 3}[name], a.ctypes.data_as(C.c_void_p))
        return a.T.astype(self.dtype)

    def compute_tendencies(self): self._call("compute_tendencies")
    def ab2_step(self, dt, euler=False): self._fn("ab2_step")(self.h, float(dt), int(euler))
    def correct_velocities_and_cache_previous_tendencies(self, dt=0.0): self._call("correct_and_cache")
    def update_state(self): self._call("update_state")
    def first_time_step(self): self._call("first_time_step")
    def time_step(self): self._call("time_step")
    def loop(self, n): self._fn("loop")(self.h, int(n))

    # unit functions
    def weno5(self, a, b, c, d, e):
        f = self._fn("weno5")
        f.restype = C.c_double
        f.argtypes = [C.c_double] * 5
        return f(a, b, c, d, e)

    def weno3(self, b, c, d):
        f = self._fn("weno3")
        f.restype = C.c_double
        f.argtypes = [C.c_double] * 3
        return f(b, c, d)

    def teos10_rho(self, T, S, Z):
        f = self._fn("teos10_rho")
        f.restype = C.c_double
        f.argtypes = [C.c_double] * 3
        return f(T, S, Z)


class CPU:
    """Oracle-backed architecture object for tests: baroclinic_instability_model(CPU(), ...)."""

    def __init__(self, precision="f64"):
        self.precision = precision

    def __call__(self, Nx, Ny, Nz, **kw):
        kw.pop("device", None)
        kw.pop("options", None)        # schedule switches of the HIP library: meaningless for the oracle
        return OracleBackend(Nx, Ny, Nz, precision=self.precision, **kw)
