"""Host-side mirror of the GordonBell25 model API for the HydrostaticFreeSurfaceModel hot path.

Mirrors (names, argument meaning, mutation-in-place semantics):
  baroclinic_instability_model(arch, Nx, Ny, Nz; dt, halo, ...)  GB-25 src/baroclinic_instability_model.jl:17-85
  first_time_step!(model) / time_step!(model) / loop!(model, Ninner)  src/timestepping_utils.jl:21-45
  set_baroclinic_instability!(model)                              src/model_utils.jl:120-127
  the *_workload! phase wrappers                                   src/precompile.jl:44-127
Julia's `!` suffix is dropped.  The model object exposes Oceananigans-shaped accessors
(model.velocities.u, model.tracers.T, model.free_surface.eta, model.timestepper.Gn.u,
model.clock, model.grid) whose `parent` / `interior` arrays have the Oceananigans layout.

The numerical engine is injected as a *backend* object (binding.HipBackend in the product).
"""
from types import SimpleNamespace

import numpy as np


class Field:
    """View of one model field: `parent` includes halos (Oceananigans `parent(field)`),
    `interior` does not.  Index order is [i, j, k]; arrays are copies fetched from the device."""

    def __init__(self, backend, name, location):
        self._b, self.name, self.location = backend, name, location

    @property
    def parent(self):
        return self._b.get_field(self.name, True)

    @property
    def interior(self):
        return self._b.get_field(self.name, False)

    def set(self, array, include_halos=False):
        a = np.asarray(array)
        if not include_halos and a.ndim >= 2:
            # A y-face field of a folded (tripolar) grid has Ny rows -- topology (Periodic, RightConnected, Bounded): the
            # faces beyond the last row of cells are halo cells.  An array shaped for a Bounded y (Ny + 1 rows) is
            # accepted and its last row, which the fold fill would overwrite anyway, is dropped.
            # Only there: a y-face field (v and its tendencies, V, G.V, ...) whose grid has no wall at the northern edge -- the
            # zipper fold, or a northern neighbour rank of a 2-D decomposition.  Any other array with a row too many is a
            # wrongly shaped array and is passed through, so that the backend rejects it.
            want = self._b.field_dims(self.name, False)
            y_face = self.name in ("v", "Gn.v", "Gm.v", "V", "V_bar", "Gn.V", "previous_v")
            no_wall = y_face and want[1] == self._b.field_dims("T", False)[1]      # (a Bounded y gives a y-face field one row more than a cell field)
            if no_wall and a.shape[0] == want[0] and a.shape[1] == want[1] + 1:
                a = np.ascontiguousarray(a[:, :want[1]])
        self._b.set_field(self.name, a, include_halos)

    def set_parent(self, array):
        self._b.set_field(self.name, array, True)

    @property
    def shape(self):
        return self._b.field_dims(self.name, False)

    def __repr__(self):
        return f"Field({self.name!r} at {self.location}, interior {self.shape})"


class Clock:
    """model.clock (time, last_dt, iteration) -- src/model_utils.jl:150-155."""

    def __init__(self, backend):
        self._b = backend

    @property
    def time(self):
        return self._b.clock()[0]

    @property
    def iteration(self):
        return self._b.clock()[1]

    @property
    def last_dt(self):
        return self._b.clock()[2]

    @last_dt.setter
    def last_dt(self, dt):
        self._b.set_dt(dt)


class LatitudeLongitudeGrid:
    """simple_latitude_longitude_grid(arch, Nx, Ny, Nz; halo) -- src/model_utils.jl:56-65."""

    def __init__(self, backend, Nx, Ny, Nz, halo):
        self._b = backend
        self.Nx, self.Ny, self.Nz = Nx, Ny, Nz
        self.halo = (halo, halo, halo)
        self.latitude, self.longitude = (-80, 80), (0, 360)

    @property
    def size(self):
        return (self.Nx, self.Ny, self.Nz)

    def metric(self, name, index):
        """1-based index like the Julia sources; names: phif phic dxc dxf azc azf fcor zf zc dzc dzf."""
        return self._b.metric(name, index)

    def z_faces(self):
        return np.array([self.metric("zf", k) for k in range(1, self.Nz + 2)])


class HydrostaticFreeSurfaceModel:
    """The object baroclinic_instability_model returns."""

    def __init__(self, backend, Nx, Ny, Nz, halo):
        b = self.backend = backend
        self.grid = LatitudeLongitudeGrid(b, Nx, Ny, Nz, halo)
        self.clock = Clock(b)
        F = lambda n, loc: Field(b, n, loc)
        self.velocities = SimpleNamespace(u=F("u", "fcc"), v=F("v", "cfc"), w=F("w", "ccf"))
        self.tracers = SimpleNamespace(T=F("T", "ccc"), S=F("S", "ccc"))
        self.pressure = SimpleNamespace(pHY=F("pHY", "ccc"))
        self.free_surface = SimpleNamespace(
            eta=F("eta", "ccf"),
            barotropic_velocities=SimpleNamespace(U=F("U", "fc"), V=F("V", "cf")),
            filtered_state=SimpleNamespace(eta=F("eta_bar", "ccf"), U=F("U_bar", "fc"), V=F("V_bar", "cf")),
            substeps=30, gravitational_acceleration=9.80665)
        G = lambda p: SimpleNamespace(u=F(p + ".u", "fcc"), v=F(p + ".v", "cfc"), T=F(p + ".T", "ccc"),
                                      S=F(p + ".S", "ccc"))
        Gn = G("Gn")
        Gn.U, Gn.V = F("Gn.U", "fc"), F("Gn.V", "cf")
        self.timestepper = SimpleNamespace(Gn=Gn, Gm=G("Gm"), chi=0.1)

        self.closure = None
        self.diffusivity_fields = None

    def enable_catke_fields(self):
        """closure = CATKEVerticalDiffusivity(): tracers = (:T, :S, :e) and model.diffusivity_fields
        (src/baroclinic_instability_model.jl:50-51, src/correctness.jl:60-67)."""
        b = self.backend
        F = lambda n, loc: Field(b, n, loc)
        self.tracers.e = F("e", "ccc")
        self.timestepper.Gn.e, self.timestepper.Gm.e = F("Gn.e", "ccc"), F("Gm.e", "ccc")
        self.diffusivity_fields = SimpleNamespace(kappa_u=F("kappa_u", "ccf"), kappa_c=F("kappa_c", "ccf"),
                                                  kappa_e=F("kappa_e", "ccf"), Le=F("Le", "ccc"), Jb=F("Jb", "cc"))

    # Oceananigans.fields(model): the set compare_states walks (src/correctness.jl:34-35)
    def fields(self):
        out = {"u": self.velocities.u, "v": self.velocities.v, "w": self.velocities.w,
               "eta": self.free_surface.eta, "T": self.tracers.T, "S": self.tracers.S}
        if hasattr(self.tracers, "e"):
            out["e"] = self.tracers.e
        return out

    def prognostic_fields(self):
        fs = self.free_surface
        return {"u": self.velocities.u, "v": self.velocities.v, "eta": fs.eta,
                "U": fs.barotropic_velocities.U, "V": fs.barotropic_velocities.V,
                "T": self.tracers.T, "S": self.tracers.S}

    def set(self, **kw):
        """set!(model, u=..., v=..., T=..., S=..., eta=...) with interior-shaped arrays."""
        allf = {**self.fields()}
        for k, v in kw.items():
            allf[k].set(v)

    def synchronize(self):
        self.backend.synchronize()

    def __repr__(self):
        Nx, Ny, Nz = self.grid.size
        return (f"HydrostaticFreeSurfaceModel({Nx}x{Ny}x{Nz} LatitudeLongitudeGrid, halo {self.grid.halo}, "
                f"{np.dtype(getattr(self.backend, 'dtype', np.float32)).name}, MI355X)")


def resolution_to_points(resolution):
    """src/model_utils.jl:45-49."""
    Nx, Ny = 384 / resolution, 192 / resolution
    if Nx != int(Nx) or Ny != int(Ny):
        raise ValueError("resolution must divide 384 and 192")
    return int(Nx), int(Ny)


class VerticalScalarDiffusivity:
    """closure = VerticalScalarDiffusivity(VerticallyImplicitTimeDiscretization(), κ=1e-5, ν=1e-4), the alternative the
    reference keeps next to `closure = nothing` (src/baroclinic_instability_model.jl:29-31): constant vertical viscosity
    ν and diffusivity κ, stepped implicitly (one tridiagonal solve per column and field after the AB2 update)."""

    def __init__(self, nu=1e-4, kappa=1e-5):
        self.nu, self.kappa = float(nu), float(kappa)


class CATKEVerticalDiffusivity:
    """closure = Oceananigans.TurbulenceClosures.CATKEVerticalDiffusivity() (src/baroclinic_instability_model.jl:30,
    sharding/less_simple_sharding_problem.jl:84-93): a third tracer e (turbulent kinetic energy), diffusivity fields
    κu, κc, κe, Lᵉ, Jᵇ (compared by src/correctness.jl:60-67), vertically implicit mixing of u, v, T, S, e.
    Keyword arguments change parameters (names of gb25_catke_parameters in include/gb25.h), e.g. Cb=0.01."""

    def __init__(self, **parameters):
        self.parameters = parameters


def default_ocean_closure():
    """ClimaOcean.OceanSimulations.default_ocean_closure() -- what ocean_simulation(grid; ...) uses when no closure is given
    (src/data_free_ocean_climate_model.jl:26): CATKEVerticalDiffusivity with CATKEMixingLength(Cᵇ = 0.01)
    [UPSTREAM-UNVERIFIED: recalled from ClimaOcean, which is not in /root/reference]."""
    return CATKEVerticalDiffusivity(Cb=0.01)


def baroclinic_instability_model(arch, Nx=None, Ny=None, Nz=None, *, dt, halo=(8, 8, 8), grid_type="simple_lat_lon",
                                 substeps=30, resolution=None, closure=None, **backend_kw):
    """baroclinic_instability_model(arch, Nx, Ny, Nz; dt, halo, grid_type, free_surface=SplitExplicit(substeps))
    -- src/baroclinic_instability_model.jl:12-85.  `arch` is a backend factory: GPU() from this package
    (or, in tests only, an oracle-backed factory).  Physics is fixed to the reference defaults:
    TEOS-10 SeawaterBuoyancy, HydrostaticSphericalCoriolis, WENOVectorInvariant(order=5),
    WENO(order=5), closure=nothing.  The initial state is all zeros, as in the reference
    (set_baroclinic_instability! is commented out at :74-80)."""
    if resolution is not None:
        Nx, Ny = resolution_to_points(resolution)
    # grid_type (src/baroclinic_instability_model.jl:19,59-65): :simple_lat_lon | :gaussian_islands, where
    # :gaussian_islands = ImmersedBoundaryGrid(TripolarGrid, GridFittedBottom(gaussian_islands)) (src/model_utils.jl:
    # 134-146).  Also: the same mountains on the lat-lon grid, the bare tripolar grid, and (tests) the lat-lon metrics
    # sent through the curvilinear code path.
    grid_types = {"simple_lat_lon": 0, "gaussian_islands_lat_lon": 1, "lat_lon_as_curvilinear": 2, "tripolar": 3,
                  "gaussian_islands": 4}
    if grid_type not in grid_types:
        raise ValueError(f"grid_type={grid_type!r} must be one of {sorted(grid_types)}")
    if grid_types[grid_type]:
        backend_kw["grid_type"] = grid_types[grid_type]
    H = halo[0] if isinstance(halo, (tuple, list)) else halo
    if isinstance(halo, (tuple, list)) and len(set(halo)) != 1:
        raise ValueError("halo must be the same in every direction")
    # backend_kw: `options` (schedule switches, binding.OPTION_IDS) and overrides of gb25_config fields
    backend = arch(Nx, Ny, Nz, dt=dt, halo=H, substeps=substeps, **backend_kw)
    model = HydrostaticFreeSurfaceModel(backend, Nx, Ny, Nz, H)
    model.free_surface.substeps = substeps
    model.grid_type = grid_type
    model.closure = closure
    if isinstance(closure, CATKEVerticalDiffusivity):
        backend.set_catke(True)
        if closure.parameters:
            backend.set_catke_parameters(**closure.parameters)
        model.enable_catke_fields()
    elif closure is not None:
        if not isinstance(closure, VerticalScalarDiffusivity):
            raise NotImplementedError("closures: None, VerticalScalarDiffusivity(nu, kappa) or CATKEVerticalDiffusivity()")
        backend.set_vertical_diffusivity(closure.nu, closure.kappa)
    return model


def set_top_flux(model, **fluxes):
    """FluxBoundaryCondition at the top of u, v, T, S (what ClimaOcean's ocean_simulation gives the ocean model and the
    coupled model fills every step: wind stress, heat and fresh-water flux; src/data_free_ocean_climate_model.jl:26,
    src/precompile.jl:52-61).  `set_top_flux(model, T=J_T, u=tau_x)`; arrays at the interior points of the field's
    horizontal location, positive upward; None restores the default no-flux condition."""
    for name, J in fluxes.items():
        if name not in ("u", "v", "T", "S"):
            raise ValueError(f"top flux boundary conditions exist for u, v, T, S, not {name!r}")
        model.backend.set_top_flux(name, J)


def set_baroclinic_instability(model):
    """set_baroclinic_instability!(model) -- src/model_utils.jl:120-127."""
    model.backend.set_baroclinic_instability()


def first_time_step(model):
    """first_time_step!(model) -- src/timestepping_utils.jl:21-27."""
    model.backend.first_time_step()


def time_step(model):
    """time_step!(model) -- src/timestepping_utils.jl:29-35."""
    model.backend.time_step()


def loop(model, Ninner):
    """loop!(model, Ninner) -- src/timestepping_utils.jl:37-45."""
    model.backend.loop(int(Ninner))


# ---- the per-phase workloads of src/precompile.jl:44-127 ----
def tupled_fill_halo_regions_workload(model): model.backend.fill_halo_regions()
def compute_tendencies_workload(model): model.backend.compute_tendencies()
def compute_boundary_tendencies_workload(model): model.backend.compute_boundary_tendencies()
def compute_interior_momentum_tendencies_workload(model): model.backend.compute_momentum_tendencies()
def compute_interior_tracer_tendencies_workload(model): model.backend.compute_tracer_tendencies()
def compute_auxiliaries_workload(model): model.backend.compute_auxiliaries()
def fill_halo_regions_workload(model): model.backend.fill_diffusivity_halos()
def mask_immersed_model_fields_workload(model): model.backend.mask_immersed_fields()
def ab2_step_workload(model, dt): model.backend.ab2_step(dt, False)
def correct_velocities_and_cache_previous_tendencies_workload(model, dt):
    model.backend.correct_velocities_and_cache_previous_tendencies(dt)
def initialize(model): model.backend.initialize()
def update_state(model): model.backend.update_state()
