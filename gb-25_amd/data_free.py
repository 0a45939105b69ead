"""The data-free ocean climate model: GB-25 src/data_free_ocean_climate_model.jl:1-70.

The reference couples ClimaOcean's `ocean_simulation` on the TripolarGrid with Gaussian islands to an ANALYTIC
`PrescribedAtmosphere` (24 identical snapshots on a 360 x 180 latitude-longitude grid), `Radiation` and
`SimilarityTheoryFluxes(solver_stop_criteria = FixedIterations(5))` inside an `OceanSeaIceModel`.  Here the host side -- this
file -- builds the atmosphere and interpolates it to the ocean's cell centres; the per-step work (the similarity-theory flux
solve per surface cell and the top flux boundary conditions it feeds) runs in the library (`gb25_set_prescribed_atmosphere`,
`k_similarity_fluxes`).  What `ocean_simulation` adds beyond `baroclinic_instability_model` + CATKE -- WENO(order = 7)
tracer advection, the quadratic bottom drag, its closure parameters -- is in the library too (`gb25_set_tracer_advection_order`,
`gb25_set_bottom_drag`, `default_ocean_closure`); all of it restated from memory of ClimaOcean 0.5.10, which is not in
/root/reference: parity unpinned (DESIGN.md sections 0 and 4).
"""
import numpy as np

from .model import baroclinic_instability_model, default_ocean_closure, resolution_to_points

ATMOSPHERE_FIELDS = ("u", "v", "T", "q", "p", "shortwave", "longwave")


# src/data_free_ocean_climate_model.jl:1-3 (degrees)
def zonal_wind(lam, phi):
    return 4 * np.sin(np.radians(2 * phi)) ** 2 - 2 * np.exp(-(np.abs(phi) - 12) ** 2 / 72)


def sunlight(lam, phi):
    return -200 - 600 * np.cos(np.radians(phi)) ** 2


def Tatm(lam, phi, z=0):
    return 30 * np.cos(np.radians(phi))


class PrescribedAtmosphere:
    """ClimaOcean.PrescribedAtmosphere on LatitudeLongitudeGrid(size = (360, 180), longitude = (0, 360), latitude = (-90, 90))
    (src/data_free_ocean_climate_model.jl:31-57): velocities u, v, tracers T [K], q, pressure, downwelling shortwave and
    longwave radiation, constant in time (the reference fills all 24 snapshots alike).  Defaults as the constructor leaves
    them: v = q = longwave = 0, p = 101325 Pa."""

    def __init__(self, size=(360, 180)):
        self.size = size
        nx, ny = size
        self.lam = (np.arange(nx) + 0.5) * 360.0 / nx
        self.phi = -90.0 + (np.arange(ny) + 0.5) * 180.0 / ny
        self.fields = {n: np.zeros(size) for n in ATMOSPHERE_FIELDS}
        self.fields["p"][:] = 101325.0

    def set(self, **functions):
        """set!(field, f) with f(lam, phi) evaluated at the cell centres of the atmosphere grid."""
        L, P = np.meshgrid(self.lam, self.phi, indexing="ij")
        for name, f in functions.items():
            self.fields[name][:] = f(L, P)

    def interpolate(self, name, lam, phi):
        """Bilinear interpolation to the points (lam, phi) [degrees]: periodic in longitude, clamped beyond the first / last
        row of cell centres (where ClimaOcean's interpolation extrapolates from the halo)."""
        nx, ny = self.size
        a = self.fields[name]
        x = (np.asarray(lam, float) % 360.0) / (360.0 / nx) - 0.5
        y = np.clip((np.asarray(phi, float) + 90.0) / (180.0 / ny) - 0.5, 0.0, ny - 1.0)
        i0 = np.floor(x).astype(int)
        j0 = np.minimum(np.floor(y).astype(int), ny - 2)
        fx, fy = x - i0, y - j0
        i0, i1 = i0 % nx, (i0 + 1) % nx
        return ((1 - fx) * (1 - fy) * a[i0, j0] + fx * (1 - fy) * a[i1, j0] + (1 - fx) * fy * a[i0, j0 + 1]
                + fx * fy * a[i1, j0 + 1])


def analytic_atmosphere():
    """The atmosphere of the reference: Tatm + 273.15, zonal_wind, sunlight (set_tracers, :5-10,44-55); q = 0 (:57)."""
    atm = PrescribedAtmosphere()
    atm.set(T=lambda l, p: Tatm(l, p) + 273.15, u=zonal_wind, shortwave=sunlight)
    return atm


def cell_centre_latitudes(model):
    """Latitude of the ocean's cell centres, halo cells included: (Nx + 2H, Ny + 2H)."""
    b, H = model.backend, model.grid.halo[0]
    Nx, Ny, _ = model.grid.size
    curvilinear = getattr(model, "grid_type", "simple_lat_lon") in ("lat_lon_as_curvilinear", "tripolar", "gaussian_islands")
    if curvilinear:
        try:
            return np.asarray(b.metric2("phicc"))[:, : Ny + 2 * H].copy()          # the library: the whole 2-D metric
        except TypeError:                                                          # (tests' oracle backend: point by point)
            return np.array([[b.metric2("phicc", i, j) for j in range(1 - H, Ny + H + 1)] for i in range(1 - H, Nx + H + 1)])
    row = np.array([b.metric("phic", j) for j in range(1 - H, Ny + H + 1)])
    return np.broadcast_to(row, (Nx + 2 * H, Ny + 2 * H)).copy()


def set_prescribed_atmosphere(model, atmosphere):
    """Interpolates the atmosphere to the ocean's cell centres and hands it to the backend: the model is coupled from
    here on (ComponentInterfaces + OceanSeaIceModel, :62-66).  The analytic fields of the reference depend on latitude only;
    the longitude of a cell is taken as 0."""
    phi = cell_centre_latitudes(model)
    for name in ATMOSPHERE_FIELDS:
        model.backend.set_prescribed_atmosphere(name, atmosphere.interpolate(name, np.zeros_like(phi), phi))
    model.atmosphere = atmosphere


def smooth_step(phi):
    return (1 - np.tanh((np.abs(phi) - 40.0) / 5.0)) / 2      # src/model_utils.jl:83-87


def set_data_free_state(model, noise=None):
    """set!(ocean.model, T = Ti, S = Si) (src/data_free_ocean_climate_model.jl:27, src/model_utils.jl:89-97; their rand()
    term is `noise`, an (Nx, Ny, Nz) array or None) and the analytic atmosphere: coupled from here on.  Works on a slab too
    (the latitudes are the slab's own)."""
    Nx, Ny, Nz = model.grid.size
    H = model.grid.halo[0]
    phi = cell_centre_latitudes(model)[H:H + Nx, H:H + Ny]
    zc = np.array([model.backend.metric("zc", k) for k in range(1, Nz + 1)])
    r = 0.0 if noise is None else np.asarray(noise)
    model.set(T=(30 + 1e-3 * zc[None, None, :]) * smooth_step(phi)[:, :, None] + r,
              S=np.broadcast_to(-5e-3 * zc, (Nx, Ny, Nz)) + r)
    set_prescribed_atmosphere(model, analytic_atmosphere())
    return model


def data_free_ocean_climate_model_init(arch, resolution=2, Nz=20, *, dt=30.0, noise=None, size=None,
                                       bottom_drag_coefficient=0.003, tracer_advection_order=7, **backend_kw):
    """data_free_ocean_climate_model_init(arch; resolution = 2, Nz = 20) -- src/data_free_ocean_climate_model.jl:12-70:
    gaussian_islands_tripolar_grid(arch, resolution, Nz), SplitExplicitFreeSurface(substeps = 30), dt = 30 s, the closure of
    ClimaOcean's ocean_simulation (default_ocean_closure: CATKE with Cᵇ = 0.01), T = Ti, S = Si, the analytic atmosphere, coupled.  size = (Nx, Ny): a grid that is
    not one of resolution_to_points (BASELINE.json configs[3]: 1440 x 720)."""
    Nx, Ny = size if size is not None else resolution_to_points(resolution)
    model = baroclinic_instability_model(arch, Nx, Ny, Nz, dt=dt, grid_type="gaussian_islands",
                                         closure=default_ocean_closure(), **backend_kw)
    # ocean_simulation(grid; ...): bottom_drag_coefficient = Default(0.003) [UPSTREAM-UNVERIFIED: recalled from ClimaOcean]
    model.backend.set_bottom_drag(bottom_drag_coefficient)
    # ... and tracer_advection = WENO(order = 7) [UPSTREAM-UNVERIFIED likewise]
    model.backend.set_tracer_advection_order(tracer_advection_order)
    return set_data_free_state(model, noise)
