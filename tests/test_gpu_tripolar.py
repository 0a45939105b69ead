"""TripolarGrid + GridFittedBottom(gaussian_islands) on the HIP path (second half; GB-25
src/model_utils.jl:129-146, grid_type = :gaussian_islands of src/baroclinic_instability_model.jl:59-65) against the
oracle's restatement of the same grid (tests/test_oracle_tripolar.py pins that one): the grid generator, the
orthogonal-curvilinear kernel variants (2-D metrics), the zipper fold in the halo fills, the stepped row of y faces on
the fold line, the fold inside the split-explicit sub-cycle."""
import numpy as np
import pytest

import gb25_amd as gb
from gb25_amd.binding import METRIC2_IDS
from helpers import SQRT_EPS32, assert_states_close, counter_rng, make_pair, set_noisy_velocities

pytestmark = pytest.mark.gpu
ALL_FIELDS = ["u", "v", "w", "T", "S", "pHY", "Gn.u", "Gn.v", "Gn.T", "Gn.S", "Gm.u", "Gm.v", "Gm.T", "Gm.S",
              "eta", "U", "V", "eta_bar", "U_bar", "V_bar", "Gn.U", "Gn.V"]
H = 8


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    n = max(np.linalg.norm(a.ravel()), np.linalg.norm(b.ravel()))
    return 0.0 if n == 0 else float(np.linalg.norm((a - b).ravel()) / n)


def start(r, v, amplitude=1e-2):
    gb.set_baroclinic_instability(v)
    set_noisy_velocities(v, amplitude)
    for n in ALL_FIELDS:
        a = v.backend.get_field(n, True).astype(np.float32)
        r.backend.set_field(n, a, True)
        v.backend.set_field(n, a.astype(v.backend.dtype), True)


@pytest.mark.parametrize("grid_type", ["lat_lon_as_curvilinear", "tripolar"])
def test_grid_generator_matches_the_oracle(grid_type):
    """Every metric, every location, halo rows and columns included (the rows beyond the fold are the mirrored cells)."""
    Nx, Ny, Nz = 72, 36, 6
    r, v = make_pair(Nx, Ny, Nz, dt=600.0, float_type="Float64", grid_type=grid_type)
    for name in METRIC2_IDS:
        a = r.backend.metric2(name)
        assert a.shape == (Nx + 2 * H, Ny + 2 * H + 1)
        # (the outermost halo row / column have no neighbour to measure to in the oracle's arrays either: same formula)
        b = np.array([[v.backend.metric2(name, i, j) for j in range(2 - H, Ny + H + 1)] for i in range(2 - H, Nx + H)])
        assert np.allclose(a[1:-1, 1:-1], b, rtol=1e-11, atol=0), (name, np.abs(a[1:-1, 1:-1] / b - 1).max())
    # an analytic bottom sees the same physical coordinates
    ri, vi = make_pair(Nx, Ny, 8, dt=600.0, grid_type="gaussian_islands" if grid_type == "tripolar" else "lat_lon_as_curvilinear")
    for i in range(1, Nx + 1):
        for j in range(1, Ny + 1):
            assert ri.backend.bottom_info("kbot", i, j) == vi.backend.bottom_info("kbot", i, j), (i, j)
            assert ri.backend.bottom_info("Hfc", i, j) == pytest.approx(vi.backend.bottom_info("Hfc", i, j), rel=1e-6)
            assert ri.backend.bottom_info("Hcf", i, j) == pytest.approx(vi.backend.bottom_info("Hcf", i, j), rel=1e-6)


@pytest.mark.parametrize("float_type,tol", [("Float64", 1e-12), ("Float32", 5e-6)])
def test_lat_lon_grid_through_the_curvilinear_kernels(float_type, tol):
    """grid_type 2: the lat-lon metrics as 2-D arrays through the CURV kernel variants (per-point face lengths in LDS,
    Az w tiles, per-point reciprocals, the per-substep barotropic kernel) against the plain kernels with their row
    tables: the same numbers to round-off (different template instances contract different FMAs; the sub-cycle divides
    where the blocked kernel multiplies by a reciprocal)."""
    Nx, Ny, Nz, dt = 150, 70, 24, 600.0
    dtype = np.float64 if float_type == "Float64" else np.float32
    a = gb.baroclinic_instability_model(gb.GPU(float_type=float_type), Nx, Ny, Nz, dt=dt)
    b = gb.baroclinic_instability_model(gb.GPU(float_type=float_type), Nx, Ny, Nz, dt=dt, grid_type="lat_lon_as_curvilinear")
    for m in (a, b):
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 0.05)
        m.set(eta=(1e-2 * counter_rng((Nx, Ny, 1), 3, 3)).astype(dtype))
        gb.first_time_step(m)
        gb.loop(m, 6)
    for n in ALL_FIELDS:
        x, y = a.backend.get_field(n, False), b.backend.get_field(n, False)
        assert rel(x, y) < tol, (n, rel(x, y))
    assert np.abs(a.velocities.u.interior).max() > 0.05


def test_fold_fill_is_the_oracles_data_movement():
    """tupled_fill_halo_regions! with the zipper fold: pure data movement (signs, mirrored columns, the symmetrised row
    of y faces on the fold line, the bottom / top layers of the rows beyond the fold): identical parents."""
    Nx, Ny, Nz = 72, 36, 8
    r, v = make_pair(Nx, Ny, Nz, dt=600.0, grid_type="tripolar")
    rng = np.random.default_rng(5)
    for n in ("u", "v", "T", "S", "eta", "U", "V"):
        shape = r.backend.get_field(n, True).shape
        a = rng.standard_normal(shape).astype(np.float32)
        r.backend.set_field(n, a, True)
        v.backend.set_field(n, a.astype(np.float64), True)
    for m in (r, v):
        m.backend.fill_halo_regions()
    for n in ("u", "v", "T", "S", "eta", "U", "V"):
        a, b = r.backend.get_field(n, True), v.backend.get_field(n, True).astype(np.float32)
        assert np.array_equal(a, b), (n, np.argwhere(a != b)[:5])
    vp = r.backend.get_field("v", True)
    assert np.array_equal(vp[H:-H, H + Ny, H:-H], -vp[H:-H, H + Ny - 1, H:-H][::-1])    # y faces beyond the pivot row: the images


def test_phase_by_phase_on_the_tripolar_grid_with_islands():
    Nx, Ny, Nz = 72, 36, 8
    r, v = make_pair(Nx, Ny, Nz, dt=600.0, grid_type="gaussian_islands")
    start(r, v)
    get = lambda m, n: m.backend.get_field(n, True)
    sync = lambda: [r.backend.set_field(n, get(v, n).astype(np.float32), True) or
                    v.backend.set_field(n, get(v, n).astype(np.float32).astype(np.float64), True) for n in ALL_FIELDS]
    for m in (r, v):
        m.backend.mask_immersed_fields()
    for n in ("u", "v", "T", "S", "U", "V"):
        assert np.array_equal(get(r, n), get(v, n).astype(np.float32)), n
    sync()
    for m in (r, v):
        m.backend.initialize()
        m.backend.update_state()
    core = (slice(H - 1, -(H - 1)), slice(H - 1, -(H - 1)), slice(H, -H))
    assert rel(get(r, "w")[core], get(v, "w")[core]) < 1e-5
    for n, tol in (("Gn.T", 2e-4), ("Gn.S", 2e-4), ("Gn.u", 2e-4), ("Gn.v", 2e-4)):
        assert rel(get(r, n), get(v, n)) < tol, (n, rel(get(r, n), get(v, n)))
    # the pivot row (the last row of cells, held twice) and its y faces have tendencies like any other row, and they agree
    piv = (slice(H, -H), H + Ny - 1, slice(H, -H))
    for n in ("Gn.v", "Gn.T"):
        assert np.abs(get(v, n)[piv]).max() > 0
        assert rel(get(r, n)[piv], get(v, n)[piv]) < 2e-4
    for n in ("Gn.u", "Gn.v"):
        assert np.array_equal(get(r, n) == 0, get(v, n) == 0), n
    for euler in (True, False):
        sync()
        for m in (r, v):
            m.backend.ab2_step(600.0, euler)
        for n in ("u", "v", "T", "S", "eta", "U", "V", "eta_bar", "U_bar", "V_bar", "Gn.U", "Gn.V"):
            assert rel(get(r, n), get(v, n)) < 2e-5, (n, euler, rel(get(r, n), get(v, n)))
        assert np.abs(get(r, "V")[H:-H, H + Ny - 1, 0]).max() > 0      # (the y faces of the pivot row are ordinary faces)
    sync()
    for m in (r, v):
        m.backend.fill_halo_regions()
        m.backend.correct_velocities_and_cache_previous_tendencies(600.0)
    for n in ("u", "v", "U_bar", "V_bar"):
        assert rel(get(r, n), get(v, n)) < 2e-6, n
        assert np.array_equal(get(r, n)[H:-H, H:-H] == 0, get(v, n)[H:-H, H:-H] == 0), n


@pytest.mark.parametrize("size", [(72, 36, 8, 600.0, 8), (180, 90, 12, 600.0, 5)])
def test_stepping_the_reference_gaussian_islands_grid(size):
    """first_time_step! + loop! on grid_type = :gaussian_islands against the oracle at the reference's tolerance, every
    compared field, halos (the fold rows among them) included."""
    Nx, Ny, Nz, dt, nsteps = size
    r, v = make_pair(Nx, Ny, Nz, dt=dt, grid_type="gaussian_islands")
    start(r, v, 1e-3)
    for m in (r, v):
        gb.first_time_step(m)
        gb.loop(m, nsteps - 1)
    assert_states_close(r, v, label=f"{Nx}x{Ny}x{Nz} tripolar islands after {nsteps} steps")
    assert np.isfinite(r.backend.get_field("eta", False)).all()
    assert np.abs(r.velocities.u.interior).max() > 1e-3
    r.backend.fill_halo_regions()
    vp = r.backend.get_field("v", True)
    assert np.array_equal(vp[H:-H, H + Ny, H:-H], -vp[H:-H, H + Ny - 1, H:-H][::-1]) and np.abs(vp[H:-H, H + Ny - 1, H:-H]).max() > 0


def test_rest_state_stays_at_rest_on_the_tripolar_grid():
    Nx, Ny, Nz = 72, 36, 8
    m = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=600.0, grid_type="gaussian_islands")
    zc = np.array([m.backend.metric("zc", k) for k in range(1, Nz + 1)])
    m.set(T=np.broadcast_to(10 + 5e-3 * zc, (Nx, Ny, Nz)).astype(np.float32),
          S=np.broadcast_to(35 - 1e-3 * zc, (Nx, Ny, Nz)).astype(np.float32))
    gb.first_time_step(m)
    gb.loop(m, 3)
    for name in ("u", "v", "w", "eta", "U", "V"):
        assert np.abs(m.backend.get_field(name, False)).max() == 0.0, name


def test_schedule_options_do_not_change_the_tripolar_step():
    """The AB2 and sub-cycle look-aheads, and the one-stream schedule, on the folded grid: the fold-line row rides along
    in every route (adopted buffers, per-chunk column sums, the sub-cycle's own row of threads)."""
    Nx, Ny, Nz = 72, 36, 12
    outs = []
    for opts in (dict(), dict(ab2_lookahead=0), dict(subcycle_lookahead=1), dict(two_streams=0)):
        m = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=600.0, grid_type="gaussian_islands", options=opts)
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 1e-2)
        gb.first_time_step(m)
        gb.loop(m, 5)
        outs.append({n: m.backend.get_field(n, False) for n in ("u", "v", "T", "S", "eta", "U", "V")})
        m.backend.close()
    for o in outs[1:]:
        for n, a in outs[0].items():
            assert rel(a, o[n]) < 2e-6, (n, rel(a, o[n]))


@pytest.mark.parametrize("grid_type,P", [("lat_lon_as_curvilinear", 3), ("gaussian_islands", 2), ("gaussian_islands", 3),
                                         ("gaussian_islands", 4), ("tripolar", 1)])
def test_slabs_of_a_curvilinear_grid_reproduce_the_single_domain_bitwise(grid_type, P):
    """x-slab decomposition of the curvilinear grids: the 2-D metrics of a slab
    come from the same generator at its global columns; on the tripolar grid the cells beyond a slab's fold line belong
    to the mirrored slab P-1-r (an odd count makes the middle slab its own partner; P = 1 is the self-ring: slab_mode 1)
    -- the rows next to the fold line travel once per step, five rows of the widened barotropic arrays once per
    substep.  Every compared field bit for bit against the single domain."""
    from gb25_amd.distributed import LocalSlabEnsemble
    Nx, Ny, Nz, dt = 96 * max(P, 2) // 2, 40, 10, 600.0       # slabs of >= 48 columns (> the 22-column barotropic halo)
    single = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt, grid_type=grid_type)
    gb.set_baroclinic_instability(single)
    single.set(u=(1e-2 * counter_rng((Nx, Ny, Nz), 42, 1)).astype(np.float32),
               v=(1e-2 * counter_rng((Nx, Ny + 1, Nz), 42, 2)).astype(np.float32),
               eta=(1e-2 * counter_rng((Nx, Ny, 1), 42, 3)).astype(np.float32))
    init = {n: single.backend.get_field(n, False) for n in ("u", "v", "T", "S", "eta")}
    gtypes = {"lat_lon_as_curvilinear": 2, "tripolar": 3, "gaussian_islands": 4}
    kw = dict(slab_mode=1) if P == 1 else {}
    ens = LocalSlabEnsemble(Nx, Ny, Nz, P, dt=dt, grid_type=gtypes[grid_type], **kw)
    for n, a in init.items():
        ens.scatter(n, a)
    gb.first_time_step(single)
    ens.first_time_step()
    steps = 2 if grid_type == "tripolar" else 5          # (the bare tripolar grid does not live long: the poles)
    gb.loop(single, steps)
    ens.loop(steps)
    for n in ALL_FIELDS:
        a, b = ens.gather(n), single.backend.get_field(n, False)
        if grid_type == "tripolar":
            # without land the two poles are singular points of the coordinates (metrics clamped there): the handful of
            # cells around them amplify the last bit; everywhere else, and everywhere on the grids one can run, bit for bit
            assert rel(a, b) < 1e-6, (grid_type, P, n, rel(a, b))
            far = np.ones(Nx, bool)
            far[:4] = far[-4:] = far[Nx // 2 - 4: Nx // 2 + 4] = False
            if n not in ("Gn.u", "Gn.v", "Gn.T", "Gn.S", "w"):
                assert np.array_equal(a[far], b[far]), (grid_type, P, n)
        else:
            assert np.array_equal(a, b), (grid_type, P, n, float(np.abs(a - b).max()), np.argwhere(a != b)[:3].tolist())
    assert np.abs(single.velocities.u.interior).max() > 1e-3
    ens.close()


def test_rccl_self_ring_on_the_folded_grid():
    """The RCCL transport with the fold: ONE rank that is its own west / east neighbour AND its own fold partner
    (slab_mode = 1): the x exchanges are ncclSend/ncclRecv to self, the partner exchanges device copies; bit for bit the
    single domain, through a changed dt."""
    from gb25_amd.distributed import SlabModel
    Nx, Ny, Nz, dt = 128, 40, 10, 600.0
    single = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt, grid_type="gaussian_islands")
    gb.set_baroclinic_instability(single)
    single.set(u=(1e-2 * counter_rng((Nx, Ny, Nz), 42, 1)).astype(np.float32),
               v=(1e-2 * counter_rng((Nx, Ny + 1, Nz), 42, 2)).astype(np.float32))
    init = {n: single.backend.get_field(n, False) for n in ("u", "v", "T", "S", "eta")}
    ring = SlabModel(Nx, Ny, Nz, dt=dt, rank=0, nranks=1, slab_mode=1, transport="rccl", grid_type=4)
    for n, a in init.items():
        ring.backend.set_field(n, a, False)
    for m in (single, ring):
        gb.first_time_step(m)
        gb.loop(m, 4)
        m.backend.set_dt(450.0)
        gb.loop(m, 3)
    for n in ALL_FIELDS:
        a, b = ring.backend.get_field(n, False), single.backend.get_field(n, False)
        assert np.array_equal(a, b), (n, float(np.abs(a - b).max()))
    ring.backend.close()
    single.backend.close()


@pytest.mark.parametrize("float_type", ["Float32", "Float64"])
def test_folded_slabs_with_closure_and_fluxes(float_type):
    """Everything at once on the reference's :gaussian_islands grid in three slabs (the middle one its own fold partner):
    the vertically implicit closure, a wind stress and a heat flux, both float types -- bit for bit the single domain."""
    from gb25_amd.distributed import LocalSlabEnsemble
    Nx, Ny, Nz, dt, P = 144, 40, 10, 600.0, 3
    dtype = np.float32 if float_type == "Float32" else np.float64
    closure = gb.VerticalScalarDiffusivity(nu=1e-2, kappa=1e-3)
    single = gb.baroclinic_instability_model(gb.GPU(float_type=float_type), Nx, Ny, Nz, dt=dt, grid_type="gaussian_islands",
                                             closure=closure)
    gb.set_baroclinic_instability(single)
    single.set(u=(1e-2 * counter_rng((Nx, Ny, Nz), 42, 1)).astype(dtype),
               v=(1e-2 * counter_rng((Nx, Ny + 1, Nz), 42, 2)).astype(dtype))
    tau = (1e-4 * np.cos(np.linspace(-1.4, 1.4, Ny))[None, :] * np.ones((Nx, 1))).astype(dtype)
    heat = (1e-5 * (counter_rng((Nx, Ny), 7, 1) - 0.5)).astype(dtype)
    gb.set_top_flux(single, u=tau, T=heat)
    init = {n: single.backend.get_field(n, False) for n in ("u", "v", "T", "S", "eta")}
    ens = LocalSlabEnsemble(Nx, Ny, Nz, P, dt=dt, grid_type=4, float_type=float_type)
    n = Nx // P
    for r, b in enumerate(ens.backends):
        b.set_vertical_diffusivity(closure.nu, closure.kappa)
        b.set_top_flux("u", tau[r * n:(r + 1) * n])
        b.set_top_flux("T", heat[r * n:(r + 1) * n])
    for name, a in init.items():
        ens.scatter(name, a)
    gb.first_time_step(single)
    ens.first_time_step()
    gb.loop(single, 4)
    ens.loop(4)
    for name in ALL_FIELDS:
        a, b = ens.gather(name), single.backend.get_field(name, False)
        assert np.array_equal(a, b), (float_type, name, float(np.abs(a - b).max()))
    ens.close()


def test_a_folded_grid_too_short_for_its_sub_cycle_is_refused():
    """(see tests/test_oracle_tripolar.py) Ny < Ns + 3 on a folded grid is an error at construction, not a silent clip."""
    from gb25_amd.binding import GB25Error
    with pytest.raises(GB25Error, match="folded grid needs"):
        gb.baroclinic_instability_model(gb.GPU(), 48, 20, 6, dt=600.0, grid_type="tripolar")
    m = gb.baroclinic_instability_model(gb.GPU(), 48, 24, 6, dt=600.0, grid_type="gaussian_islands")
    gb.set_baroclinic_instability(m)
    gb.first_time_step(m)
    assert np.isfinite(m.free_surface.eta.interior).all()


@pytest.mark.parametrize("float_type,tol", [("Float32", 3.4527e-4), ("Float64", 1e-9)])
def test_fold_pivot_slaved_option_matches_the_oracle(float_type, tol):
    """Option fold_pivot_slaved (the later upstream fix as recalled: every fold fill overwrites the eastern half of the pivot
    row with the image of the western half) in the library and in the oracle; afterwards the two halves are exact images."""
    from helpers import assert_states_close, make_pair, set_noisy_velocities
    r, v = make_pair(48, 24, 6, dt=600.0, float_type=float_type, grid_type="gaussian_islands")
    for m in (r, v):
        m.backend.set_option("fold_pivot_slaved", 1)
    gb.set_baroclinic_instability(v)
    set_noisy_velocities(v, 1e-3)          # (noise: the two copies of the pivot row would drift apart without the option)
    gb.sync_states(r, v)
    for m in (r, v):
        gb.first_time_step(m)
        gb.loop(m, 4)
    assert_states_close(r, v, state_rtol=tol, tendency_rtol=tol, label="fold_pivot_slaved")
    r.backend.fill_halo_regions()
    T, u = r.backend.get_field("T", False), r.backend.get_field("u", False)
    Nx = T.shape[0]
    assert np.array_equal(T[Nx // 2:, -1], T[:Nx // 2, -1][::-1])
    assert np.array_equal(u[Nx // 2 + 1:, -1], -u[1:Nx // 2, -1][::-1])
