"""The host's grid through the C ABI (gb25_set_curvilinear_grid, gb25_set_vertical_faces, gb25_set_bottom_height): in the
reference the grid is built by Oceananigans on the Julia side (TripolarGrid(arch; size, halo, z), GridFittedBottom,
exponential_z_faces: src/model_utils.jl:56-62,129-146) and handed to the model; the library's own generators are stand-ins.
 (a) feeding the built-in generator's own output back through the setters changes no bit;
 (b) a DIFFERENT grid -- perturbed metrics, other vertical faces, another bottom -- steps like the oracle fed the same arrays
     (the reference's tolerance, halos included), as a single domain and, bit for bit, in 2 and 4 slabs."""
import numpy as np
import pytest

import gb25_amd as gb
from gb25_amd.binding import METRIC2_IDS
from helpers import SQRT_EPS32, assert_states_close, counter_rng, make_pair, set_noisy_velocities

pytestmark = pytest.mark.gpu
H = 8
ALL = ["u", "v", "w", "T", "S", "pHY", "Gn.u", "Gn.v", "Gn.T", "Gn.S", "Gm.u", "Gm.v", "Gm.T", "Gm.S", "eta", "U", "V",
       "eta_bar", "U_bar", "V_bar", "Gn.U", "Gn.V"]


def z_faces(m, Nz):
    return np.array([m.backend.metric("zf", k) for k in range(1, Nz + 2)])


@pytest.mark.parametrize("grid_type", ["lat_lon_as_curvilinear", "tripolar", "gaussian_islands"])
def test_the_generators_own_output_through_the_setters_is_bitwise_neutral(grid_type):
    Nx, Ny, Nz = 96, 44, 10
    models = []
    for fed in (False, True):
        m = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=300.0, grid_type=grid_type)
        if fed:
            metrics = {n: m.backend.metric2(n) for n in METRIC2_IDS}
            if grid_type == "gaussian_islands":
                kb = np.array([[m.backend.bottom_info("kbot", i, j) for j in range(1, Ny + 1)] for i in range(1, Nx + 1)], int)
                zf = z_faces(m, Nz)
                m.backend.set_bottom_height(np.where(kb > 0, zf[kb] - 1e-6 * np.abs(zf[kb]), -1e30))   # the materialised bottom again
            m.backend.set_curvilinear_grid(metrics)
            m.backend.set_vertical_faces(z_faces(m, Nz))
            for n in METRIC2_IDS:
                assert np.array_equal(m.backend.metric2(n), metrics[n]), n
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 1e-2)
        gb.first_time_step(m)
        gb.loop(m, 5)
        models.append(m)
    a, b = models
    for n in ALL:   # (equal_nan: the bare tripolar grid, whose poles are singular points without land, does not stay finite)
        assert np.array_equal(a.backend.get_field(n, True), b.backend.get_field(n, True), equal_nan=True), n


def another_grid(v, Nx, Ny, Nz, folded):
    """A grid the generators do not make: their metrics scaled by smooth positive factors (a different factor per metric, so
    that no two of them stay tied), other vertical faces, a bottom with a ridge and a few land columns."""
    x = (np.arange(-H, Nx + H) + 0.5)[:, None] * 2 * np.pi / Nx
    y = (np.arange(-H, Ny + H + 1) + 0.5)[None, :] * np.pi / Ny
    metrics = {}
    for q, n in enumerate(METRIC2_IDS):
        a = v.backend.metric2_array(n)
        if n not in ("fff", "phicc"):
            a = a * (1 + 0.03 * np.sin(x * (1 + q % 3) + 0.3 * q) * np.cos(y * (1 + q % 2)))
        metrics[n] = a
    zf = -4200.0 * (1 - np.linspace(0, 1, Nz + 1)) ** 1.7
    lam = (np.arange(Nx) + 0.5)[:, None] * 2 * np.pi / Nx
    phi = (np.arange(Ny) + 0.5)[None, :] * np.pi / Ny
    zb = -4200.0 + 2500.0 * np.exp(-((lam - 2.0) / 0.5) ** 2) * np.sin(phi) ** 2 + 4500.0 * np.exp(-(((lam - 4.5) / 0.25) ** 2 + ((phi - 1.2) / 0.2) ** 2))
    if folded:   # the two copies of a pivot-row cell share their bottom; land over the two poles (singular points of the grid)
        zb[:, Ny - 1] = 0.5 * (zb[:, Ny - 1] + zb[::-1, Ny - 1])
        for ip in (0, Nx // 2):
            for di in (-2, -1, 0, 1):
                zb[(ip + di) % Nx, Ny - 6:] = 100.0
    return metrics, zf, zb


@pytest.mark.parametrize("float_type", ["Float32", "Float64"])
@pytest.mark.parametrize("grid_type", ["lat_lon_as_curvilinear", "tripolar"])
def test_a_different_host_grid_steps_like_the_oracle(grid_type, float_type):
    Nx, Ny, Nz = 96, 44, 10
    r, v = make_pair(Nx, Ny, Nz, dt=300.0, float_type=float_type, grid_type=grid_type)
    metrics, zf, zb = another_grid(v, Nx, Ny, Nz, grid_type == "tripolar")
    for m in (r, v):
        m.backend.set_curvilinear_grid(metrics)
        m.backend.set_vertical_faces(zf)
        m.backend.set_bottom_height(zb)
    for n in METRIC2_IDS:   # what the library steps on is what the host gave it (rows beyond a fold's pivot row: the images)
        a, b = r.backend.metric2(n), v.backend.metric2_array(n)
        assert np.allclose(a, b, rtol=1e-6 if float_type == "Float32" else 1e-14, atol=0), n
        rows = slice(0, Ny + H) if grid_type == "tripolar" else slice(None)
        assert np.array_equal(a[:, rows], metrics[n][:, rows].astype(r.backend.dtype).astype(np.float64)), n
    assert np.allclose(z_faces(r, Nz), zf) and np.allclose(z_faces(v, Nz), zf)
    gb.set_baroclinic_instability(v)
    set_noisy_velocities(v, 1e-2)
    gb.sync_states(r, v)
    for m in (r, v):
        gb.first_time_step(m)
        gb.loop(m, 7)
    tol = SQRT_EPS32 if float_type == "Float32" else 1e-7
    assert_states_close(r, v, state_rtol=tol, tendency_rtol=tol, label=f"host grid on {grid_type}, {float_type}")
    assert np.abs(r.velocities.u.interior).max() > 1e-3 and np.isfinite(r.backend.get_field("eta", False)).all()


@pytest.mark.parametrize("grid_type,P,Ry,Ny", [("tripolar", 2, 1, 44), ("tripolar", 4, 1, 44), ("lat_lon_as_curvilinear", 3, 1, 44),
                                               ("tripolar", 4, 2, 88), ("lat_lon_as_curvilinear", 4, 2, 88)])
def test_slabs_take_their_columns_from_the_hosts_global_arrays(grid_type, P, Ry, Ny):
    """... and the ranks of a 2-D mesh (Ry = 2) their columns AND rows: every rank is handed the same GLOBAL arrays."""
    from gb25_amd.distributed import LocalSlabEnsemble
    from helpers import make_oracle
    Nx, Nz, dt = 192, 10, 300.0
    v = make_oracle(Nx, Ny, Nz, dt, grid_type=grid_type)
    metrics, zf, zb = another_grid(v, Nx, Ny, Nz, grid_type == "tripolar")
    single = gb.baroclinic_instability_model(gb.GPU(), Nx, Ny, Nz, dt=dt, grid_type=grid_type)
    ens = LocalSlabEnsemble(Nx, Ny, Nz, P, dt=dt, ranks_y=Ry, grid_type={"tripolar": 3, "lat_lon_as_curvilinear": 2}[grid_type])
    for b in [single.backend] + list(ens.backends):
        b.set_curvilinear_grid(metrics)
        b.set_vertical_faces(zf)
        b.set_bottom_height(zb)
    gb.set_baroclinic_instability(single)
    single.set(u=(1e-2 * counter_rng((Nx, Ny, Nz), 42, 1)).astype(np.float32),
               v=(1e-2 * counter_rng((Nx, Ny + 1, Nz), 42, 2)).astype(np.float32))
    for n in ("u", "v", "T", "S", "eta"):
        ens.scatter(n, single.backend.get_field(n, False))
    gb.first_time_step(single)
    ens.first_time_step()
    gb.loop(single, 4)
    ens.loop(4)
    for n in ALL:
        a, b = ens.gather(n), single.backend.get_field(n, False)
        assert a.shape == b.shape and np.array_equal(a, b), (grid_type, P, Ry, n, float(np.abs(a - b).max()))
