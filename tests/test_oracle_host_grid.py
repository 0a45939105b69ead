"""The oracle side of the host-grid setters (tests/test_gpu_host_grid.py compares the library with it): the generator's own
output fed back changes nothing; other vertical faces re-materialise the bottom on the new levels."""
import numpy as np

import gb25_amd as gb
from gb25_amd.binding import METRIC2_IDS
from helpers import make_oracle, set_noisy_velocities


def test_the_generators_own_metrics_fed_back_change_nothing():
    Nx, Ny, Nz = 48, 24, 6
    out = []
    for fed in (False, True):
        m = make_oracle(Nx, Ny, Nz, 600.0, grid_type="gaussian_islands")
        if fed:
            m.backend.set_curvilinear_grid({n: m.backend.metric2_array(n) for n in METRIC2_IDS})
            m.backend.set_vertical_faces([m.backend.metric("zf", k) for k in range(1, Nz + 2)])
        gb.set_baroclinic_instability(m)
        set_noisy_velocities(m, 1e-2)
        gb.first_time_step(m)
        gb.loop(m, 3)
        out.append({n: m.backend.get_field(n, True) for n in ("u", "v", "T", "eta", "Gn.u")})
    for n in out[0]:
        assert np.array_equal(out[0][n], out[1][n]), n


def test_other_vertical_faces_rematerialise_the_bottom():
    Nx, Ny, Nz = 32, 20, 8
    m = make_oracle(Nx, Ny, Nz, 600.0)
    zb = np.full((Nx, Ny), -2600.0)
    m.backend.set_bottom_height(zb)
    zf = -4000.0 * (1 - np.linspace(0, 1, Nz + 1))          # uniform 500 m levels: centres at -3750, -3250, -2750, ...
    m.backend.set_vertical_faces(zf)
    assert [m.backend.metric("zf", k) for k in (1, Nz + 1)] == [-4000.0, 0.0]
    assert m.backend.metric("dzc", 3) == 500.0
    assert m.backend.bottom_info("kbot", 5, 5) == 3                 # z_center <= bottom: the three lowest cells
    assert m.backend.bottom_info("Hfc", 5, 5) == 2500.0
