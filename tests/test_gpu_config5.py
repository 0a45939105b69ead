"""BASELINE.json configs[4] at its full size: baroclinic_instability_model 4320x2160x100 (1/12 degree) on the TripolarGrid
with the Gaussian mountains, in x slabs -- all of them on the ONE GPU of this box (the library's local transport: the
same stages, pack / unpack kernels, fold partner exchanges and two streams as one rank per GPU over RCCL).  The grid does
not fit a single domain (32-bit byte offsets: 2 GB per 3-D array; the library says so), so the size-independent property
checked is decomposition invariance itself: eight slabs of 540 columns against four of 1080, bit for bit, plus exact
antisymmetry on the fold line and finiteness.  About 90 GB of HBM per ensemble."""
import numpy as np
import pytest

import gb25_amd as gb
from gb25_amd.binding import GB25Error
from gb25_amd.distributed import LocalSlabEnsemble

pytestmark = pytest.mark.gpu
NX, NY, NZ, DT = 4320, 2160, 100, 60.0
FIELDS = ("u", "v", "T", "eta", "V", "Gn.u", "Gn.T")


def run(P, steps):
    ens = LocalSlabEnsemble(NX, NY, NZ, P, dt=DT, grid_type=4)
    for b in ens.backends:
        b.set_baroclinic_instability()
    ens.first_time_step()
    ens.loop(steps)
    ens.synchronize()
    n = NX // P
    out = {}
    for f in FIELDS:                      # (kept per 540-column piece: a gathered field would be 3.7 GB)
        pieces = []
        for b in ens.backends:
            a = b.get_field(f, False)
            pieces += [a[q * 540:(q + 1) * 540].copy() for q in range(n // 540)]
        out[f] = pieces
    ens.close()
    return out


def test_config5_grid_in_eight_and_in_four_slabs():
    with pytest.raises(GB25Error, match="2 GB per array"):
        gb.baroclinic_instability_model(gb.GPU(), NX, NY, NZ, dt=DT, grid_type="gaussian_islands")
    a = run(8, 2)
    for f in FIELDS:
        assert all(np.isfinite(p).all() for p in a[f]), f
    assert max(np.abs(p).max() for p in a["u"]) > 1e-4          # the fronts have started to move the water
    assert max(np.abs(p[:, NY - 1, 0]).max() for p in a["V"]) > 0     # the y faces of the pivot row (slab r's rows beyond it are slab 7-r's)
    b = run(4, 2)
    for f in FIELDS:
        for q, (x, y) in enumerate(zip(a[f], b[f])):
            assert np.array_equal(x, y), (f, q, float(np.abs(x - y).max()))
