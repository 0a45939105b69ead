"""BASELINE.json configs[4] at its full size: baroclinic_instability_model 4320x2160x100 (1/12 degree) on the TripolarGrid
with the Gaussian mountains -- as ONE domain on the one MI355X of this box (4.4 GB per 3-D field, ~100 GB of the 288 GB: the
tendency kernels reach arrays beyond 4 GB through rebased pointers per run of level chunks / per-block buffer views) and in
eight x slabs of 540 columns on the same GPU (the library's local transport: the same stages, pack / unpack kernels, fold
partner exchanges and two streams as one rank per GPU over RCCL).  The size-independent property checked is decomposition
invariance: the single domain against the eight slabs, bit for bit, plus finiteness."""
import numpy as np
import pytest

import gb25_amd as gb
from gb25_amd.binding import GB25Error
from gb25_amd.distributed import LocalSlabEnsemble

pytestmark = pytest.mark.gpu
NX, NY, NZ, DT = 4320, 2160, 100, 60.0
FIELDS = ("u", "T", "eta", "V", "Gn.u")   # (4.4 GB per 3-D field and copy: the suite's longest test by far)


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


def run_single(steps):
    # (bit for bit against the slabs: w from the stand-alone kernel, as they compute it -- w on the fly changes the last bits)
    m = gb.baroclinic_instability_model(gb.GPU(), NX, NY, NZ, dt=DT, grid_type="gaussian_islands", options=dict(w_on_the_fly=0))
    gb.set_baroclinic_instability(m)
    gb.first_time_step(m)
    gb.loop(m, steps)
    out = {}
    for f in FIELDS:
        a = m.backend.get_field(f, False)
        out[f] = [a[q * 540:(q + 1) * 540].copy() for q in range(NX // 540)]
        del a
    m.backend.close()
    return out


def test_config5_grid_as_one_domain_and_in_eight_slabs():
    with pytest.raises(GB25Error, match="2\\^31"):     # (what still does not fit: 2^31 elements per array)
        gb.baroclinic_instability_model(gb.GPU(), 2 * NX, NY, NZ, dt=DT, grid_type="gaussian_islands")
    a = run(8, 1)
    for f in FIELDS:
        assert all(np.isfinite(p).all() for p in a[f]), f
    assert max(np.abs(p).max() for p in a["u"]) > 1e-4          # the fronts have started to move the water
    assert max(np.abs(p[:, NY - 1, 0]).max() for p in a["V"]) > 0     # the y faces of the pivot row (slab r's rows beyond it are slab 7-r's)
    b = run_single(1)
    for f in FIELDS:
        for q, (x, y) in enumerate(zip(a[f], b[f])):
            assert np.array_equal(x, y), (f, q, float(np.abs(x - y).max()))
