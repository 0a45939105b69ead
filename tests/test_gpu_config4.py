"""BASELINE.json configs[3] at its full size: the data-free climate model -- 1440x720x60 on the TripolarGrid with the Gaussian
islands, CATKE, the analytic atmosphere with similarity-theory fluxes after every step (GB-25
src/data_free_ocean_climate_model.jl:12-70 at the resolution of simulations/ocean_climate_simulation.jl).  The reference
decomposes it 4x2 over eight GPUs; here it is a single domain, eight x slabs of 180 columns and the 4 x 2 mesh itself -- all on
the ONE GPU of this box, the library's local transport -- which must agree bit for bit (fold partner exchanges, e and J^b in the bundles, kappa
and the fluxes of the halo column computed locally)."""
import numpy as np
import pytest

import gb25_amd as gb
from gb25_amd.data_free import ATMOSPHERE_FIELDS
from gb25_amd.distributed import LocalSlabEnsemble

pytestmark = pytest.mark.gpu
NX, NY, NZ, DT, H = 1440, 720, 60, 30.0, 8
FIELDS = ("u", "v", "T", "S", "e", "eta", "U", "V", "kappa_u", "Gn.u", "Gn.T", "Gn.e")


@pytest.fixture(scope="module")
def reference():
    m = gb.data_free_ocean_climate_model_init(gb.GPU(), Nz=NZ, dt=DT, size=(NX, NY))
    for o in ("momentum_chunk_levels", "tracer_chunk_levels"):      # (the chunking of its narrower ranks: bit for bit)
        m.backend.set_option(o, 12)
    m.backend.set_option("w_on_the_fly", 0)     # (... and w from the stand-alone kernel, as the ranks compute it: w on the fly changes the last bits)
    init = {n: m.backend.get_field(n, False) for n in ("T", "S")}
    gb.first_time_step(m)
    gb.loop(m, 3)
    ref = {n: m.backend.get_field(n, False) for n in FIELDS}
    flux = {n: m.backend.top_flux(n) for n in ("u", "T", "S")}
    m.backend.close()
    for n, a in ref.items():
        assert np.isfinite(a).all(), n
    assert np.abs(ref["u"]).max() > 1e-6 and np.abs(flux["u"]).max() > 1e-6 and ref["kappa_u"].max() > 0
    return init, ref, flux


@pytest.mark.parametrize("Rx,Ry", [(8, 1), (4, 2)])
def test_config4_grid_single_domain_and_eight_ranks(reference, Rx, Ry):
    """(8, 1): eight x slabs of 180 columns; (4, 2): the reference's own decomposition of this configuration,
    Partition(4, 2, 1) -- ranks of 360 columns x 360 rows, the fold partners within the top row."""
    init, ref, flux = reference
    ens = LocalSlabEnsemble(NX, NY, NZ, Rx * Ry, dt=DT, grid_type=4, ranks_y=Ry)
    atm = gb.analytic_atmosphere()
    for b in ens.backends:
        b.set_catke(True)
        b.set_catke_parameters(**gb.default_ocean_closure().parameters)
        b.set_bottom_drag(0.003)
        b.set_tracer_advection_order(7)
        phi = np.asarray(b.metric2("phicc"))[:, : b.Ny_local + 2 * H]
        for n in ATMOSPHERE_FIELDS:
            b.set_prescribed_atmosphere(n, atm.interpolate(n, np.zeros_like(phi), phi))
    for n, a in init.items():
        ens.scatter(n, a)
    ens.first_time_step()
    ens.loop(3)
    for n, a in ref.items():
        assert np.array_equal(ens.gather(n), a), n
    for n, a in flux.items():
        rows = [np.concatenate([ens.backends[ry * Rx + rx].top_flux(n) for rx in range(Rx)], axis=0) for ry in range(Ry)]
        assert np.array_equal(np.concatenate(rows, axis=1), a), n
    ens.close()
