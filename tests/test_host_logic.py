"""Host-side logic that needs neither GPU nor oracle arithmetic."""
import math

import numpy as np
import pytest

import gb25_amd as gb
from gb25_amd.correctness import approx_equal
from helpers import make_oracle, set_noisy_velocities


def test_factors_matches_reference_rule():
    # src/sharding_utils.jl:39-62
    assert gb.factors(4) == (2, 2) and gb.factors(16) == (4, 4) and gb.factors(9180) == (135, 68)
    assert gb.factors(8) == (4, 2) and gb.factors(32) == (8, 4) and gb.factors(2048) == (64, 32)
    with pytest.raises(ValueError):
        gb.factors(7)
    with pytest.raises(ValueError):
        gb.factors(12)


def test_resolution_to_points():
    assert gb.resolution_to_points(2) == (192, 96) and gb.resolution_to_points(0.25) == (1536, 768)
    with pytest.raises(ValueError):
        gb.resolution_to_points(5)


def test_isapprox_is_a_norm_test():
    a = np.zeros(100); b = np.zeros(100)
    a[0], b[0] = 1.0, 1.0
    b[5] = 1e-5          # element-wise relative error is infinite, norm-wise it is 1e-5
    assert approx_equal(a, b, rtol=1e-4, atol=0)
    assert not approx_equal(a, b, rtol=1e-6, atol=0)
    assert not approx_equal(a, np.full(100, np.nan), rtol=1, atol=0)
    assert approx_equal(np.zeros(3), np.zeros(3), rtol=0, atol=0)


def test_compare_and_sync_states_walk_the_reference_field_set():
    m1, m2 = make_oracle(16, 12, 4, 1.0), make_oracle(16, 12, 4, 1.0)
    set_noisy_velocities(m2)
    ok, report = gb.compare_states(m1, m2, include_halos=True, verbose=False)
    names = [r["name"] for r in report]
    assert names == ["u", "Gn.u", "Gm.u", "v", "Gn.v", "Gm.v", "w", "eta", "T", "Gn.T", "Gm.T", "S", "Gn.S", "Gm.S",
                     "filtered.U", "filtered.V", "filtered.eta"]      # src/correctness.jl:37-58
    assert not ok
    with pytest.raises(AssertionError):
        gb.compare_states(m1, m2, throw_error=True, verbose=False)
    gb.sync_states(m1, m2)
    ok, _ = gb.compare_states(m1, m2, rtol=0.0, include_halos=True, verbose=False)
    assert ok


def test_model_api_shapes():
    m = make_oracle(16, 12, 4, 7.0)
    assert m.velocities.u.shape == (16, 12, 4) and m.velocities.v.shape == (16, 13, 4)
    assert m.velocities.w.shape == (16, 12, 5) and m.free_surface.eta.shape == (16, 12, 1)
    assert m.velocities.v.parent.shape == (32, 29, 20) and m.free_surface.barotropic_velocities.V.parent.shape == (32, 29, 1)
    assert m.clock.last_dt == 7.0 and m.clock.iteration == 0
    assert list(m.fields()) == ["u", "v", "w", "eta", "T", "S"]
    assert math.isclose(m.grid.z_faces()[0], -4000.0)


def test_arg_parsing_mirrors_the_reference_flags():
    a = gb.parse_baroclinic_instability_args([], grid_x_default=128, grid_y_default=128, grid_z_default=16)
    assert (a["grid-x"], a["grid-y"], a["grid-z"], a["float-type"]) == (128, 128, 16, "Float64")   # src/arg_parsing.jl:28-31
    assert gb.float_type_from_args(a) == "f64" and gb.multifloat_from_args(a) is None
    assert gb.interior_size(a) == (112, 112, 16)       # correctness script: Nx = Ny = 128 - 16
    b = gb.parse_baroclinic_instability_args(["--grid-x", "768", "--grid-y", "768", "--grid-z", "64", "--float-type", "f32"],
                                             grid_x_default=1, grid_y_default=1, grid_z_default=1)
    assert gb.float_type_from_args(b) == "f32" and gb.interior_size(b, Rx=8, Ry=4) == (6128, 3056, 64)  # alps-weak-scaling.jl:9
    with pytest.raises(AssertionError):
        gb.float_type_from_string("Float128")


def test_state_dump_roundtrip(tmp_path):
    # two "ranks" of a 16x12x4 model written separately and re-assembled offline (src/sharded_io.jl:122-213)
    whole = make_oracle(16, 12, 4, 1.0)
    set_noisy_velocities(whole)
    parts = []
    for r in range(2):
        m = make_oracle(8, 12, 4, 1.0)
        m.set(u=whole.velocities.u.interior[8 * r:8 * r + 8], v=whole.velocities.v.interior[8 * r:8 * r + 8])
        parts.append(gb.save_model_state(str(tmp_path), m, rank=r, nranks=2, label="after_loop"))
    assert [p.endswith(f"fields_rank{r}.npz") for r, p in enumerate(parts)] == [True, True]
    got = gb.load_all_fields(str(tmp_path / "after_loop"))
    assert np.array_equal(got["u"], whole.velocities.u.interior) and np.array_equal(got["v"], whole.velocities.v.interior)
    assert got["iteration"] == 0 and set(got) >= {"u", "v", "w", "eta", "T", "S"}


def test_library_npz_writer_roundtrip(tmp_path):
    """The state dump of the library (csrc/state_io.hpp: an uncompressed .npz written by hand, CRC-32 and all) as a
    host-only harness: np.load and zipfile must accept it, arrays come back in [i, j, k] order."""
    import os
    import subprocess
    import zipfile
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "t.cpp"
    out = tmp_path / "t.npz"
    src.write_text('#include "%s"\n' % os.path.join(ROOT, "gb-25_amd", "csrc", "state_io.hpp") + r'''
int main() {
  NpzWriter z;
  if (!z.open("%s")) return 2;
  long long it = 42; double t = 3.5;
  z.add_array("iteration", "<i8", {}, &it);
  z.add_array("time", "<f8", {}, &t);
  std::vector<float> a(5 * 3 * 4);
  for (size_t q = 0; q < a.size(); q++) a[q] = (float)q;
  z.add_array("u.data", "<f4", {5, 3, 4}, a.data());
  std::vector<double> b(7, 1.25);
  z.add_array("w.data", "<f8", {7}, b.data());
  z.add("field_names", "u\nw\n", nullptr, 0);
  return z.close() ? 0 : 1;
}''' % out)
    exe = tmp_path / "t"
    subprocess.run(["g++", "-std=c++17", "-O1", "-o", str(exe), str(src)], check=True)
    subprocess.run([str(exe)], check=True)
    assert zipfile.ZipFile(out).testzip() is None                      # every CRC-32 checks out
    z = np.load(out)
    assert int(z["iteration"]) == 42 and float(z["time"]) == 3.5
    u = z["u.data"]
    assert u.shape == (5, 3, 4) and u.dtype == np.float32
    assert u[2, 1, 3] == 2 + 5 * 1 + 15 * 3                             # i fastest
    assert np.array_equal(z["w.data"], np.full(7, 1.25)) and z["field_names"] == b"u\nw\n"
