"""CPU-side checks of the C-ABI boundary: the library builds for gfx950, loads, and exports every symbol
include/gb25.h declares (no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

import gb25_amd as gb
from gb25_amd import binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", params=["Float32", "Float64"])
def lib(request):
    gb.build_library()
    return binding.load_library(request.param)


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gb25.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gb25_[a-z_0-9]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert declared_symbols() == sorted(binding.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_config_struct_layout_matches_header(lib):
    cfg = binding.Config()
    lib.gb25_default_config(ctypes.byref(cfg), 1440, 720, 48)
    assert (cfg.Nx, cfg.Ny, cfg.Nz, cfg.halo, cfg.substeps, cfg.nranks) == (1440, 720, 48, 8, 30, 1)
    assert (cfg.lat_south, cfg.lat_north, cfg.depth, cfg.zexp_h) == (-80.0, 80.0, 4000.0, 30.0)
    assert (cfg.g, cfg.Omega, cfg.radius, cfg.rho0, cfg.chi) == (9.80665, 7.292115e-5, 6371e3, 1020.0, 0.1)
    assert lib.gb25_version().decode().startswith("gb25hip")
    # the library states the size of its structs: a binding whose mirror has another size refuses to load it
    assert lib.gb25_config_bytes() == ctypes.sizeof(binding.Config) and cfg.ranks_y == 1
    assert lib.gb25_catke_parameters_bytes() == ctypes.sizeof(binding.CatkeParameters)


def test_no_cpu_fallback_without_device(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(binding.GB25Error, match="no HIP device"):
        gb.baroclinic_instability_model(gb.GPU(), 32, 16, 8, dt=1.0)


def test_element_size_matches_the_float_type(lib):
    assert lib.gb25_real_bytes() == (8 if b"Float64" in lib.gb25_version() else 4)
    assert binding.load_library("Float32").gb25_real_bytes() == 4
    assert binding.load_library("Float64").gb25_real_bytes() == 8
    with pytest.raises(binding.GB25Error):
        binding.load_library("Float16")


@pytest.mark.parametrize("float_type", ["Float32", "Float64"])
def test_code_object_targets_gfx950(float_type):
    blob = open(binding.LIB_PATHS[float_type], "rb").read()
    assert b"gfx950" in blob and b"k_gu" in blob
