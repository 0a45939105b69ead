"""CLI flags of the reference's run scripts (GB-25 src/arg_parsing.jl:9-46,54-82): `--grid-x/-y/-z` are PER-DEVICE
totals including halos (interior Nx = grid-x * Rx - 2H, sharding/sharded_..._run.jl:82-88); `--float-type` selects
the model float type.  Float32 (BASELINE.json's configurations) and Float64 (the reference's default) have a HIP library each
(libgb25hip.so / libgb25hip_f64.so, one source built twice); Float16 / BFloat16 are parsed as the reference parses them and
rejected at model construction.  The multifloat flags are Reactant features and are accepted but unused."""
import argparse

_FLOAT_TYPES = {"Float64": "f64", "f64": "f64", "Float32": "f32", "f32": "f32", "Float16": "f16", "f16": "f16",
                "BFloat16": "bf16", "bf16": "bf16"}


def parse_baroclinic_instability_args(argv=None, *, grid_x_default, grid_y_default, grid_z_default):
    """Returns a dict with the reference's keys "grid-x", "grid-y", "grid-z", "float-type", ..."""
    p = argparse.ArgumentParser()
    p.add_argument("--grid-x", type=int, default=grid_x_default, help="grid points per device on the x axis (halos included)")
    p.add_argument("--grid-y", type=int, default=grid_y_default, help="grid points per device on the y axis (halos included)")
    p.add_argument("--grid-z", type=int, default=grid_z_default, help="vertical levels")
    p.add_argument("--float-type", type=str, default="Float64", help="Float64/f64, Float32/f32, Float16/f16, BFloat16/bf16")
    p.add_argument("--target-float-type", type=str, default="")
    p.add_argument("--limbs", type=int, default=2)
    p.add_argument("--dimension", type=str, default="first")
    a = p.parse_args(argv)
    return {"grid-x": a.grid_x, "grid-y": a.grid_y, "grid-z": a.grid_z, "float-type": a.float_type,
            "target-float-type": a.target_float_type, "limbs": a.limbs, "dimension": a.dimension}


def float_type_from_string(s):
    if s not in _FLOAT_TYPES:
        raise AssertionError(f"Unknown float type {s}")
    return _FLOAT_TYPES[s]


def float_type_from_args(parsed_args):
    """float_type_from_args(parsed_args) -- src/arg_parsing.jl:79-82; returns "f64" / "f32" / "f16" / "bf16"."""
    return float_type_from_string(parsed_args["float-type"])


def multifloat_from_args(parsed_args):
    """Reactant's multifloat lowering has no counterpart on this path: always None (src/arg_parsing.jl:100-107)."""
    return None


def interior_size(parsed_args, Rx=1, Ry=1, H=8):
    """(Nx, Ny, Nz) from per-device totals: Tx = grid-x * Rx, Nx = Tx - 2H (sharded_..._run.jl:82-88)."""
    return parsed_args["grid-x"] * Rx - 2 * H, parsed_args["grid-y"] * Ry - 2 * H, parsed_args["grid-z"]
