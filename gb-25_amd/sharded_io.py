"""Per-rank state dump and offline re-assembly: the role of GB-25 src/sharded_io.jl:70-138,146-213
(`save_model_state` after each loop of the benchmark script, `load_all_fields` offline).  Each rank writes only
its own slab -- no communication -- as `fields_rank{R}.npz` holding, per field, the local interior array, its
slice in the global array and the global shape, plus iteration and time."""
import glob
import os
import re

import numpy as np


def save_model_state(directory, model, rank=0, nranks=1, label="checkpoint"):
    """save_model_state(dir, model, arch; label) -- src/sharded_io.jl:122-138.  Returns the file path.
    The HIP library writes the file itself (gb25_save_state, one ABI call, no Python in the data path); a backend
    without that entry point (the test-side oracle) is dumped from here in the same format."""
    b = model.backend
    if hasattr(b, "save_state"):
        return b.save_state(directory, label)
    outdir = os.path.join(directory, label)
    os.makedirs(outdir, exist_ok=True)
    payload = {"iteration": np.int64(model.clock.iteration), "time": np.float64(model.clock.time),
               "rank": np.int64(rank), "nranks": np.int64(nranks)}
    names = []
    for name, field in model.fields().items():
        a = field.interior
        nx = a.shape[0]
        payload[f"{name}.data"] = a
        payload[f"{name}.slice"] = np.array([rank * nx, (rank + 1) * nx, 0, a.shape[1], 0, a.shape[2]], np.int64)
        payload[f"{name}.global_shape"] = np.array([nx * nranks, a.shape[1], a.shape[2]], np.int64)
        names.append(name)
    payload["field_names"] = np.array(names)
    path = os.path.join(outdir, f"fields_rank{rank}.npz")
    np.savez(path, **payload)
    return path


def _field_names(z):
    names = z["field_names"]
    if isinstance(names, (bytes, bytearray)):          # written by the library: a text member, one name per line
        return names.decode().split()
    return [str(n) for n in names]


def load_global_field(directory, name):
    """load_global_field -- src/sharded_io.jl:146-196: assemble one field from every rank file in `directory`."""
    files = sorted(glob.glob(os.path.join(directory, "fields_rank*.npz")),
                   key=lambda f: int(re.search(r"rank(\d+)", f).group(1)))
    if not files:
        raise FileNotFoundError(f"no fields_rank*.npz in {directory}")
    out = None
    for f in files:
        z = np.load(f)
        if out is None:
            out = np.full(tuple(z[f"{name}.global_shape"]), np.nan, dtype=z[f"{name}.data"].dtype)
        i0, i1, j0, j1, k0, k1 = z[f"{name}.slice"]
        out[i0:i1, j0:j1, k0:k1] = z[f"{name}.data"]
    return out


def load_all_fields(directory):
    """load_all_fields(dir) -- src/sharded_io.jl:198-213: dict name -> global array, plus iteration and time."""
    first = np.load(sorted(glob.glob(os.path.join(directory, "fields_rank*.npz")))[0])
    out = {n: load_global_field(directory, n) for n in _field_names(first)}
    out["iteration"], out["time"] = int(first["iteration"]), float(first["time"])
    return out
