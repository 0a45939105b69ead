#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) into
per-kernel HBM bytes per launch, with the gfx950 FETCH_SIZE corrections calibrated on kernels of known byte count.
usage: pmc_traffic.py <fetch> <write> <out.json> Nx Ny Nz
<fetch>/<write>: a rocprofv3 CSV output directory, or a per-kernel summary CSV written by tools/rocpd_summary.py"""
import collections, csv, glob, json, os, sys

def load(d, ctr):
    agg = collections.defaultdict(list)
    if os.path.isfile(d):     # summary of a rocpd database: Kernel,Counter,Launches,MeanValue,...
        for r in csv.DictReader(open(d)):
            if r["Counter"] == ctr:
                agg[r["Kernel"]].append(float(r["MeanValue"]) * 1024)
    else:
        f = glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True)[0]
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                name = r["Kernel_Name"].replace("void ", "").split("(")[0].replace("gb25::", "")
                agg[name].append(float(r["Counter_Value"]) * 1024)
    return {k: sum(v) / len(v) for k, v in agg.items()}

fetch_dir, write_dir, out, Nx, Ny, Nz = sys.argv[1], sys.argv[2], sys.argv[3], *map(int, sys.argv[4:7])
F, W = load(fetch_dir, "FETCH_SIZE"), load(write_dir, "WRITE_SIZE")
H = 8
plane = (Nx + 2 * H) * (Ny + 2 * H) * 4
t4 = [k for k in F if k.startswith("k_ab2_tracers4")]
c16 = 6 * Nz * plane / F[t4[0]] if t4 else 2.0       # float4 stream of known size (guide: exactly 2)
c4 = 6 * Nx * Ny * Nz * 4 / F["k_ab2_velocities"]   # scalar row accesses of known size
res = {}
for k in F:
    corr = c16 if k.startswith("k_ab2_tracers4") else c4
    res[k] = {"fetch_bytes_raw": F[k], "fetch_bytes_corrected": F[k] * corr, "write_bytes": W.get(k, 0.0),
              "hbm_bytes": F[k] * corr + W.get(k, 0.0)}
json.dump({"workload": f"{Nx}x{Ny}x{Nz}", "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes",
           "fetch_correction": {"16B_per_lane": c16, "4B_per_lane": c4,
                                "method": "calibrated on k_ab2_tracers4 (float4 stream) and k_ab2_velocities (scalar rows), "
                                          "both of known byte count; WRITE_SIZE needs none"},
           "kernels": res}, open(out, "w"), indent=1)
cells = Nx * Ny * Nz
for k, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes"])[:12]:
    print(f"{k[:44]:44s} {v['hbm_bytes'] / 1e6:9.1f} MB/launch {v['hbm_bytes'] / cells:6.1f} B/cell")
print("corrections", c16, c4)
