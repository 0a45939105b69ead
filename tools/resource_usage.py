"""Print VGPR / spill / LDS / occupancy per kernel for the Float32 and Float64 builds (cross-compiles, no GPU)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gb-25_amd", "csrc", "gb25_api.hip")
for t in sys.argv[1:] or ["float", "double"]:
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-amdgpu-use-amdgpu-trackers",
                            "-Wno-unused-value", "-Wno-pass-failed", f"-DGB25_REAL={t}", "--cuda-device-only", "-c",
                            "-Rpass-analysis=kernel-resource-usage", SRC, "-o", os.path.join(d, "o.o")],
                           capture_output=True, text=True)
    txt = r.stderr
    names = re.findall(r"Function Name: (\S+)", txt)
    vg = re.findall(r" VGPRs: (\d+)", txt)
    sp = re.findall(r"VGPRs Spill: (\d+)", txt)
    lds = re.findall(r"LDS Size \[bytes/block\]: (\d+)", txt)
    occ = re.findall(r"Occupancy \[waves/SIMD\]: (\d+)", txt)
    print(t)
    for n, v, s, l, o in zip(names, vg, sp, lds, occ):
        n = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().split("(")[0]
        print(f"  {n[:86]:86s} vgpr {v:>4} spill {s:>4} lds {l:>7} occ {o}")
