#!/usr/bin/env python3
"""Instruction mix of one kernel of the library (cross-compiles to gfx950 assembly; no GPU needed).
usage: kernel_isa.py <substring of the mangled kernel name> [float|double]"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gb-25_amd", "csrc", "gb25_api.hip")
pat, t = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "float")
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-amdgpu-use-amdgpu-trackers",
                    "-Wno-unused-value", "-Wno-pass-failed", f"-DGB25_REAL={t}", "--cuda-device-only", "-S", SRC,
                    "-o", out], check=True, capture_output=True)
    lines = open(out).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l]
for st in start:
    end = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
    body = [l.split()[0] for l in lines[st + 1:end] if re.match(r"^\s+[a-z]", l)]
    c = collections.Counter(body)
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    pk = sum(v for k, v in c.items() if k.startswith("v_pk_"))
    print(lines[st][:90])
    print(f"  instructions {len(body)}  VALU {valu} (packed {pk})  VMEM {sum(v for k, v in c.items() if k.startswith(('global_', 'buffer_', 'scratch_')))}"
          f"  LDS {sum(v for k, v in c.items() if k.startswith('ds_'))}  SALU {sum(v for k, v in c.items() if k.startswith('s_'))}")
    print("  " + "  ".join(f"{k}:{v}" for k, v in c.most_common(18)))
