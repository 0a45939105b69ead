#!/usr/bin/env python3
"""Instruction mix of one kernel of the library (cross-compiles to gfx950 assembly; no GPU needed): the whole kernel and its
main loop (the backward branch with the largest span: the march over the levels of a chunk), by bucket.
usage: kernel_isa.py <substring of the mangled kernel name> [float|double]
The mangled name of the headline momentum instance: k_momentum_tendencies_v5ILi4ELi4ELb1ELb0ELb0ELb1ELb0ELb1E"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gb-25_amd", "csrc", "gb25_api.hip")
if len(sys.argv) < 2:
    sys.exit(__doc__)
pat, t = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "float")

BUCKETS = [
    ("packed fp arithmetic (v_pk_fma/mul/add)", lambda k: k.startswith("v_pk_")),
    ("scalar fp arithmetic (fma/fmac/mul/add/sub/mad)", lambda k: re.match(r"v_(fma|fmac|mul|add|sub|subrev|mad)_f(32|64)", k) is not None),
    ("reciprocal / sqrt / rsq (two issue slots each)", lambda k: re.match(r"v_(rcp|sqrt|rsq)_", k) is not None),
    ("min / max / abs-free clamps", lambda k: re.match(r"v_(min|max|med3)", k) is not None),
    ("compares", lambda k: k.startswith("v_cmp")),
    ("selects (v_cndmask)", lambda k: k.startswith("v_cndmask")),
    ("moves (v_mov, v_accvgpr, v_swap)", lambda k: re.match(r"v_(mov|accvgpr|swap)", k) is not None),
    ("lane reads / writes (SGPR spills, shuffles: v_readlane, v_writelane, v_readfirstlane, ds_bpermute, dpp moves)",
     lambda k: re.match(r"v_(readlane|writelane|readfirstlane|permlane)|ds_bpermute|ds_permute", k) is not None),
    ("integer / address VALU (v_add_u32, v_lshl, v_mad_u, v_and, ...)", lambda k: k.startswith("v_")),
    ("LDS (ds_read / ds_write)", lambda k: k.startswith("ds_")),
    ("global / buffer memory", lambda k: k.startswith(("global_", "buffer_", "flat_"))),
    ("scratch (VGPR spills)", lambda k: k.startswith("scratch_")),
    ("waits and barriers (s_waitcnt, s_barrier, s_nop)", lambda k: re.match(r"s_(waitcnt|barrier|nop|sleep|setprio)", k) is not None),
    ("scalar ALU / branches", lambda k: k.startswith("s_")),
]


def mix(ops):
    left = collections.Counter(ops)
    rows = []
    for name, f in BUCKETS:
        n = sum(v for k, v in left.items() if f(k))
        for k in [k for k in left if f(k)]:
            del left[k]
        rows.append((name, n))
    return rows, left


with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-mllvm", "-amdgpu-use-amdgpu-trackers",
                    "-Wno-unused-value", "-Wno-pass-failed", f"-DGB25_REAL={t}", "--cuda-device-only", "-S", SRC,
                    "-o", out], check=True, capture_output=True)
    lines = open(out).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and pat in l]
for st in start:
    end = next(i for i in range(st, len(lines)) if "s_endpgm" in lines[i])
    body = lines[st + 1:end]
    ops = [(n, l.split()[0]) for n, l in enumerate(body) if re.match(r"^\s+[a-z]", l)]
    labels = {l.split(":")[0]: n for n, l in enumerate(body) if re.match(r"^\.LBB\S+:", l)}
    loops = []
    for n, l in enumerate(body):
        m = re.match(r"^\s+s_cbranch\S*\s+(\.LBB\S+)|^\s+s_branch\s+(\.LBB\S+)", l)
        if m:
            tgt = labels.get(m.group(1) or m.group(2))
            if tgt is not None and tgt < n:
                loops.append((n - tgt, tgt, n))
    print(lines[st][:120])
    whole, _ = mix([o for _, o in ops])
    total = len(ops)
    print(f"  whole kernel: {total} instructions")
    if loops:
        span, a, b = max(loops)
        inner = [o for n, o in ops if a <= n <= b]
        rows, _ = mix(inner)
        nvalu = sum(n for (name, n) in rows[:9])
        print(f"  main loop (lines {a}..{b}): {len(inner)} instructions, {nvalu} of them VALU")
        for (name, n), (_, nw) in zip(rows, whole):
            print(f"    {n:6d}  ({100.0 * n / max(1, len(inner)):5.1f} %)   whole kernel {nw:6d}   {name}")
        c = collections.Counter(inner)
        print("    top opcodes: " + "  ".join(f"{k}:{v}" for k, v in c.most_common(24)))
