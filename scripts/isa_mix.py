#!/usr/bin/env python3
"""Tuning aid: static instruction mix per kernel of the gfx950 ISA of artp_capi.hip (hipcc -S).
usage: scripts/isa_mix.py [kernel-name-substring ...]"""
import re, subprocess, sys, os, collections
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(root, "gpurun_out", "isa")
os.makedirs(out, exist_ok=True)
asm = os.path.join(out, "artp.s")
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC",
                "-Wno-unused-function", "-Wno-unused-command-line-argument", "-S", "--cuda-device-only", "-o", asm,
                os.path.join(root, "art_planner_amd/csrc/artp_capi.hip")] + [a for a in sys.argv[1:] if a.startswith("-D")], check=True)
want = [a for a in sys.argv[1:] if not a.startswith("-D")] or ["classify_states", "feet_stream", "resolve_boxes", "sample_states"]
L = open(asm).read().split("\n")
for i, l in enumerate(L):
    m = re.match(r"^(_Z\w+):", l)
    if not m or not any(w in m.group(1) for w in want):
        continue
    c = collections.Counter()
    n = 0
    for t in L[i + 1:]:
        t = t.strip()
        if t.startswith("s_endpgm"):
            break
        if not t or t[0] in ".;/" or t.endswith(":"):
            continue
        op = t.split()[0]
        n += 1
        c["valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "lds" if op.startswith("ds_") else op] += 1
        if op in ("v_lshl_add_u64", "v_mad_u64_u32", "v_mov_b32_e32", "v_ashrrev_i32_e32"):
            c[op] += 1
    name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
    print("%-70s %5d %s" % (name[:70], n, dict(c)))
