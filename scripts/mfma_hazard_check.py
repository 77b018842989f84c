#!/usr/bin/env python3
"""ISA-level guard for the gfx950 MFMA srcC hazard (cost_kernels.h FCM_WAIT).

ROCm 7.2's compiler pads the "VALU writes a register an in-flight MFMA still reads as srcC" hazard of
v_mfma_f32_16x16x32_f16 (8 passes on gfx950) as if the instruction had fewer passes: measured on the device,
fewer than 8 wait states between the MFMA and the overwriting VALU instruction give wrong sums.  This script compiles
artp_capi.hip to gfx950 assembly and, for every MFMA of the named kernels, walks the straight-line code behind it and
reports the smallest number of wait states before
  (a) a VALU (or LDS-return / SALU-to-VGPR) instruction WRITES one of the MFMA's srcC registers that is not also its
      destination (write-after-read on srcC: the round-4 bug), and
  (b) a non-MFMA VALU instruction reads or writes the MFMA's DESTINATION registers (the ordinary result hazard; reported,
      the compiler's own table covers it).
A wait state = one issued instruction; `s_nop N` = N + 1; a branch / label ends the walk (conservative: the walk also
follows fall-through labels).  usage: mfma_hazard_check.py [--asm FILE] [--min N] kernel-substring ...
exit status 1 if any (a) distance is below --min (default 8)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_KERNELS = ["fc_cost_mfma_kernel", "conv345_kernel", "conv_ksplit_kernel", "conv_kwalk_kernel", "conv345p_kernel"]
WINDOW = 24  # wait states looked at behind an MFMA


def compile_asm(path, defs=()):
    os.makedirs(os.path.dirname(path), exist_ok=True)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                    "-fPIC", "-Wno-unused-function", "-Wno-unused-command-line-argument", "-S", "--cuda-device-only", "-o", path,
                    os.path.join(ROOT, "art_planner_amd/csrc/artp_capi.hip")] + list(defs), check=True)
    return path


def regs_of(tok):
    """'v[10:13]' -> {10..13}; 'v7' -> {7}; anything else (sgpr, literal, acc) -> empty"""
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    return set()


def operands(line):
    body = line.split(";")[0].strip()
    parts = body.split(None, 1)
    if len(parts) < 2:
        return parts[0], []
    return parts[0], [t.strip() for t in parts[1].split(",")]


def vgpr_writes(op, ops):
    """registers an instruction writes (first operand for VALU / LDS reads / global loads are NOT counted: their data
    returns hundreds of cycles later and is ordered by s_waitcnt, not by wait states)"""
    if op.startswith("v_") and not op.startswith("v_cmp") and not op.startswith("v_cmpx"):
        w = regs_of(ops[0]) if ops else set()
        if op.startswith("v_readlane") or op.startswith("v_readfirstlane"):
            return set()
        return w
    return set()


def vgpr_reads(op, ops):
    if not op.startswith("v_"):
        return set()
    r = set()
    for t in ops[1:]:
        r |= regs_of(t)
    return r


def wait_states(op, ops):
    if op == "s_nop":
        return int(ops[0], 0) + 1 if ops else 1
    return 1


def kernels_of(asm_lines, wanted):
    i = 0
    n = len(asm_lines)
    while i < n:
        m = re.match(r"^(_Z\w+):", asm_lines[i])
        if m and any(w in m.group(1) for w in wanted):
            j = i + 1
            while j < n and not asm_lines[j].strip().startswith("s_endpgm"):
                j += 1
            yield m.group(1), asm_lines[i + 1:j + 1]
            i = j
        i += 1


def check_kernel(body):
    ins = []
    for l in body:
        t = l.strip()
        if not t or t[0] in ".;/":
            continue
        if t.endswith(":") or re.match(r"^\.?\w+:", t):
            ins.append(("label", []))
            continue
        ins.append(operands(t))
    res = {"mfma": 0, "min_srcc_war": None, "min_dst": None, "worst": None}
    for i, (op, ops) in enumerate(ins):
        if not op.startswith("v_mfma"):
            continue
        res["mfma"] += 1
        dst, srcc = regs_of(ops[0]), regs_of(ops[3]) if len(ops) > 3 else set()
        war = srcc - dst
        ws = 0
        for j in range(i + 1, len(ins)):
            o2, p2 = ins[j]
            if o2 == "label":
                continue
            if o2.startswith("s_cbranch") or o2.startswith("s_branch") or o2 == "s_endpgm" or o2 == "s_barrier":
                break
            if not o2.startswith("v_mfma"):
                w = vgpr_writes(o2, p2)
                if war and (w & war):
                    if res["min_srcc_war"] is None or ws < res["min_srcc_war"]:
                        res["min_srcc_war"] = ws
                        res["worst"] = (" ".join([op] + ops), " ".join([o2] + p2))
                    war = war - w
                if (w | vgpr_reads(o2, p2)) & dst:
                    if res["min_dst"] is None or ws < res["min_dst"]:
                        res["min_dst"] = ws
            ws += wait_states(o2, p2)
            if ws >= WINDOW:
                break
    return res


def main(argv):
    asm = None
    minimum = 8
    names = []
    it = iter(argv)
    for a in it:
        if a == "--asm":
            asm = next(it)
        elif a == "--min":
            minimum = int(next(it))
        else:
            names.append(a)
    names = names or DEFAULT_KERNELS
    if asm is None:
        asm = compile_asm(os.path.join(ROOT, "gpurun_out", "isa", "artp.s"))
    lines = open(asm).read().split("\n")
    bad = 0
    for name, body in kernels_of(lines, names):
        r = check_kernel(body)
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        flag = ""
        if r["min_srcc_war"] is not None and r["min_srcc_war"] < minimum:
            bad += 1
            flag = "  <-- srcC overwritten too early: " + " | ".join(r["worst"])
        print(f"{short[:90]:90s} mfma {r['mfma']:4d}  min wait states: srcC-WAR {r['min_srcc_war']}  dst-use {r['min_dst']}{flag}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
