#!/usr/bin/env python3
"""ISA-level guard for the gfx950 MFMA hazards the compiler leaves to the author (cost_kernels.h FCM_SHAPE_CHANGE).

Measured on the device by tests/cpp/mfma_hazard_probe.hip (profiles/r05_mfma_hazard_probe.txt), wait states needed behind a
v_mfma_f32_16x16x32_f16 / 16x16x16_f16:
  MIX  a dependent MFMA of the OTHER shape (its srcC overlaps the first one's destination)   >= 5   hipcc emits 0-1
  RD   a VALU instruction reads the destination                                             >= 7   hipcc pads it (s_nop 7)
  WR   a VALU instruction overwrites the destination                                        >= 4   hipcc pads it
  (a VALU write to a srcC register that is not the destination needs none: probe A -- round 4's theory, refuted)
This script compiles artp_capi.hip to gfx950 assembly and, for every MFMA of the named kernels, walks the straight-line
code behind it and reports the smallest distance of each kind; it fails (exit 1) when one is below its measured minimum.
A wait state = one issued instruction; `s_nop N` = N + 1; a branch / barrier ends the walk (labels are walked through).
usage: mfma_hazard_check.py [--asm FILE] kernel-substring ..."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_KERNELS = ["fc_cost_mfma_kernel", "conv345_kernel", "conv_ksplit_kernel"]              # libartp.so
VARIANT_KERNELS = ["conv12_mfma_kernel", "conv_kwalk_kernel", "conv15_pair32_kernel"]             # + with -DARTP_VARIANTS
MIN_MIX, MIN_RD, MIN_WR = 5, 7, 4
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


def mfma_shape(op):
    m = re.search(r"_(\d+x\d+x\d+)_?", op)
    return m.group(1) if m else op


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
    res = {"mfma": 0, "shapes": set(), "mix": None, "rd": None, "wr": None, "srcc_war": None, "worst": {}}

    def note(kind, ws, a, b):
        if res[kind] is None or ws < res[kind]:
            res[kind] = ws
            res["worst"][kind] = (" ".join([a[0]] + a[1]), " ".join([b[0]] + b[1]))

    for i, (op, ops) in enumerate(ins):
        if not op.startswith("v_mfma"):
            continue
        res["mfma"] += 1
        res["shapes"].add(mfma_shape(op))
        dst, srcc = regs_of(ops[0]), regs_of(ops[3]) if len(ops) > 3 else set()
        war = srcc - dst
        live = set(dst)   # destination registers not yet rewritten by something later
        ws = 0
        for j in range(i + 1, len(ins)):
            o2, p2 = ins[j]
            if o2 == "label":
                continue
            if o2.startswith("s_cbranch") or o2.startswith("s_branch") or o2 == "s_endpgm" or o2 == "s_barrier":
                break
            if o2.startswith("v_mfma"):
                c2 = regs_of(p2[3]) if len(p2) > 3 else set()
                if (c2 & live) and mfma_shape(o2) != mfma_shape(op):
                    note("mix", ws, (op, ops), (o2, p2))
                live -= regs_of(p2[0])
            else:
                w = vgpr_writes(o2, p2)
                r = vgpr_reads(o2, p2)
                if war and (w & war):
                    note("srcc_war", ws, (op, ops), (o2, p2))
                    war = war - w
                if r & live:
                    note("rd", ws, (op, ops), (o2, p2))
                if w & live:
                    note("wr", ws, (op, ops), (o2, p2))
                    live -= w
            if not live and not war:
                break
            ws += wait_states(o2, p2)
            if ws >= WINDOW:
                break
    return res


def check_file(asm, names):
    """[(kernel, result dict, [violations])] for the kernels of an assembly file whose mangled name holds one of `names`"""
    lines = open(asm).read().split("\n")
    out = []
    for name, body in kernels_of(lines, names):
        r = check_kernel(body)
        bad = []
        for kind, lim in (("mix", MIN_MIX), ("rd", MIN_RD), ("wr", MIN_WR)):
            if r[kind] is not None and r[kind] < lim:
                bad.append(f"{kind} {r[kind]} < {lim}: " + " | ".join(r["worst"][kind]))
        out.append((name, r, bad))
    return out


def main(argv):
    asm = None
    names = []
    it = iter(argv)
    for a in it:
        if a == "--asm":
            asm = next(it)
        else:
            names.append(a)
    names = names or DEFAULT_KERNELS
    if asm is None:
        asm = compile_asm(os.path.join(ROOT, "gpurun_out", "isa", "artp.s"))
    nbad = 0
    for name, r, bad in check_file(asm, names):
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
        print(f"{short[:84]:84s} mfma {r['mfma']:4d} {'+'.join(sorted(r['shapes'])):17s} min wait states: other-shape acc {r['mix']}  "
              f"VALU reads dst {r['rd']}  VALU writes dst {r['wr']}  (srcC overwritten {r['srcc_war']}: harmless)")
        for b in bad:
            nbad += 1
            print("    VIOLATION " + b)
    return 1 if nbad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
