"""Guards for the software-managed MFMA hazards of gfx950 (cost_kernels.h FCM_SHAPE_CHANGE).

CPU: scripts/mfma_hazard_check.py over the assembly hipcc produces for the MFMA kernels -- no dependent MFMA of the other
shape closer than 5 wait states, no VALU read / write of an MFMA's destination closer than 7 / 4 -- and the checker itself on
hand-written snippets.  GPU: tests/cpp/mfma_hazard_probe.hip measures those minima on the device, so a part or a toolchain
that needs more than the checker assumes is caught."""
import os
import re
import shutil
import subprocess
import sys

import pytest

import common

sys.path.insert(0, os.path.join(common.ROOT, "scripts"))
import mfma_hazard_check as H  # noqa: E402

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
needs_hipcc = pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")


def _check_snippet(tmp_path, body):
    f = tmp_path / "k.s"
    f.write_text("_Z6kernelv:\n" + body + "\n\ts_endpgm\n")
    return H.check_file(str(f), ["kernel"])


def test_checker_flags_a_dependent_mfma_of_the_other_shape(tmp_path):
    bad = """\tv_mfma_f32_16x16x32_f16 v[8:11], v[0:3], v[4:7], v[8:11]
\ts_nop 2
\tv_mfma_f32_16x16x16_f16 v[8:11], v[0:1], v[4:5], v[8:11]"""
    (_, r, viol), = _check_snippet(tmp_path, bad)
    assert r["mix"] == 3 and len(viol) == 1 and viol[0].startswith("mix 3 < 5")
    good = bad.replace("s_nop 2", "s_nop 4")
    (_, r, viol), = _check_snippet(tmp_path, good)
    assert r["mix"] == 5 and not viol
    # the same shape is interlocked by the hardware (probe B2): back to back is fine
    same = """\tv_mfma_f32_16x16x32_f16 v[8:11], v[0:3], v[4:7], v[8:11]
\tv_mfma_f32_16x16x32_f16 v[8:11], v[0:3], v[4:7], v[8:11]"""
    (_, r, viol), = _check_snippet(tmp_path, same)
    assert r["mix"] is None and not viol
    # the accumulator may move to other registers in between (dst != srcC): the dependency is on the registers
    moved = """\tv_mfma_f32_16x16x16_f16 v[12:15], v[0:1], v[4:5], v[8:11]
\tv_mfma_f32_16x16x32_f16 v[16:19], v[0:3], v[4:7], v[12:15]"""
    (_, r, viol), = _check_snippet(tmp_path, moved)
    assert r["mix"] == 0 and viol


def test_checker_flags_early_valu_use_of_the_destination(tmp_path):
    rd = """\tv_mfma_f32_16x16x32_f16 v[8:11], v[0:3], v[4:7], v[8:11]
\ts_nop 4
\tv_add_f32_e32 v20, v8, v21"""
    (_, r, viol), = _check_snippet(tmp_path, rd)
    assert r["rd"] == 5 and any(v.startswith("rd 5 < 7") for v in viol)
    (_, r, viol), = _check_snippet(tmp_path, rd.replace("s_nop 4", "s_nop 6"))
    assert r["rd"] == 7 and not viol
    wr = """\tv_mfma_f32_16x16x32_f16 v[8:11], v[0:3], v[4:7], v[8:11]
\ts_nop 1
\tv_mov_b32_e32 v9, 0"""
    (_, r, viol), = _check_snippet(tmp_path, wr)
    assert r["wr"] == 2 and any(v.startswith("wr 2 < 4") for v in viol)
    # overwriting a srcC register that is not the destination is harmless (probe A)
    war = """\tv_mfma_f32_16x16x32_f16 v[8:11], v[0:3], v[4:7], v[12:15]
\tv_mov_b32_e32 v12, 0"""
    (_, r, viol), = _check_snippet(tmp_path, war)
    assert r["srcc_war"] == 0 and not viol


@needs_hipcc
def test_mfma_kernels_keep_the_measured_distances(tmp_path):
    """Every MFMA kernel of the product as hipcc compiles it today: no violation of the measured minima."""
    asm = H.compile_asm(str(tmp_path / "artp.s"))
    res = H.check_file(asm, H.DEFAULT_KERNELS)
    names = " ".join(n for n, _, _ in res)
    for k in H.DEFAULT_KERNELS:
        assert k in names, f"{k} not found in the assembly"
    assert not any(k in open(asm).read() for k in H.VARIANT_KERNELS)   # the lost forms are not in the product
    assert sum(r["mfma"] for _, r, _ in res) > 2000
    bad = [(n, v) for n, _, viol in res for v in viol]
    assert not bad, bad
    # the cost-query MLP is the one kernel that mixes shapes: it must show the guarded distance
    fc = [r for n, r, _ in res if "fc_cost_mfma_kernel" in n]
    assert fc and all(r["shapes"] == {"16x16x16", "16x16x32"} and r["mix"] is not None and r["mix"] >= H.MIN_MIX for r in fc)
    # the convolutions use one shape only
    assert all(r["shapes"] == {"16x16x32"} and r["mix"] is None for n, r, _ in res if "conv" in n)


@needs_hipcc
def test_variant_mfma_kernels_keep_the_measured_distances(tmp_path):
    """The same for the variants build (-DARTP_VARIANTS: conv1 o conv2 as its own launch, the persistent 15 x 15 kernel, the
    15 x 15 layer on v_mfma_f32_32x32x16_f16): one shape per kernel, no violation."""
    asm = H.compile_asm(str(tmp_path / "artp_variants.s"), defs=("-DARTP_VARIANTS",))
    res = H.check_file(asm, H.VARIANT_KERNELS)
    names = " ".join(n for n, _, _ in res)
    for k in H.VARIANT_KERNELS:
        assert k in names, f"{k} not found in the assembly"
    bad = [(n, v) for n, _, viol in res for v in viol]
    assert not bad, bad
    for n, r, _ in res:
        assert r["mix"] is None and r["shapes"] == ({"32x32x16"} if "conv15_pair32" in n else {"16x16x32"}), (n, r["shapes"])


@pytest.mark.gpu
@needs_hipcc
def test_device_needs_no_more_wait_states_than_the_checker_assumes(tmp_path):
    exe = str(tmp_path / "mfma_hazard_probe")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O1", "-std=c++17", "-o", exe,
                           os.path.join(common.ROOT, "tests", "cpp", "mfma_hazard_probe.hip")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    os.makedirs(os.path.join(common.ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(common.ROOT, "gpurun_out", "mfma_hazard_probe.txt"), "w") as f:
        f.write(out.stdout)
    need = {}
    for line in out.stdout.splitlines():
        m = re.match(r"probe (\S+) .*pre_mfma=(\d).*min_safe=(\d+)", line)
        if m:
            need[m.group(1)] = max(need.get(m.group(1), 0), int(m.group(3)))
    assert set(need) == {"A", "B", "B3", "B2", "C", "E"}, out.stdout
    assert need["A"] == 0                      # a VALU write to a dead srcC register: no hazard (round 4's theory)
    assert need["B2"] == 0                     # same-shape accumulation is interlocked
    assert 1 <= max(need["B"], need["B3"]) <= H.MIN_MIX   # the other shape is not -- the reason for FCM_SHAPE_CHANGE
    assert need["C"] <= H.MIN_RD and need["E"] <= H.MIN_WR


@pytest.mark.gpu
@needs_hipcc
def test_what_the_matrix_cores_sustain(tmp_path):
    """tests/cpp/mfma_clock_probe.hip: every SIMD issues independent v_mfma_f32_16x16x32_f16 with no memory traffic.  The figures
    the CNN's roofline is read against: two wavefronts per SIMD reach ~17 cycles per MFMA at the clock the part keeps
    under that load (~2.0 GHz, not the 2.4 GHz of the nominal 2.5 PFLOP/s); ONE wavefront per SIMD issues only every ~26 cycles --
    why a workgroup alone on a CU cannot fill the pipe.  The bounds are wide: this pins the mechanism, not a number."""
    import json
    exe = str(tmp_path / "mfma_clock_probe")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O2", "-o", exe,
                           os.path.join(common.ROOT, "tests", "cpp", "mfma_clock_probe.hip")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    os.makedirs(os.path.join(common.ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(common.ROOT, "gpurun_out", "mfma_clock_probe.txt"), "w") as f:
        f.write(out.stdout)
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert 15.5 <= r["two_per_simd_cycles_per_mfma"] <= 20.0, r          # the pipe takes one per 16 cycles at best
    assert r["one_per_simd_cycles_per_mfma"] >= 1.2 * r["two_per_simd_cycles_per_mfma"], r   # a lone wavefront cannot fill it
    assert 0.5 * r["nominal_dense_f16_tflops"] <= r["two_per_simd_tflops"] <= 1.02 * r["nominal_dense_f16_tflops"], r
