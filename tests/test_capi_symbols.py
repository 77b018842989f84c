"""CPU suite: the C-ABI library loads and exports every symbol include/artp_c.h declares (no compute
calls without a GPU), and refuses to create a context without a device."""
import ctypes as C
import os
import re

import pytest

import common
from art_planner_amd import _capi


def _declared_symbols():
    src = open(os.path.join(common.ROOT, "include", "artp_c.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(artp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    import subprocess
    # always through make: a no-op when libartp.so is current, a rebuild when a header changed
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(_capi.LIB_PATH)])
    L = _capi.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(L, name), f"libartp.so does not export {name}"
    assert sorted(_capi.SYMBOLS) == declared


def test_params_presets_match_reference_values():
    L = _capi.load()
    p = _capi.Params()
    L.artp_params_defaults(C.byref(p))
    assert (p.torso_length, p.torso_width, p.torso_height) == (1.05, 0.55, 0.2)   # params.h:93-95
    assert (p.feet_off_x, p.feet_off_y, p.feet_off_z) == (0.362, 0.225, -0.525)   # params.h:107-111
    assert (p.reach_x, p.reach_y, p.reach_z) == (0.25, 0.1, 0.15)
    L.artp_params_yaml(C.byref(p))
    assert (p.torso_length, p.torso_width, p.torso_height) == (1.31, 0.65, 0.3)   # params.yaml:57-59
    assert (p.feet_off_x, p.feet_off_y, p.feet_off_z) == (0.51, 0.2, -0.475)
    assert p.torso_off_z == 0.04 and p.unknown_space_untraversable == 1


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L = _capi.load()
    p = _capi.Params()
    L.artp_params_yaml(C.byref(p))
    h = C.c_void_p()
    rc = L.artp_create(0, C.byref(p), C.byref(h))
    assert rc == -2 and not h.value   # ARTP_ERR_NO_DEVICE: the product never computes on the CPU
    with pytest.raises(_capi.ArtpError):
        from art_planner_amd.context import Context
        Context(0, "yaml")
