"""ctypes bindings for the CPU oracle (oracle/liboracle.so) and, where it was built, the real
patched ODE (oracle/_ref/libartp_ref.so).  TEST INFRASTRUCTURE: imported by tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() only -- never by art_planner_amd/."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "liboracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libartp_ref.so")


class Field(C.Structure):
    _fields_ = [("nW", C.c_int), ("nD", C.c_int),
                ("width", C.c_float), ("depth", C.c_float), ("half_w", C.c_float), ("half_d", C.c_float),
                ("sample_w", C.c_float), ("sample_d", C.c_float), ("inv_w", C.c_float),
                ("inv_d", C.c_float), ("zx_aspect", C.c_float),
                ("pos", C.c_float * 3), ("R", C.c_float * 12), ("data", C.c_void_p)]


class Robot(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("torso_length", "torso_width", "torso_height", "torso_off_x", "torso_off_y",
                 "torso_off_z", "feet_off_x", "feet_off_y", "feet_off_z", "reach_x", "reach_y",
                 "reach_z")] + [("unknown_space_untraversable", C.c_int),
                                ("max_pitch_pert", C.c_double), ("max_roll_pert", C.c_double)]

    @property
    def torso(self):
        return np.array([self.torso_length, self.torso_width, self.torso_height], np.float32)

    @property
    def foot(self):
        return np.array([self.reach_x, self.reach_y, self.reach_z], np.float32)


class Map(C.Structure):
    _fields_ = [("len_x", C.c_double), ("len_y", C.c_double), ("pos_x", C.c_double),
                ("pos_y", C.c_double), ("res", C.c_double), ("rows", C.c_int), ("cols", C.c_int),
                ("body", Field), ("feet", Field)]


class SamplerMap(C.Structure):
    _fields_ = [("rows", C.c_int), ("cols", C.c_int), ("len_x", C.c_double), ("len_y", C.c_double),
                ("pos_x", C.c_double), ("pos_y", C.c_double), ("res", C.c_double),
                ("cum_prob", C.c_void_p), ("cum_prob_rowwise", C.c_void_p), ("elevation", C.c_void_p),
                ("normal_x", C.c_void_p), ("normal_y", C.c_void_p), ("normal_z", C.c_void_p),
                ("plane_fit_std_dev", C.c_void_p), ("sample_uniform", C.c_int)]


def build_oracle(force: bool = False) -> str:
    src = os.path.join(ORACLE_DIR, "artp_oracle.c")
    if force or not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "liboracle.so"])
    return ORACLE_SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build_oracle())
        L.artp_oracle_field_init.argtypes = [C.POINTER(Field), C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                             C.c_double, C.c_double, C.c_double, C.c_double]
        L.artp_oracle_check_boxes.argtypes = [C.POINTER(Field), C.c_void_p, C.c_void_p, C.c_size_t,
                                              C.c_void_p, C.c_void_p, C.c_void_p]
        L.artp_oracle_check_boxes.restype = C.c_int
        L.artp_oracle_robot_defaults.argtypes = [C.POINTER(Robot)]
        L.artp_oracle_robot_yaml.argtypes = [C.POINTER(Robot)]
        L.artp_oracle_state_poses.argtypes = [C.POINTER(Map), C.POINTER(Robot), C.c_void_p, C.c_void_p,
                                              C.c_void_p]
        L.artp_oracle_state_valid.argtypes = [C.POINTER(Map), C.POINTER(Robot), C.c_void_p, C.c_void_p]
        L.artp_oracle_state_valid.restype = C.c_int
        L.artp_oracle_states_valid.argtypes = [C.POINTER(Map), C.POINTER(Robot), C.c_void_p, C.c_size_t,
                                               C.c_void_p, C.c_void_p]
        L.artp_oracle_uniform01.argtypes = [C.c_uint64, C.c_uint64, C.c_uint]
        L.artp_oracle_uniform01.restype = C.c_double
        L.artp_oracle_samples.argtypes = [C.POINTER(SamplerMap), C.POINTER(Robot), C.c_uint64,
                                          C.c_uint64, C.c_size_t, C.c_void_p, C.c_void_p]
        L.artp_oracle_interpolate.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p]
        L.artp_oracle_valid_segment_count.argtypes = [C.POINTER(Map), C.c_double, C.c_void_p, C.c_void_p]
        L.artp_oracle_valid_segment_count.restype = C.c_uint
        L.artp_oracle_check_motion.argtypes = [C.POINTER(Map), C.POINTER(Robot), C.c_double, C.c_void_p,
                                               C.c_void_p, C.c_void_p]
        L.artp_oracle_check_motion.restype = C.c_int
        L.artp_oracle_check_motion_last_valid.argtypes = [C.POINTER(Map), C.POINTER(Robot), C.c_double, C.c_void_p,
                                                          C.c_void_p, C.c_void_p, C.c_void_p]
        L.artp_oracle_check_motion_last_valid.restype = C.c_int
        L.artp_oracle_edge_interp_valid.argtypes = [C.POINTER(Map), C.POINTER(Robot), C.c_void_p,
                                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint]
        L.artp_oracle_edge_interp_valid.restype = C.c_int
        _lib = L
    return _lib


def robot(kind: str = "yaml") -> Robot:
    r = Robot()
    (lib().artp_oracle_robot_yaml if kind == "yaml" else lib().artp_oracle_robot_defaults)(C.byref(r))
    return r


def _f32F(a):
    a = np.asfortranarray(a, dtype=np.float32)
    return a


class OracleField:
    """One HeightMapBoxChecker's heightfield (keeps the storage alive)."""

    def __init__(self, layer, len_x, len_y, pos_x=0.0, pos_y=0.0):
        layer = _f32F(layer)
        self.rows, self.cols = layer.shape
        self.storage = np.empty(layer.size, np.float32)
        self.f = Field()
        lib().artp_oracle_field_init(C.byref(self.f), self.storage.ctypes.data, layer.ctypes.data,
                                     self.rows, self.cols, len_x, len_y, pos_x, pos_y)

    def check_boxes(self, side, poses, want_detail=False):
        side = np.ascontiguousarray(side, np.float32)
        poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
        n = poses.shape[0]
        hit = np.empty(n, np.uint8)
        ec = np.empty(n, np.uint8)
        nv = np.empty(n, np.uint32)
        lib().artp_oracle_check_boxes(C.byref(self.f), side.ctypes.data, poses.ctypes.data, n,
                                      hit.ctypes.data, ec.ctypes.data, nv.ctypes.data)
        return (hit, ec, nv) if want_detail else hit


class OracleMap:
    """Map + both heightfields, as StateValidityChecker sees them."""

    def __init__(self, gm, body_layer="elevation", feet_layer="elevation_masked"):
        self.gm = gm
        self.body = OracleField(gm[body_layer], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y)
        self.feet = OracleField(gm[feet_layer], gm.len_x, gm.len_y, gm.pos_x, gm.pos_y)
        self.m = Map(gm.len_x, gm.len_y, gm.pos_x, gm.pos_y, gm.res, gm.rows, gm.cols,
                     self.body.f, self.feet.f)
        elev = gm[body_layer]
        fin = elev[np.isfinite(elev)]
        self._zminmax = (float(fin.min()), float(fin.max())) if fin.size else (0.0, 0.0)

    def z_extent(self, rob: Robot) -> float:
        lo = self._zminmax[0] - rob.reach_z / 2
        hi = self._zminmax[1] + rob.reach_z / 2
        return hi - lo

    def state_poses(self, rob, se3):
        se3 = np.ascontiguousarray(se3, np.float64).reshape(-1, 7)
        poses = np.empty((se3.shape[0], 5, 16), np.float32)
        inside = np.empty((se3.shape[0], 5), np.int32)
        for i in range(se3.shape[0]):
            lib().artp_oracle_state_poses(C.byref(self.m), C.byref(rob), se3[i].ctypes.data,
                                          poses[i].ctypes.data, inside[i].ctypes.data)
        return poses, inside

    def states_valid(self, rob, se3, want_vertices=False):
        se3 = np.ascontiguousarray(se3, np.float64).reshape(-1, 7)
        valid = np.empty(se3.shape[0], np.uint8)
        verts = C.c_uint64(0)
        lib().artp_oracle_states_valid(C.byref(self.m), C.byref(rob), se3.ctypes.data, se3.shape[0],
                                       valid.ctypes.data, C.byref(verts) if want_vertices else None)
        return (valid, verts.value) if want_vertices else valid

    def state_detail(self, rob, se3):
        se3 = np.ascontiguousarray(se3, np.float64).reshape(7)
        d = np.empty(6, np.int32)
        v = lib().artp_oracle_state_valid(C.byref(self.m), C.byref(rob), se3.ctypes.data, d.ctypes.data)
        return v, d

    def check_motions(self, rob, s1, s2):
        s1 = np.ascontiguousarray(s1, np.float64).reshape(-1, 7)
        s2 = np.ascontiguousarray(s2, np.float64).reshape(-1, 7)
        out = np.empty(s1.shape[0], np.uint8)
        nchk = np.empty(s1.shape[0], np.uint32)
        ze = self.z_extent(rob)
        c = C.c_uint(0)
        for i in range(s1.shape[0]):
            out[i] = lib().artp_oracle_check_motion(C.byref(self.m), C.byref(rob), ze, s1[i].ctypes.data,
                                                    s2[i].ctypes.data, C.byref(c))
            nchk[i] = c.value
        return out, nchk

    def check_motions_last_valid(self, rob, s1, s2):
        """DiscreteMotionValidator::checkMotion(s1, s2, lastValid): (valid, t, state); t = 1 / state = s2 where
        the motion is valid (the convention of artp_check_motions_last_valid)."""
        s1 = np.ascontiguousarray(s1, np.float64).reshape(-1, 7)
        s2 = np.ascontiguousarray(s2, np.float64).reshape(-1, 7)
        n = s1.shape[0]
        out = np.empty(n, np.uint8)
        t = np.ones(n, np.float64)
        st = s2.copy()
        ze = self.z_extent(rob)
        tt = C.c_double(0)
        for i in range(n):
            out[i] = lib().artp_oracle_check_motion_last_valid(C.byref(self.m), C.byref(rob), ze, s1[i].ctypes.data,
                                                               s2[i].ctypes.data, C.byref(tt), st[i].ctypes.data)
            if not out[i]:
                t[i] = tt.value
        return out, t, st

    def segment_counts(self, rob, s1, s2):
        s1 = np.ascontiguousarray(s1, np.float64).reshape(-1, 7)
        s2 = np.ascontiguousarray(s2, np.float64).reshape(-1, 7)
        ze = self.z_extent(rob)
        return np.array([lib().artp_oracle_valid_segment_count(C.byref(self.m), ze, s1[i].ctypes.data,
                                                               s2[i].ctypes.data)
                         for i in range(s1.shape[0])], np.uint32)

    def edges_interp_valid(self, rob, s1, s2):
        s1 = np.ascontiguousarray(s1, np.float64).reshape(-1, 7)
        s2 = np.ascontiguousarray(s2, np.float64).reshape(-1, 7)
        out = np.empty(s1.shape[0], np.uint8)
        nint = np.empty(s1.shape[0], np.uint32)
        c = C.c_uint(0)
        for i in range(s1.shape[0]):
            out[i] = lib().artp_oracle_edge_interp_valid(C.byref(self.m), C.byref(rob), s1[i].ctypes.data,
                                                         s2[i].ctypes.data, C.byref(c), None, 0)
            nint[i] = c.value
        return out, nint


class OracleSampler:
    def __init__(self, gm, elevation_layer="elevation", sample_uniform=False):
        self.gm = gm
        self.keep = [_f32F(gm["cum_prob"]), np.ascontiguousarray(gm["cum_prob_rowwise"], np.float32),
                     _f32F(gm[elevation_layer]), _f32F(gm["normal_x"]), _f32F(gm["normal_y"]),
                     _f32F(gm["normal_z"]), _f32F(gm["plane_fit_std_dev"])]
        self.m = SamplerMap(gm.rows, gm.cols, gm.len_x, gm.len_y, gm.pos_x, gm.pos_y, gm.res,
                            *[a.ctypes.data for a in self.keep], 1 if sample_uniform else 0)

    def sample(self, rob, seed, first, n):
        se3 = np.empty((n, 7), np.float64)
        rc = np.empty((n, 2), np.int32)
        lib().artp_oracle_samples(C.byref(self.m), C.byref(rob), seed, first, n, se3.ctypes.data,
                                  rc.ctypes.data)
        return se3, rc


def interpolate(a, b, t):
    a = np.ascontiguousarray(a, np.float64)
    b = np.ascontiguousarray(b, np.float64)
    out = np.empty(7, np.float64)
    lib().artp_oracle_interpolate(a.ctypes.data, b.ctypes.data, t, out.ctypes.data)
    return out


# ----- the real patched ODE (only where oracle/_ref was built) -----------------------------------
def have_ref() -> bool:
    return os.path.exists(REF_SO)


_ref = None


def ref_lib():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        L.artp_ref_create.restype = C.c_void_p
        L.artp_ref_create.argtypes = [C.c_float] * 3
        L.artp_ref_destroy.argtypes = [C.c_void_p]
        L.artp_ref_set_field.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int] + [C.c_double] * 4
        L.artp_ref_check.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.artp_ref_check.restype = C.c_int
        _ref = L
    return _ref


class RefChecker:
    """art_planner::HeightMapBoxChecker on the real ODE."""

    def __init__(self, side, layer, len_x, len_y, pos_x=0.0, pos_y=0.0):
        self.layer = _f32F(layer)
        self.h = ref_lib().artp_ref_create(float(side[0]), float(side[1]), float(side[2]))
        ref_lib().artp_ref_set_field(self.h, self.layer.ctypes.data, self.layer.shape[0],
                                     self.layer.shape[1], len_x, len_y, pos_x, pos_y)

    def check(self, poses):
        poses = np.ascontiguousarray(poses, np.float32).reshape(-1, 16)
        hit = np.empty(poses.shape[0], np.uint8)
        ref_lib().artp_ref_check(self.h, poses.ctypes.data, poses.shape[0], hit.ctypes.data)
        return hit

    def close(self):
        if self.h:
            ref_lib().artp_ref_destroy(self.h)
            self.h = None
