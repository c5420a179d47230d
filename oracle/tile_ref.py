"""ctypes front-end of oracle/tile_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
It restates the rasterizer the reference reaches through
gaussian_renderer/__init__.py:85-93 (forward) and train.py:149 (backward); see the header of
tile_ref.c for the parity statement ("parity unpinned").
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libtile_ref.so")
    src = os.path.join(_HERE, "tile_ref.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libtile_ref.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        assert _LIB.orc_sizeof_scene() == C.sizeof(_Scene), "OrcScene layout drifted"
        assert _LIB.orc_sizeof_geom() == C.sizeof(_Geom), "OrcGeom layout drifted"
    return _LIB


_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_up = C.POINTER(C.c_uint32)
_bp = C.POINTER(C.c_uint8)


class _Scene(C.Structure):
    _fields_ = [("P", C.c_int), ("D", C.c_int), ("M", C.c_int), ("W", C.c_int), ("H", C.c_int),
                ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
                ("prefiltered", C.c_int),
                ("bg", _fp), ("means3D", _fp), ("shs", _fp), ("colors_precomp", _fp), ("opacities", _fp),
                ("scales", _fp), ("rotations", _fp), ("cov3D_precomp", _fp), ("viewmatrix", _fp),
                ("projmatrix", _fp), ("campos", _fp)]


class _Geom(C.Structure):
    _fields_ = [("depths", _fp), ("radii", _ip), ("means2D", _fp), ("cov3D", _fp), ("conic_opacity", _fp),
                ("rgb", _fp), ("clamped", _bp), ("rect", _ip), ("tiles_touched", _up), ("point_offsets", _up)]


class _PixGrads(C.Structure):
    _fields_ = [("dL_dmean2D", _dp), ("dL_dconic", _dp), ("dL_dopacity", _dp), ("dL_dcolor", _dp),
                ("dL_ddepth", _dp)]


class _Grads(C.Structure):
    _fields_ = [("dL_dmeans3D", _fp), ("dL_dcov3D", _fp), ("dL_dsh", _fp), ("dL_dcolors", _fp),
                ("dL_dscales", _fp), ("dL_drots", _fp), ("dL_dopacity", _fp), ("dL_dmeans2D", _fp)]


def _f32(a):
    return None if a is None else np.ascontiguousarray(np.asarray(a, dtype=np.float32))


def _ptr(a, t):
    return C.cast(None, t) if a is None else a.ctypes.data_as(t)


class State:
    """Everything the forward produced; the backward needs all of it."""


def forward(*, means3D, opacities, viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy, sh_degree=0,
            shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None, scale_modifier=1.0,
            prefiltered=False, threads=None) -> State:
    L = lib()
    if threads is not None:
        os.environ["OMP_NUM_THREADS"] = str(int(threads))
    st = State()
    st.means3D = _f32(means3D).reshape(-1, 3)
    P = st.means3D.shape[0]
    st.shs = None if shs is None else _f32(shs).reshape(P, -1, 3)
    st.colors_precomp = None if colors_precomp is None else _f32(colors_precomp).reshape(P, 3)
    st.opacities = _f32(opacities).reshape(P)
    st.scales = None if scales is None else _f32(scales).reshape(P, 3)
    st.rotations = None if rotations is None else _f32(rotations).reshape(P, 4)
    st.cov3D_precomp = None if cov3D_precomp is None else _f32(cov3D_precomp).reshape(P, 6)
    assert (st.shs is None) != (st.colors_precomp is None)
    assert (st.cov3D_precomp is None) != (st.scales is None or st.rotations is None)
    st.viewmatrix = _f32(viewmatrix).reshape(16)
    st.projmatrix = _f32(projmatrix).reshape(16)
    st.campos = _f32(campos).reshape(3)
    st.bg = _f32(bg).reshape(3)
    st.W, st.H, st.P = int(W), int(H), P
    st.M = 0 if st.shs is None else st.shs.shape[1]
    st.D = int(sh_degree)

    sc = _Scene(P, st.D, st.M, st.W, st.H, float(tanfovx), float(tanfovy), float(scale_modifier),
                int(bool(prefiltered)), _ptr(st.bg, _fp), _ptr(st.means3D, _fp), _ptr(st.shs, _fp),
                _ptr(st.colors_precomp, _fp), _ptr(st.opacities, _fp), _ptr(st.scales, _fp),
                _ptr(st.rotations, _fp), _ptr(st.cov3D_precomp, _fp), _ptr(st.viewmatrix, _fp),
                _ptr(st.projmatrix, _fp), _ptr(st.campos, _fp))
    st._scene = sc

    st.depths = np.zeros(P, np.float32)
    st.radii = np.zeros(P, np.int32)
    st.means2D = np.zeros((P, 2), np.float32)
    st.cov3D = np.zeros((P, 6), np.float32)
    st.conic_opacity = np.zeros((P, 4), np.float32)
    st.rgb = np.zeros((P, 3), np.float32)
    st.clamped = np.zeros((P, 3), np.uint8)
    st.rect = np.zeros((P, 4), np.int32)
    st.tiles_touched = np.zeros(P, np.uint32)
    st.point_offsets = np.zeros(P, np.uint32)
    gm = _Geom(_ptr(st.depths, _fp), _ptr(st.radii, _ip), _ptr(st.means2D, _fp), _ptr(st.cov3D, _fp),
               _ptr(st.conic_opacity, _fp), _ptr(st.rgb, _fp), _ptr(st.clamped, _bp), _ptr(st.rect, _ip),
               _ptr(st.tiles_touched, _up), _ptr(st.point_offsets, _up))
    st._geom = gm
    L.orc_preprocess(C.byref(sc), C.byref(gm))

    N = int(st.point_offsets[-1]) if P else 0
    st.N = N
    gx, gy = (st.W + 15) // 16, (st.H + 15) // 16
    st.grid = (gx, gy)
    st.keys = np.zeros(max(N, 1), np.uint64)
    st.point_list = np.zeros(max(N, 1), np.uint32)
    st.ranges = np.zeros((gx * gy, 2), np.uint32)
    L.orc_bin(C.byref(sc), C.byref(gm), st.keys.ctypes.data_as(C.POINTER(C.c_uint64)),
              _ptr(st.point_list, _up), _ptr(st.ranges, _up))
    st.keys, st.point_list = st.keys[:N], st.point_list[:N]
    st._pl = np.ascontiguousarray(st.point_list) if N else np.zeros(1, np.uint32)

    st.color = np.zeros((3, st.H, st.W), np.float32)
    st.depth = np.zeros((1, st.H, st.W), np.float32)
    st.alpha = np.zeros((1, st.H, st.W), np.float32)
    st.final_T = np.zeros((st.H, st.W), np.float32)
    st.n_contrib = np.zeros((st.H, st.W), np.uint32)
    L.orc_render_forward(C.byref(sc), C.byref(gm), _ptr(st._pl, _up), _ptr(st.ranges, _up), _ptr(st.color, _fp),
                         _ptr(st.depth, _fp), _ptr(st.alpha, _fp), _ptr(st.final_T, _fp), _ptr(st.n_contrib, _up))
    return st


def backward(st: State, dL_dcolor, dL_ddepth=None, dL_dalpha=None) -> dict:
    L = lib()
    P = st.P
    dC = _f32(dL_dcolor).reshape(3, st.H, st.W)
    dD = None if dL_ddepth is None else _f32(dL_ddepth).reshape(st.H, st.W)
    dA = None if dL_dalpha is None else _f32(dL_dalpha).reshape(st.H, st.W)
    pg_arr = dict(dL_dmean2D=np.zeros((P, 2)), dL_dconic=np.zeros((P, 3)), dL_dopacity=np.zeros(P),
                  dL_dcolor=np.zeros((P, 3)), dL_ddepth=np.zeros(P))
    pg = _PixGrads(*[_ptr(pg_arr[k], _dp) for k in ("dL_dmean2D", "dL_dconic", "dL_dopacity", "dL_dcolor", "dL_ddepth")])
    L.orc_render_backward(C.byref(st._scene), C.byref(st._geom), _ptr(st._pl, _up), _ptr(st.ranges, _up),
                          _ptr(st.final_T, _fp), _ptr(st.n_contrib, _up), _ptr(dC, _fp), _ptr(dD, _fp), _ptr(dA, _fp),
                          C.byref(pg))
    out = dict(dL_dmeans3D=np.zeros((P, 3), np.float32), dL_dcov3D=np.zeros((P, 6), np.float32),
               dL_dsh=None if st.shs is None else np.zeros((P, st.M, 3), np.float32),
               dL_dcolors=np.zeros((P, 3), np.float32),
               dL_dscales=None if st.scales is None else np.zeros((P, 3), np.float32),
               dL_drotations=None if st.rotations is None else np.zeros((P, 4), np.float32),
               dL_dopacity=np.zeros((P, 1), np.float32), dL_dmeans2D=np.zeros((P, 3), np.float32))
    gr = _Grads(_ptr(out["dL_dmeans3D"], _fp), _ptr(out["dL_dcov3D"], _fp), _ptr(out["dL_dsh"], _fp),
                _ptr(out["dL_dcolors"], _fp), _ptr(out["dL_dscales"], _fp), _ptr(out["dL_drotations"], _fp),
                _ptr(out["dL_dopacity"], _fp), _ptr(out["dL_dmeans2D"], _fp))
    L.orc_preprocess_backward(C.byref(st._scene), C.byref(st._geom), C.byref(pg), C.byref(gr))
    out["pix"] = pg_arr
    return out
