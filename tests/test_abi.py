"""CPU: the C-ABI library loads here (no GPU needed to dlopen it) and exports exactly the
symbols include/b3gs_raster.h declares; sizes/argument checks that do not touch the device."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    h = open(os.path.join(ROOT, "include", "b3gs_raster.h")).read()
    h = re.sub(r"/\*.*?\*/", "", h, flags=re.S)
    return sorted(set(re.findall(r"\b(b3gs_[a-z0-9_]+)\s*\(", h)) - {"b3gs_alloc_fn"})


def test_library_exports_every_declared_symbol():
    from binocular3dgs_amd import _lib
    from binocular3dgs_amd.build import build
    build()
    L = C.CDLL(_lib.LIB_PATH)
    declared = _declared()
    assert declared, "header parse failed"
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/b3gs_raster.h but not exported"
    assert sorted(_lib.EXPORTS) == declared, "binding list and header disagree"
    assert _lib.lib().b3gs_abi_version() == _lib.ABI_VERSION


def test_scene_struct_layout_matches_header():
    from binocular3dgs_amd import _lib
    # 10 x 4-byte scalars then 11 pointers
    assert C.sizeof(_lib.B3gsScene) == 10 * 4 + 11 * 8
    assert _lib.B3gsScene.background.offset == 40 and _lib.B3gsScene.campos.offset == 40 + 10 * 8


def test_buffer_sizes_are_monotone_and_aligned():
    from binocular3dgs_amd import _lib
    L = _lib.lib()
    prev = 0
    for P in (0, 1, 1000, 10 ** 6):
        b = L.b3gs_geometry_bytes(P)
        assert b % 256 == 0 and b >= prev
        prev = b
    assert L.b3gs_image_bytes(800, 600) % 256 == 0
    assert L.b3gs_binning_bytes(1000, 10 ** 7) > L.b3gs_binning_bytes(1000, 10 ** 6) >= 4 * 4 * 10 ** 6
    # 1M Gaussians: geometry state stays well below 200 B per Gaussian
    assert L.b3gs_geometry_bytes(10 ** 6) < 200 * 10 ** 6


def test_argument_errors_are_reported_not_thrown():
    from binocular3dgs_amd import _lib
    L = _lib.lib()
    sc = _lib.B3gsScene()          # all zero: W = H = 0
    n = C.c_int32(0)
    cb = _lib.ALLOC_FN(lambda u, b: None)
    rc = L.b3gs_forward(C.byref(sc), cb, None, cb, None, cb, None, None, None, None, None, C.byref(n), None)
    assert rc == -1 and b"W/H" in L.b3gs_last_error()
    with pytest.raises(_lib.B3gsError):
        _lib.check(rc, "b3gs_forward")


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "binocular3dgs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "libtile_ref" not in src, f


def test_argument_errors_are_reported_without_touching_a_device():
    """Every entry point validates its arguments before the first HIP call and returns B3GS_ERR_ARG (-1):
    checkable on a machine without a GPU."""
    from binocular3dgs_amd import _lib
    L = _lib.lib()
    ERR_ARG = -1
    assert L.b3gs_binocular_loss(None, None) == ERR_ARG
    assert L.b3gs_binocular_loss_batch(0, None, None) == ERR_ARG
    io = _lib.B3gsLossIO()
    io.W, io.H = 16, 16                                   # all image pointers NULL
    assert L.b3gs_binocular_loss_batch(1, C.byref(io), None) == ERR_ARG
    assert L.b3gs_binocular_loss_batch(9, C.byref(io), None) == ERR_ARG
    assert L.b3gs_densify_classify(None, None, None) == ERR_ARG
    dio = _lib.B3gsDensifyIO()
    dio.P, dio.M = 5, 4                                   # parameters missing
    assert L.b3gs_densify_classify(C.byref(dio), None, None) == ERR_ARG
    assert L.b3gs_knn_mean_dist2(-1, None, None, None, None) == ERR_ARG
    assert L.b3gs_knn_mean_dist2(10, None, None, None, None) == ERR_ARG
    assert L.b3gs_knn_mean_dist2(0, None, None, None, None) == 0          # empty set: nothing to do
    assert L.b3gs_adam_step(0, None, None, 0.9, 0.999, 1e-15, 0.0, -1, 0, 1, None, None, None) == ERR_ARG
    assert L.b3gs_forward_raw_batch(0, None, None, 3, None) == ERR_ARG
    assert L.b3gs_forward_raw_batch(9, None, None, 3, None) == ERR_ARG
    assert L.b3gs_backward_raw_accumulate_range(0, None, None, None, 1, None, 0, 0, None) == ERR_ARG
    sc = _lib.B3gsScene()
    sc.P, sc.W, sc.H = 1 << 24, 64, 64                    # too many Gaussians for the 24-bit row offsets
    assert L.b3gs_forward_capacity(C.byref(sc), None, None, 0, None, None, None, None, None, None, None) == ERR_ARG
    assert b"2^24" in L.b3gs_last_error()
    assert L.b3gs_knn_workspace_bytes(1000) > 0 and L.b3gs_loss_workspace_floats(800, 600) == 8 * 64 + 9 * 800 * 600
