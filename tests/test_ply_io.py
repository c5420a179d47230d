"""CPU: PLY interchange (binocular3dgs_amd/init_points.py) -- header/property ORDER as the reference writes it
(scene/gaussian_model.py:177-208), channel-major feature flattening, binary and ASCII reading, round trip."""
import os

import numpy as np
import pytest
import torch


def _model(P=7, deg=2):
    from binocular3dgs_amd.gaussian_model import GaussianModel
    g = torch.Generator().manual_seed(0)
    K = (deg + 1) ** 2
    r = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    return GaussianModel.from_tensors(r(P, 3), r(P, 1, 3), r(P, K - 1, 3), r(P, 3), r(P, 4), r(P, 1), sh_degree=deg, device="cpu")


def test_header_and_layout_match_the_reference_writer(tmp_path):
    from binocular3dgs_amd.init_points import save_ply
    m = _model()
    path = str(tmp_path / "point_cloud" / "iteration_7" / "point_cloud.ply")
    save_ply(m, path)
    raw = open(path, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().strip().split("\n")
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 7"]
    props = [ln.split()[-1] for ln in lines[3:]]
    assert all(ln.startswith("property float ") for ln in lines[3:])
    assert props == (["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"] + [f"f_rest_{i}" for i in range(24)] +
                     ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"])
    rows = np.frombuffer(body, dtype="<f4").reshape(7, len(props))
    np.testing.assert_array_equal(rows[:, 0:3], m._xyz.detach().numpy())
    np.testing.assert_array_equal(rows[:, 3:6], 0)
    np.testing.assert_array_equal(rows[:, 6:9], m._features_dc.detach().numpy()[:, 0, :])
    # f_rest_{c*8 + k} = coefficient k (1-based in SH order) of channel c
    fr = m._features_rest.detach().numpy()           # [P, 8, 3]
    for c in range(3):
        for k in range(8):
            np.testing.assert_array_equal(rows[:, 9 + c * 8 + k], fr[:, k, c])
    np.testing.assert_array_equal(rows[:, 33], m._opacity.detach().numpy()[:, 0])


def test_round_trip_and_degree_check(tmp_path):
    from binocular3dgs_amd.init_points import load_ply, save_ply
    m = _model(P=11, deg=1)
    path = str(tmp_path / "pc.ply")
    save_ply(m, path)
    m2 = load_ply(path, sh_degree=1, device="cpu")
    for a in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
        assert torch.equal(getattr(m, a).detach(), getattr(m2, a).detach()), a
    assert m2.active_sh_degree == 1 and m2._features_rest.shape == (11, 3, 3)
    with pytest.raises(ValueError):
        load_ply(path, sh_degree=3, device="cpu")


def test_reads_ascii_and_uchar_colours(tmp_path):
    from binocular3dgs_amd.init_points import fetch_point_cloud
    path = str(tmp_path / "points3D.ply")
    open(path, "w").write("ply\nformat ascii 1.0\ncomment made by hand\nelement vertex 2\nproperty float x\nproperty float y\n"
                          "property float z\nproperty float nx\nproperty float ny\nproperty float nz\nproperty uchar red\n"
                          "property uchar green\nproperty uchar blue\nend_header\n0.5 1 -2 0 0 0 255 0 51\n1 2 3 0 0 0 0 102 255\n")
    pts, rgb = fetch_point_cloud(path)
    np.testing.assert_allclose(pts, [[0.5, 1, -2], [1, 2, 3]])
    np.testing.assert_allclose(rgb, [[1.0, 0.0, 0.2], [0.0, 0.4, 1.0]])
    # the same through a binary file with mixed property types
    dt = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("red", "u1"), ("green", "u1"), ("blue", "u1")])
    arr = np.array([(0.5, 1, -2, 255, 0, 51)], dtype=dt)
    p2 = str(tmp_path / "b.ply")
    with open(p2, "wb") as f:
        f.write(b"ply\nformat binary_little_endian 1.0\nelement vertex 1\nproperty float x\nproperty float y\nproperty float z\n"
                b"property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
        f.write(arr.tobytes())
    pts, rgb = fetch_point_cloud(p2)
    np.testing.assert_allclose(pts, [[0.5, 1, -2]])
    np.testing.assert_allclose(rgb, [[1.0, 0.0, 0.2]])
