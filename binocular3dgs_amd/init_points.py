"""Initialisation of a Gaussian set from a point cloud (scene/gaussian_model.py:124-147 create_from_pcd) and the
PLY interchange format of trained models (save_ply / load_ply, :177-256) -- SURVEY 8f-4.

`knn_mean_dist2` is the reference's simple_knn._C.distCUDA2 (exact 3-nearest-neighbour mean squared distance) on
the HIP device through the C ABI.  The PLY reader / writer is plain numpy (plyfile is not a dependency): binary
little-endian or ASCII, `vertex` element, property names and ORDER exactly as the reference writes them, so
point_cloud.ply files of reference-trained models load here and vice versa.
"""
from __future__ import annotations

import os
from typing import Dict, List, Tuple

import numpy as np
import torch

from . import _lib

SH_C0 = 0.28209479177387814


def knn_mean_dist2(points: torch.Tensor) -> torch.Tensor:
    """[P,3] device tensor -> [P] mean squared distance to the 3 nearest other points."""
    if not points.is_cuda:
        raise _lib.B3gsError("knn_mean_dist2 needs a device tensor (no CPU fallback)")
    L = _lib.lib()
    pts = points.detach().float().contiguous()
    P = pts.shape[0]
    out = torch.empty(P, dtype=torch.float32, device=pts.device)
    ws = torch.empty(max(L.b3gs_knn_workspace_bytes(P), 1), dtype=torch.uint8, device=pts.device)
    rc = L.b3gs_knn_mean_dist2(P, pts.data_ptr(), out.data_ptr(), ws.data_ptr(), torch.cuda.current_stream(pts.device).cuda_stream)
    _lib.check(rc, "b3gs_knn_mean_dist2")
    return out


def rgb_to_sh(rgb):
    """utils/sh_utils.py:114-115"""
    return (rgb - 0.5) / SH_C0


def create_from_points(points, colors, sh_degree: int, device="cuda"):
    """GaussianModel as create_from_pcd builds it: SH DC from the colours, isotropic scales
    log(sqrt(max(dist2, 1e-7))), identity rotations, opacity logit(0.1)."""
    from .gaussian_model import GaussianModel, inverse_sigmoid
    xyz = torch.as_tensor(np.asarray(points), dtype=torch.float32).to(device)
    col = rgb_to_sh(torch.as_tensor(np.asarray(colors), dtype=torch.float32).to(device))
    P, K = xyz.shape[0], (sh_degree + 1) ** 2
    feats = torch.zeros((P, 3, K), device=device)
    feats[:, :3, 0] = col
    dist2 = torch.clamp_min(knn_mean_dist2(xyz), 1e-7)
    scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
    rots = torch.zeros((P, 4), device=device)
    rots[:, 0] = 1
    opac = inverse_sigmoid(0.1 * torch.ones((P, 1), device=device))
    m = GaussianModel.from_tensors(xyz, feats[:, :, 0:1].transpose(1, 2).contiguous(), feats[:, :, 1:].transpose(1, 2).contiguous(),
                                   scales, rots, opac, sh_degree=sh_degree, active_sh_degree=0, device=device)
    m.max_radii2D = torch.zeros((P,), device=device)
    return m


# ---- PLY ------------------------------------------------------------------------------------------------
_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2",
              "uint16": "u2", "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4",
              "double": "f8", "float64": "f8"}


def read_ply_vertices(path: str) -> np.ndarray:
    """Structured array of the `vertex` element (binary_little_endian, binary_big_endian or ascii)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError(f"{path}: not a PLY file")
        fmt, props, count, in_vertex, elements = None, [], 0, False, []
        while True:
            line = f.readline()
            if not line:
                raise ValueError(f"{path}: truncated header")
            tok = line.decode("ascii", "replace").split()
            if not tok or tok[0] == "comment":
                continue
            if tok[0] == "format":
                fmt = tok[1]
            elif tok[0] == "element":
                in_vertex = tok[1] == "vertex"
                elements.append(tok[1])
                if in_vertex:
                    count = int(tok[2])
                    if len(elements) != 1:
                        raise ValueError(f"{path}: `vertex` must be the first element")
            elif tok[0] == "property" and in_vertex:
                if tok[1] == "list":
                    raise ValueError(f"{path}: list properties in `vertex` are not supported")
                props.append((tok[2], _PLY_TYPES[tok[1]]))
            elif tok[0] == "end_header":
                break
        if fmt == "ascii":
            data = np.loadtxt(f, max_rows=count, ndmin=2) if count else np.zeros((0, len(props)))
            out = np.zeros(count, dtype=[(n, t) for n, t in props])
            for k, (n, _t) in enumerate(props):
                out[n] = data[:, k]
            return out
        order = "<" if fmt == "binary_little_endian" else ">"
        dt = np.dtype([(n, order + t) for n, t in props])
        return np.frombuffer(f.read(count * dt.itemsize), dtype=dt, count=count)


def write_ply_vertices(path: str, names: List[str], columns: np.ndarray) -> None:
    """Binary little-endian PLY, one float32 property per column of `columns` [P, len(names)]."""
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    columns = np.ascontiguousarray(columns, dtype="<f4")
    assert columns.ndim == 2 and columns.shape[1] == len(names)
    head = ["ply", "format binary_little_endian 1.0", f"element vertex {columns.shape[0]}"]
    head += [f"property float {n}" for n in names] + ["end_header"]
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode("ascii"))
        f.write(columns.tobytes())


def attribute_names(n_dc: int, n_rest: int, n_scale: int = 3, n_rot: int = 4) -> List[str]:
    """scene/gaussian_model.py:177-190"""
    return (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(n_dc)] + [f"f_rest_{i}" for i in range(n_rest)] +
            ["opacity"] + [f"scale_{i}" for i in range(n_scale)] + [f"rot_{i}" for i in range(n_rot)])


def save_ply(model, path: str) -> None:
    """scene/gaussian_model.py:192-208: features stored CHANNEL-major ([P,K,3] -> transpose -> [P,3,K] -> flatten)."""
    xyz = model._xyz.detach().cpu().numpy()
    f_dc = model._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    f_rest = model._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous().cpu().numpy()
    cols = np.concatenate((xyz, np.zeros_like(xyz), f_dc, f_rest, model._opacity.detach().cpu().numpy(),
                           model._scaling.detach().cpu().numpy(), model._rotation.detach().cpu().numpy()), axis=1)
    write_ply_vertices(path, attribute_names(f_dc.shape[1], f_rest.shape[1]), cols)


def load_ply(path: str, sh_degree: int, device="cuda"):
    """scene/gaussian_model.py:215-256 (active_sh_degree = max_sh_degree afterwards)."""
    from .gaussian_model import GaussianModel
    v = read_ply_vertices(path)
    names = v.dtype.names
    col = lambda ns: np.stack([np.asarray(v[n], dtype=np.float32) for n in ns], axis=1)  # noqa: E731
    by_index = lambda prefix: sorted([n for n in names if n.startswith(prefix)], key=lambda n: int(n.split("_")[-1]))  # noqa: E731
    xyz = col(["x", "y", "z"])
    P = xyz.shape[0]
    f_dc = col(["f_dc_0", "f_dc_1", "f_dc_2"]).reshape(P, 3, 1)
    rest_names = by_index("f_rest_")
    K = (sh_degree + 1) ** 2
    if len(rest_names) != 3 * K - 3:
        raise ValueError(f"{path}: {len(rest_names)} f_rest_* properties, sh_degree {sh_degree} needs {3 * K - 3}")
    f_rest = (col(rest_names) if rest_names else np.zeros((P, 0), np.float32)).reshape(P, 3, K - 1)
    t = lambda a: torch.tensor(a, dtype=torch.float32)  # noqa: E731
    return GaussianModel.from_tensors(t(xyz), t(f_dc).transpose(1, 2).contiguous(), t(f_rest).transpose(1, 2).contiguous(),
                                      t(col(by_index("scale_"))), t(col(by_index("rot"))), t(col(["opacity"])),
                                      sh_degree=sh_degree, active_sh_degree=sh_degree, device=device)


def fetch_point_cloud(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """scene/dataset_readers.py fetchPly: positions and colours/255 of an input point cloud (x y z [nx ny nz] red green blue)."""
    v = read_ply_vertices(path)
    pts = np.stack([v["x"], v["y"], v["z"]], axis=1).astype(np.float32)
    rgb = np.stack([v["red"], v["green"], v["blue"]], axis=1).astype(np.float32) / 255.0
    return pts, rgb
