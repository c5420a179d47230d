"""Camera matrices exactly as the reference builds them, plus the binocular (shifted) view.

Mirrors scene/cameras.py:17-83 (Camera / MiniCam attribute names: `world_view_transform`,
`full_proj_transform`, `camera_center`, `FoVx`, `FoVy`, `image_width`, `image_height`),
utils/graphics_utils.py:38-77 (getWorld2View2, getProjectionMatrix, fov2focal) and
scene/__init__.py:96-115 (getShiftedCamera).  Convention: ROW-vector, i.e. the stored matrices
are the transposes of the conventional ones: [x y z 1] @ world_view_transform = view coords.
"""
from __future__ import annotations

import math

import numpy as np
import torch


def world_to_view(R: np.ndarray, t: np.ndarray, translate=np.array([0.0, 0.0, 0.0]), scale: float = 1.0) -> np.ndarray:
    """utils/graphics_utils.py:38-49.  `R` is the camera-to-world rotation (stored transposed in
    the W2C matrix), `t` the W2C translation; returns the conventional 4x4 W2C in float32."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = np.asarray(R, dtype=np.float64).transpose()
    Rt[:3, 3] = np.asarray(t, dtype=np.float64)
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    C2W[:3, 3] = (C2W[:3, 3] + np.asarray(translate, dtype=np.float64)) * scale
    return np.float32(np.linalg.inv(C2W))


def projection_matrix(znear: float, zfar: float, fovX: float, fovY: float) -> torch.Tensor:
    """utils/graphics_utils.py:51-71 (conventional, un-transposed; float32)."""
    tanHalfFovY = math.tan(fovY / 2)
    tanHalfFovX = math.tan(fovX / 2)
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = torch.zeros(4, 4)
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = 1.0
    P[2, 2] = zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def fov2focal(fov: float, pixels: int) -> float:
    return pixels / (2 * math.tan(fov / 2))


def focal2fov(focal: float, pixels: int) -> float:
    return 2 * math.atan(pixels / (2 * focal))


_pinned = {}     # device -> _PinnedRing


class _PinnedRing:
    """Rows of pinned host memory for asynchronous uploads of a few dozen floats.  A row is handed out again `rows` uploads
    later -- by then its copy has normally run; an event recorded behind every copy makes that CERTAIN (a loop
    that never synchronises, e.g. shifted cameras precomputed for a whole dataset behind a busy queue, can run more than 64
    uploads ahead of the stream): a row whose copy is still in flight is waited for, not overwritten."""

    def __init__(self, rows: int, cols: int):
        self.buf = torch.empty((rows, cols), dtype=torch.float32).pin_memory()
        self.events = [None] * rows
        self.next = 0

    def take(self):
        """-> (row index, host row): safe to write"""
        k = self.next % self.buf.shape[0]
        self.next += 1
        ev = self.events[k]
        if ev is not None and not ev.query():
            ev.synchronize()
        return k, self.buf[k]

    def uploaded(self, k: int, device):
        """call right behind the non_blocking copy of row k"""
        ev = self.events[k]
        if ev is None:
            ev = self.events[k] = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(device))


def _staging(device):
    """(host tensor, its numpy view, done): 35 floats the next shifted camera is assembled in.  On a HIP device: a row of a
    pinned ring, so that the upload is an asynchronous copy (a pageable source makes the runtime block the host until the
    stream has drained: ~170 us per iteration of the two-view schedule); `done()` is called behind the copy."""
    device = torch.device(device)
    if device.type != "cuda":
        row = torch.empty(35, dtype=torch.float32)
        return row, row.numpy(), lambda: None
    ring = _pinned.get(device)
    if ring is None:
        ring = _pinned[device] = _PinnedRing(64, 35)
    k, row = ring.take()
    return row, row.numpy(), lambda: ring.uploaded(k, device)


class Camera:
    """Holds what render() and the loss block read from a reference Camera (scene/cameras.py:17-70)."""

    zfar = 100.0
    znear = 0.01

    def __init__(self, R, T, FoVx, FoVy, width, height, image=None, gt_alpha_mask=None, uid=0,
                 trans=np.array([0.0, 0.0, 0.0]), scale=1.0, device="cpu"):
        self.uid = uid
        self.R = np.asarray(R, dtype=np.float64)
        self.T = np.asarray(T, dtype=np.float64)
        self.FoVx = float(FoVx)
        self.FoVy = float(FoVy)
        self.image_width = int(width)
        self.image_height = int(height)
        self.trans = np.asarray(trans, dtype=np.float64)
        self.scale = scale
        self.device = torch.device(device)
        self.original_image = None if image is None else image.clamp(0.0, 1.0).to(self.device)
        self.gt_alpha_mask = None if gt_alpha_mask is None else gt_alpha_mask.to(self.device)
        if self.original_image is not None and self.gt_alpha_mask is not None:
            self.original_image = self.original_image * self.gt_alpha_mask
        wvt = torch.tensor(world_to_view(self.R, self.T, self.trans, scale)).transpose(0, 1)
        proj = projection_matrix(self.znear, self.zfar, self.FoVx, self.FoVy).transpose(0, 1)
        self._set(wvt, proj)

    def _set(self, wvt: torch.Tensor, proj_t: torch.Tensor):
        self.world_view_transform = wvt.contiguous().to(self.device)
        self.projection_matrix = proj_t.contiguous().to(self.device)
        self.full_proj_transform = (self.world_view_transform.unsqueeze(0)
                                    .bmm(self.projection_matrix.unsqueeze(0))).squeeze(0).contiguous()
        self.camera_center = self.world_view_transform.inverse()[3, :3].contiguous()

    def _host_matrices(self):
        """Host copies of this camera's matrices (one device->host read per camera object, cached): what shifted() derives
        the partner from."""
        h = getattr(self, "_host", None)
        if h is None:
            h = self._host = {"wvt": self.world_view_transform.detach().cpu().numpy().copy(),
                              "proj": self.projection_matrix.detach().cpu().numpy().copy(),
                              "full": self.full_proj_transform.detach().cpu().numpy().copy(),
                              "center": self.camera_center.detach().cpu().numpy().copy()}
        return h

    def _ones_image(self):
        o = getattr(self, "_ones", None)
        if o is None or o.shape != self.original_image.shape:
            o = self._ones = torch.ones_like(self.original_image)
        return o

    def get_focal(self):
        return fov2focal(self.FoVx, self.image_width), fov2focal(self.FoVy, self.image_height)

    def to(self, device):
        self.device = torch.device(device)
        for k in ("world_view_transform", "projection_matrix", "full_proj_transform", "camera_center",
                  "original_image", "gt_alpha_mask"):
            v = getattr(self, k)
            if v is not None:
                setattr(self, k, v.to(self.device))
        return self

    def shifted(self, trans_dist: float) -> "Camera":
        """Binocular partner: camera centre moved by `trans_dist` along the camera's own +x axis
        (scene/__init__.py:96-115).  The reference rebuilds a Camera through two host-side 4x4
        inversions and a device->host copy every iteration; in row-vector form the only entry that
        changes is world_view_transform[3, 0] -= trans_dist, which is done here in place on the
        device (golden vectors G4 check it against the reference's construction)."""
        cam = Camera.__new__(Camera)
        cam.__dict__.update(self.__dict__)
        cam.__dict__.pop("_b3gs_zkey", None)
        cam.original_image = None if self.original_image is None else self._ones_image()
        cam.gt_alpha_mask = None
        # All three matrices in closed form on the HOST, one upload: with the matrices in row-vector form only row 3 moves,
        #   wvt'[3,0] = wvt[3,0] - t,   full'[3,:] = wvt'[3,:] @ proj,   centre' = centre + t * wvt[:3,0]  (the camera x axis)
        # (round 3 cloned the view matrix, patched it and re-ran bmm + inverse on the device: ~25 tiny kernels, 54 us of
        # stream time per iteration of the reference's two-view schedule -- bench_ref_schedule.py)
        row, buf, done = _staging(self.device)
        _shifted_block(self._host_matrices(), trans_dist, buf)
        dev_buf = row.to(self.device, non_blocking=True) if self.device.type == "cuda" else row
        done()
        cam.world_view_transform = dev_buf[:16].view(4, 4)
        cam.full_proj_transform = dev_buf[16:32].view(4, 4)
        cam.camera_center = dev_buf[32:35]
        cam._host = None
        # only world_view_transform[3, 0] differs: every Gaussian has the same view-space z in both cameras, so
        # the pair can share one depth sort (FusedRasterizer, b3gs_forward_raw_batch depth_order_from)
        cam.same_depth_as = getattr(self, "same_depth_as", None) or self
        return cam


def _shifted_block(h, trans_dist: float, buf: np.ndarray):
    """[wvt' (16) | full' (16) | centre' (3)] of the camera `h` (host matrices) moved by trans_dist along its own x axis."""
    t = np.float32(trans_dist)
    w, f, c = buf[:16].reshape(4, 4), buf[16:32].reshape(4, 4), buf[32:35]
    w[:] = h["wvt"]
    w[3, 0] = w[3, 0] - t
    f[:] = h["full"]
    f[3, :] = w[3, :] @ h["proj"]
    c[:] = h["center"] + t * h["wvt"][:3, 0]


class CameraPairSlots:
    """An (input, shifted) camera pair whose matrices live in ONE static device block that is rewritten in place every
    iteration -- what an iteration replayed as a HIP graph needs: the reference draws a new input view and a new shift per
    iteration (train.py:92,125-128), a captured graph only sees fixed addresses.  `set(camera, trans_dist)` assembles
    [input: wvt, full, centre | shifted: wvt', full', centre' | trans_dist] on the host (the shifted camera in closed form,
    Camera.shifted) and uploads the 71 floats with one asynchronous copy from a pinned ring.  `cam`, `shifted` are Camera
    objects over views of the block (same intrinsics and image size as `template`); `trans_dist_dev` is the device float the
    loss block reads (B3gsLossIO::trans_dist_dev)."""

    def __init__(self, template: Camera, trans_dist: float = 0.1):
        dev = template.device
        self.device = dev
        self.block = torch.zeros(72, dtype=torch.float32, device=dev)
        self._ring = _PinnedRing(64, 72) if dev.type == "cuda" else None
        self._host_row = None if dev.type == "cuda" else torch.empty(72, dtype=torch.float32)
        self.cam, self.shifted = Camera.__new__(Camera), Camera.__new__(Camera)
        for k, c in enumerate((self.cam, self.shifted)):
            c.__dict__.update({a: v for a, v in template.__dict__.items() if a not in ("_host", "_b3gs_zkey", "same_depth_as")})
            b = self.block[35 * k:35 * k + 35]
            c.world_view_transform, c.full_proj_transform, c.camera_center = b[:16].view(4, 4), b[16:32].view(4, 4), b[32:35]
            c._host = None
        self.shifted.original_image, self.shifted.gt_alpha_mask = None, None
        self.shifted.same_depth_as = self.cam          # by construction: one depth sort per pair
        self.trans_dist_dev = self.block[70:71]
        self.set(template, trans_dist)

    def set(self, camera: Camera, trans_dist: float):
        k, row = self._ring.take() if self._ring is not None else (0, self._host_row)
        buf = row.numpy()
        h = camera._host_matrices()
        buf[:16] = h["wvt"].reshape(-1)
        buf[16:32] = h["full"].reshape(-1)
        buf[32:35] = h["center"]
        _shifted_block(h, trans_dist, buf[35:70])
        buf[70] = np.float32(trans_dist)
        buf[71] = 0.0
        self.block.copy_(row, non_blocking=True)
        if self._ring is not None:
            self._ring.uploaded(k, self.device)
        # (what rasterizer.camera_depth_key derives the z row from follows the pose: its cache is dropped)
        self.cam.R, self.cam.T, self.cam.trans, self.cam.scale = camera.R, camera.T, camera.trans, camera.scale
        self.cam.__dict__.pop("_b3gs_zkey", None)
        self.shifted.__dict__.pop("_b3gs_zkey", None)
        self.cam.original_image, self.cam.gt_alpha_mask = camera.original_image, camera.gt_alpha_mask


def look_at_orbit(yaw_deg: float, pivot=(0.0, 0.0, 6.0)):
    """Camera that starts at the origin looking down +z and is rotated about `pivot` by `yaw_deg`
    around the world y axis.  Returns (R, T) in the reference's convention:
    R = camera-to-world rotation, T = world-to-camera translation."""
    th = math.radians(yaw_deg)
    Ry = np.array([[math.cos(th), 0.0, math.sin(th)], [0.0, 1.0, 0.0], [-math.sin(th), 0.0, math.cos(th)]])
    pivot = np.asarray(pivot, dtype=np.float64)
    center = pivot + Ry @ (np.zeros(3) - pivot)
    R_c2w = Ry
    T = -R_c2w.T @ center
    return R_c2w, T
