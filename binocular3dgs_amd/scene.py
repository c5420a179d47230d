"""The part of the reference's `scene/__init__.py` the training loop touches per iteration (train.py:59,127):

    scene.getTrainCameras() / getTestCameras()          scene/__init__.py:90-94
    scene.getShiftedCamera(camera, trans_dist=0.1)      scene/__init__.py:96-115   (the binocular partner)
    scene.cameras_extent                                 scene/__init__.py:64       (densify_and_prune's `extent`)

Dataset loading (COLMAP / Blender readers, image resizing, the dense-matcher point cloud) is outside the hot path
(SURVEY section 2, rows 8 and 15): a Scene here is built from camera.Camera objects the caller already has.

`getShiftedCamera` is the closed form SURVEY 8a-9 asks for: in row-vector form only `world_view_transform[3, 0]` moves,
the full projection's last row and the camera centre follow (camera.Camera.shifted, golden G4 against the reference's
construction).  The reference inverts the extrinsic on the device, copies the offset to the host (`.cpu()`: a sync per
iteration) and rebuilds a Camera through two host-side 4x4 inversions, three pageable uploads, a bmm and a device
inverse -- ~0.4 ms of device-idle host work per iteration at the reference's own iteration shape; here: host arithmetic on
cached matrices and ONE asynchronous upload from a pinned ring, no synchronisation.  The returned camera remembers that its
view-space depths equal its parent's (`same_depth_as`), which lets the pair share one depth sort (checked on the device).
"""
from __future__ import annotations

from typing import Sequence


class Scene:
    def __init__(self, train_cameras: Sequence, gaussians=None, test_cameras: Sequence = (), cameras_extent: float = 1.0,
                 model_path: str = ""):
        self.gaussians = gaussians
        self.model_path = model_path
        self.cameras_extent = cameras_extent
        self.train_cameras = {1.0: list(train_cameras)}
        self.test_cameras = {1.0: list(test_cameras)}

    def getTrainCameras(self, scale=1.0):
        return self.train_cameras[scale]

    def getTestCameras(self, scale=1.0):
        return self.test_cameras[scale]

    def getShiftedCamera(self, camera, trans_dist=0.1):
        return getShiftedCamera(camera, trans_dist)

    def save(self, iteration):
        """scene/__init__.py:86-88"""
        import os
        self.gaussians.save_ply(os.path.join(self.model_path, "point_cloud", f"iteration_{iteration}", "point_cloud.ply"))


def getShiftedCamera(camera, trans_dist=0.1):
    """scene/__init__.py:96-115 for a camera.Camera: centre moved by `trans_dist` along the camera's own +x axis, image of
    ones, no alpha mask, same R / T / FoV / uid."""
    return camera.shifted(float(trans_dist))
