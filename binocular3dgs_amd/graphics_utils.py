"""Drop-in for the pieces of the reference's `utils/graphics_utils.py` the training loop uses
(train.py:19: `from utils.graphics_utils import inverse_warp_images`; scene/cameras.py: the matrix helpers):

    inverse_warp_images(image, disparity, row_indices, column_indices)     utils/graphics_utils.py:80-125
    getWorld2View2 / getProjectionMatrix / fov2focal / focal2fov          :38-77  (camera.py, pinned by golden G3)

`inverse_warp_images` is one launch forward and one backward (`csrc/lossfn.hip`, ABI 9) where the reference loops over
batch and channels in Python (~25 PyTorch kernels per call, index tensors cloned per channel); gradients for the image
(bilinear scatter) and for the disparity.  No CPU fallback: `loss.inverse_warp_images` is the PyTorch statement.
"""
from __future__ import annotations

from . import _C
from .rasterizer import touch_pending
from .camera import focal2fov, fov2focal  # noqa: F401
from .camera import projection_matrix as getProjectionMatrix  # noqa: F401


def getWorld2View2(R, t, translate=None, scale=1.0):
    """utils/graphics_utils.py:38-49"""
    import numpy as np
    from .camera import world_to_view
    return world_to_view(R, t, np.array([0.0, 0.0, 0.0]) if translate is None else translate, scale)


def inverse_warp_images(image, disparity, row_indices=None, column_indices=None):
    """utils/graphics_utils.py:80-125: out[b,ch,r,c] = (x1 - d) image[b,ch,r,c+x0] + (d - x0) image[b,ch,r,c+x1] with
    d = disparity[b,0,r,c], x0 = floor(d), x1 = x0 + 1; zero where either column leaves the image.  image [B,C,H,W],
    disparity [B,1,H,W].  `row_indices`, `column_indices` (the reference's meshgrid of pixel coordinates, train.py:56-57)
    are accepted for signature compatibility; the kernel knows where its pixels are.  The node is C++
    (csrc/host/loss.cpp: WarpFn)."""
    touch_pending(image, disparity)
    return _C.inverse_warp_images(image, disparity, row_indices, column_indices)
