"""The loss block that consumes the rasterizer's outputs and defines its upstream gradients:
train.py:123-149 of the reference (photometric L1 + D-SSIM, binocular warp L1 + edge-aware
disparity smoothness, alpha / background-mask loss), with utils/loss_utils.py:18-91,
utils/graphics_utils.py:80-125 (inverse_warp_images) and utils/image_utils.py:18-24 (psnr).

PyTorch-ROCm ops (this is a "next" row; values AND pixel gradients are pinned by
the golden fixture tests/golden/loss_block.npz generated from the reference's own Python).  The
warp is vectorised over batch and channels (the reference loops in Python).
"""
from __future__ import annotations

from math import exp
from typing import Optional

import torch
import torch.nn.functional as F


def l1_loss(network_output, gt, mask=None):
    if mask is not None:
        return torch.abs(network_output * mask - gt * mask).mean()
    return torch.abs(network_output - gt).mean()


_WINDOWS = {}


def _gaussian_window(window_size: int, sigma: float, channel: int, like: torch.Tensor) -> torch.Tensor:
    """utils/loss_utils.py:23-34; built once per (size, channels, device, dtype) instead of on every call
    (a host->device copy per iteration, which also cannot be captured in a HIP graph)."""
    key = (window_size, sigma, channel, like.device, like.dtype)
    if key not in _WINDOWS:
        _WINDOWS[key] = _build_window(window_size, sigma, channel, like)
    return _WINDOWS[key]


def _build_window(window_size: int, sigma: float, channel: int, like: torch.Tensor) -> torch.Tensor:
    g = torch.tensor([exp(-(x - window_size // 2) ** 2 / float(2 * sigma ** 2)) for x in range(window_size)])
    g = (g / g.sum()).unsqueeze(1)
    w2d = g.mm(g.t()).float().unsqueeze(0).unsqueeze(0)
    return w2d.expand(channel, 1, window_size, window_size).contiguous().to(device=like.device, dtype=like.dtype)


def ssim(img1, img2, window_size=11, size_average=True):
    channel = img1.size(-3)
    window = _gaussian_window(window_size, 1.5, channel, img1)
    pad = window_size // 2
    mu1 = F.conv2d(img1, window, padding=pad, groups=channel)
    mu2 = F.conv2d(img2, window, padding=pad, groups=channel)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img1 * img1, window, padding=pad, groups=channel) - mu1_sq
    sigma2_sq = F.conv2d(img2 * img2, window, padding=pad, groups=channel) - mu2_sq
    sigma12 = F.conv2d(img1 * img2, window, padding=pad, groups=channel) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return ssim_map.mean() if size_average else ssim_map.mean(1).mean(1).mean(1)


def smooth_loss(disparity: torch.Tensor, image: torch.Tensor) -> torch.Tensor:
    """Edge-aware disparity smoothness (utils/loss_utils.py:68-91): central differences without
    padding, weighted by exp(-0.33 |sum_c d image|).  disparity [B,1,H,W], image [B,3,H,W]."""
    def dx(t):
        return 0.5 * (t[..., 1:-1, 2:] - t[..., 1:-1, :-2])

    def dy(t):
        return 0.5 * (t[..., 2:, 1:-1] - t[..., :-2, 1:-1])
    edge_x_im = torch.exp(dx(image).sum(1, keepdim=True).abs() * -0.33)
    edge_y_im = torch.exp(dy(image).sum(1, keepdim=True).abs() * -0.33)
    return (edge_x_im * dx(disparity)).abs().mean() + (edge_y_im * dy(disparity)).abs().mean()


def inverse_warp_images(image: torch.Tensor, disparity: torch.Tensor) -> torch.Tensor:
    """Sample `image` [B,C,H,W] at column (c + disparity) with linear interpolation between
    floor and floor+1; pixels whose either tap falls outside the image are zero
    (utils/graphics_utils.py:80-125).  disparity [B,1,H,W]."""
    B, C, H, W = image.shape
    x0 = torch.floor(disparity).long()
    x1 = x0 + 1
    cols = torch.arange(W, device=image.device).view(1, 1, 1, W)
    c0, c1 = cols + x0, cols + x1
    invalid = (c0 < 0) | (c0 >= W) | (c1 < 0) | (c1 >= W)
    c0 = c0.clamp(0, W - 1).expand(B, C, H, W)
    c1 = c1.clamp(0, W - 1).expand(B, C, H, W)
    out = (x1 - disparity) * torch.gather(image, 3, c0) + (disparity - x0) * torch.gather(image, 3, c1)
    return out.masked_fill(invalid.expand(B, C, H, W), 0.0)


def psnr(img1, img2):
    mse = ((img1 - img2) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)
    return 20 * torch.log10(1.0 / torch.sqrt(mse))


def dtu_background_mask(gt_image: torch.Tensor, threshold: float = 30 / 255, dilate: int = 49) -> torch.Tensor:
    """train.py:110-120: dark-pixel mask [1,H,W], AND-ed with its 1..49-ROW downward shifts (the
    reference's `bg_mask[:, i:] *= clone[:, :-i]` slices dim 1 = image rows)."""
    m = (gt_image.max(0, keepdim=True).values < threshold)
    out = m.clone()
    for i in range(1, dilate + 1):
        out[:, i:] &= m[:, :-i]
    return out.float()


def binocular_loss(image, depth, alpha, gt_image, *, lambda_dssim: float = 0.2, shifted_image=None,
                   focal_x: Optional[float] = None, trans_dist: Optional[float] = None, gt_alpha_mask=None,
                   bg_mask=None):
    """total_loss of train.py:123-148 for one (input view, shifted view) pair.  Returns
    (total, dict of parts).  `depth` is NOT detached (train.py:131): the binocular term drives
    dL/d(depth) of the primary render; the shifted render only receives dL/d(colour)."""
    parts = {}
    disparity_loss = image.new_zeros(())
    if shifted_image is not None:
        disparity = focal_x * (-trans_dist) / (depth + 1e-5)
        warped = inverse_warp_images(shifted_image.unsqueeze(0), disparity.unsqueeze(0))
        shift_mask = inverse_warp_images(torch.ones_like(depth).unsqueeze(0), disparity.unsqueeze(0))
        parts["l1_masked"] = l1_loss(warped, gt_image.unsqueeze(0), mask=shift_mask)
        parts["smooth"] = smooth_loss(disparity.unsqueeze(0) * shift_mask, gt_image.unsqueeze(0))
        parts["warped"], parts["shift_mask"] = warped, shift_mask
        disparity_loss = parts["l1_masked"] + 0.05 * parts["smooth"]
    alpha_loss = image.new_zeros(())
    if gt_alpha_mask is not None:
        alpha_loss = torch.mean(torch.abs(alpha) * (1 - gt_alpha_mask))
    elif bg_mask is not None:
        alpha_loss = torch.mean(torch.abs(alpha) * bg_mask)
    parts["alpha_loss"] = alpha_loss
    parts["Ll1"] = l1_loss(image, gt_image)
    parts["ssim"] = ssim(image, gt_image)
    loss = (1.0 - lambda_dssim) * parts["Ll1"] + lambda_dssim * (1.0 - parts["ssim"])
    total = loss + disparity_loss + alpha_loss
    parts["loss"] = loss
    return total, parts


def expon_lr(step: int, lr_init: float, lr_final: float, lr_delay_steps: int = 0, lr_delay_mult: float = 1.0,
             max_steps: int = 1000000) -> float:
    """utils/general_utils.py:29-62 (log-linear interpolation with optional delayed warm-up)."""
    import numpy as np
    if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
        return 0.0
    if lr_delay_steps > 0:
        delay_rate = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
    else:
        delay_rate = 1.0
    t = min(max(step / max_steps, 0.0), 1.0)
    # (numpy's float64 exp / log, as the reference: golden G7 pins the values bit for bit)
    return float(delay_rate * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t))
