"""One training iteration of a Binocular3DGS-style trainer as a sequence of CALLS into pluggable callables.

What is pinned is the contract, not anybody's text: golden G11 (tests/golden/train_trace.json, recorded from the
reference's loop train.py:65-202 with stand-ins for everything it calls) holds, per iteration, the ordered calls, their
argument shapes / constants, which earlier output feeds which later input, and the weights the loss terms carry into
backward().  `IterationSchedule.run_iteration` drives that sequence; tests/test_schedule_trace.py asserts that it produces
G11 event for event when handed the same stand-ins.  With the default `ops` it drives this build's modules (render,
loss_utils, graphics_utils -- each one HIP launch behind the reference's signature), which is what the "unchanged" surface
of bench_ref_schedule.py and the lock-step trainer of tests/ref_schedule.py run.

Branches of the reference's loop that are configuration, not call sequence, and are NOT driven here: the DTU background
mask (train.py:112-120, a dataset switch), random backgrounds, the network GUI, test-view reports, saving.
"""
from __future__ import annotations

import types

import torch


def default_ops():
    """The build's own callables behind the five names the loop uses."""
    from .graphics_utils import inverse_warp_images
    from .loss_utils import SmoothLoss, l1_loss, ssim
    from .render import render
    return types.SimpleNamespace(render=render, l1_loss=l1_loss, ssim=ssim, SmoothLoss=SmoothLoss,
                                 inverse_warp_images=inverse_warp_images)


class IterationSchedule:
    """model: update_learning_rate / oneupSHdegree / opacity_decay / add_densification_stats / densify_and_prune /
    optimizer / max_radii2D (the reference's GaussianModel surface); scene: getTrainCameras / getShiftedCamera /
    cameras_extent.  `opacity_decay_factor=None` and `binocular=False` switch the two optional phases off."""

    def __init__(self, model, scene, pipe, background, *, ops=None, iterations=30_000, shift_cam_start=20_000,
                 binocular=True, opacity_decay_factor=0.995, lambda_dssim=0.2, densify_from_iter=500,
                 densify_until_iter=15_000, densification_interval=100, densify_grad_threshold=0.0002, min_opacity=0.005,
                 sh_interval=1000, smooth_weight=0.05, log_item=False, before_densify=None):
        self.model, self.scene, self.pipe, self.background = model, scene, pipe, background
        self.ops = ops if ops is not None else default_ops()
        self.iterations, self.shift_cam_start, self.binocular = iterations, shift_cam_start, binocular
        self.decay, self.lam, self.smooth_weight = opacity_decay_factor, lambda_dssim, smooth_weight
        self.densify_from_iter, self.densify_until_iter = densify_from_iter, densify_until_iter
        self.densification_interval, self.grad_threshold, self.min_opacity = densification_interval, densify_grad_threshold, \
            min_opacity
        self.sh_interval, self.log_item, self.before_densify = sh_interval, log_item, before_densify
        self.views = list(scene.getTrainCameras())
        H, W = self.views[0].image_height, self.views[0].image_width
        dev = background.device
        # pixel-coordinate grids and the all-ones image whose warp says which pixels the shifted view reaches
        ys, xs = torch.meshgrid(torch.arange(H, device=dev), torch.arange(W, device=dev), indexing="ij")
        self.rows, self.cols = ys.contiguous(), xs.contiguous()
        self.ones = torch.ones((1, H, W), dtype=torch.float32, device=dev)
        self.smooth = self.ops.SmoothLoss()
        self.ema = 0.0
        self.densified = False      # did the iteration just run change the Gaussian set?

    # ---- the terms of the loss --------------------------------------------------------------------------------------
    def _stereo_term(self, cam, first, target, shift):
        """Render the partner `shift` along the camera's x axis, warp it back with the primary's depth, compare."""
        ops = self.ops
        partner = self.scene.getShiftedCamera(cam, shift)
        second = ops.render(partner, self.model, self.pipe, self.background)
        fx = cam.get_focal()[0]
        disp = (fx * -shift) / (first["rendered_depth"] + 1e-5)
        disp4, target4 = disp[None], target[None]
        warped = ops.inverse_warp_images(second["render"][None], disp4, self.rows, self.cols)
        reach = ops.inverse_warp_images(self.ones[None], disp4, self.rows, self.cols)
        return ops.l1_loss(warped, target4, mask=reach) + self.smooth_weight * self.smooth.forward(disparity=disp * reach,
                                                                                                   image=target4)

    def run_iteration(self, it: int, view_index: int, shift: float | None = None):
        """-> the total loss (attached).  `view_index` / `shift`: the caller's draws (an input view; the signed baseline)."""
        m, ops = self.model, self.ops
        m.update_learning_rate(it)
        if it % self.sh_interval == 0:
            m.oneupSHdegree()
        cam = self.views[view_index]
        first = ops.render(cam, m, self.pipe, self.background)
        target = cam.original_image.to(self.background.device)
        stereo = 0.0
        if self.binocular and it > self.shift_cam_start:
            stereo = self._stereo_term(cam, first, target, float(shift))
        coverage = 0.0
        if getattr(cam, "gt_alpha_mask", None) is not None:
            coverage = (first["rendered_alpha"].abs() * (1 - cam.gt_alpha_mask)).mean()
        photo = ops.l1_loss(first["render"], target)
        recon = (1.0 - self.lam) * photo + self.lam * (1.0 - ops.ssim(first["render"], target))
        total = recon + stereo + coverage
        total.backward()
        self.densified = False
        with torch.no_grad():
            if self.log_item:
                self.ema = 0.4 * recon.item() + 0.6 * self.ema
            self._after_backward(it, first)
        return total

    def _after_backward(self, it, first):
        m = self.model
        if self.decay is not None and it > self.densify_from_iter:
            self.densify_until_iter = self.iterations      # with the decay on, densification runs to the end
            m.opacity_decay(factor=self.decay)
        if it < self.densify_until_iter:
            seen = first["visibility_filter"]
            widest = m.max_radii2D
            widest[seen] = torch.max(widest[seen], first["radii"][seen])
            m.add_densification_stats(first["viewspace_points"], seen)
            if it > self.densify_from_iter and it % self.densification_interval == 0:
                if self.before_densify is not None:
                    self.before_densify(it)
                m.densify_and_prune(self.grad_threshold, self.min_opacity, self.scene.cameras_extent, None)
                self.densified = True
        if it < self.iterations:
            opt = m.optimizer
            opt.step()
            opt.zero_grad(set_to_none=True)
