"""Gaussian parameter store: the part of scene/gaussian_model.py the hot path reads.

Same attribute and accessor names as the reference (`_xyz`, `_features_dc`, `_features_rest`,
`_scaling`, `_rotation`, `_opacity`; `get_xyz`, `get_scaling`, `get_rotation`, `get_features`,
`get_opacity`, `get_covariance`, `active_sh_degree`, `max_sh_degree`, `oneupSHdegree`:
scene/gaussian_model.py:24-60,95-122), so render() accepts either class.  The "next" rows of SURVEY.md section 8f live
beside it: densification (densify.py), PLY I/O and k-NN initialisation (init_points.py), the optimisers (step.py) and
the checkpoint tuple (`capture()` / `restore()`, checkpoint.py).

Round 5: the training-side methods an unchanged train.py calls on `gaussians` carry the reference's own signatures --
`training_setup(training_args)`, `update_learning_rate(iteration)`, `opacity_decay(factor)`,
`add_densification_stats(viewspace_point_tensor, update_filter)`, `densify_and_prune(max_grad, min_opacity, extent,
max_screen_size)`, `reset_opacity()`, `create_from_pcd`, `save_ply` / `load_ply`, `capture()` / `restore(model_args,
training_args)` -- on top of the HIP rows (csrc/optim.hip, lossfn.hip, densify.hip, knn.hip); `self.optimizer` is a
torch.optim.Adam subclass (optim.Adam) whose step() is one launch.
"""
from __future__ import annotations

import torch
from torch import nn


def build_rotation(r: torch.Tensor) -> torch.Tensor:
    """utils/general_utils.py:78-99: normalise, then (w,x,y,z) -> 3x3."""
    q = r / torch.sqrt((r * r).sum(dim=1, keepdim=True))
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.reshape(-1, 3, 3)


def covariance_from_scaling_rotation(scaling, scaling_modifier, rotation) -> torch.Tensor:
    """scene/gaussian_model.py:27-31 + utils/general_utils.py:64-73,101-110:
    L = R diag(mod*s); Sigma = L L^T; 6-vector (00,01,02,11,12,22)."""
    L = build_rotation(rotation) * (scaling_modifier * scaling).unsqueeze(1)
    S = L @ L.transpose(1, 2)
    return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], dim=1)


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


class GaussianModel:
    def __init__(self, sh_degree: int):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        e = torch.empty(0)
        self._xyz, self._features_dc, self._features_rest = e, e, e
        self._scaling, self._rotation, self._opacity = e, e, e
        self.max_radii2D, self.xyz_gradient_accum, self.denom = e, e, e
        self.optimizer = None            # scene/gaussian_model.py:56-58
        self.percent_dense = 0
        self.spatial_lr_scale = 0
        # scene/gaussian_model.py:33-43 (setup_functions): render() recognises a raw-parameter model by these
        self.scaling_activation = torch.exp
        self.scaling_inverse_activation = torch.log
        self.covariance_activation = covariance_from_scaling_rotation
        self.opacity_activation = torch.sigmoid
        self.inverse_opacity_activation = inverse_sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    @classmethod
    def from_tensors(cls, xyz, features_dc, features_rest, scaling, rotation, opacity, sh_degree=1,
                     active_sh_degree=None, device=None, requires_grad=True):
        m = cls(sh_degree)
        for name, t in (("_xyz", xyz), ("_features_dc", features_dc), ("_features_rest", features_rest),
                        ("_scaling", scaling), ("_rotation", rotation), ("_opacity", opacity)):
            t = t.detach().clone().float()
            if device is not None:
                t = t.to(device)
            setattr(m, name, nn.Parameter(t.contiguous(), requires_grad=requires_grad))
        m.active_sh_degree = sh_degree if active_sh_degree is None else active_sh_degree
        return m

    # ---- densification statistics (the consumers of radii / means2D gradients) ----
    def init_densification_stats(self):
        """scene/gaussian_model.py:147-152: max_radii2D [P], xyz_gradient_accum [P,1], denom [P,1]."""
        P, dev = self._xyz.shape[0], self._xyz.device
        self.max_radii2D = torch.zeros((P,), device=dev)
        self.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
        self.denom = torch.zeros((P, 1), device=dev)

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        """scene/gaussian_model.py:409-411, the reference's signature: the tensor render() returned as `viewspace_points`
        (its `.grad` holds the screen-space gradient after backward) and a boolean mask / index tensor.
            xyz_gradient_accum[update_filter] += norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1, keepdim=True)
            denom[update_filter] += 1
        The reference indexes with the boolean mask, which makes the host wait for the mask's population count on every
        call; on the device this is one launch (b3gs_add_densification_stats) that leaves the rows outside the mask
        untouched, bit for bit."""
        grad = viewspace_point_tensor.grad
        if grad is None:
            raise AttributeError("add_densification_stats: viewspace_point_tensor.grad is None (call it after backward(), "
                                 "with the tensor render() returned as 'viewspace_points')")
        self._accumulate_stats(grad, update_filter)

    def _accumulate_stats(self, grad, update_filter):
        if self.xyz_gradient_accum.shape[0] != grad.shape[0] or self.denom.shape[0] != grad.shape[0]:
            raise RuntimeError(f"add_densification_stats: the statistics hold {self.xyz_gradient_accum.shape[0]} rows for "
                               f"{grad.shape[0]} Gaussians -- training_setup() (or restore()) allocates them, as in the reference")
        if update_filter.dtype != torch.bool:
            self.xyz_gradient_accum[update_filter] += torch.norm(grad[update_filter, :2], dim=-1, keepdim=True)
            self.denom[update_filter] += 1
            return
        if grad.is_cuda and grad.dtype == torch.float32 and grad.dim() == 2 and grad.stride(1) == 1 and \
                self.xyz_gradient_accum.is_contiguous() and self.denom.is_contiguous():
            # (the compiled entry checks that the mask has one element per row and lives on the gradient's device --
            # a short mask raises IndexError like the reference's indexing, a host mask raises instead of being read as a
            # device pointer)
            from . import _C
            from .rasterizer import touch_pending
            touch_pending(update_filter)
            if update_filter.device != grad.device:
                update_filter = update_filter.to(grad.device)        # (the reference accepts a CPU mask)
            _C.add_densification_stats(grad, update_filter, self.xyz_gradient_accum, self.denom)
            return
        m = update_filter.unsqueeze(-1)
        n = torch.norm(grad[:, :2], dim=-1, keepdim=True)
        torch.where(m, self.xyz_gradient_accum + n, self.xyz_gradient_accum, out=self.xyz_gradient_accum)
        self.denom += m

    def update_max_radii(self, radii, visibility_filter):
        """train.py:178 (sync-free like add_densification_stats)"""
        if visibility_filter.dtype != torch.bool:
            self.max_radii2D[visibility_filter] = torch.max(self.max_radii2D[visibility_filter],
                                                            radii[visibility_filter].float())
            return
        torch.where(visibility_filter, torch.maximum(self.max_radii2D, radii.float()), self.max_radii2D, out=self.max_radii2D)

    # ---- the training-side methods of scene/gaussian_model.py, reference signatures -------------------------------------
    def training_setup(self, training_args):
        """scene/gaussian_model.py:149-167: six single-tensor parameter groups in the reference's order and with its names,
        Adam(lr=0.0, eps=1e-15) -- here optim.Adam, a torch.optim.Adam whose step() is one HIP launch -- and the exponential
        position schedule."""
        from .optim import Adam
        self.percent_dense = training_args.percent_dense
        P, dev = self.get_xyz.shape[0], self.get_xyz.device
        self.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
        self.denom = torch.zeros((P, 1), device=dev)
        if self.max_radii2D.shape[0] != P:
            self.max_radii2D = torch.zeros((P,), device=dev)
        groups = [
            {"params": [self._xyz], "lr": training_args.position_lr_init * self.spatial_lr_scale, "name": "xyz"},
            {"params": [self._features_dc], "lr": training_args.feature_lr, "name": "f_dc"},
            {"params": [self._features_rest], "lr": training_args.feature_lr / 20.0, "name": "f_rest"},
            {"params": [self._opacity], "lr": training_args.opacity_lr, "name": "opacity"},
            {"params": [self._scaling], "lr": training_args.scaling_lr, "name": "scaling"},
            {"params": [self._rotation], "lr": training_args.rotation_lr, "name": "rotation"},
        ]
        self.optimizer = Adam(groups, lr=0.0, eps=1e-15)
        from .loss import expon_lr
        lr_init = training_args.position_lr_init * self.spatial_lr_scale
        lr_final = training_args.position_lr_final * self.spatial_lr_scale
        mult, steps = training_args.position_lr_delay_mult, training_args.position_lr_max_steps
        self.xyz_scheduler_args = lambda step: expon_lr(step, lr_init, lr_final, lr_delay_mult=mult, max_steps=steps)

    def update_learning_rate(self, iteration):
        """scene/gaussian_model.py:169-175"""
        for param_group in self.optimizer.param_groups:
            if param_group["name"] == "xyz":
                lr = self.xyz_scheduler_args(iteration)
                param_group["lr"] = lr
                return lr

    def opacity_decay(self, factor=0.99):
        """scene/gaussian_model.py:307-309: `_opacity.data = inverse_sigmoid(get_opacity * factor)` -- one launch, in place."""
        o = self._opacity
        if not o.is_cuda:
            o.data = self.inverse_opacity_activation(self.get_opacity * factor)
            return
        from . import _C
        _C.opacity_decay(o, float(factor))        # (in place through the raw pointer; bumps the version counter)

    def reset_opacity(self):
        """scene/gaussian_model.py:210-213 (commented out of this fork's train.py:188-193, kept for the interface):
        opacity <- min(opacity, 0.01) in logit space, Adam moments of the group reset."""
        new = inverse_sigmoid(torch.min(self.get_opacity, torch.ones_like(self.get_opacity) * 0.01))
        for group in self.optimizer.param_groups:
            if group["name"] == "opacity":
                old = group["params"][0]
                st = self.optimizer.state.pop(old, None)
                p = nn.Parameter(new.detach().contiguous().requires_grad_(True))
                group["params"][0] = p
                if st is not None:
                    st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(p), torch.zeros_like(p)
                    self.optimizer.state[p] = st
                self._opacity = p

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size):
        """scene/gaussian_model.py:393-407: clone / split / prune with the optimiser state carried along, as one
        classification launch, three prefix sums and one scatter launch (densify.py, csrc/densify.hip; the reference's row
        order; the split offsets are drawn with torch.randn on the device instead of torch.normal)."""
        from .densify import densify_and_prune
        # `split_noise` (optional attribute, [2, P, 3] standard-normal samples addressed by the ORIGINAL Gaussian index, used
        # once): replicas of a data-parallel run and lock-step tests hand every copy the same split offsets
        noise, self.split_noise = getattr(self, "split_noise", None), None
        densify_and_prune(self, self.optimizer, max_grad, min_opacity, extent, max_screen_size,
                          percent_dense=self.percent_dense, noise=noise)
        torch.cuda.empty_cache()

    def create_from_pcd(self, pcd, spatial_lr_scale: float):
        """scene/gaussian_model.py:124-147: `pcd` with `.points` / `.colors` ([P,3] arrays)."""
        from .init_points import create_from_points
        self.spatial_lr_scale = spatial_lr_scale
        m = create_from_points(pcd.points, pcd.colors, self.max_sh_degree)
        for a in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity", "max_radii2D"):
            setattr(self, a, getattr(m, a))

    def save_ply(self, path):
        from .init_points import save_ply
        save_ply(self, path)

    def load_ply(self, path):
        from .init_points import load_ply
        m = load_ply(path, self.max_sh_degree)
        for a in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity"):
            setattr(self, a, getattr(m, a))
        self.active_sh_degree = self.max_sh_degree

    # ---- checkpoint tuple (scene/gaussian_model.py:61-93, train.py:41-43,200-202) ----------------------------
    def capture(self, optimizer=None):
        """The reference's 12-tuple; the optimiser entry has torch.optim.Adam's state_dict() layout whatever optimiser
        of this build is in use (checkpoint.py)."""
        from .checkpoint import capture
        return capture(self, optimizer)

    def restore(self, model_args, training_args=None, optimizer=None, optimizer_factory=None):
        """Put a captured tuple (of this build or of the reference) back.  `restore(model_args, training_args)` is the
        reference's call (train.py:41-43): the optimiser is rebuilt by training_setup(training_args) once the parameters
        are in place and receives the captured state.  Alternatively an optimiser of this build is passed in, or built by
        `optimizer_factory(model)`."""
        from .checkpoint import restore
        if training_args is not None and optimizer is None and optimizer_factory is None:
            # the reference's flow (scene/gaussian_model.py:77-93): parameters in place, training_setup(training_args)
            # rebuilds the optimiser over them, then load_state_dict
            def optimizer_factory(model):
                model.training_setup(training_args)
                return model.optimizer
            return restore(self, model_args, None, optimizer_factory)
        return restore(self, model_args, optimizer if optimizer is not None else self.optimizer, optimizer_factory)

    def parameters(self):
        return [self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity]

    @property
    def get_scaling(self):
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    def get_covariance(self, scaling_modifier=1):
        return covariance_from_scaling_rotation(self.get_scaling, scaling_modifier, self._rotation)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1
