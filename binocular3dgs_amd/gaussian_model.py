"""Gaussian parameter store: the part of scene/gaussian_model.py the hot path reads.

Same attribute and accessor names as the reference (`_xyz`, `_features_dc`, `_features_rest`,
`_scaling`, `_rotation`, `_opacity`; `get_xyz`, `get_scaling`, `get_rotation`, `get_features`,
`get_opacity`, `get_covariance`, `active_sh_degree`, `max_sh_degree`, `oneupSHdegree`:
scene/gaussian_model.py:24-60,95-122), so render() accepts either class.  The "next" rows of SURVEY.md section 8f live
beside it: densification (densify.py), PLY I/O and k-NN initialisation (init_points.py), the optimisers (step.py) and
the checkpoint tuple (`capture()` / `restore()`, checkpoint.py).
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
        self.optimizer = None            # scene/gaussian_model.py:56-58
        self.spatial_lr_scale = 0
        # scene/gaussian_model.py:33-43 (setup_functions): render() recognises a raw-parameter model by these
        self.scaling_activation = torch.exp
        self.opacity_activation = torch.sigmoid
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

    def add_densification_stats(self, viewspace_point_grad, update_filter):
        """scene/gaussian_model.py:409-411 (takes the gradient tensor itself: [P,3]).  The reference indexes with the boolean
        mask, which makes the host wait for the mask's population count on every call; the same sums are formed here without
        leaving the device (rows outside the mask keep their bits)."""
        if update_filter.dtype != torch.bool:
            self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_grad[update_filter, :2], dim=-1,
                                                                 keepdim=True)
            self.denom[update_filter] += 1
            return
        m = update_filter.unsqueeze(-1)
        n = torch.norm(viewspace_point_grad[:, :2], dim=-1, keepdim=True)
        torch.where(m, self.xyz_gradient_accum + n, self.xyz_gradient_accum, out=self.xyz_gradient_accum)
        self.denom += m

    def update_max_radii(self, radii, visibility_filter):
        """train.py:178 (sync-free like add_densification_stats)"""
        if visibility_filter.dtype != torch.bool:
            self.max_radii2D[visibility_filter] = torch.max(self.max_radii2D[visibility_filter],
                                                            radii[visibility_filter].float())
            return
        torch.where(visibility_filter, torch.maximum(self.max_radii2D, radii.float()), self.max_radii2D, out=self.max_radii2D)

    # ---- checkpoint tuple (scene/gaussian_model.py:61-93, train.py:41-43,200-202) ----------------------------
    def capture(self, optimizer=None):
        """The reference's 12-tuple; the optimiser entry has torch.optim.Adam's state_dict() layout whatever optimiser
        of this build is in use (checkpoint.py)."""
        from .checkpoint import capture
        return capture(self, optimizer)

    def restore(self, model_args, training_args=None, optimizer=None, optimizer_factory=None):
        """Put a captured tuple (of this build or of the reference) back.  `training_args` is accepted for signature
        compatibility with the reference (which rebuilds its optimiser from it); here the optimiser is passed in, or
        built by `optimizer_factory(model)` once the parameters are in place."""
        from .checkpoint import restore
        del training_args
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
