"""`torch.optim.Adam`-compatible optimiser whose `step()` is ONE launch of libb3gs_raster.so for all parameter groups
(`b3gs_adam_step_at`, csrc/optim.hip) -- what sits behind `gaussians.optimizer.step()` / `.zero_grad(set_to_none=True)` of
an unchanged train.py:196-198 and behind the optimiser-state surgery of scene/gaussian_model.py:258-340.

It IS a torch.optim.Adam (subclass): `param_groups` (the reference's six named groups, scene/gaussian_model.py:154-161),
`state[param] = {"step", "exp_avg", "exp_avg_sq"}`, `state_dict()` / `load_state_dict()` (the checkpoint tuple of
scene/gaussian_model.py:61-93, golden G9), `add_param_group`, `zero_grad` are torch's own; only the arithmetic of `step()`
moves: torch issues two foreach kernels per group plus scalar bookkeeping (12+ launches, ~0.35 ms per iteration at 1M
Gaussians), here every group's tensors are segments of one launch (28 bytes per parameter float at 5-6.5 TB/s).
No CPU path: host parameters raise.  amsgrad / weight_decay / maximize are not what the reference uses and raise.
"""
from __future__ import annotations

import torch

from . import _lib
from ._cuda import device_guard, raw_stream


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, **kw):
        kw.pop("foreach", None), kw.pop("fused", None), kw.pop("capturable", None)
        if kw.pop("maximize", False) or kw.pop("differentiable", False):
            raise NotImplementedError("binocular3dgs_amd.optim.Adam: maximize / differentiable are not supported")
        if kw:
            raise TypeError(f"unexpected arguments {sorted(kw)}")
        if weight_decay or amsgrad:
            raise NotImplementedError("binocular3dgs_amd.optim.Adam: weight_decay / amsgrad are not supported "
                                      "(scene/gaussian_model.py:163 uses neither)")
        # (foreach / fused stay None, torch's defaults: state_dict() then carries exactly the hyper-parameters the reference's
        # own torch.optim.Adam(l, lr=0.0, eps=1e-15) carries -- this class's step() consults neither)
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False)

    def _init_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            # torch.optim.Adam's own layout (non-capturable: the step counter is a host tensor)
            st["step"] = torch.tensor(0.0, dtype=torch.float32)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        # segments grouped by (device, betas, eps, step number): ONE launch per group of up to 8 tensors -- the reference's
        # six single-tensor groups share everything but the learning rate, which travels per segment
        buckets = {}
        for group in self.param_groups:
            if group.get("weight_decay", 0) or group.get("amsgrad", False) or group.get("maximize", False):
                raise NotImplementedError("binocular3dgs_amd.optim.Adam: weight_decay / amsgrad / maximize are not supported")
            b1, b2 = group["betas"]
            lr = group["lr"]
            lr = float(lr.item()) if torch.is_tensor(lr) else float(lr)
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if not p.is_cuda:
                    raise _lib.B3gsError("binocular3dgs_amd.optim.Adam: parameters must live on the HIP device (no CPU path)")
                if g.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                if p.dtype != torch.float32 or not p.is_contiguous():
                    raise _lib.B3gsError("binocular3dgs_amd.optim.Adam: parameters must be contiguous float32 tensors")
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.float().contiguous()
                st = self._init_state(p)
                st["step"] += 1
                m, v = st["exp_avg"], st["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous() and m.dtype == torch.float32 and m.device == p.device):
                    raise _lib.B3gsError("binocular3dgs_amd.optim.Adam: exp_avg / exp_avg_sq must be contiguous float32 "
                                         "tensors on the parameter's device")
                key = (p.device.index, float(b1), float(b2), float(group["eps"]), int(st["step"]))
                buckets.setdefault(key, []).append((p, g, m, v, lr))
        L = _lib.lib()
        cache = self.__dict__.setdefault("_b3gs_segs", {})
        for (di, b1, b2, eps, step), segs_py in buckets.items():
            dev = torch.device("cuda", di)
            stream = raw_stream(dev)
            with device_guard(dev):
                for c0 in range(0, len(segs_py), 8):
                    chunk = segs_py[c0:c0 + 8]
                    # the ctypes array of a chunk is kept while its tensors stay where they are (a field write costs ~1 us:
                    # nine fields x six tensors per step otherwise); only gradient pointers and learning rates move
                    sig = tuple((p.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel()) for p, _g, m, v, _lr in chunk)
                    ent = cache.get((di, c0))
                    if ent is None or ent[0] != sig:
                        segs = (_lib.B3gsAdamSegment * len(chunk))()
                        for k, (p, _g, m, v, _lr) in enumerate(chunk):
                            n = p.numel()
                            s = segs[k]
                            s.param = (p.data_ptr() or None) if n else None
                            s.exp_avg, s.exp_avg_sq = ((m.data_ptr() or None), (v.data_ptr() or None)) if n else (None, None)
                            s.count, s.row_len, s.first_row, s.lr_dev = n, 0, 0, None
                        ent = cache[(di, c0)] = (sig, segs)
                    segs = ent[1]
                    for k, (p, g, _m, _v, lr) in enumerate(chunk):
                        segs[k].grad = (g.data_ptr() or None) if p.numel() else None
                        segs[k].lr = lr
                    _lib.check(L.b3gs_adam_step_at(len(chunk), segs, step, b1, b2, eps, stream), "b3gs_adam_step_at")
        # the kernel wrote through raw pointers: autograd's version counters have to hear about it (saved-tensor checks;
        # the depth-order hint of rasterizer._RasterizeRaw keys on the position tensor's version)
        for segs_py in buckets.values():
            for p, _g, _m, _v, _lr in segs_py:
                torch.autograd.graph.increment_version(p)
        return loss
