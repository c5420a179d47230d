"""Python surface of the rasterizer, name-for-name what the reference imports:

    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
                                                        (gaussian_renderer/__init__.py:14)
    GaussianRasterizationSettings(image_height=..., ..., debug=...)   (:36-49, 12 keywords)
    rasterizer = GaussianRasterizer(raster_settings=...)               (:51)
    color, radii, depth, alpha = rasterizer(means3D=, means2D=, shs=, colors_precomp=,
                                            opacities=, scales=, rotations=, cov3D_precomp=)  (:85-93)

and the two `_C` functions of the un-vendored extension:
`_C.rasterize_gaussians(...)`, `_C.rasterize_gaussians_backward(...)` (+ `_C.mark_visible`).
The arithmetic lives in libb3gs_raster.so (hand-written HIP for gfx950) behind the C ABI of
include/b3gs_raster.h; this module only moves pointers.  No CPU / PyTorch fallback exists:
CPU tensors raise.
"""
from __future__ import annotations

import ctypes as C
import atexit
import os
from typing import NamedTuple, Optional

import torch
from torch import nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _dev_f32(t: torch.Tensor, name: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a tensor")
    if not t.is_cuda:
        raise _lib.B3gsError(f"{name} is on {t.device}: the rasterizer runs on an MI355X (HIP) device only; "
                             "there is no CPU path")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None or t.numel() == 0 else t.data_ptr()


def _stream(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def _scene(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
           projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos, prefiltered, debug):
    """Validate like the upstream binding and fill the C struct.  Returns (struct, keepalive)."""
    means3D = _dev_f32(means3D, "means3D")
    if means3D.dim() != 2 or means3D.shape[1] != 3:
        raise ValueError("means3D must have dimensions (num_points, 3)")
    P = means3D.shape[0]
    dev = means3D.device
    keep = [means3D]

    def opt(t, name, shape_tail):
        if t is None or t.numel() == 0:
            return None
        t = _dev_f32(t, name)
        if t.shape[0] != P or tuple(t.shape[1:]) not in shape_tail:
            raise ValueError(f"{name} has shape {tuple(t.shape)}, expected ({P}, {shape_tail})")
        keep.append(t)
        return t

    sh_t = None
    M = 0
    if sh is not None and sh.numel() != 0:
        sh_t = _dev_f32(sh, "sh")
        if sh_t.dim() != 3 or sh_t.shape[0] != P or sh_t.shape[2] != 3:
            raise ValueError("sh must have dimensions (num_points, K, 3)")
        M = sh_t.shape[1]
        keep.append(sh_t)
    colors_t = opt(colors, "colors_precomp", [(3,)])
    opacity_t = _dev_f32(opacity, "opacities").reshape(-1)
    if opacity_t.numel() != P:
        raise ValueError("opacities must have num_points elements")
    scales_t = opt(scales, "scales", [(3,)])
    rot_t = opt(rotations, "rotations", [(4,)])
    cov_t = opt(cov3D_precomp, "cov3D_precomp", [(6,)])
    bg = _dev_f32(background, "bg").reshape(-1)
    vm = _dev_f32(viewmatrix, "viewmatrix").reshape(-1)
    pm = _dev_f32(projmatrix, "projmatrix").reshape(-1)
    cp = _dev_f32(campos, "campos").reshape(-1)
    if bg.numel() != 3 or vm.numel() != 16 or pm.numel() != 16 or cp.numel() != 3:
        raise ValueError("bg/campos must have 3 and viewmatrix/projmatrix 16 elements")
    keep += [opacity_t, bg, vm, pm, cp]
    sc = _lib.B3gsScene(P, int(degree), int(M), int(image_width), int(image_height), float(tan_fovx),
                        float(tan_fovy), float(scale_modifier), int(bool(prefiltered)), int(bool(debug)),
                        _ptr(bg), _ptr(means3D), _ptr(sh_t), _ptr(colors_t), _ptr(opacity_t), _ptr(scales_t),
                        _ptr(rot_t), _ptr(cov_t), _ptr(vm), _ptr(pm), _ptr(cp))
    return sc, keep, dev, P, M


class _LazyN:
    """Sync-free forward of the autograd surface, TRAINING renders only: the one blocking read-back of num_rendered per
    forward (upstream sizes the binning buffer from it; `api.hip`: b3gs_forward) is paid by the FIRST render of a
    (device, P, W, H) shape and by every render that will not be differentiated (no input requires a gradient: evaluation
    loops, render.py-style scripts -- they get the exact, synchronous forward).  Later differentiated renders of the shape
    run b3gs_forward_capacity with a binning buffer of twice the largest N seen; N stays on the device and travels to
    pinned host memory behind the kernels.  It is looked at
      * at the entry of that render's BACKWARD (the loss sits between the two, so the copy has long completed): an N
        above the capacity -- the image and the loss were computed from truncated tile lists -- raises B3gsError there,
        i.e. before any gradient exists and before `optimizer.step()` of train.py:149-197 can consume it;
      * at the next render (capacity grown while N is still inside it; renders whose backward never ran are checked here);
      * at interpreter exit (a warning for anything still unchecked).
    P changes at every densification, so each new Gaussian set starts with an exact, synchronous render.
    B3GS_DROPIN_SYNC=1 selects the synchronous forward everywhere."""
    HEADROOM, REGROW_AT, RING = 2.0, 0.6, 64

    def __init__(self):
        self.capacity = {}     # (device index, P, W, H) -> instances the binning buffer is sized for
        self.pending = []      # [token] = [key, capacity used, pinned int32[1], event, checked]
        self.pinned = None
        self.slot = 0
        self.enabled = os.environ.get("B3GS_DROPIN_SYNC", "0") != "1"
        self.key_bits = 27     # fused render() node: depth sort on 27-bit keys until a render reports a key outside the span

    def note(self, key, n):
        self.capacity[key] = max(self.capacity.get(key, 0), int(n * self.HEADROOM), 1 << 16)

    def _resolve(self, tok):
        """wait for the token's copy, grow the capacity; returns (n, cap) when that render overflowed, else None"""
        key, cap, host, ev, _ = tok
        ev.synchronize()
        tok[4] = True
        n = int(host[0])
        if n > cap * self.REGROW_AT:
            self.note(key, n)
        if host.numel() > 1 and int(host[1]) & 2:     # fused render() node: a depth key outside the 27-bit span
            self.key_bits = 0
            return (n, cap, "a depth key outside the 27-bit span of the three-pass sort (z > ~13107); the full 32-bit "
                            "sort is used from now on")
        return (n, cap) if n > cap else None

    def _error(self, key, n, cap, when, why=None):
        if why is not None:
            return _lib.B3gsError(f"B3GS_ERR_CAPACITY: {when} render of shape {key[1:]} had {why} -- repeat the step")
        return _lib.B3gsError(f"B3GS_ERR_CAPACITY: {when} render of shape {key[1:]} produced {n} tile instances, its "
                              f"binning buffer held {cap} (truncated lists); the capacity is now {self.capacity[key]} "
                              f"-- repeat the step")

    def poll(self, force=False):
        keep, over = [], None
        for tok in self.pending:
            if tok[4]:
                continue
            if not (force or tok[3].query()):
                keep.append(tok)
                continue
            r = self._resolve(tok)
            if r is not None and over is None:
                over = (tok[0],) + r
        self.pending = keep
        if over is not None:
            raise self._error(over[0], over[1], over[2], "an earlier", *over[3:])

    def confirm(self, tok):
        """Backward entry: this render's N must have fitted its buffer."""
        if tok is None or tok[4]:
            return
        r = self._resolve(tok)
        self.pending = [t for t in self.pending if t is not tok]
        if r is not None:
            raise self._error(tok[0], r[0], r[1], "this", *r[2:])

    def track(self, key, cap, n_dev):
        if self.pinned is None:
            self.pinned = torch.zeros((self.RING, 2), dtype=torch.int32).pin_memory()
        if len(self.pending) >= self.RING - 1:
            self.poll(force=True)
        host = self.pinned[self.slot, :n_dev.numel()]     # n_dev: [N] or [N, overflow word]
        self.slot = (self.slot + 1) % self.RING
        host.copy_(n_dev, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(n_dev.device))
        tok = [key, cap, host, ev, False]
        self.pending.append(tok)
        return tok

    def flush_at_exit(self):
        try:
            self.poll(force=True)
        except Exception as exc:   # nothing can repeat the step any more: say so
            import sys
            print(f"[binocular3dgs_amd] WARNING at exit: {exc}", file=sys.stderr)


_lazy = _LazyN()
atexit.register(_lazy.flush_at_exit)


class _CModule:
    """Stands in for the `_C` torch-extension module of the upstream package."""
    last_lazy_token = None   # set by a sync-free forward: what _RasterizeGaussians.backward confirms

    @staticmethod
    def rasterize_gaussians(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                            viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree,
                            campos, prefiltered, debug, lazy_num_rendered=False):
        """-> (num_rendered, color[3,H,W], depth[1,H,W], alpha[1,H,W], radii[P] int32,
               geomBuffer, binningBuffer, imgBuffer)   (uint8 state tensors, opaque)

        lazy_num_rendered (not part of the upstream signature; set by the autograd surface, which never shows
        num_rendered to its caller): sync-free forward, see _LazyN -- the returned count is then the CAPACITY of the
        binning buffer (what the backward needs to find its arrays), not N."""
        L = _lib.lib()
        sc, keep, dev, P, _ = _scene(background, means3D, colors, opacity, scales, rotations, scale_modifier,
                                     cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height,
                                     image_width, sh, degree, campos, prefiltered, debug)
        H, W = int(image_height), int(image_width)
        color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
        depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
        alpha = torch.empty((1, H, W), dtype=torch.float32, device=dev)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        key = (dev.index if dev.index is not None else torch.cuda.current_device(), P, W, H)
        if lazy_num_rendered and _lazy.enabled and not debug and P > 0:
            _lazy.poll()
            cap = _lazy.capacity.get(key)
            if cap is not None:
                u8 = dict(dtype=torch.uint8, device=dev)
                geom = torch.empty((L.b3gs_geometry_bytes(P),), **u8)
                binning = torch.empty((L.b3gs_binning_bytes(P, cap),), **u8)
                img = torch.empty((L.b3gs_image_bytes(W, H),), **u8)
                n_dev = torch.empty((1,), dtype=torch.int32, device=dev)
                with torch.cuda.device(dev):
                    rc = L.b3gs_forward_capacity(C.byref(sc), geom.data_ptr(), binning.data_ptr(), cap, img.data_ptr(),
                                                 color.data_ptr(), depth.data_ptr(), alpha.data_ptr(), _ptr(radii),
                                                 n_dev.data_ptr(), _stream(dev))
                    _lib.check(rc, "b3gs_forward_capacity")
                    _CModule.last_lazy_token = _lazy.track(key, cap, n_dev)
                del keep
                return cap, color, depth, alpha, radii, geom, binning, img
        bufs = {}

        def mk(key):
            def fn(_user, nbytes):
                t = torch.empty((max(int(nbytes), 1),), dtype=torch.uint8, device=dev)
                bufs[key] = t
                return t.data_ptr()
            return _lib.ALLOC_FN(fn)

        cbs = [mk("geom"), mk("binning"), mk("img")]
        n = C.c_int32(0)
        with torch.cuda.device(dev):
            rc = L.b3gs_forward(C.byref(sc), cbs[0], None, cbs[1], None, cbs[2], None, color.data_ptr(),
                                depth.data_ptr(), alpha.data_ptr(), _ptr(radii), C.byref(n), _stream(dev))
        _lib.check(rc, "b3gs_forward")
        del keep
        if P > 0:
            _lazy.note(key, int(n.value))     # every exact render teaches the capacity of its shape
        return (int(n.value), color, depth, alpha, radii, bufs["geom"],
                bufs.get("binning", torch.empty(0, dtype=torch.uint8, device=dev)), bufs["img"])

    @staticmethod
    def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier,
                                     cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color,
                                     dL_dout_depth, dL_dout_alpha, sh, degree, campos, geomBuffer, R, binningBuffer,
                                     imageBuffer, alpha, debug, opacities=None):
        """-> (dL_dmeans2D[P,3], dL_dcolors[P,3], dL_dopacity[P,1], dL_dmeans3D[P,3], dL_dcov3D[P,6],
               dL_dsh[P,M,3], dL_dscales[P,3], dL_drotations[P,4])

        `opacities` is not part of the upstream signature (the kernels read opacity from the saved
        geometry state); it is accepted so the scene struct can be validated the same way."""
        L = _lib.lib()
        P = means3D.shape[0]
        dev = means3D.device
        if opacities is None:
            opacities = torch.empty((P, 1), dtype=torch.float32, device=dev)  # unused by the backward kernels
        H, W = dL_dout_color.shape[-2], dL_dout_color.shape[-1]
        sc, keep, dev, P, M = _scene(background, means3D, colors, opacities, scales, rotations, scale_modifier,
                                     cov3D_precomp, viewmatrix, projmatrix, tan_fovx, tan_fovy, H, W, sh, degree,
                                     campos, False, debug)
        dC = _dev_f32(dL_dout_color, "dL_dout_color")
        dD = None if dL_dout_depth is None or dL_dout_depth.numel() == 0 else _dev_f32(dL_dout_depth, "dL_dout_depth")
        dA = None if dL_dout_alpha is None or dL_dout_alpha.numel() == 0 else _dev_f32(dL_dout_alpha, "dL_dout_alpha")
        f = dict(dtype=torch.float32, device=dev)
        dL_dmeans2D = torch.empty((P, 3), **f)
        dL_dcolors = torch.empty((P, 3), **f)
        dL_dopacity = torch.empty((P, 1), **f)
        dL_dmeans3D = torch.empty((P, 3), **f)
        dL_dcov3D = torch.empty((P, 6), **f)
        dL_dsh = torch.empty((P, M, 3), **f)
        has_sr = sc.scales is not None
        dL_dscales = torch.empty((P, 3) if has_sr else (0, 3), **f)
        dL_drot = torch.empty((P, 4) if has_sr else (0, 4), **f)
        radii_i = radii.to(torch.int32).contiguous()
        with torch.cuda.device(dev):
            rc = L.b3gs_backward(C.byref(sc), int(R), _ptr(radii_i), _ptr(geomBuffer), _ptr(binningBuffer),
                                 _ptr(imageBuffer), _ptr(dC), _ptr(dD), _ptr(dA), _ptr(dL_dmeans2D),
                                 _ptr(dL_dcolors), _ptr(dL_dopacity), _ptr(dL_dmeans3D), _ptr(dL_dcov3D),
                                 _ptr(dL_dsh), _ptr(dL_dscales), _ptr(dL_drot), _stream(dev))
        _lib.check(rc, "b3gs_backward")
        del keep
        return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drot

    @staticmethod
    def mark_visible(means3D, viewmatrix, projmatrix):
        L = _lib.lib()
        m = _dev_f32(means3D, "means3D")
        vm = _dev_f32(viewmatrix, "viewmatrix")
        pm = _dev_f32(projmatrix, "projmatrix")
        present = torch.zeros((m.shape[0],), dtype=torch.bool, device=m.device)
        with torch.cuda.device(m.device):
            rc = L.b3gs_mark_visible(m.shape[0], _ptr(m), _ptr(vm), _ptr(pm), _ptr(present), _stream(m.device))
        _lib.check(rc, "b3gs_mark_visible")
        return present


_C = _CModule()


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    # will this render be differentiated?  (grad mode is invisible inside autograd.Function.forward, and
    # ctx.needs_input_grad reflects requires_grad even under torch.no_grad())
    differentiated = torch.is_grad_enabled() and any(
        torch.is_tensor(t) and t.requires_grad
        for t in (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp))
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings, differentiated)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings, differentiated=None):
        """`differentiated` (set by rasterize_gaussians(); absent when the Function is applied directly with the
        upstream's nine arguments): the caller will run backward() on this render -- only then may the forward skip the
        read-back of num_rendered (_LazyN); otherwise it is the exact synchronous forward."""
        rs = raster_settings
        ctx.n_inputs = 9 if differentiated is None else 10
        # sync-free only when this render will be differentiated: its backward then checks N before any gradient is
        # produced; a render nobody differentiates (evaluation) takes the exact synchronous forward
        _CModule.last_lazy_token = None
        num_rendered, color, depth, alpha, radii, geom, binning, img = _C.rasterize_gaussians(
            rs.bg, means3D, colors_precomp, opacities, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, sh,
            rs.sh_degree, rs.campos, rs.prefiltered, rs.debug, lazy_num_rendered=bool(differentiated))
        ctx.lazy_token, _CModule.last_lazy_token = _CModule.last_lazy_token, None
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom,
                              binning, img, alpha)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        rs = ctx.raster_settings
        _lazy.confirm(ctx.lazy_token)   # raises when this render's tile lists were truncated: no gradient leaves here
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geom, binning, img, alpha = \
            ctx.saved_tensors
        if grad_color is None:
            grad_color = torch.zeros((3, rs.image_height, rs.image_width), dtype=torch.float32,
                                     device=means3D.device)
        empty = torch.empty(0, device=means3D.device)
        (grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh,
         grad_scales, grad_rotations) = _C.rasterize_gaussians_backward(
            rs.bg, means3D, radii, colors_precomp, scales, rotations, rs.scale_modifier, cov3Ds_precomp,
            rs.viewmatrix, rs.projmatrix, rs.tanfovx, rs.tanfovy, grad_color,
            empty if grad_depth is None else grad_depth, empty if grad_alpha is None else grad_alpha, sh,
            rs.sh_degree, rs.campos, geom, ctx.num_rendered, binning, img, alpha, rs.debug)
        needs = ctx.needs_input_grad

        def pick(g, had_input, i):
            return g if (had_input and needs[i]) else None

        return (pick(grad_means3D, True, 0), pick(grad_means2D, True, 1), pick(grad_sh, sh.numel() != 0, 2),
                pick(grad_colors_precomp, colors_precomp.numel() != 0, 3), pick(grad_opacities, True, 4),
                pick(grad_scales, scales.numel() != 0, 5), pick(grad_rotations, rotations.numel() != 0, 6),
                pick(grad_cov3Ds_precomp, cov3Ds_precomp.numel() != 0, 7), None, None)[:ctx.n_inputs]


# ---------------------------------------------------------------------------------------------------------------
# render() handed a GaussianModel: one autograd node on the RAW parameters
# ---------------------------------------------------------------------------------------------------------------
_INPLACE_GRADS = os.environ.get("B3GS_DROPIN_INPLACE_GRADS", "1") != "0"
_raw_scratch = {}     # (device index, P) -> zeroed [P * 10] floats, left clean by every backward (b3gs_backward_raw_accumulate)


def raw_model_ok(pc) -> bool:
    """True when `pc` stores the reference's six raw parameter tensors and activates them the reference's way
    (scene/gaussian_model.py:33-43: exp / sigmoid / normalize; get_features = cat(_features_dc, _features_rest)) -- then
    render() can hand the RAW tensors to the library, which evaluates the activations and their backward in-kernel."""
    try:
        return (pc.scaling_activation is torch.exp and pc.opacity_activation is torch.sigmoid and
                pc.rotation_activation is torch.nn.functional.normalize and
                all(torch.is_tensor(getattr(pc, a)) and getattr(pc, a).is_cuda and getattr(pc, a).dtype == torch.float32
                    and getattr(pc, a).is_contiguous()
                    for a in ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")))
    except AttributeError:
        return False


_word_ring = {}       # device index -> [zeroed int32 ring, next pair]


def _zero_words(dev):
    """Two zeroed int32 words on `dev` without a fill kernel per render: pairs of a ring that is zeroed once per lap (a
    pair handed out is read by the host long before the ring comes round: 4096 renders later)."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    ent = _word_ring.get(idx)
    if ent is None or ent[1] >= ent[0].shape[0]:
        ent = _word_ring[idx] = [torch.zeros((4096, 2), dtype=torch.int32, device=dev), 0]
    w = ent[0][ent[1]]
    ent[1] += 1
    return w


class _RasterizeRaw(torch.autograd.Function):
    """gaussian_renderer/__init__.py:53-93 as ONE node: the accessors (get_scaling / get_rotation / get_opacity /
    get_features: ~30 PyTorch kernels with their autograd per render) run inside the projection and chain-rule kernels
    (b3gs_forward_raw_batch / b3gs_backward_raw_accumulate), the depth sort takes three passes, and the Gaussians are
    binned into the tiles their alpha >= 1/255 footprint reaches (same images; `_C.rasterize_gaussians` keeps the
    reference's binning rule and bit-exact lists).  Gradients are RETURNED to autograd (they land in `.grad` of the six
    parameters and of the `means2D` dummy exactly like the reference's), not written behind its back."""

    @staticmethod
    def forward(ctx, xyz, f_dc, f_rest, scaling, rotation, opacity, means2D, cfg):
        L = _lib.lib()
        dev, P = xyz.device, xyz.shape[0]
        W, H = cfg["W"], cfg["H"]
        K = f_dc.shape[1] + f_rest.shape[1]
        sc = _lib.B3gsScene(P, int(cfg["sh_degree"]), int(K), W, H, float(cfg["tanfovx"]), float(cfg["tanfovy"]),
                            float(cfg["scale_modifier"]), 0, int(bool(cfg["debug"])), cfg["bg"].data_ptr(), None, None, None,
                            None, None, None, None, cfg["viewmatrix"].data_ptr(), cfg["projmatrix"].data_ptr(),
                            cfg["campos"].data_ptr())
        rp = _lib.B3gsRawParams()
        rp.xyz, rp.features_dc = xyz.data_ptr(), f_dc.data_ptr()
        rp.features_rest = f_rest.data_ptr() if f_rest.numel() else None
        rp.scaling, rp.rotation, rp.opacity = scaling.data_ptr(), rotation.data_ptr(), opacity.data_ptr()
        u8 = dict(dtype=torch.uint8, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        key = ("raw", dev.index if dev.index is not None else torch.cuda.current_device(), P, W, H)
        lazy = bool(cfg["differentiated"]) and _lazy.enabled and not cfg["debug"]
        if lazy:
            _lazy.poll()
        geom = torch.empty((L.b3gs_geometry_bytes(P),), **u8)
        # recycled allocator memory: B3gsForwardView::fresh_image tells the library to read nothing from it (the batched
        # forward otherwise trusts a tile-order array it finds behind a signature in a PERSISTENT image buffer)
        img = torch.empty((L.b3gs_image_bytes(W, H),), **u8)
        color, depth, alpha = torch.empty((3, H, W), **f32), torch.empty((1, H, W), **f32), torch.empty((1, H, W), **f32)
        radii = torch.empty((P,), dtype=torch.int32, device=dev)
        cap = _lazy.capacity.get(key) or max(1 << 20, 12 * P)
        while True:
            binning = torch.empty((L.b3gs_binning_bytes(P, cap),), **u8)
            words = _zero_words(dev)                                       # [N, overflow word], zero
            fv = (_lib.B3gsForwardView * 1)()
            fv[0].view = C.pointer(sc)
            fv[0].geometry, fv[0].binning, fv[0].image = geom.data_ptr(), binning.data_ptr(), img.data_ptr()
            fv[0].binning_capacity = cap
            fv[0].out_color, fv[0].out_depth, fv[0].out_alpha = color.data_ptr(), depth.data_ptr(), alpha.data_ptr()
            fv[0].radii, fv[0].device_num_rendered = radii.data_ptr(), words.data_ptr()
            fv[0].depth_order_from, fv[0].seg1_fraction = -1, 0.0
            fv[0].high_water, fv[0].overflow_flag = None, words[1:].data_ptr()
            fv[0].depth_key_bits = _lazy.key_bits
            fv[0].fresh_image = 1
            with torch.cuda.device(dev):
                _lib.check(L.b3gs_forward_raw_batch(1, fv, C.byref(rp), 3, _stream(dev)), "b3gs_forward_raw_batch")
            if lazy and key in _lazy.capacity:
                ctx.lazy_token = _lazy.track(key, cap, words)
                break
            n, flag = (int(v) for v in words.tolist())                    # exact render: one read-back
            _lazy.note(key, n)
            if flag & 2:
                _lazy.key_bits = 0
            ctx.lazy_token = None
            if n <= cap and not (flag & 2):
                break
            cap = max(cap, _lazy.capacity[key])                           # repeat with what it needs
        ctx.cfg, ctx.sc, ctx.rp = cfg, sc, rp
        ctx.save_for_backward(xyz, f_dc, f_rest, scaling, rotation, opacity, radii, geom, binning, img)
        ctx.cap = cap
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        _lazy.confirm(ctx.lazy_token)     # truncated lists / key span: raises before any gradient exists
        L = _lib.lib()
        xyz, f_dc, f_rest, scaling, rotation, opacity, radii, geom, binning, img = ctx.saved_tensors
        cfg, sc, dev, P = ctx.cfg, ctx.sc, xyz.device, xyz.shape[0]
        W, H = cfg["W"], cfg["H"]
        f32 = dict(dtype=torch.float32, device=dev)
        if grad_color is None:
            grad_color = torch.zeros((3, H, W), **f32)
        gc = _dev_f32(grad_color, "dL_dout_color")
        gd = None if grad_depth is None else _dev_f32(grad_depth, "dL_dout_depth")
        ga = None if grad_alpha is None else _dev_f32(grad_alpha, "dL_dout_alpha")
        skey = (dev.index if dev.index is not None else torch.cuda.current_device(), P)
        scratch = _raw_scratch.get(skey)
        if scratch is None:
            if len(_raw_scratch) > 4:
                _raw_scratch.clear()
            scratch = _raw_scratch[skey] = torch.zeros((max(L.b3gs_backward_scratch_floats(P), 1),), **f32)
        bv = (_lib.B3gsBlendView * 1)()
        bv[0].view = C.pointer(sc)
        bv[0].geometry, bv[0].binning, bv[0].image = geom.data_ptr(), binning.data_ptr(), img.data_ptr()
        bv[0].dL_dcolor = gc.data_ptr()
        bv[0].dL_ddepth = None if gd is None else gd.data_ptr()
        bv[0].dL_dalpha = None if ga is None else ga.data_ptr()
        bv[0].scratch, bv[0].binning_capacity = scratch.data_ptr(), ctx.cap
        # Where the six gradients go.  Default: fresh tensors, RETURNED to autograd (AccumulateGrad keeps the first one a
        # parameter receives and adds the later ones: one more pass over 92 B per Gaussian per render).  When every
        # parameter is a plain leaf that already holds a dense fp32 `.grad` of its own shape and nothing hooks into its
        # gradient, the chain-rule kernel adds into `.grad` directly (`+=`, what AccumulateGrad would do) and the node
        # returns None for them -- same values, one pass less.  B3GS_DROPIN_INPLACE_GRADS=0 disables it.
        params = (xyz, f_dc, f_rest, scaling, rotation, opacity)
        inplace = _INPLACE_GRADS and all(
            (not ctx.needs_input_grad[i]) or
            (t.is_leaf and t.grad is not None and t.grad.dtype == torch.float32 and t.grad.shape == t.shape and
             t.grad.is_contiguous() and t.grad.device == t.device and not t._backward_hooks and
             not getattr(t, "_post_accumulate_grad_hooks", None))
            for i, t in enumerate(params)) and all(ctx.needs_input_grad[:6])
        grads = [t.grad if inplace else torch.empty_like(t) for t in params]
        g_m2d = torch.empty((P, 3), **f32)
        gr = _lib.B3gsRawGrads()
        gr.xyz, gr.features_dc = grads[0].data_ptr(), grads[1].data_ptr()
        gr.features_rest = grads[2].data_ptr() if grads[2].numel() else None
        gr.scaling, gr.rotation, gr.opacity = grads[3].data_ptr(), grads[4].data_ptr(), grads[5].data_ptr()
        gr.touched_rows = None
        av = (_lib.B3gsFusedView * 1)()
        av[0].view = C.pointer(sc)
        av[0].radii, av[0].geometry, av[0].scratch = radii.data_ptr(), geom.data_ptr(), scratch.data_ptr()
        av[0].dL_dmeans2D, av[0].densify_stats = g_m2d.data_ptr(), 0
        with torch.cuda.device(dev):
            s = _stream(dev)
            _lib.check(L.b3gs_blend_backward_batch(1, bv, s), "b3gs_blend_backward_batch")
            # overwrite mode: every row of every gradient tensor is stored (zeros for Gaussians without a contribution);
            # accumulate mode (in place): only the rows that received something are touched
            _lib.check(L.b3gs_backward_raw_accumulate(1, av, C.byref(ctx.rp), C.byref(gr), 0 if inplace else 1, None, s),
                       "b3gs_backward_raw_accumulate")
        needs = ctx.needs_input_grad
        out = [None if (inplace or not needs[i]) else g for i, g in enumerate(grads)]
        return (*out, g_m2d if needs[6] else None, None)


def rasterize_raw(pc, means2D, raster_settings):
    """render()'s fast path: `pc` = a model raw_model_ok() accepts.  -> (color, radii, depth, alpha)."""
    rs = raster_settings
    dev = pc._xyz.device
    t = [_dev_f32(x, n).reshape(-1) for x, n in ((rs.bg, "bg"), (rs.viewmatrix, "viewmatrix"), (rs.projmatrix, "projmatrix"),
                                                 (rs.campos, "campos"))]
    if t[0].numel() != 3 or t[1].numel() != 16 or t[2].numel() != 16 or t[3].numel() != 3:
        raise ValueError("bg/campos must have 3 and viewmatrix/projmatrix 16 elements")
    params = (pc._xyz, pc._features_dc, pc._features_rest, pc._scaling, pc._rotation, pc._opacity)
    cfg = dict(W=int(rs.image_width), H=int(rs.image_height), tanfovx=rs.tanfovx, tanfovy=rs.tanfovy, bg=t[0],
               scale_modifier=rs.scale_modifier, viewmatrix=t[1], projmatrix=t[2], campos=t[3], sh_degree=rs.sh_degree,
               debug=rs.debug,
               differentiated=torch.is_grad_enabled() and any(p.requires_grad for p in params + (means2D,)))
    del dev
    return _RasterizeRaw.apply(*params, means2D, cfg)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            return _C.mark_visible(positions, rs.viewmatrix, rs.projmatrix)

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        e = torch.Tensor([]).to(means3D.device)
        shs = e if shs is None else shs
        colors_precomp = e if colors_precomp is None else colors_precomp
        scales = e if scales is None else scales
        rotations = e if rotations is None else rotations
        cov3D_precomp = e if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, self.raster_settings)
