"""Fast path of the build's own training step: fused activations, persistent scratch, no host sync,
views rendered concurrently on separate HIP streams.

`FusedRasterizer.render()` returns the same dict as the drop-in `render()` (render.py /
gaussian_renderer/__init__.py:97-103) but goes through b3gs_forward_raw / b3gs_backward_raw:
  * the parameter accessors of scene/gaussian_model.py:95-115 (exp / normalize / sigmoid / cat) and
    their backward run inside the per-Gaussian HIP kernels -- no PyTorch elementwise kernels,
  * gradients are accumulated (+=) directly into the `.grad` views of the flat gradient slab
    (step.FlatGradSlab): no per-view AccumulateGrad adds, nothing to pack before the all-reduce,
  * geometry / binning / image state lives in persistent per-slot buffers sized once (288 GB of HBM
    make over-allocation free), N stays on the device: zero allocations and zero host syncs per
    view, so a whole iteration can be captured in one HIP graph (torch.cuda.graph),
  * every slot owns a HIP stream: the views of an iteration are independent until their gradients
    meet in the slab, and the binning kernels (a few hundred workgroups each) leave most of the 256
    CUs idle, so running the 6 views of an iteration concurrently hides them behind other views'
    blend kernels.  Autograd runs each view's backward on the stream its forward used; the
    non-atomic `+=` into the shared slab is ordered across streams by an event chain.
Results equal the drop-in path up to activation rounding (tests/test_gpu_fused.py).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence

import torch

from . import _lib


class _Slot:
    def __init__(self, P, W, H, capacity, dev, want_means2D, scratch_floats, stream):
        L = _lib.lib()
        u8 = dict(dtype=torch.uint8, device=dev)
        self.geom = torch.empty(L.b3gs_geometry_bytes(P), **u8)
        self.binning = torch.empty(L.b3gs_binning_bytes(P, capacity), **u8)
        self.img = torch.empty(L.b3gs_image_bytes(W, H), **u8)
        self.capacity = capacity
        f = dict(dtype=torch.float32, device=dev)
        self.color = torch.empty((3, H, W), **f)
        self.depth = torch.empty((1, H, W), **f)
        self.alpha = torch.empty((1, H, W), **f)
        self.radii = torch.zeros((P,), dtype=torch.int32, device=dev)
        self.n_dev = torch.zeros((1,), dtype=torch.int32, device=dev)
        self.means2D_grad = torch.zeros((P, 3), **f) if want_means2D else None
        self.scratch = torch.zeros(max(scratch_floats, 1), **f)   # zero on entry / exit of every backward
        self.stream = stream


class _ViewOutputs(dict):
    """render() dict whose `visibility_filter` (= radii > 0, gaussian_renderer/__init__.py:99) is only
    materialised when somebody reads it (one elementwise kernel per view otherwise wasted)."""

    def __missing__(self, key):
        if key == "visibility_filter":
            self[key] = self["radii"] > 0
            return self[key]
        raise KeyError(key)


class _RasterizeRaw(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, f_dc, f_rest, scaling, rotation, opacity, owner, slot_idx, view):
        owner._forward(slot_idx, view)
        s = owner.slots[slot_idx]
        ctx.owner, ctx.slot_idx, ctx.view = owner, slot_idx, view
        ctx.mark_non_differentiable(s.radii)
        return s.color, s.radii, s.depth, s.alpha

    @staticmethod
    def backward(ctx, g_color, g_radii, g_depth, g_alpha):
        ctx.owner._backward(ctx.slot_idx, ctx.view, g_color, g_depth, g_alpha)
        # gradients were accumulated in place into the parameters' .grad (flat slab views)
        return (None,) * 9


class FusedRasterizer:
    def __init__(self, model, width: int, height: int, num_slots: int = 2, binning_capacity: Optional[int] = None,
                 want_means2D: bool = True, concurrent: bool = True):
        self.model = model
        self.W, self.H = int(width), int(height)
        p = model.get_xyz
        if not p.is_cuda:
            raise _lib.B3gsError("FusedRasterizer needs the model on an MI355X (HIP) device")
        self.dev = p.device
        self.P = p.shape[0]
        self.K = model._features_dc.shape[1] + model._features_rest.shape[1]
        self.capacity = int(binning_capacity) if binning_capacity else max(4_000_000, 12 * self.P)
        self.concurrent = bool(concurrent)
        self._want_m2d = want_means2D
        self.slots: List[_Slot] = []
        for _ in range(num_slots):
            self.slots.append(self._new_slot(torch.cuda.Stream(self.dev) if self.concurrent else None))
        self._acc_event: Optional[torch.cuda.Event] = None   # tail of the accumulate chain
        self._deferred = None   # list of (slot_idx, B3gsScene) while a deferred-accumulate section is open
        self._params = _lib.B3gsRawParams()
        self._grads = _lib.B3gsRawGrads()

    def _new_slot(self, stream):
        L = _lib.lib()
        return _Slot(self.P, self.W, self.H, self.capacity, self.dev, self._want_m2d,
                     L.b3gs_backward_scratch_floats(self.P), stream)

    # ---- C-ABI calls ------------------------------------------------------------------------
    def _scene(self, view) -> _lib.B3gsScene:
        cam, bg, scaling_modifier, debug = view[:4]
        m = self.model
        return _lib.B3gsScene(self.P, int(m.active_sh_degree), int(self.K), self.W, self.H,
                              math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), float(scaling_modifier), 0,
                              int(bool(debug)), bg.data_ptr(), None, None, None, None, None, None, None,
                              cam.world_view_transform.data_ptr(), cam.full_proj_transform.data_ptr(),
                              cam.camera_center.data_ptr())

    def _bind_params(self):
        m, rp = self.model, self._params
        rp.xyz, rp.features_dc = m._xyz.data_ptr(), m._features_dc.data_ptr()
        rp.features_rest = m._features_rest.data_ptr() if m._features_rest.numel() else None
        rp.scaling, rp.rotation, rp.opacity = m._scaling.data_ptr(), m._rotation.data_ptr(), m._opacity.data_ptr()
        return rp

    def _forward(self, slot_idx, view):
        L, s = _lib.lib(), self.slots[slot_idx]
        sc = self._scene(view)
        rc = L.b3gs_forward_raw(C.byref(sc), C.byref(self._bind_params()), s.geom.data_ptr(), s.binning.data_ptr(),
                                s.capacity, s.img.data_ptr(), s.color.data_ptr(), s.depth.data_ptr(),
                                s.alpha.data_ptr(), s.radii.data_ptr(), s.n_dev.data_ptr(),
                                torch.cuda.current_stream(self.dev).cuda_stream)
        _lib.check(rc, "b3gs_forward_raw")

    def _backward(self, slot_idx, view, g_color, g_depth, g_alpha):
        L, s, m = _lib.lib(), self.slots[slot_idx], self.model
        sc = self._scene(view)
        gr = self._grads
        for name, p in (("xyz", m._xyz), ("features_dc", m._features_dc), ("features_rest", m._features_rest),
                        ("scaling", m._scaling), ("rotation", m._rotation), ("opacity", m._opacity)):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            setattr(gr, name, p.grad.data_ptr() if p.numel() else None)
        if g_color is None:
            g_color = torch.zeros((3, self.H, self.W), dtype=torch.float32, device=self.dev)
        gc = g_color.contiguous()
        gd = None if g_depth is None else g_depth.contiguous()
        ga = None if g_alpha is None else g_alpha.contiguous()
        stream = torch.cuda.current_stream(self.dev)   # autograd runs this node on the forward's stream

        def call(phases):
            rc = L.b3gs_backward_raw(C.byref(sc), C.byref(self._bind_params()), s.radii.data_ptr(), s.geom.data_ptr(),
                                     s.binning.data_ptr(), s.img.data_ptr(), gc.data_ptr(),
                                     None if gd is None else gd.data_ptr(), None if ga is None else ga.data_ptr(),
                                     s.scratch.data_ptr(), C.byref(gr),
                                     None if s.means2D_grad is None else s.means2D_grad.data_ptr(), phases,
                                     stream.cuda_stream)
            _lib.check(rc, "b3gs_backward_raw")

        if self._deferred is not None:            # phase 2 of all views is fused into finish_deferred()
            call(1)
            self._deferred.append((slot_idx, sc, bool(view[4]) if len(view) > 4 else False))
            return
        if not self.concurrent:
            call(3)
            return
        call(1)                                   # blend backward: private scratch, runs concurrently
        if self._acc_event is not None:
            stream.wait_event(self._acc_event)    # the += into the shared slab is not atomic: one view at a time
        call(2)
        ev = torch.cuda.Event()
        ev.record(stream)
        self._acc_event = ev

    # ---- public -----------------------------------------------------------------------------
    def _render_on_current_stream(self, viewpoint_camera, bg_color, slot, scaling_modifier, debug,
                                  densify_stats=False) -> dict:
        m = self.model
        view = (viewpoint_camera, bg_color, scaling_modifier, debug, densify_stats)
        color, radii, depth, alpha = _RasterizeRaw.apply(m._xyz, m._features_dc, m._features_rest, m._scaling,
                                                         m._rotation, m._opacity, self, slot, view)
        s = self.slots[slot]
        return _ViewOutputs({"render": color, "viewspace_points_grad": s.means2D_grad, "radii": radii,
                             "rendered_depth": depth, "rendered_alpha": alpha})

    def render(self, viewpoint_camera, bg_color: torch.Tensor, slot: int = 0, scaling_modifier: float = 1.0,
               debug: bool = False) -> dict:
        """One view.  Same keys as render(); `viewspace_points_grad` ([P,3], filled by backward) replaces
        the `.grad` of the reference's dummy `viewspace_points` tensor."""
        return self.render_batch([(viewpoint_camera, slot)], bg_color, scaling_modifier, debug)[0]

    def render_batch(self, views: Sequence, bg_color: torch.Tensor, scaling_modifier: float = 1.0,
                     debug: bool = False) -> List[dict]:
        """Render [(camera, slot[, densify_stats]), ...] concurrently (one stream per slot); views with
        densify_stats=True feed the model's densification statistics (init_densification_stats) in
        finish_deferred(), as the primary view does at train.py:178-179.  On return the current stream
        has been made to wait for all of them, so the outputs can be consumed normally.  Calling
        backward ONCE on a loss that depends on several of the views lets autograd run their backward
        passes concurrently as well."""
        views = [(v[0], v[1], (v[2] if len(v) > 2 else False)) for v in views]
        if not self.concurrent:
            return [self._render_on_current_stream(cam, bg_color, slot, scaling_modifier, debug, ds)
                    for cam, slot, ds in views]
        main = torch.cuda.current_stream(self.dev)
        out = []
        self._acc_event = None
        for cam, slot, ds in views:
            st = self.slots[slot].stream
            st.wait_stream(main)
            with torch.cuda.stream(st):
                out.append(self._render_on_current_stream(cam, bg_color, slot, scaling_modifier, debug, ds))
        for _, slot, _ds in views:
            main.wait_stream(self.slots[slot].stream)
        return out

    def begin_deferred(self):
        """Open a section in which every view's backward only runs the blend backward (phase 1, own
        scratch, own stream); finish_deferred() then does the per-Gaussian chain rule of ALL those
        views in one kernel (b3gs_backward_raw_accumulate)."""
        self._deferred = []

    def finish_deferred(self, overwrite: bool = True):
        """Join the view streams and accumulate every deferred view in one pass over the Gaussians.
        overwrite=True stores the gradients (no zero-fill of the slab needed), False adds to them."""
        pend, self._deferred = self._deferred, None
        self.join()
        if not pend:
            if overwrite:
                for p in self.model.parameters():
                    if p.grad is not None:
                        p.grad.zero_()
            return
        L, m = _lib.lib(), self.model
        gr = self._grads
        for name, p in (("xyz", m._xyz), ("features_dc", m._features_dc), ("features_rest", m._features_rest),
                        ("scaling", m._scaling), ("rotation", m._rotation), ("opacity", m._opacity)):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            setattr(gr, name, p.grad.data_ptr() if p.numel() else None)
        stats = None
        if getattr(m, "denom", None) is not None and m.denom.numel() == self.P:
            stats = _lib.B3gsDensifyStats(m.xyz_gradient_accum.data_ptr(), m.denom.data_ptr(), m.max_radii2D.data_ptr())
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        for c0 in range(0, len(pend), 8):
            chunk = pend[c0:c0 + 8]
            arr = (_lib.B3gsFusedView * len(chunk))()
            for k, (slot_idx, sc, want_stats) in enumerate(chunk):
                sl = self.slots[slot_idx]
                arr[k].densify_stats = int(want_stats and stats is not None)
                arr[k].view = C.pointer(sc)
                arr[k].radii, arr[k].geometry = sl.radii.data_ptr(), sl.geom.data_ptr()
                arr[k].scratch = sl.scratch.data_ptr()
                arr[k].dL_dmeans2D = None if sl.means2D_grad is None else sl.means2D_grad.data_ptr()
            rc = L.b3gs_backward_raw_accumulate(len(chunk), arr, C.byref(self._bind_params()), C.byref(gr),
                                                int(bool(overwrite) and c0 == 0),
                                                None if stats is None else C.byref(stats), stream)
            _lib.check(rc, "b3gs_backward_raw_accumulate")

    def join(self):
        """Make the current stream wait for every slot stream (call before consuming the slab)."""
        if self.concurrent:
            main = torch.cuda.current_stream(self.dev)
            for s in self.slots:
                main.wait_stream(s.stream)

    def num_rendered(self):
        """Host copy of every slot's N (synchronises).  N > capacity means that view was rendered
        from a truncated list: call grow() and repeat the step."""
        torch.cuda.synchronize(self.dev)
        return [int(s.n_dev.item()) for s in self.slots]

    def overflowed(self) -> bool:
        return any(n > self.capacity for n in self.num_rendered())

    def grow(self, factor: float = 1.5):
        need = max(self.num_rendered() + [self.capacity])
        self.capacity = int(need * factor)
        for i in range(len(self.slots)):
            self.slots[i] = self._new_slot(self.slots[i].stream)
