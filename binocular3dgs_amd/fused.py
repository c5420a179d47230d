"""Fast path of the build's own training step: fused activations, persistent scratch, no host sync,
every stage of the forward and of the backward ONE launch for all views of an iteration.

`FusedRasterizer.render_batch()` returns, per view, the same dict as the drop-in `render()`
(render.py / gaussian_renderer/__init__.py:97-103) but goes through the raw-parameter entry points of
include/b3gs_raster.h:
  * the parameter accessors of scene/gaussian_model.py:95-115 (exp / normalize / sigmoid / cat) and
    their backward run inside the per-Gaussian HIP kernels -- no PyTorch elementwise kernels;
  * geometry / binning / image state lives in persistent per-slot buffers sized once (288 GB of HBM
    make over-allocation free), N stays on the device: zero allocations and zero host syncs per
    view, so a whole iteration can be captured in one HIP graph (torch.cuda.graph);
  * forward: ONE b3gs_forward_raw_batch for all views (a single view's binning kernels are a few hundred
    workgroups each and leave most of the 256 CUs idle): multi-view projection, batched radix / scan / emit /
    blend launches, one depth sort per binocular pair (schedule="batched"; "streams" = one HIP stream per
    view and "serial" are kept for A/B);
  * backward: ONE blend-backward launch for all views (b3gs_blend_backward_batch, per-view scratch),
    then ONE pass over the Gaussians for all views (b3gs_backward_raw_accumulate): gradients are
    stored (or accumulated) straight into the `.grad` views of the flat slab, together with the
    densification statistics of the views that ask for them.
Results equal the drop-in path up to activation rounding (tests/test_gpu_fused.py).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence

import torch

from . import _lib

MAX_BATCH = 8   # views per batched launch (B3GS_MAX_FUSED_VIEWS)


class _Slot:
    def __init__(self, P, W, H, capacity, dev, want_means2D, scratch_floats, stream, n_dev):
        L = _lib.lib()
        u8 = dict(dtype=torch.uint8, device=dev)
        self.geom = torch.empty(L.b3gs_geometry_bytes(P), **u8)
        self.binning = torch.empty(L.b3gs_binning_bytes(P, capacity), **u8)
        # zeroed: the image buffer carries the open-tile prediction of two-round binning from one forward to the next
        # (any content is safe -- the prediction only moves work between the binning rounds -- but zero is deterministic)
        self.img = torch.zeros(L.b3gs_image_bytes(W, H), **u8)
        self.capacity = capacity
        f = dict(dtype=torch.float32, device=dev)
        self.color = torch.empty((3, H, W), **f)
        self.depth = torch.empty((1, H, W), **f)
        self.alpha = torch.empty((1, H, W), **f)
        self.radii = torch.zeros((P,), dtype=torch.int32, device=dev)
        self.n_dev = n_dev      # [1] int32 view into the rasterizer's per-slot N array (device-side num_rendered)
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


class _RasterizeBatch(torch.autograd.Function):
    """All views of a render_batch() call as one autograd node: 4 outputs per view
    (colour, radii, depth, alpha); the backward receives every view's pixel gradients at once."""

    @staticmethod
    def forward(ctx, xyz, f_dc, f_rest, scaling, rotation, opacity, owner, specs):
        owner._forward_batch(specs)
        ctx.owner, ctx.specs = owner, specs
        ctx.set_materialize_grads(False)   # outputs without an upstream gradient arrive as None, not as zero images
        outs = []
        for sp in specs:
            s = owner.slots[sp["slot"]]
            ctx.mark_non_differentiable(s.radii)
            outs += [s.color, s.radii, s.depth, s.alpha]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        per_view = [(grads[4 * k], grads[4 * k + 2], grads[4 * k + 3]) for k in range(len(ctx.specs))]
        ctx.owner._backward_batch(ctx.specs, per_view)
        # gradients are written in place into the parameters' .grad (flat slab views)
        return (None,) * 8


class FusedRasterizer:
    def __init__(self, model, width: int, height: int, num_slots: int = 2, binning_capacity: Optional[int] = None,
                 want_means2D: bool = True, concurrent: bool = True, schedule: Optional[str] = None,
                 seg1_fraction="auto", reference_binning: bool = False):
        self.model = model
        # reference_binning: bin every tile of the reference's 3-sigma rectangle instead of the tiles the alpha >= 1/255
        # footprint reaches (B3gsForwardView::reference_binning, ABI 10): tile lists bit-identical to the reference's rule
        # (north_star: "tile/bin indices bit-exact") at the price of ~3x the instances; images and gradients are the same
        self.reference_binning = bool(reference_binning)
        self.W, self.H = int(width), int(height)
        p = model.get_xyz
        if not p.is_cuda:
            raise _lib.B3gsError("FusedRasterizer needs the model on an MI355X (HIP) device")
        self.dev = p.device
        self.P = p.shape[0]
        self.K = model._features_dc.shape[1] + model._features_rest.shape[1]
        self.capacity = int(binning_capacity) if binning_capacity else max(4_000_000, 12 * self.P)
        # how the views of one render_batch() are issued:
        #   "batched": every stage ONE launch for all views (b3gs_forward_raw_batch), binocular pairs share
        #              one depth sort;  "streams": each view's forward on its own stream;  "serial": one stream,
        #              per-view projection + binning, one blend launch
        self.schedule = schedule or ("batched" if concurrent else "serial")
        assert self.schedule in ("batched", "groups", "streams", "serial")
        self.concurrent = self.schedule == "streams"
        self._want_m2d = want_means2D
        # Two-round binning (schedule "batched"): bin the nearest seg1_fraction of the depth order, blend, bin the rest
        # only into the tiles that are not finished (B3gsForwardView.seg1_fraction; 0 or >= 1: one round).  It removes
        # the emission and the tile split of every instance behind a tile's saturation point.  The tiles a slot's previous
        # forward left unterminated -- or saw terminate only in the last quarter of their segment-1 prefix -- are predicted
        # open and get their complete list in round 1; the second round is ONE persistent launch (binning.hip
        # repair_kernel) that exits at once when the prediction held.  "auto" is a deterministic rule applied whenever
        # fit_capacity() runs: two rounds from `two_round_min_instances` tile instances per view on (measured on MI355X,
        # 800x600, 6 views, tools/sizes_ab.sh: 0.44M instances per view -3 %, 1.2M -2 %, 2.4M +3 %, 4.8M +7 %, 9.6M
        # +16 %), with segment 1 sized for ~0.75M instances of a view.
        self.two_round_min_instances = 2_000_000
        self.two_round_max_miss_rate = 0.3
        self.two_round_disabled = None     # (missed, total) when check_overflow() sent "auto" back to one round
        self._seg1_auto = seg1_fraction == "auto"
        self.seg1_fraction = 0.0 if self._seg1_auto else float(seg1_fraction)
        # N of every slot's last forward lives on the device (no read-back per view); `high_water` keeps the largest N
        # seen since the last check_overflow(), updated by one tiny kernel per forward (graph-capturable)
        self._n_all = torch.zeros((num_slots,), dtype=torch.int32, device=self.dev)
        self.high_water = torch.zeros((num_slots,), dtype=torch.int32, device=self.dev)
        # ... and a sticky device flag set by the binning kernels when a view's N exceeded the capacity (truncated lists):
        # the optimiser and the densification statistics skip every step while it is set (B3gsForwardView::overflow_flag,
        # b3gs_adam_step(skip_if_nonzero)), so check_overflow() finds the state of the last complete step
        self.overflow_flag = torch.zeros((1,), dtype=torch.int32, device=self.dev)
        # depth sort on 27-bit keys (three 9-bit passes instead of four 8-bit ones): valid while every visible Gaussian
        # has view z < ~13107; the projection checks it and raises bit 1 of overflow_flag otherwise -> check_overflow()
        # falls back to the full 32-bit sort for good
        self.depth_key_bits = 27
        self.slots: List[_Slot] = [self._new_slot(torch.cuda.Stream(self.dev), k) for k in range(num_slots)]
        self._deferred = None   # [(spec, B3gsScene)] while a deferred-accumulate section is open
        self._params = _lib.B3gsRawParams()
        self._grads = _lib.B3gsRawGrads()

    def _new_slot(self, stream, index):
        L = _lib.lib()
        return _Slot(self.P, self.W, self.H, self.capacity, self.dev, self._want_m2d,
                     L.b3gs_backward_scratch_floats(self.P), stream, self._n_all[index:index + 1])

    # ---- C-ABI plumbing -----------------------------------------------------------------------
    def _scene(self, sp) -> _lib.B3gsScene:
        cam, m = sp["cam"], self.model
        return _lib.B3gsScene(self.P, int(m.active_sh_degree), int(self.K), self.W, self.H,
                              math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), float(sp["scaling_modifier"]), 0,
                              int(bool(sp["debug"])), sp["bg"].data_ptr(), None, None, None, None, None, None, None,
                              cam.world_view_transform.data_ptr(), cam.full_proj_transform.data_ptr(),
                              cam.camera_center.data_ptr())

    def _bind_params(self):
        m, rp = self.model, self._params
        rp.xyz, rp.features_dc = m._xyz.data_ptr(), m._features_dc.data_ptr()
        rp.features_rest = m._features_rest.data_ptr() if m._features_rest.numel() else None
        rp.scaling, rp.rotation, rp.opacity = m._scaling.data_ptr(), m._rotation.data_ptr(), m._opacity.data_ptr()
        return rp

    def _bind_grads(self):
        m, gr = self.model, self._grads
        for name, p in (("xyz", m._xyz), ("features_dc", m._features_dc), ("features_rest", m._features_rest),
                        ("scaling", m._scaling), ("rotation", m._rotation), ("opacity", m._opacity)):
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            setattr(gr, name, p.grad.data_ptr() if p.numel() else None)
        return gr

    def _blend_table(self, specs, scenes, grads=None):
        arr = (_lib.B3gsBlendView * len(specs))()
        keep = []
        for k, (sp, sc) in enumerate(zip(specs, scenes)):
            s = self.slots[sp["slot"]]
            arr[k].view = C.pointer(sc)
            arr[k].geometry, arr[k].binning, arr[k].image = s.geom.data_ptr(), s.binning.data_ptr(), s.img.data_ptr()
            arr[k].out_color, arr[k].out_depth, arr[k].out_alpha = s.color.data_ptr(), s.depth.data_ptr(), s.alpha.data_ptr()
            arr[k].binning_capacity = s.capacity
            if grads is not None:
                gc, gd, ga = grads[k]
                if gc is None:
                    gc = torch.zeros((3, self.H, self.W), dtype=torch.float32, device=self.dev)
                gc = gc.contiguous()
                gd = None if gd is None else gd.contiguous()
                ga = None if ga is None else ga.contiguous()
                keep += [gc, gd, ga]
                arr[k].dL_dcolor = gc.data_ptr()
                arr[k].dL_ddepth = None if gd is None else gd.data_ptr()
                arr[k].dL_dalpha = None if ga is None else ga.data_ptr()
                arr[k].scratch = s.scratch.data_ptr()
        return arr, keep

    def _forward_batch(self, specs):
        self._forward_batch_launch(specs)
        if self.schedule not in ("batched", "groups"):    # (the batched launches update both words inside the binning kernels)
            torch.maximum(self.high_water, self._n_all, out=self.high_water)
            self.overflow_flag.bitwise_or_((self._n_all > self.capacity).any().to(torch.int32))

    def _forward_batch_launch(self, specs):
        L = _lib.lib()
        main = torch.cuda.current_stream(self.dev)
        scenes = [self._scene(sp) for sp in specs]
        rp = self._bind_params()
        if self.schedule in ("batched", "groups"):
            # "groups" (experiment, round 4): the views are cut into `self.groups` batches that run their whole forward on
            # streams of their own -- the ~25 launches of a forward are a serial chain of mostly latency-bound kernels, two
            # independent chains interleave in each other's ramp-up / drain gaps.  Pairs stay together (shared depth sort).
            if self.schedule == "groups" and len(specs) > 2:
                # (never more views per launch than the library takes: more than 16 views in two groups are cut further)
                per = min(MAX_BATCH - MAX_BATCH % 2, max(2, 2 * (-(-len(specs) // (2 * getattr(self, "groups", 2))))))
                bounds = list(range(0, len(specs), per))
            else:
                per, bounds = MAX_BATCH, list(range(0, len(specs), MAX_BATCH))
            streams = [main] + [self.slots[k].stream for k in range(1, len(bounds))] if len(bounds) > 1 and self.schedule == "groups" else None
            for gi, c0 in enumerate(bounds):
                chunk = specs[c0:c0 + per]
                st = main
                if streams is not None:
                    st = streams[gi]
                    if st is not main:
                        st.wait_stream(main)
                arr = (_lib.B3gsForwardView * len(chunk))()
                for k, sp in enumerate(chunk):
                    s = self.slots[sp["slot"]]
                    arr[k].view = C.pointer(scenes[c0 + k])
                    arr[k].geometry, arr[k].binning, arr[k].image = s.geom.data_ptr(), s.binning.data_ptr(), s.img.data_ptr()
                    arr[k].binning_capacity = s.capacity
                    arr[k].out_color, arr[k].out_depth, arr[k].out_alpha = (s.color.data_ptr(), s.depth.data_ptr(),
                                                                            s.alpha.data_ptr())
                    arr[k].radii, arr[k].device_num_rendered = s.radii.data_ptr(), s.n_dev.data_ptr()
                    # a camera made by Camera.shifted() has the z row of its parent's view matrix: same depth order
                    donor = getattr(sp["cam"], "same_depth_as", None)
                    arr[k].depth_order_from = -1
                    arr[k].seg1_fraction = self.seg1_fraction
                    arr[k].reference_binning = int(self.reference_binning)
                    arr[k].high_water = self.high_water[sp["slot"]:sp["slot"] + 1].data_ptr()
                    arr[k].overflow_flag = self.overflow_flag.data_ptr()
                    arr[k].depth_key_bits = sp.get("key_bits", self.depth_key_bits)
                    for j in range(k):
                        if donor is not None and chunk[j]["cam"] is donor and arr[j].depth_order_from == -1:
                            arr[k].depth_order_from = j
                            # (ABI 8: for an adjacent pair the projection compares the two depth keys of every Gaussian -- a
                            # camera that names a `same_depth_as` whose z row it does not have raises bit 3: check_overflow)
                            arr[k].hint_trusted = 1 if j == k - 1 else 0
                rc = L.b3gs_forward_raw_batch(len(chunk), arr, C.byref(rp), 3, st.cuda_stream)
                _lib.check(rc, "b3gs_forward_raw_batch")
            if streams is not None:
                for st in streams[1:]:
                    main.wait_stream(st)
            return
        if self.concurrent:
            # Every view runs its whole forward on its own stream.  Measured (1M Gaussians, 6 views): the
            # binning kernels of one view overlap well with the BLEND kernel of another (forward phase
            # 1.79 ms), but not with each other (binning on 6 streams + one batched blend: 2.26 ms).
            for sp, sc in zip(specs, scenes):
                s = self.slots[sp["slot"]]
                s.stream.wait_stream(main)
                rc = L.b3gs_forward_raw(C.byref(sc), C.byref(rp), s.geom.data_ptr(), s.binning.data_ptr(), s.capacity,
                                        s.img.data_ptr(), s.color.data_ptr(), s.depth.data_ptr(), s.alpha.data_ptr(),
                                        s.radii.data_ptr(), s.n_dev.data_ptr(), 3, s.stream.cuda_stream)
                _lib.check(rc, "b3gs_forward_raw")
            for sp in specs:
                main.wait_stream(self.slots[sp["slot"]].stream)
            return
        # single stream: projection + binning per view, then one blend launch per <= 8 views
        for sp, sc in zip(specs, scenes):
            s = self.slots[sp["slot"]]
            rc = L.b3gs_forward_raw(C.byref(sc), C.byref(rp), s.geom.data_ptr(), s.binning.data_ptr(), s.capacity,
                                    s.img.data_ptr(), None, None, None, s.radii.data_ptr(), s.n_dev.data_ptr(), 1,
                                    main.cuda_stream)
            _lib.check(rc, "b3gs_forward_raw(phase 1)")
        for c0 in range(0, len(specs), MAX_BATCH):
            arr, _ = self._blend_table(specs[c0:c0 + MAX_BATCH], scenes[c0:c0 + MAX_BATCH])
            _lib.check(L.b3gs_blend_forward_batch(len(arr), arr, main.cuda_stream), "b3gs_blend_forward_batch")

    def _backward_batch(self, specs, grads):
        L = _lib.lib()
        stream = torch.cuda.current_stream(self.dev)
        scenes = [self._scene(sp) for sp in specs]
        for c0 in range(0, len(specs), MAX_BATCH):
            arr, keep = self._blend_table(specs[c0:c0 + MAX_BATCH], scenes[c0:c0 + MAX_BATCH], grads[c0:c0 + MAX_BATCH])
            _lib.check(L.b3gs_blend_backward_batch(len(arr), arr, stream.cuda_stream), "b3gs_blend_backward_batch")
            del keep
        pend = list(zip(specs, scenes))
        if self._deferred is not None:
            self._deferred += pend            # per-Gaussian pass happens once, in finish_deferred()
        else:
            self._accumulate(pend, overwrite=False)

    def _accumulate(self, pend, overwrite: bool, grads=None, first: int = 0, count: Optional[int] = None,
                    touched_rows: Optional[torch.Tensor] = None, stats_target=None):
        """grads: a B3gsRawGrads whose pointers are indexed by the global Gaussian index (default: the parameters'
        .grad); [first, first+count): the Gaussians to process (default: all).  touched_rows (int64, ceil(P/64) words):
        sparse-row mode of B3gsRawGrads -- rows of Gaussians without any gradient are NOT stored, their bit is clear."""
        L, m = _lib.lib(), self.model
        gr = grads if grads is not None else self._bind_grads()
        if touched_rows is not None:
            if not overwrite or len(pend) > MAX_BATCH:
                raise _lib.B3gsError("sparse gradient rows need overwrite mode and at most 8 views (one accumulate call)")
            assert touched_rows.dtype == torch.int64 and touched_rows.numel() >= (self.P + 63) // 64
        if grads is None or touched_rows is not None:
            gr.touched_rows = None if touched_rows is None else touched_rows.data_ptr()
        count = self.P - first if count is None else count
        stats = None
        if getattr(m, "denom", None) is not None and m.denom.numel() == self.P:
            # stats_target: (accum, denom, max_radii) STAGING arrays of the data-parallel tail (step.py) instead of the model's
            ta, td, tr = stats_target if stats_target is not None else (m.xyz_gradient_accum, m.denom, m.max_radii2D)
            stats = _lib.B3gsDensifyStats(ta.data_ptr(), td.data_ptr(), tr.data_ptr(), self.overflow_flag.data_ptr())
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        for c0 in range(0, len(pend), MAX_BATCH):
            chunk = pend[c0:c0 + MAX_BATCH]
            arr = (_lib.B3gsFusedView * len(chunk))()
            for k, (sp, sc) in enumerate(chunk):
                sl = self.slots[sp["slot"]]
                arr[k].view = C.pointer(sc)
                arr[k].radii, arr[k].geometry = sl.radii.data_ptr(), sl.geom.data_ptr()
                arr[k].scratch = sl.scratch.data_ptr()
                arr[k].dL_dmeans2D = None if sl.means2D_grad is None else sl.means2D_grad.data_ptr()
                arr[k].densify_stats = int(bool(sp["densify_stats"]) and stats is not None)
            rc = L.b3gs_backward_raw_accumulate_range(len(chunk), arr, C.byref(self._bind_params()), C.byref(gr),
                                                      int(bool(overwrite) and c0 == 0),
                                                      None if stats is None else C.byref(stats), first, count, stream)
            _lib.check(rc, "b3gs_backward_raw_accumulate_range")

    # ---- public -----------------------------------------------------------------------------
    def render(self, viewpoint_camera, bg_color: torch.Tensor, slot: int = 0, scaling_modifier: float = 1.0,
               debug: bool = False) -> dict:
        """One view.  Same keys as render(); `viewspace_points_grad` ([P,3], filled by backward) replaces
        the `.grad` of the reference's dummy `viewspace_points` tensor."""
        return self.render_batch([(viewpoint_camera, slot)], bg_color, scaling_modifier, debug)[0]

    def render_batch(self, views: Sequence, bg_color: torch.Tensor, scaling_modifier: float = 1.0,
                     debug: bool = False, _span_checked: bool = False) -> List[dict]:
        """Render [(camera, slot[, densify_stats]), ...]: binning concurrently (one stream per slot), then
        one blend launch; the outputs live on the current stream.  Views with densify_stats=True feed
        the model's densification statistics (init_densification_stats), as the primary view does at
        train.py:178-179.  All views of the call form ONE autograd node."""
        m = self.model
        # The three-pass depth sort rests on a CHECKED key span whose verdict sits in the overflow word until somebody reads
        # it (check_overflow(): the training step's protocol).  A render nobody will differentiate -- evaluation, a viewer --
        # has no such reader: it sorts all 32 key bits (fit_capacity() reads the word itself and keeps the three passes).
        key_bits = self.depth_key_bits if (torch.is_grad_enabled() or _span_checked) else 0
        specs = tuple({"cam": v[0], "slot": int(v[1]), "densify_stats": bool(v[2]) if len(v) > 2 else False,
                       "bg": bg_color, "scaling_modifier": scaling_modifier, "debug": debug, "key_bits": key_bits}
                      for v in views)
        assert len({sp["slot"] for sp in specs}) == len(specs), "each view of a batch needs its own slot"
        flat = _RasterizeBatch.apply(m._xyz, m._features_dc, m._features_rest, m._scaling, m._rotation, m._opacity,
                                     self, specs)
        out = []
        for k, sp in enumerate(specs):
            color, radii, depth, alpha = flat[4 * k:4 * k + 4]
            out.append(_ViewOutputs({"render": color, "viewspace_points_grad": self.slots[sp["slot"]].means2D_grad,
                                     "radii": radii, "rendered_depth": depth, "rendered_alpha": alpha}))
        return out

    def begin_deferred(self):
        """Open a section in which backward passes only run the blend backward (per-view scratch);
        finish_deferred() then does the per-Gaussian chain rule of ALL their views in one kernel."""
        self._deferred = []

    def take_deferred(self):
        """Close the deferred section WITHOUT running the chain rule and return the recorded views, for
        accumulate_range() (the list stays valid for replays of a HIP graph that captured the section)."""
        pend, self._deferred = self._deferred, None
        return pend

    def accumulate_range(self, pend, grads, first: int, count: int, overwrite: bool = True, stats_target=None):
        """Per-Gaussian chain rule of the views in `pend` (from take_deferred()) for the Gaussians
        [first, first+count) only, into `grads` (B3gsRawGrads, pointers indexed by the global Gaussian index).
        Call it for disjoint ranges covering all Gaussians."""
        self._accumulate(pend, overwrite, grads, first, count, stats_target=stats_target)

    def finish_views(self, pend, overwrite: bool = True):
        """Per-Gaussian chain rule of the views in `pend` (from take_deferred()) for ALL Gaussians into the parameters'
        `.grad`, dense rows -- the un-pipelined data-parallel tail (step.ViewShardedStep.reduce_and_update)."""
        if not pend:
            if overwrite:
                for p in self.model.parameters():
                    if p.grad is not None:
                        p.grad.zero_()
            return
        self._accumulate(pend, overwrite)

    def finish_deferred(self, overwrite: bool = True, touched_rows: Optional[torch.Tensor] = None):
        """overwrite=True stores the gradients (no zero-fill of the slab needed), False adds to them.  With
        `touched_rows` (int64 bitmap, one bit per Gaussian) the rows of Gaussians that received nothing are not stored
        at all: only a consumer that reads the bitmap (FusedAdam / ShardedAdam `row_mask`) may use the gradients."""
        pend, self._deferred = self._deferred, None
        if not pend:
            if overwrite:
                for p in self.model.parameters():
                    if p.grad is not None:
                        p.grad.zero_()
                if touched_rows is not None:
                    touched_rows.fill_(-1)          # every row valid (zero)
            return
        self._accumulate(pend, overwrite, touched_rows=touched_rows)

    def resize(self):
        """The model's Gaussian count changed (densification): re-create every slot for the new P."""
        self.P = self.model.get_xyz.shape[0]
        self.capacity = max(self.capacity, 12 * self.P)
        for i in range(len(self.slots)):
            self.slots[i] = self._new_slot(self.slots[i].stream, i)
        self.high_water.zero_()
        self.overflow_flag.zero_()

    def num_rendered(self):
        """Host copy of every slot's N (synchronises).  N > capacity means that view was rendered
        from a truncated list: call grow() and repeat the step."""
        torch.cuda.synchronize(self.dev)
        return [int(s.n_dev.item()) for s in self.slots]

    def overflowed(self) -> bool:
        return any(n > self.capacity for n in self.num_rendered())

    def grow(self, factor: float = 1.5, need: Optional[int] = None):
        need = max(self.num_rendered() + [self.capacity, int(need or 0)])
        self.capacity = int(need * factor)
        for i in range(len(self.slots)):
            self.slots[i] = self._new_slot(self.slots[i].stream, i)
        self.high_water.zero_()

    def check_overflow(self) -> int:
        """One small read-back: the largest N any view produced since the last check, and the sticky overflow flag.
        Returns 0 when every list fitted; otherwise grows the buffers (1.5 x what was needed), clears the flag and
        returns that N.  From the overflowing step on, the optimiser update and the densification statistics were
        dropped ON THE DEVICE (the Adam launch and the chain-rule pass read the flag), so the model is in the state of
        the last complete step and the caller simply repeats from there.  (The reference sizes the binning buffer from
        N on every render: one blocking read-back per view.)"""
        hw = int(self.high_water.max().item())
        flag = int(self.overflow_flag.item())
        self.high_water.zero_()
        timed_out = self._check_repair_status()
        if hw <= self.capacity and not flag and not timed_out:
            return 0
        self.overflow_flag.zero_()
        if flag & 8:
            raise _lib.B3gsError("B3GS_ERR_ARG: a view was rendered from the depth order of the camera it names as "
                                 "`same_depth_as`, but its depth keys differ (the z rows of the two view matrices are not the "
                                 "same bits); the steps since the last check were dropped on the device")
        if flag & 2:                    # a depth key outside the 27-bit span: sort all 32 bits from now on
            self.depth_key_bits = 0
        if (flag & 4) or timed_out:
            # the second binning round's persistent launch timed out at its grid barrier (its workgroups were not resident
            # together: a shared / partitioned device, co-resident persistent kernels): the repaired tiles of that forward
            # were wrong and the step was dropped on the device (bit 2 of the overflow word); one round from now on
            self.seg1_fraction = 0.0
            self._seg1_auto = False
            self.two_round_disabled = "grid-barrier time-out of the second binning round"
        if hw > self.capacity or (flag & 1):
            self.grow(need=hw)
        return max(hw, 1)

    def repair_rate(self, reset: bool = False):
        """(forwards whose open-tile prediction missed, two-round forwards) since the counters were last reset: words 10 /
        11 of slot 0's image header, kept by the binning kernels (they also count inside HIP-graph replays)."""
        w = self.slots[0].img[:48].view(torch.int32)
        missed, total = int(w[10].item()), int(w[11].item())
        if reset:
            w[10:12] = 0
        return missed, total

    def _check_repair_status(self):
        """Word 9 of slot 0's image header: bit 0 is set when a grid barrier of the second binning round's persistent
        kernel timed out (a workgroup of its grid never became resident) -- that forward's repaired tiles are wrong; the
        kernel also raised bit 2 of the overflow word, so the step was dropped on the device.  Returns that bit (cleared)."""
        if self.seg1_fraction <= 0.0 or not self.slots:
            return False
        # the rule's safety net: two rounds pay only while the prediction holds (the repair round's slow path costs about
        # twice what the second round of one-launch-per-stage binning did).  Cameras that change too much from one forward
        # of a slot to the next make it miss; when more than `two_round_max_miss_rate` of the forwards since the last check
        # needed a repair, "auto" goes back to one round (outside a graph capture: a captured step keeps what it captured)
        if self._seg1_auto and not torch.cuda.is_current_stream_capturing():
            missed, total = self.repair_rate(reset=True)
            if total >= 8 and missed > self.two_round_max_miss_rate * total:
                self.seg1_fraction = 0.0
                self.two_round_disabled = (missed, total)
        st = int(self.slots[0].img[:48].view(torch.int32)[9].item())
        if st:
            self.slots[0].img[:48].view(torch.int32)[9] = 0
        return bool(st)

    def fit_capacity(self, views: Sequence, bg_color: torch.Tensor, margin: float = 1.3) -> int:
        """Size the persistent binning buffers from the actual N of `views` ([(camera, slot), ...]): one forward without
        gradients, one read-back.  Called at start-up and after every densification (the Gaussian set changed), so the
        capacity follows the scene the way the reference's per-render allocation does."""
        frac, self.seg1_fraction = self.seg1_fraction, 0.0      # one round: the N of the complete lists
        with torch.no_grad():
            self.render_batch([(v[0], v[1]) for v in views], bg_color, _span_checked=True)
        need = max(self.num_rendered())
        self.high_water.zero_()
        if int(self.overflow_flag.item()) & 2:     # a depth key outside the 27-bit span: sort all 32 bits from now on
            self.depth_key_bits = 0
        self.overflow_flag.zero_()       # (a too-small start-up capacity is what this call is here to fix)
        if need * margin > self.capacity:
            self.grow(factor=margin, need=need)
        # two rounds: the nearest fraction of the depth order everywhere + the rest into the tiles predicted open
        # (the ones the previous forward of that slot left unterminated); segment 1 sized for ~0.75M instances of a view
        self.seg1_fraction = (0.0 if (need < self.two_round_min_instances or self.two_round_disabled)
                              else min(0.125, max(0.02, 0.75e6 / need))) if self._seg1_auto else frac
        if self.seg1_fraction > 0.0:
            with torch.no_grad():     # one forward settles the open-tile prediction (the first one repairs many tiles)
                self.render_batch([(v[0], v[1]) for v in views], bg_color, _span_checked=True)
            self.high_water.zero_()
        return self.capacity
