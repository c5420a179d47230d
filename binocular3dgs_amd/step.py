"""View-sharded training step: the data-parallel form of one `train.py` iteration.

Reference semantics (train.py:92-149,196-198): one input view per iteration, optionally a second
render of its binocular-shifted partner; the per-Gaussian gradients of the renders simply add up
in autograd, then Adam steps.  Here the VIEWS of an iteration are spread over the ranks
(SURVEY.md section 8e), the Gaussian parameters are replicated, and the only per-Gaussian exchange
is the sum of a flat, pre-packed fp32 gradient slab (92 B per Gaussian at K=4) over RCCL/xGMI
(torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests):

  * `ViewShardedStep(model, pairs, ...)`        this rank's whole (input, shifted) pairs (`shard_pairs`);
  * `ViewShardedStep.from_global(...)`          VIEW-granular: `assign_views` cuts the iteration's view list
    [in0, sh0, in1, sh1, ...] into balanced contiguous blocks, so a pair may straddle two ranks (6 views on 4
    GPUs: 2,2,1,1; 8 views on 8 GPUs: one each).  The binocular loss couples the members of a pair through
    the primary's depth map (train.py:128-136): the rank that owns the shifted view sends its rendered image
    (3 H W floats) to the rank that owns the input view, which forms the loss and sends d(loss)/d(shifted image)
    back -- two point-to-point messages per split pair over the direct xGMI link, no collective;
  * gradient sum + optimiser: either ONE all-reduce + replicated Adam (`FusedAdam`), or `ShardedAdam`:
    reduce-scatter of the slab, Adam on this rank's 1/N of the flat parameter buffer (Adam state and its
    28 B per float of traffic shrink by N), all-gather of the updated parameters -- the same bytes on the
    links as the all-reduce.

There is no reference counterpart for the exchange (the reference is single-GPU): semantics are defined in
SURVEY.md section 8e / DESIGN.md "Multi-GPU".  One rank with one pair reproduces the reference schedule.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from .render import PipelineParams, render


def _dist_on(group=None) -> bool:
    return dist.is_available() and dist.is_initialized()


class FlatGradSlab:
    """All parameter gradients live in one contiguous fp32 buffer: `p.grad` of every parameter is
    a view into it, so autograd accumulates the views of all renders in place, zeroing is one
    memset and the data-parallel exchange is one collective without a pack step.  `padded_numel`
    (>= the number of gradient floats) lets a sharded optimiser cut the buffer into equal chunks."""

    def __init__(self, params: Sequence[torch.nn.Parameter], padded_numel: int = 0):
        self.params = list(params)
        self.force_collective = False   # issue the all-reduce even in a 1-rank group (path check)
        total = sum(p.numel() for p in self.params)
        self.numel = total
        dev = self.params[0].device
        self.flat = torch.zeros(max(total, int(padded_numel)), dtype=torch.float32, device=dev)
        off = 0
        self.views: List[torch.Tensor] = []
        for p in self.params:
            v = self.flat[off:off + p.numel()].view_as(p)
            p.grad = v
            self.views.append(v)
            off += p.numel()

    def rebind(self):
        for p, v in zip(self.params, self.views):
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v

    def zero(self):
        self.flat.zero_()
        self.rebind()

    def all_reduce(self, average: bool = False, group=None):
        if _dist_on() and (dist.get_world_size(group) > 1 or self.force_collective):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            if average:
                self.flat.div_(dist.get_world_size(group))

    def nbytes(self) -> int:
        return self.flat.numel() * 4


ADAM_STEP_WORDS = 2 + 64 * 32     # B3GS_ADAM_STEP_WORDS of include/b3gs_raster.h


def _step_words(dev):
    """{step, completion counter, 64 first-level completion counters a cache line apart}: the int32 words
    b3gs_adam_step's `device_step` points to (ABI 7); `[:1]` of it is the optimiser's `step_count`."""
    return torch.zeros(ADAM_STEP_WORDS, dtype=torch.int32, device=dev)


def _bump_versions(params):
    """The HIP optimiser writes the parameters through raw pointers: tell autograd's version counters (saved-tensor checks
    of graphs that outlive the step; the depth-order hint of rasterizer._RasterizeRaw keys on them)."""
    for p in params:
        torch.autograd.graph.increment_version(p)


def _adam_launch(segs_py, step_count, betas, eps, opacity_decay, opacity_seg, decay_first, bump, dev, row_mask=None,
                 skip_flag=None):
    """One b3gs_adam_step launch over [(param_ptr, grad_ptr, m_ptr, v_ptr, count, lr[, row_len, first_row]), ...] (at
    most 8).  row_mask: the int64 touched-rows bitmap of a sparse-row gradient slab (segments then carry row_len).
    skip_flag: device int32; != 0 at launch time turns the call into a no-op (overflowed step, B3gsForwardView)."""
    from . import _lib
    segs = (_lib.B3gsAdamSegment * max(len(segs_py), 1))()
    for k, seg in enumerate(segs_py):
        p, g, m, v, n, lr = seg[:6]
        segs[k].param, segs[k].grad, segs[k].exp_avg, segs[k].exp_avg_sq = p or None, g or None, m or None, v or None
        segs[k].count, segs[k].lr = int(n), float(lr)
        segs[k].row_len, segs[k].first_row = (int(seg[6]), int(seg[7])) if len(seg) > 6 else (0, 0)
        segs[k].lr_dev = (seg[8] or None) if len(seg) > 8 else None
    rc = _lib.lib().b3gs_adam_step(len(segs_py), segs, step_count.data_ptr(), betas[0], betas[1], eps,
                                   float(opacity_decay), int(opacity_seg), int(bool(decay_first)), int(bool(bump)),
                                   None if row_mask is None else row_mask.data_ptr(),
                                   None if skip_flag is None else skip_flag.data_ptr(),
                                   torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "b3gs_adam_step")


class FusedAdam:
    """torch.optim.Adam semantics (betas, eps, per-tensor learning rates; no weight decay / amsgrad) for
    all parameter tensors of the model in one HIP launch (b3gs_adam_step), step counter on the device
    (graph-replayable).  `opacity_decay` (e.g. 0.995, train.py:171-173,278-279) is applied to the tensor
    with index `opacity_index` when set: after the Adam update by default, or -- `decay_first=True`, the
    reference's order (train.py:171-173 runs before optimizer.step() at :196-198) -- to the value the
    update is then subtracted from.  State lives in two flat fp32 buffers."""

    def __init__(self, params: Sequence[torch.nn.Parameter], lrs: Sequence[float], betas=(0.9, 0.999), eps=1e-15,
                 opacity_decay: float = 0.0, opacity_index: int = -1, decay_first: bool = False):
        self.params = list(params)
        self.lrs = [float(x) for x in lrs]
        assert len(self.params) == len(self.lrs) <= 8
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self._step_words = _step_words(dev)
        self.step_count = self._step_words[:1]
        self.betas, self.eps = betas, float(eps)
        self.opacity_decay, self.opacity_index = float(opacity_decay), int(opacity_index)
        self.decay_first = bool(decay_first)
        self.skip_flag = None    # device int32: != 0 drops the update on the device (ViewShardedStep sets it)
        # optional [len(params)] float32 tensor on the device: the learning rates the launch reads INSTEAD of `lrs` -- a step
        # replayed as a HIP graph follows a schedule (train.py:83) through device memory (B3gsAdamSegment::lr_dev)
        self.lr_device = None

    def step(self, row_mask: Optional[torch.Tensor] = None):
        """row_mask: the touched-rows bitmap of a sparse-row gradient slab (FusedRasterizer.finish_deferred)."""
        P = self.params[0].shape[0]
        self.step_rows(0, P, [p.grad.data_ptr() for p in self.params], last=True, row_mask=row_mask)

    def step_rows(self, first: int, count: int, grad_ptrs: Sequence[int], last: bool, row_mask=None):
        """Adam for rows [first, first+count) of every tensor (row = one Gaussian); grad_ptrs[k] is the address of
        the gradient of row `first` of tensor k (rows contiguous).  The step counter advances when `last`."""
        segs, off = [], 0
        for k, (p, lr, gp) in enumerate(zip(self.params, self.lrs, grad_ptrs)):
            assert p.is_contiguous()
            w = p.numel() // max(p.shape[0], 1)
            segs.append((p.data_ptr() + 4 * w * first, gp, self.exp_avg.data_ptr() + 4 * (off + w * first),
                         self.exp_avg_sq.data_ptr() + 4 * (off + w * first), w * count, lr, w, first,
                         None if self.lr_device is None else self.lr_device.data_ptr() + 4 * k))
            off += p.numel()
        _adam_launch(segs, self.step_count, self.betas, self.eps, self.opacity_decay, self.opacity_index,
                     self.decay_first, last, self.params[0].device, row_mask, self.skip_flag)
        if last:
            _bump_versions(self.params)

    def zero_grad(self, set_to_none: bool = False):
        for p in self.params:
            if p.grad is not None:
                p.grad.zero_()


class ShardedAdam:
    """Adam with the state sharded over the ranks (ZeRO-1 style) on top of flat buffers:

        reduce_scatter(gradient slab)  ->  Adam on this rank's 1/N of the flat parameter buffer  ->  all_gather(parameters)

    The six parameter tensors become views into ONE flat fp32 buffer `pflat` (tensor-major, padded to N equal
    chunks of a multiple of 64 floats); rank r owns flat[r*chunk, (r+1)*chunk) whatever tensors that range cuts
    through (the one-launch kernel takes up to 8 (pointer, count, lr) segments).  Same link bytes as an all-reduce
    (2 (N-1)/N of the slab), but exp_avg / exp_avg_sq and the 28 B per float the Adam kernel moves are divided by N.
    With one rank it degenerates to FusedAdam on flat buffers.  `adam_impl` replaces the HIP launch in the CPU
    tests of the host logic (tests/test_dp_gloo.py); the product default has no CPU path."""

    def __init__(self, params: Sequence[torch.nn.Parameter], lrs: Sequence[float], betas=(0.9, 0.999), eps=1e-15,
                 opacity_decay: float = 0.0, opacity_index: int = -1, decay_first: bool = False, group=None,
                 adam_impl: Optional[Callable] = None):
        self.lrs = [float(x) for x in lrs]
        self.betas, self.eps = betas, float(eps)
        self.opacity_decay, self.opacity_index = float(opacity_decay), int(opacity_index)
        self.decay_first = bool(decay_first)
        self.group = group
        self.world = dist.get_world_size(group) if _dist_on() else 1
        self.rank = dist.get_rank(group) if _dist_on() else 0
        self.adam_impl = adam_impl
        self.force_collective = False   # issue the collectives even in a 1-rank group (path check on one GPU)
        self.step_count = None
        self.skip_flag = None           # device int32: != 0 drops the update on the device (ViewShardedStep sets it)
        self.record_events = False      # bench: keep HIP events around the reduce-scatter and the all-gather
        self.events = []
        self._flatten(list(params), None, None)

    # ---- layout ---------------------------------------------------------------------------------
    def _flatten(self, params, full_m, full_v):
        assert len(params) == len(self.lrs) <= 8
        self.params = params
        dev = params[0].device
        if dev.type != "cuda" and self.adam_impl is None:
            from . import _lib
            raise _lib.B3gsError("ShardedAdam needs the parameters on the HIP device (no CPU fallback)")
        total = sum(p.numel() for p in params)
        per = -(-total // self.world)
        self.chunk = -(-per // 64) * 64
        self.padded_numel = self.chunk * self.world
        self.numel = total
        self.pflat = torch.zeros(self.padded_numel, dtype=torch.float32, device=dev)
        self.bounds, off = [], 0
        for p in params:
            n = p.numel()
            self.pflat[off:off + n].copy_(p.detach().reshape(-1))
            p.data = self.pflat[off:off + n].view(p.shape)          # the parameter now lives in the flat buffer
            self.bounds.append((off, off + n))
            off += n
        lo = self.rank * self.chunk
        f = dict(dtype=torch.float32, device=dev)
        self.exp_avg = torch.zeros(self.chunk, **f)
        self.exp_avg_sq = torch.zeros(self.chunk, **f)
        if full_m is not None:                                     # carried state (densification): take this rank's part
            n = max(0, min(total, lo + self.chunk) - lo)
            self.exp_avg[:n].copy_(full_m[lo:lo + n])
            self.exp_avg_sq[:n].copy_(full_v[lo:lo + n])
        self.gshard = None              # reduce-scatter output, allocated on first use
        if self.step_count is None:
            self._step_words = _step_words(dev)
            self.step_count = self._step_words[:1]

    def my_segments(self) -> List[Tuple[int, int, int]]:
        """[(tensor index, lo, hi)] in flat coordinates: the pieces of the tensors inside this rank's chunk."""
        lo, hi = self.rank * self.chunk, (self.rank + 1) * self.chunk
        out = []
        for k, (a, b) in enumerate(self.bounds):
            s, e = max(a, lo), min(b, hi)
            if e > s:
                out.append((k, s, e))
        return out

    # ---- one optimisation step ----------------------------------------------------------------------
    def collective(self) -> bool:
        return _dist_on() and (self.world > 1 or self.force_collective)

    def step(self, slab: FlatGradSlab, average: bool = False, row_mask: Optional[torch.Tensor] = None):
        """row_mask: touched-rows bitmap of a sparse-row slab -- only without a collective (the reduce-scatter needs
        dense rows) and with the HIP Adam."""
        assert slab.flat.numel() == self.padded_numel, "gradient slab must be padded to ShardedAdam.padded_numel"
        lo = self.rank * self.chunk
        collective = self.collective()
        assert row_mask is None or (not collective and self.adam_impl is None and self.world == 1)
        ev = None
        if collective and self.record_events:      # bench: events around the two collectives on the compute stream
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
            ev[0].record()
        if collective:
            if self.gshard is None:
                self.gshard = torch.zeros(self.chunk, dtype=torch.float32, device=self.pflat.device)
            dist.reduce_scatter_tensor(self.gshard, slab.flat, op=dist.ReduceOp.SUM, group=self.group)
            if ev:
                ev[1].record()
            g = self.gshard
            if average:
                g.div_(self.world)
        else:
            g = slab.flat[:self.chunk]
        segs, opacity_seg = [], -1
        for k, s, e in self.my_segments():
            if k == self.opacity_index:
                opacity_seg = len(segs)
            segs.append((self.pflat[s:e], g[s - lo:e - lo], self.exp_avg[s - lo:e - lo], self.exp_avg_sq[s - lo:e - lo],
                         self.lrs[k]))
        decay = self.opacity_decay if opacity_seg >= 0 else 0.0
        if self.adam_impl is not None:
            # (host-logic tests: the stand-in honours the skip word the way the HIP launch does on the device)
            if self.skip_flag is None or int(self.skip_flag.item()) == 0:
                self.adam_impl(segs, self.step_count, self.betas, self.eps, decay, opacity_seg, self.decay_first)
        else:
            P = max(self.params[0].shape[0], 1)
            widths = [self.params[k].numel() // P for k, _, _ in self.my_segments()]
            _adam_launch([(p.data_ptr(), gg.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, w, 0)
                          for (p, gg, m, v, lr), w in zip(segs, widths)], self.step_count, self.betas, self.eps, decay,
                         opacity_seg, self.decay_first, True, self.pflat.device, row_mask, self.skip_flag)
        _bump_versions(self.params)
        if collective:
            if ev:
                ev[2].record()
            dist.all_gather_into_tensor(self.pflat, self.pflat[lo:lo + self.chunk], group=self.group)
            if ev:
                ev[3].record()
                self.events.append(ev)

    def zero_grad(self, set_to_none: bool = False):
        for p in self.params:
            if p.grad is not None:
                p.grad.zero_()

    # ---- densification support (densify.py): the full moments, tensor-major ----------------------------
    def gather_full_state(self):
        if self.world == 1:
            return self.exp_avg, self.exp_avg_sq
        m = torch.empty(self.padded_numel, dtype=torch.float32, device=self.pflat.device)
        v = torch.empty_like(m)
        dist.all_gather_into_tensor(m, self.exp_avg, group=self.group)
        dist.all_gather_into_tensor(v, self.exp_avg_sq, group=self.group)
        return m, v

    def rebuild(self, new_params, full_m=None, full_v=None):
        """New parameter tensors (densification changed P): re-flatten, keep the carried moments."""
        self._flatten(list(new_params), full_m, full_v)


def auto_pipeline_ranges(P: int, row_floats: int = 23) -> int:
    """How many Gaussian ranges the pipelined data-parallel tail cuts the gradients into: one all-reduce per ~48 MB, at most
    four.  Every extra range costs a chain-rule launch pair and an Adam launch over a shorter slice (1M Gaussians, one rank:
    chain rule 0.12 ms in one pass, 0.24 ms in four ranges) and buys overlap of the collective with its neighbours' compute;
    below a few tens of MB a ring all-reduce over xGMI is latency-bound and splitting it only adds launches.  92 B per
    Gaussian at K = 4: 500k -> 1 range, 1M -> 2, 2M -> 4."""
    return max(1, min(4, round(4 * row_floats * P / float(48 << 20))))


class RangeGradSlab:
    """Gradient buffer laid out RANGE-major for the pipelined data-parallel tail: the Gaussians are cut into K
    index ranges and all six tensors' gradients of a range are contiguous, so one all-reduce per range can start
    as soon as that range's chain rule is done, while the next range is still being computed:
        [range 0: xyz | f_dc | f_rest | scaling | rotation | opacity][range 1: ...] ..."""

    def __init__(self, params: Sequence[torch.nn.Parameter], num_ranges: int):
        self.params = list(params)
        self.P = self.params[0].shape[0]
        self.widths = [p.numel() // max(self.P, 1) for p in self.params]
        self.K = max(1, min(int(num_ranges), max(self.P, 1)))
        self.bounds = [(r * self.P) // self.K for r in range(self.K + 1)]
        self.row = sum(self.widths)
        # PAD floats behind range 0, part of ITS collective: word 0 carries this rank's overflow flag into the sum (a step
        # is dropped by all replicas or by none without a collective of its own in front of the chain rule)
        self.PAD = 64
        self.flat = torch.zeros(self.row * self.P + self.PAD, dtype=torch.float32, device=self.params[0].device)
        n0 = self.bounds[1] - self.bounds[0]
        self.flag = self.flat[self.row * n0: self.row * n0 + 1]          # float: sum over the ranks of (overflow flag != 0)
        self.flag_word = self.flag.view(torch.int32)                     # the same 4 bytes as the "!= 0" word the kernels read

    def rows(self, r: int):
        return self.bounds[r], self.bounds[r + 1] - self.bounds[r]

    def _offset(self, r: int) -> int:
        """first float of range r (ranges behind range 0 sit behind its pad)"""
        return self.row * self.bounds[r] + (self.PAD if r > 0 else 0)

    def chunk(self, r: int) -> torch.Tensor:
        s, n = self.rows(r)
        o = self._offset(r)
        return self.flat[o: o + self.row * n + (self.PAD if r == 0 else 0)]

    def grad_ptrs(self, r: int) -> List[int]:
        """address of the first row of range r for each tensor"""
        s, n = self.rows(r)
        base, out, acc = self.flat.data_ptr() + 4 * self._offset(r), [], 0
        for w in self.widths:
            out.append(base + 4 * n * acc)
            acc += w
        return out

    def tensor_view(self, r: int, k: int) -> torch.Tensor:
        s, n = self.rows(r)
        off = self._offset(r) + n * sum(self.widths[:k])
        return self.flat[off: off + n * self.widths[k]].view((n,) + tuple(self.params[k].shape[1:]))

    def gather(self) -> List[torch.Tensor]:
        """the gradients as ordinary per-parameter tensors (tests / inspection)"""
        return [torch.cat([self.tensor_view(r, k) for r in range(self.K)], dim=0) for k in range(len(self.params))]


def shard_pairs(num_pairs: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment of whole view pairs to ranks (pair i -> rank i % world)."""
    return [i for i in range(num_pairs) if i % world == rank]


def assign_views(has_partner: Sequence[bool], world: int) -> List[List[Tuple[int, int]]]:
    """View-granular sharding.  The iteration's views in order [(pair, role)] -- role 0 = input view, 1 = its
    binocular partner -- are cut into `world` contiguous blocks whose sizes differ by at most one, larger blocks
    first: 6 views on 2 ranks -> 3+3 (pair 1 split), on 4 ranks -> 2,2,1,1 (only pair 2 split), on 8 ranks -> one
    view each and two idle ranks; 8 partner-less views on 8 ranks -> one each.  A block boundary splits at most one
    pair, so two ranks exchange at most one image each way.  Returns the block of every rank."""
    views = []
    for i, hp in enumerate(has_partner):
        views.append((i, 0))
        if hp:
            views.append((i, 1))
    base, extra = divmod(len(views), world)
    out, pos = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append(views[pos:pos + n])
        pos += n
    return out


class _View:
    __slots__ = ("key", "pair", "role", "cam", "t", "peer", "slot", "mate")

    def __init__(self, key, pair, role, cam, t, peer):
        self.key, self.pair, self.role, self.cam, self.t, self.peer = key, pair, role, cam, t, peer
        self.slot = None   # FusedRasterizer slot
        self.mate = None   # index (in the local view list) of the other member of the pair when it is local


class ViewShardedStep:
    """One optimisation step over this rank's views.

    pair_grad_fn(pair_index, primary_pkg_or_None, shifted_pkg_or_None) -> list of (output_tensor, grad_tensor)
    supplies the upstream pixel gradients (bench: seeded synthetic gradients; a member of a pair that lives on
    another rank arrives as None).  `loss_fn(pair_index, camera, primary_pkg, shifted_pkg_or_None, trans_dist)`
    returns a scalar loss per pair, `batch_loss_fn([(pair_index, camera, pkg, shifted_pkg, trans_dist), ...])` the
    sum over all pairs in one call; both see detached leaves of the rendered images, so that the pixel gradients
    of a split pair can travel between the ranks before the rasterizer backward runs.
    """

    def __init__(self, model, pairs, bg: torch.Tensor, pipe: Optional[PipelineParams] = None, optimizer=None,
                 average_over_world: bool = False, render_fn: Callable = render, fused=None, pipeline_ranges: int = 0,
                 group=None, _views: Optional[List[_View]] = None, overflow_check_every: int = 32,
                 sparse_grad_rows: bool = True):
        self.model = model
        self.pairs = list(pairs)          # [(camera, shifted_camera_or_None, trans_dist)]  (whole local pairs)
        self.bg = bg
        self.pipe = pipe or PipelineParams()
        self.optimizer = optimizer
        self.average = average_over_world
        self.group = group
        self.fused = fused                # FusedRasterizer: fused activations + persistent scratch
        if _views is None:
            _views = []
            for i, (cam, scam, t) in enumerate(self.pairs):
                _views.append(_View(i, i, 0, cam, t, None))
                if scam is not None:
                    _views.append(_View(i, i, 1, scam, t, None))
        self.views = _views
        by_pair = {}
        for k, v in enumerate(self.views):
            by_pair.setdefault(v.pair, {})[v.role] = k
        for k, v in enumerate(self.views):
            v.mate = by_pair[v.pair].get(1 - v.role)
            v.slot = k
        self.slab = FlatGradSlab(model.parameters(), getattr(optimizer, "padded_numel", 0))
        if fused is not None:
            assert len(fused.slots) >= len(self.views), "FusedRasterizer needs one slot per view of the step"
            if self.views:     # size the persistent binning buffers from the actual N of this rank's views
                fused.fit_capacity([(v.cam, v.slot) for v in self.views], bg)
            # an overflowing step must not reach the parameters: the optimiser reads the rasterizer's device-side flag
            if hasattr(optimizer, "skip_flag"):
                optimizer.skip_flag = fused.overflow_flag
        self.render = render_fn
        self.last_stats = {}
        self.overflow_check_every = int(overflow_check_every)
        self._steps_since_check = 0
        self._recv_img = {}               # primary view index -> persistent receive buffer for the partner's image
        self._recv_grad = {}              # shifted view index -> persistent receive buffer for d(loss)/d(image)
        # Pipelined data-parallel tail (fused path + FusedAdam): compute_grads() stops after the blend backward;
        # reduce_and_update() then walks K Gaussian ranges -- chain rule of range r+1 overlaps the all-reduce of
        # range r, Adam of range r overlaps the all-reduce of range r+1 -- instead of accumulate -> all-reduce ->
        # Adam back to back (the all-reduce of 92 B/Gaussian is ~20 % of an iteration on 8 GPUs).
        # pipeline_ranges: 0 = off, K >= 1 = that many ranges (K = 1: one all-reduce, still with the overflow word in-band and
        # the statistics staged -- no blocking agreement collective in front of the chain rule), "auto" / < 0 = from P
        if pipeline_ranges == "auto" or (isinstance(pipeline_ranges, int) and pipeline_ranges < 0):
            pipeline_ranges = auto_pipeline_ranges(model.get_xyz.shape[0], sum(p.numel() for p in model.parameters()) //
                                                   max(model.get_xyz.shape[0], 1))
        self.pipeline_ranges = int(pipeline_ranges) if (fused is not None and isinstance(optimizer, FusedAdam)) else 0
        self.range_slab = RangeGradSlab(model.parameters(), self.pipeline_ranges) if self.pipeline_ranges >= 1 else None
        # Sparse gradient rows (single rank, fused path, HIP Adam): ~80 % of the Gaussians receive no gradient in an
        # iteration; their slab rows are then neither written (chain-rule pass) nor read (Adam) -- a bitmap says which
        # rows are valid.  After such a step `p.grad` holds STALE rows: use sparse_grad_rows=False to inspect it.
        self.sparse_grad_rows = bool(sparse_grad_rows)
        self.touched_rows = None
        self._staged_stats = None         # pipelined data-parallel tail: (accum, denom, max_radii) staging arrays
        self._row_mask = None             # bitmap of the gradients now in the slab (None = dense)
        self._pending_views = None        # views whose chain rule reduce_and_update() still has to run (data parallel)
        self.tail_events = None           # bench: list that receives the HIP events of every pipelined tail

    @classmethod
    def from_global(cls, model, global_pairs, bg, rank: Optional[int] = None, world: Optional[int] = None, **kw):
        """View-granular sharding of the GLOBAL pair list (identical on every rank): this rank takes block `rank` of
        assign_views().  Pair indices handed to the loss / gradient callbacks are GLOBAL."""
        group = kw.get("group")
        if world is None:
            world = dist.get_world_size(group) if _dist_on() else 1
        if rank is None:
            rank = dist.get_rank(group) if _dist_on() else 0
        blocks = assign_views([scam is not None for _, scam, _ in global_pairs], world)
        owner = {v: r for r, blk in enumerate(blocks) for v in blk}
        views = []
        for (i, role) in blocks[rank]:
            cam, scam, t = global_pairs[i]
            peer = owner.get((i, 1 - role))
            views.append(_View(i, i, role, cam if role == 0 else scam, t, None if peer in (None, rank) else peer))
        st = cls(model, [], bg, _views=views, **kw)
        st.global_pairs, st.rank, st.world, st.blocks = list(global_pairs), rank, world, blocks
        return st

    # ---- the iteration ------------------------------------------------------------------------------------
    def step(self, pair_grad_fn=None, loss_fn=None, batch_loss_fn=None):
        n = self.compute_grads(pair_grad_fn, loss_fn, batch_loss_fn)
        self.reduce_and_update()
        self._steps_since_check += 1
        if self.fused is not None and self.overflow_check_every > 0 and \
                self._steps_since_check >= self.overflow_check_every and not torch.cuda.is_current_stream_capturing():
            self.check_capacity()
        return n

    def check_capacity(self):
        """The persistent binning buffers hold `capacity` tile instances per view; a view with more renders (and
        back-propagates) a truncated list.  The binning kernels record the largest N and raise a sticky device-side
        flag when a view overflowed; from that step on the Adam launch and the densification statistics drop their
        updates ON THE DEVICE (b3gs_adam_step(skip_if_nonzero), B3gsDensifyStats::skip_if_nonzero), so nothing computed
        from truncated lists ever reaches the parameters, the moments or the statistics.  This method -- every
        `overflow_check_every` steps, at every densification, at resize -- reads both words back, grows the buffers,
        clears the flag and raises: the model is in the state of the last complete step, the caller repeats from there.
        (The reference sizes the binning buffer from N on every render: one host sync per view.)"""
        self._steps_since_check = 0
        if self.fused is None:
            return
        over = self.fused.check_overflow()
        if over:
            from . import _lib
            raise _lib.B3gsError(f"B3GS_ERR_CAPACITY: a view needed {over} tile instances, the binning buffers held "
                                 f"fewer; they have been grown to {self.fused.capacity}.  The optimiser updates and "
                                 f"statistics from the overflowing step on were dropped on the device: the model is in "
                                 f"the state of the last complete step -- repeat from there (at most "
                                 f"{self.overflow_check_every} steps)")

    def _sparse_rows(self) -> Optional[torch.Tensor]:
        """The touched-rows bitmap when this step may leave untouched gradient rows unwritten, else None."""
        opt = self.optimizer
        if not (self.sparse_grad_rows and self.fused is not None and self.range_slab is None and self.views
                and len(self.views) <= 8):
            return None
        if isinstance(opt, ShardedAdam):
            if opt.collective() or opt.adam_impl is not None or opt.world != 1:
                return None
        elif isinstance(opt, FusedAdam):
            if _dist_on() and (dist.get_world_size(self.group) > 1 or self.slab.force_collective):
                return None
        else:
            return None
        words = (self.model.get_xyz.shape[0] + 63) // 64
        if self.touched_rows is None or self.touched_rows.numel() != words:
            self.touched_rows = torch.zeros(words, dtype=torch.int64, device=self.bg.device)
        return self.touched_rows

    def _collective(self, group=None) -> bool:
        group = group if group is not None else self.group
        return _dist_on() and (dist.get_world_size(group) > 1 or self.slab.force_collective)

    def _chain_rule_after_agreement(self) -> bool:
        """Data parallel + fused path: compute_grads() stops behind the blend backward and reduce_and_update() runs the
        per-Gaussian chain rule AFTER the ranks have agreed on the overflow word -- the chain-rule pass is what folds the
        densification statistics in, and a rank that did not overflow itself must not count a step the others drop
        (the statistics of the replicas diverged by exactly those steps)."""
        return self.fused is not None and (self.range_slab is not None or self._collective())

    def reduce_and_update(self):
        """The exchange step of the data-parallel path (collectives over the gradient slab) + optimiser."""
        if self.range_slab is not None:          # (the overflow word travels inside the first range's all-reduce)
            return self._reduce_and_update_pipelined()
        self._agree_on_overflow()
        if self._pending_views is not None:          # chain rule of this rank's views, now that the overflow word is agreed
            # (the list is NOT cleared: a HIP graph that captured compute_grads() is replayed without re-running its Python)
            self.slab.rebind()
            self.fused.finish_views(self._pending_views, overwrite=True)
        mask, self._row_mask = self._row_mask, None
        if isinstance(self.optimizer, ShardedAdam):
            self.slab.rebind()
            return self.optimizer.step(self.slab, average=self.average, row_mask=mask)
        assert mask is None or isinstance(self.optimizer, FusedAdam)
        self.slab.all_reduce(self.average, self.group)
        if self.optimizer is not None:
            self.slab.rebind()
            if mask is not None:
                self.optimizer.step(row_mask=mask)
            else:
                self.optimizer.step()

    def _agree_on_overflow(self, group=None):
        """Data parallel: a step is dropped by ALL replicas or by none -- the overflow flag is max-reduced (4 bytes)
        before the optimiser reads it."""
        group = group if group is not None else self.group
        if self.fused is not None and _dist_on() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.fused.overflow_flag, op=dist.ReduceOp.MAX, group=group)

    def densify_and_prune(self, max_grad: float, min_opacity: float, extent: float, max_screen_size=None,
                          percent_dense: float = 0.01, noise=None, generator=None, group=None) -> int:
        """train.py:180-185 for this step object: replicas agree on the statistics, the Gaussian set is rebuilt
        on the device (densify.densify_and_prune) and everything sized by P -- gradient slab, rasterizer
        slots -- is re-created.  Any HIP graph captured around step() must be re-captured afterwards."""
        from .densify import densify_and_prune
        group = group if group is not None else self.group
        self.check_capacity()
        self.sync_densify_stats(group)
        if noise is None and generator is None and _dist_on() and dist.get_world_size(group) > 1:
            # the split offsets must be the same on every replica (scene/gaussian_model.py:364 draws them from the
            # device RNG): rank 0 draws, everybody else receives
            P, dev = self.model.get_xyz.shape[0], self.model.get_xyz.device
            noise = torch.randn((2, P, 3), device=dev) if dist.get_rank(group) == 0 else torch.empty((2, P, 3), device=dev)
            dist.broadcast(noise, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        newP = densify_and_prune(self.model, self.optimizer, max_grad, min_opacity, extent, max_screen_size,
                                 percent_dense, noise, generator)
        self.slab = FlatGradSlab(self.model.parameters(), getattr(self.optimizer, "padded_numel", 0))
        if self.range_slab is not None:
            self.range_slab = RangeGradSlab(self.model.parameters(), self.pipeline_ranges)
        if self.fused is not None:
            self.fused.resize()
            if self.views:     # the Gaussian set changed: size the binning buffers from the new N (one read-back)
                self.fused.fit_capacity([(v.cam, v.slot) for v in self.views], self.bg)
        return newP

    def _reduce_and_update_pipelined(self, group=None):
        """Chain rule of range r+1 overlaps the all-reduce of range r; Adam of range r overlaps the all-reduce of range r+1
        (the collectives run on the backend's own stream: an async all-reduce waits for what the compute stream had queued
        when it was issued, and `wait()` makes the compute stream wait for that one collective only)."""
        from . import _lib
        rs, fr, opt = self.range_slab, self.fused, self.optimizer
        group = group if group is not None else self.group
        collective = self._collective(group)
        # One collective chain per step: this rank's overflow word rides in the pad of range 0 and comes back as the SUM over
        # the ranks.  What must not happen before that sum is known -- the statistics of a step somebody else drops -- is
        # staged: the chain rule folds the step's statistics into three zeroed arrays, b3gs_apply_staged_densify_stats adds
        # them to the model's (or discards them, and raises this rank's sticky word) right behind the first range's
        # all-reduce; the Adam launches of all ranges read the summed word.
        m = self.model
        staged = None
        if fr is not None and getattr(m, "denom", None) is not None and m.denom.numel() == rs.P:
            if self._staged_stats is None or self._staged_stats[0].numel() != rs.P:
                self._staged_stats = tuple(torch.zeros(rs.P, dtype=torch.float32, device=rs.flat.device) for _ in range(3))
            staged = self._staged_stats
        if fr is not None:
            rs.flag.copy_(fr.overflow_flag)          # int32 -> float: any set bit is a non-zero float
        else:
            rs.flag.zero_()
        skip_keep = getattr(opt, "skip_flag", None)
        if fr is not None and hasattr(opt, "skip_flag"):
            opt.skip_flag = rs.flag_word
        ev = None
        if self.tail_events is not None:     # bench: issue / completion stamps of every range on the compute stream
            mk = lambda: torch.cuda.Event(enable_timing=True)   # noqa: E731
            ev = {"start": mk(), "issued": [mk() for _ in range(rs.K)], "reduced": [mk() for _ in range(rs.K)], "end": mk()}
            ev["start"].record()
        works = []
        for r in range(rs.K):
            first, count = rs.rows(r)
            gr = _lib.B3gsRawGrads()
            for name, ptr, w in zip(("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity"),
                                    rs.grad_ptrs(r), rs.widths):
                setattr(gr, name, (ptr - 4 * w * first) if w else None)     # indexed by the GLOBAL Gaussian index
            if self._pending_views:
                fr.accumulate_range(self._pending_views, gr, first, count, overwrite=True, stats_target=staged)
            if ev:
                ev["issued"][r].record()
            works.append(dist.all_reduce(rs.chunk(r), op=dist.ReduceOp.SUM, group=group, async_op=True) if collective else None)
        for r in range(rs.K):
            if works[r] is not None:
                works[r].wait()                     # the compute stream waits for that range's all-reduce only
            if r == 0 and staged is not None:
                rc = _lib.lib().b3gs_apply_staged_densify_stats(
                    rs.P, staged[0].data_ptr(), staged[1].data_ptr(), staged[2].data_ptr(), m.xyz_gradient_accum.data_ptr(),
                    m.denom.data_ptr(), m.max_radii2D.data_ptr(), rs.flag_word.data_ptr(), fr.overflow_flag.data_ptr(),
                    torch.cuda.current_stream(rs.flat.device).cuda_stream)
                _lib.check(rc, "b3gs_apply_staged_densify_stats")
            elif r == 0 and fr is not None:
                fr.overflow_flag.bitwise_or_((rs.flag_word != 0).to(torch.int32))      # (no statistics: keep the word sticky)
            if ev:
                ev["reduced"][r].record()
            if self.average and collective:
                rs.chunk(r).div_(dist.get_world_size(group))
            first, count = rs.rows(r)
            opt.step_rows(first, count, rs.grad_ptrs(r), last=(r == rs.K - 1))
        if fr is not None and hasattr(opt, "skip_flag"):
            opt.skip_flag = skip_keep
        if ev:
            ev["end"].record()
            self.tail_events.append(ev)

    def sync_densify_stats(self, group=None):
        """Make the densification statistics identical on every replica before a densify/prune decision
        (SURVEY 8e): sums of xyz_gradient_accum / denom, maximum of max_radii2D.  Call it every
        densification interval, not every step."""
        m = self.model
        if not _dist_on() or getattr(m, "denom", None) is None:
            return
        group = group if group is not None else self.group
        dist.all_reduce(m.xyz_gradient_accum, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(m.denom, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(m.max_radii2D, op=dist.ReduceOp.MAX, group=group)

    # ---- rendering + pixel gradients --------------------------------------------------------------------------
    def _exchange(self, sends, recvs):
        """Point-to-point messages of the split pairs: [(tensor, peer)] each way, one batch (deadlock-free)."""
        if not sends and not recvs:
            return
        ops = [dist.P2POp(dist.isend, t, p, self.group) for t, p in sends] + \
              [dist.P2POp(dist.irecv, t, p, self.group) for t, p in recvs]
        for w in dist.batch_isend_irecv(ops):
            w.wait()

    def _render_views(self):
        if self.fused is not None:
            return self.fused.render_batch([(v.cam, v.slot, v.role == 0) for v in self.views], self.bg)
        return [self.render(v.cam, self.model, self.pipe, self.bg) for v in self.views]

    def compute_grads(self, pair_grad_fn=None, loss_fn=None, batch_loss_fn=None):
        """Render this rank's views forward+backward; leaves the summed gradients in the slab.  Without split
        pairs it contains no communication, no host sync and (fused path) no allocation: capturable as one HIP
        graph."""
        assert sum(f is not None for f in (pair_grad_fn, loss_fn, batch_loss_fn)) == 1
        assert batch_loss_fn is None or self.fused is not None
        if not self.views:
            # a rank without views (more ranks than views) contributes zeros to the collective
            self.slab.zero()
            self._pending_views = None
            if self.range_slab is not None:
                self.range_slab.flat.zero_()
                self._pending_views = []
            self.last_stats = {"views": 0}
            return 0
        if self.fused is not None:
            # the fused multi-view accumulate STORES the gradients: no zero-fill of the slab
            self.slab.rebind()
            self.fused.begin_deferred()
        else:
            self.slab.zero()
        pkgs = self._render_views()
        outs, grads = [], []
        if pair_grad_fn is not None:
            for k, v in enumerate(self.views):
                if v.role == 1 and v.mate is not None:
                    continue                                  # handled together with its primary
                prim = pkgs[k] if v.role == 0 else None
                shif = pkgs[v.mate] if (v.role == 0 and v.mate is not None) else (pkgs[k] if v.role == 1 else None)
                for o, g in pair_grad_fn(v.key, prim, shif):
                    outs.append(o)
                    grads.append(g)
        else:
            outs, grads = self._loss_pixel_grads(pkgs, loss_fn, batch_loss_fn)
        if outs:
            torch.autograd.backward(outs, grads)
        if self.fused is None:
            self._update_densify_stats(pkgs)
        if self.fused is not None:
            if not self._chain_rule_after_agreement():
                self._pending_views = None
                self._row_mask = self._sparse_rows()
                self.fused.finish_deferred(overwrite=True, touched_rows=self._row_mask)
            else:   # reduce_and_update() runs the chain rule (range by range when the tail is pipelined)
                self._pending_views = self.fused.take_deferred()
        self.last_stats = {"views": len(self.views)}
        return len(self.views)

    def _update_densify_stats(self, pkgs):
        """train.py:178-179 for the reference-shaped render(): statistics from the INPUT views only (the shifted
        render's radii / screen-space gradients are discarded, train.py:128).  The fused path does this inside
        its per-Gaussian backward kernel."""
        m = self.model
        if getattr(m, "denom", None) is None or m.denom.shape[0] != m.get_xyz.shape[0]:
            return
        with torch.no_grad():
            for v, pkg in zip(self.views, pkgs):
                g = pkg["viewspace_points"].grad if v.role == 0 else None
                if g is None:
                    continue
                vis = pkg["visibility_filter"]
                m.update_max_radii(pkg["radii"], vis)
                m._accumulate_stats(g, vis)

    def _loss_pixel_grads(self, pkgs, loss_fn, batch_loss_fn):
        """Loss of every pair whose INPUT view lives here, on detached leaves of the rendered images; returns the
        rasterizer outputs of the local views with their pixel gradients.  Split pairs: the partner's image is
        received before the loss, its gradient sent back afterwards (and vice versa for a local shifted view whose
        input view is remote)."""
        def leaf(t):
            return t.detach().requires_grad_(True)

        dev = self.bg.device
        H, W = pkgs[0]["render"].shape[-2:]
        sends, recvs = [], []
        for k, v in enumerate(self.views):
            if v.peer is None:
                continue
            if v.role == 1:
                sends.append((pkgs[k]["render"].detach(), v.peer))
            else:
                if k not in self._recv_img:
                    self._recv_img[k] = torch.empty((3, H, W), dtype=torch.float32, device=dev)
                recvs.append((self._recv_img[k], v.peer))
        self._exchange(sends, recvs)
        leaves, items = {}, []
        for k, v in enumerate(self.views):
            if v.role != 0:
                continue
            lp = dict(pkgs[k])
            for name in ("render", "rendered_depth", "rendered_alpha"):
                lp[name] = leaf(pkgs[k][name])
            leaves[k] = lp
            sp = None
            if v.mate is not None:
                sp = dict(pkgs[v.mate])
                sp["render"] = leaf(pkgs[v.mate]["render"])
                leaves[v.mate] = sp
            elif v.peer is not None:
                sp = {"render": leaf(self._recv_img[k])}
                leaves[("remote", k)] = sp
            items.append((v.key, v.cam, lp, sp, v.t))
        if items:
            if batch_loss_fn is not None:
                total = batch_loss_fn(items)
            else:
                losses = [loss_fn(*it) for it in items]
                total = losses[0]
                for extra in losses[1:]:
                    total = total + extra
            # (a cached root gradient: `total.backward()` fills a fresh ones_like(total) every iteration -- one more launch
            # in a two-view iteration that is a chain of ~30 small ones)
            one = self._one if getattr(self, "_one", None) is not None and self._one.device == total.device and \
                self._one.dtype == total.dtype and self._one.shape == total.shape else None
            if one is None:
                one = self._one = torch.ones_like(total)
            torch.autograd.backward([total], [one])
        sends, recvs = [], []
        for k, v in enumerate(self.views):
            if v.peer is None:
                continue
            if v.role == 0:
                g = leaves[("remote", k)]["render"].grad
                sends.append((g if g is not None else torch.zeros((3, H, W), dtype=torch.float32, device=dev), v.peer))
            else:
                if k not in self._recv_grad:
                    self._recv_grad[k] = torch.empty((3, H, W), dtype=torch.float32, device=dev)
                recvs.append((self._recv_grad[k], v.peer))
        self._exchange(sends, recvs)
        outs, grads = [], []
        for k, v in enumerate(self.views):
            if v.role == 1 and v.peer is not None:
                outs.append(pkgs[k]["render"])
                grads.append(self._recv_grad[k])
                continue
            lp = leaves.get(k)
            if lp is None:
                continue
            for name in (("render", "rendered_depth", "rendered_alpha") if v.role == 0 else ("render",)):
                if lp[name].grad is not None:
                    outs.append(pkgs[k][name])
                    grads.append(lp[name].grad)
        return outs, grads
