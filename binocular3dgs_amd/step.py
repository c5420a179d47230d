"""View-sharded training step: the data-parallel form of one `train.py` iteration.

Reference semantics (train.py:92-149,196-198): one input view per iteration, optionally a second
render of its binocular-shifted partner; the per-Gaussian gradients of the renders simply add up
in autograd, then Adam steps.  Here every rank owns whole (input view, shifted view) PAIRS --
the binocular loss couples the two members of a pair through the primary depth map, so a pair is
never split -- the Gaussian parameters are replicated, and the only exchange is ONE all-reduce
(sum) of a flat, pre-packed fp32 gradient slab (92 B per Gaussian at K=4) over RCCL/xGMI
(torch.distributed backend "nccl" on ROCm; "gloo" in the CPU tests).  There is no reference
counterpart for the collective (the reference is single-GPU): semantics are defined in
SURVEY.md section 8e / DESIGN.md "Multi-GPU".

`views_per_step == 1` on one rank reproduces the reference schedule.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import torch
import torch.distributed as dist

from .render import PipelineParams, render


class FlatGradSlab:
    """All parameter gradients live in one contiguous fp32 buffer: `p.grad` of every parameter is
    a view into it, so autograd accumulates the views of all renders in place, zeroing is one
    memset and the data-parallel exchange is one all-reduce without a pack step."""

    def __init__(self, params: Sequence[torch.nn.Parameter]):
        self.params = list(params)
        self.force_collective = False   # issue the all-reduce even in a 1-rank group (path check)
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
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
        if dist.is_available() and dist.is_initialized() and \
                (dist.get_world_size(group) > 1 or self.force_collective):
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            if average:
                self.flat.div_(dist.get_world_size(group))

    def nbytes(self) -> int:
        return self.flat.numel() * 4


class FusedAdam:
    """torch.optim.Adam semantics (betas, eps, per-tensor learning rates; no weight decay / amsgrad) for
    all parameter tensors of the model in one HIP launch (b3gs_adam_step), step counter on the device
    (graph-replayable).  `opacity_decay` (e.g. 0.995, train.py:171-173,278-279) is applied to the tensor
    with index `opacity_index` after its update when set.  State lives in two flat fp32 buffers."""

    def __init__(self, params: Sequence[torch.nn.Parameter], lrs: Sequence[float], betas=(0.9, 0.999), eps=1e-15,
                 opacity_decay: float = 0.0, opacity_index: int = -1):
        import ctypes as C

        from . import _lib
        self._C, self._lib = C, _lib
        self.params = list(params)
        self.lrs = [float(x) for x in lrs]
        assert len(self.params) == len(self.lrs) <= 8
        dev = self.params[0].device
        total = sum(p.numel() for p in self.params)
        self.exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self.step_count = torch.zeros(1, dtype=torch.int32, device=dev)
        self.betas, self.eps = betas, float(eps)
        self.opacity_decay, self.opacity_index = float(opacity_decay), int(opacity_index)

    def step(self):
        P = self.params[0].shape[0]
        self.step_rows(0, P, [p.grad.data_ptr() for p in self.params], last=True)

    def step_rows(self, first: int, count: int, grad_ptrs: Sequence[int], last: bool):
        """Adam for rows [first, first+count) of every tensor (row = one Gaussian); grad_ptrs[k] is the address of
        the gradient of row `first` of tensor k (rows contiguous).  The step counter advances when `last`."""
        C, _lib = self._C, self._lib
        segs = (_lib.B3gsAdamSegment * len(self.params))()
        off = 0
        for k, (p, lr) in enumerate(zip(self.params, self.lrs)):
            assert p.is_contiguous()
            w = p.numel() // max(p.shape[0], 1)
            segs[k].param, segs[k].grad = p.data_ptr() + 4 * w * first, grad_ptrs[k]
            segs[k].exp_avg = self.exp_avg.data_ptr() + 4 * (off + w * first)
            segs[k].exp_avg_sq = self.exp_avg_sq.data_ptr() + 4 * (off + w * first)
            segs[k].count, segs[k].lr = w * count, lr
            off += p.numel()
        dev = self.params[0].device
        rc = _lib.lib().b3gs_adam_step(len(self.params), segs, self.step_count.data_ptr(), self.betas[0], self.betas[1],
                                       self.eps, self.opacity_decay, self.opacity_index, int(bool(last)),
                                       torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(rc, "b3gs_adam_step")

    def zero_grad(self, set_to_none: bool = False):
        for p in self.params:
            if p.grad is not None:
                p.grad.zero_()


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
        self.flat = torch.zeros(self.row * self.P, dtype=torch.float32, device=self.params[0].device)

    def rows(self, r: int):
        return self.bounds[r], self.bounds[r + 1] - self.bounds[r]

    def chunk(self, r: int) -> torch.Tensor:
        s, n = self.rows(r)
        return self.flat[self.row * s: self.row * (s + n)]

    def grad_ptrs(self, r: int) -> List[int]:
        """address of the first row of range r for each tensor"""
        s, n = self.rows(r)
        base, out, acc = self.flat.data_ptr() + 4 * self.row * s, [], 0
        for w in self.widths:
            out.append(base + 4 * n * acc)
            acc += w
        return out

    def tensor_view(self, r: int, k: int) -> torch.Tensor:
        s, n = self.rows(r)
        off = self.row * s + n * sum(self.widths[:k])
        return self.flat[off: off + n * self.widths[k]].view((n,) + tuple(self.params[k].shape[1:]))

    def gather(self) -> List[torch.Tensor]:
        """the gradients as ordinary per-parameter tensors (tests / inspection)"""
        return [torch.cat([self.tensor_view(r, k) for r in range(self.K)], dim=0) for k in range(len(self.params))]


def shard_pairs(num_pairs: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment of view pairs to ranks (pair i -> rank i % world)."""
    return [i for i in range(num_pairs) if i % world == rank]


class ViewShardedStep:
    """One optimisation step over this rank's view pairs.

    pair_grad_fn(pair_index, primary_pkg, shifted_pkg_or_None) -> list of (output_tensor, grad_tensor)
    supplies the upstream pixel gradients (bench: seeded synthetic gradients; training: autograd of
    the loss block).  When `loss_fn` is given instead, it returns a scalar loss per pair and
    ordinary autograd is used.
    """

    def __init__(self, model, pairs, bg: torch.Tensor, pipe: Optional[PipelineParams] = None, optimizer=None,
                 average_over_world: bool = False, render_fn: Callable = render, fused=None, pipeline_ranges: int = 0):
        self.model = model
        self.pairs = list(pairs)          # [(camera, shifted_camera_or_None, trans_dist)]
        self.bg = bg
        self.pipe = pipe or PipelineParams()
        self.slab = FlatGradSlab(model.parameters())
        self.optimizer = optimizer
        self.average = average_over_world
        self.fused = fused                # FusedRasterizer: fused activations + persistent scratch
        if fused is not None:
            slot = iter(range(10 ** 9))
            self._slots = [(next(slot), next(slot) if sc is not None else None) for _, sc, _ in self.pairs]
            assert len(fused.slots) >= 2 * len(self.pairs), "FusedRasterizer needs one slot per view of the step"
        self.render = render_fn
        self.last_stats = {}
        # Pipelined data-parallel tail (fused path + FusedAdam): compute_grads() stops after the blend backward;
        # reduce_and_update() then walks K Gaussian ranges -- chain rule of range r+1 overlaps the all-reduce of
        # range r, Adam of range r overlaps the all-reduce of range r+1 -- instead of accumulate -> all-reduce ->
        # Adam back to back (the all-reduce of 92 B/Gaussian is ~20 % of an iteration on 8 GPUs).
        self.pipeline_ranges = int(pipeline_ranges) if (fused is not None and isinstance(optimizer, FusedAdam)) else 0
        self.range_slab = RangeGradSlab(model.parameters(), self.pipeline_ranges) if self.pipeline_ranges > 1 else None

    def step(self, pair_grad_fn=None, loss_fn=None, batch_loss_fn=None):
        """batch_loss_fn([(pair_index, camera, pkg, shifted_pkg, trans_dist), ...]) -> scalar: all pairs' loss in one
        call (fused path only; e.g. fused_loss.binocular_loss_fused_batch)."""
        n = self.compute_grads(pair_grad_fn, loss_fn, batch_loss_fn)
        self.reduce_and_update()
        return n

    def reduce_and_update(self):
        """The exchange step of the data-parallel path (one all-reduce of the flat slab) + optimiser."""
        if self.range_slab is not None:
            return self._reduce_and_update_pipelined()
        self.slab.all_reduce(self.average)
        if self.optimizer is not None:
            self.slab.rebind()
            self.optimizer.step()

    def densify_and_prune(self, max_grad: float, min_opacity: float, extent: float, max_screen_size=None,
                          percent_dense: float = 0.01, noise=None, generator=None, group=None) -> int:
        """train.py:180-185 for this step object: replicas agree on the statistics, the Gaussian set is rebuilt
        on the device (densify.densify_and_prune) and everything sized by P -- gradient slab, rasterizer
        slots -- is re-created.  Any HIP graph captured around step() must be re-captured afterwards."""
        from .densify import densify_and_prune
        self.sync_densify_stats(group)
        newP = densify_and_prune(self.model, self.optimizer, max_grad, min_opacity, extent, max_screen_size,
                                 percent_dense, noise, generator)
        self.slab = FlatGradSlab(self.model.parameters())
        if self.range_slab is not None:
            self.range_slab = RangeGradSlab(self.model.parameters(), self.pipeline_ranges)
        if self.fused is not None:
            self.fused.resize()
        return newP

    def _reduce_and_update_pipelined(self, group=None):
        from . import _lib
        rs, fr, opt = self.range_slab, self.fused, self.optimizer
        collective = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or self.slab.force_collective)
        works = []
        for r in range(rs.K):
            first, count = rs.rows(r)
            gr = _lib.B3gsRawGrads()
            for name, ptr, w in zip(("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity"),
                                    rs.grad_ptrs(r), rs.widths):
                setattr(gr, name, (ptr - 4 * w * first) if w else None)     # indexed by the GLOBAL Gaussian index
            if self._pending_views:
                fr.accumulate_range(self._pending_views, gr, first, count, overwrite=True)
            works.append(dist.all_reduce(rs.chunk(r), op=dist.ReduceOp.SUM, group=group, async_op=True) if collective else None)
        for r in range(rs.K):
            if works[r] is not None:
                works[r].wait()                     # the compute stream waits for that range's all-reduce only
            if self.average and collective:
                rs.chunk(r).div_(dist.get_world_size(group))
            first, count = rs.rows(r)
            opt.step_rows(first, count, rs.grad_ptrs(r), last=(r == rs.K - 1))

    def sync_densify_stats(self, group=None):
        """Make the densification statistics identical on every replica before a densify/prune decision
        (SURVEY 8e): sums of xyz_gradient_accum / denom, maximum of max_radii2D.  Call it every
        densification interval, not every step."""
        m = self.model
        if not (dist.is_available() and dist.is_initialized()) or getattr(m, "denom", None) is None:
            return
        dist.all_reduce(m.xyz_gradient_accum, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(m.denom, op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(m.max_radii2D, op=dist.ReduceOp.MAX, group=group)

    def compute_grads(self, pair_grad_fn=None, loss_fn=None, batch_loss_fn=None):
        """Render this rank's views forward+backward; leaves the summed gradients in the slab.  Contains
        no collective, no host sync and (fused path) no allocation: capturable as one HIP graph."""
        assert sum(f is not None for f in (pair_grad_fn, loss_fn, batch_loss_fn)) == 1
        assert batch_loss_fn is None or self.fused is not None
        if not self.pairs:
            # a rank without views (more ranks than pairs) contributes zeros to the all-reduce
            self.slab.zero()
            if self.range_slab is not None:
                self.range_slab.flat.zero_()
                self._pending_views = []
            self.last_stats = {"views": 0}
            return 0
        if self.fused is not None:
            # the fused multi-view accumulate STORES the gradients: no zero-fill of the slab
            self.slab.rebind()
            n_rendered = self._step_fused(pair_grad_fn, loss_fn, batch_loss_fn)
        else:
            self.slab.zero()
            n_rendered = 0
            for i, (cam, scam, t) in enumerate(self.pairs):
                pkg = self.render(cam, self.model, self.pipe, self.bg)
                spkg = self.render(scam, self.model, self.pipe, self.bg) if scam is not None else None
                n_rendered += 1 + (scam is not None)
                if loss_fn is not None:
                    loss_fn(i, cam, pkg, spkg, t).backward()
                else:
                    outs, grads = zip(*pair_grad_fn(i, pkg, spkg))
                    torch.autograd.backward(list(outs), list(grads))
        self.last_stats = {"views": n_rendered}
        return n_rendered

    def _step_fused(self, pair_grad_fn, loss_fn, batch_loss_fn=None):
        """All views of the step in one render_batch() (binning concurrent, one blend launch), the loss /
        upstream gradients of every pair formed on the current stream, ONE backward call (one blend-
        backward launch for all views), then one per-Gaussian pass for all views (finish_deferred)."""
        views = []
        for i, (cam, scam, t) in enumerate(self.pairs):
            a, b = self._slots[i]
            views.append((cam, a, True))      # densification statistics come from the primary view only
            if scam is not None:             # (train.py:102-104,128: the shifted render's are discarded)
                views.append((scam, b, False))
        self.fused.begin_deferred()
        pkgs = iter(self.fused.render_batch(views, self.bg))
        outs, grads, total = [], [], None
        if batch_loss_fn is not None:
            items = []
            for i, (cam, scam, t) in enumerate(self.pairs):
                pkg = next(pkgs)
                items.append((i, cam, pkg, next(pkgs) if scam is not None else None, t))
            batch_loss_fn(items).backward()
            if self.range_slab is None:
                self.fused.finish_deferred(overwrite=True)
            else:
                self._pending_views = self.fused.take_deferred()
            return len(views)
        for i, (cam, scam, t) in enumerate(self.pairs):
            pkg = next(pkgs)
            spkg = next(pkgs) if scam is not None else None
            if loss_fn is not None:
                li = loss_fn(i, cam, pkg, spkg, t)
                total = li if total is None else total + li
            else:
                o, g = zip(*pair_grad_fn(i, pkg, spkg))
                outs += list(o)
                grads += list(g)
        if loss_fn is not None:
            total.backward()
        else:
            torch.autograd.backward(outs, grads)
        if self.range_slab is None:
            self.fused.finish_deferred(overwrite=True)
        else:   # reduce_and_update() runs the chain rule range by range
            self._pending_views = self.fused.take_deferred()
        return len(views)
