"""Python surface of the rasterizer, name-for-name what the reference imports:

    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
                                                        (gaussian_renderer/__init__.py:14)
    GaussianRasterizationSettings(image_height=..., ..., debug=...)   (:36-49, 12 keywords)
    rasterizer = GaussianRasterizer(raster_settings=...)               (:51)
    color, radii, depth, alpha = rasterizer(means3D=, means2D=, shs=, colors_precomp=,
                                            opacities=, scales=, rotations=, cov3D_precomp=)  (:85-93)

and the two `_C` functions of the un-vendored extension:
`_C.rasterize_gaussians(...)`, `_C.rasterize_gaussians_backward(...)` (+ `_C.mark_visible`) -- `_C` is the COMPILED module
binocular3dgs_amd/_C*.so (csrc/host/*.cpp: host-only C++ against the torch headers, like the reference's own extension),
which also holds the autograd node of this surface and the launch assembly of the raw-parameter node below; it calls
libb3gs_raster.so (hand-written HIP for gfx950) through the C ABI of include/b3gs_raster.h.  This file keeps POLICY only:
which renders share a launch, which capacity a sync-free forward gets, when its N is checked.  No CPU / PyTorch fallback
exists: CPU tensors raise.
"""
from __future__ import annotations

import atexit
import contextlib
import os
import threading
import weakref
from typing import NamedTuple

import torch
from torch import nn

from . import _lib
from . import _C            # the compiled module (fails loudly when it has not been built: python -m binocular3dgs_amd.build)
from ._cuda import device_guard, raw_stream


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


_stream = raw_stream


class _LazyN:
    """Sync-free forward of the autograd surface, TRAINING renders only: the one blocking read-back of num_rendered per
    forward (upstream sizes the binning buffer from it; `api.hip`: b3gs_forward) is paid by the FIRST render of a
    (device, P, W, H) shape and by every render that will not be differentiated (no input requires a gradient: evaluation
    loops, render.py-style scripts -- they get the exact, synchronous forward).  Later differentiated renders of the shape
    run b3gs_forward_capacity with a binning buffer of twice the largest N seen; N stays on the device and travels to
    pinned host memory behind the kernels.  It is looked at
      * in that render's BACKWARD: an N above the capacity -- the image and the loss were computed from truncated tile
        lists -- raises B3gsError out of `loss.backward()`, i.e. before `optimizer.step()` of train.py:149-197 can consume
        anything (`_C.rasterize_gaussians` / GaussianRasterizer: at the entry of the node, before any gradient exists; the
        render() node of a raw-parameter model: as the last act of that backward(), behind its launches, so that the wait
        for the read-back does not drain the device's queue between forward and backward -- whatever the interrupted
        backward() left in `.grad` is to be discarded, as after any backward() that raised);
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
        self.pinned4 = None
        self.slot4 = 0
        self.enabled = os.environ.get("B3GS_DROPIN_SYNC", "0") != "1"
        self.key_bits = 27     # fused render() node: depth sort on 27-bit keys until a render reports a key outside the span
        self.trust_hints = True    # ... and host-side knowledge of equal z rows is trusted until the device contradicts it

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
        if host.numel() > 1 and int(host[1]) & 8:     # fused render() node: a trusted depth-order hint was wrong
            self.trust_hints = False
            return (n, cap, "the depth order of another view although its keys differ (the camera's matrices are not what "
                            "its R / T / trans say); depth-order hints are verified before use from now on")
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
        keep, over, waiting = [], None, False
        for tok in self.pending:
            if tok[4]:
                continue
            # (not forced: an early look only -- the copies of one stream complete in order, so behind the first one that is
            # still in flight nothing is asked: an event query costs microseconds and every forward comes through here)
            if waiting or not (force or tok[3].query()):
                waiting = not force
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
        with device_guard(n_dev.device):
            ev.record()                                   # (the current stream: where the copy was just enqueued)
        tok = [key, cap, host, ev, False]
        self.pending.append(tok)
        return tok

    def track_rows(self, keys, caps, rows):
        """The same for the views of ONE batched forward: `rows` = their [n, 4] word rows, consecutive in device memory ->
        one device-to-host copy and one event for all of them (a small copy occupies the stream for 10-20 us)."""
        n = rows.shape[0]
        if self.pinned4 is None:
            self.pinned4 = torch.zeros((self.RING, 4), dtype=torch.int32).pin_memory()
        if len(self.pending) >= self.RING - n:
            self.poll(force=True)
        if self.slot4 + n > self.RING:
            self.slot4 = 0
        host = self.pinned4[self.slot4:self.slot4 + n]
        self.slot4 += n
        host.copy_(rows, non_blocking=True)
        ev = torch.cuda.Event()
        with device_guard(rows.device):
            ev.record()
        toks = [[keys[k], caps[k], host[k, :2], ev, False] for k in range(n)]
        self.pending.extend(toks)
        return toks

    def flush_at_exit(self):
        try:
            self.poll(force=True)
        except Exception as exc:   # nothing can repeat the step any more: say so
            import sys
            print(f"[binocular3dgs_amd] WARNING at exit: {exc}", file=sys.stderr)


_lazy = _LazyN()
atexit.register(_lazy.flush_at_exit)


def _module_node(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs, differentiated):
    """The autograd node of the module surface (csrc/host/raster.cpp: RasterizeFn) + the sync-free-N policy around it.
    differentiated: the caller will run backward() on this render -- only then may the forward skip the read-back of
    num_rendered (_LazyN: a binning buffer of known capacity, N checked at the entry of the node's backward); a render nobody
    differentiates (evaluation) takes the exact synchronous forward."""
    P = means3D.shape[0] if means3D.dim() == 2 else 0
    key, cap = None, 0
    if P > 0 and means3D.is_cuda:
        dev = means3D.device
        key = (dev.index if dev.index is not None else torch.cuda.current_device(), P, int(rs.image_width), int(rs.image_height))
        if differentiated and _lazy.enabled and not rs.debug:
            _lazy.poll()
            cap = _lazy.capacity.get(key) or 0
    color, radii, depth, alpha, n_info = _C.rasterize_gaussians_autograd(
        means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs.bg, rs.viewmatrix, rs.projmatrix,
        rs.campos, int(rs.image_height), int(rs.image_width), rs.tanfovx, rs.tanfovy, rs.scale_modifier, int(rs.sh_degree),
        bool(rs.prefiltered), bool(rs.debug), cap)
    if cap:
        tok = _lazy.track(key, cap, n_info)
        node = color.grad_fn
        if node is not None:
            # truncated tile lists raise at the ENTRY of the node's backward: no gradient leaves it
            node.register_prehook(lambda grads, _tok=tok: _lazy.confirm(_tok))
    elif key is not None:
        _lazy.note(key, int(n_info))          # every exact render teaches the capacity of its shape
    return color, radii, depth, alpha


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    # will this render be differentiated?
    differentiated = torch.is_grad_enabled() and any(
        torch.is_tensor(t) and t.requires_grad
        for t in (means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp))
    return _module_node(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                        differentiated)


class _RasterizeGaussians:
    """Upstream's `_RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
    cov3Ds_precomp, raster_settings)`: the node itself is C++ (`_C.rasterize_gaussians_autograd`); applied directly with the
    upstream's nine arguments it is the exact synchronous forward."""

    @staticmethod
    def apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
              differentiated=None):
        return _module_node(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                            bool(differentiated))


# ---------------------------------------------------------------------------------------------------------------
# render() handed a GaussianModel: one autograd node on the RAW parameters
# ---------------------------------------------------------------------------------------------------------------
# B3GS_DROPIN_INPLACE_GRADS=0: the node always returns its gradients to autograd (no self-accumulation, no batching of
# the backward of several renders); B3GS_DROPIN_ORDER_HINT=0: every render sorts its own depth keys
_INPLACE_GRADS = os.environ.get("B3GS_DROPIN_INPLACE_GRADS", "1") != "0"
_ORDER_HINT = os.environ.get("B3GS_DROPIN_ORDER_HINT", "1") != "0"
# B3GS_DROPIN_LAZY=0: every render() launches its forward before it returns (see _LazyOut / _launch_forward)
_LAZY_FWD = os.environ.get("B3GS_DROPIN_LAZY", "1") != "0"
_LAZY_MAX = max(1, min(8, int(os.environ.get("B3GS_DROPIN_LAZY_MAX", "2"))))    # renders per batched forward (a pair)
# Round 6: the number of renders a pending forward waits for FOLLOWS THE LOOP unless B3GS_DROPIN_LAZY_MAX pins it: train.py
# renders a pair and consumes it (2: the second render launches the batch when it returns, no NaN fill, nothing moves behind the
# first touch -- measured 1.4-4 % better for that loop than "wait until touched"); a loop that renders its six views before it
# consumes any gets ONE six-view forward (+18 % on the 6-view iteration).  Rule (_lazy_adapt_step, once per backward()): grow to
# the number of differentiated renders since the previous backward (at most 8) when none of the pending batches of that
# iteration was consumed before it was full; a batch consumed early (a touch while fewer than the current maximum are pending)
# shrinks the maximum to that size and blocks growth for a while (4, 8, ... 64 iterations).  B3GS_DROPIN_LAZY_ADAPT=0: fixed.
_LAZY_ADAPT = ("B3GS_DROPIN_LAZY_MAX" not in os.environ) and os.environ.get("B3GS_DROPIN_LAZY_ADAPT", "1") != "0"
# B3GS_DROPIN_LAZY_IDLE=0: a render waits for its partner only while the stream is busy or while an adaptive rule (see
# _RasterizeRaw.forward) finds that starting at once does not overlap anything.  Default 1 = always wait: measured INTERLEAVED
# inside one process (tools/ab_interleaved.py lazy: blocks of 40 iterations cycling through the settings, medians of 12 --
# separate runs on the shared host differ by +-15 %), train.py's loop at 500k Gaussians: always 793 / 576 iters/s
# (504x378 / 800x600), the adaptive rule 744 / 543, never waiting 779 / 564; render() + fused loss: 893 / 700, 891 / 716, 787 / 632.
# (The rule had become the default earlier in round 5 on the strength of separate runs.)
_LAZY_WHEN_IDLE = os.environ.get("B3GS_DROPIN_LAZY_IDLE", "1") != "0"


class _DropinState:
    """Everything the raw-parameter surface remembers between calls, in ONE object behind ONE re-entrant lock (round 4 kept
    seven module-level dicts that the caller's thread and autograd's device thread both wrote).  The entry points that
    touch it -- the forward and the backward of _RasterizeRaw, the flush of pending forwards (any thread that first
    touches a pending output), the final callback of a backward() -- hold the lock for their whole body, so renders issued
    from several Python threads (their own models, their own streams) interleave as whole calls.  Per-backward state is
    keyed by the engine's graph-task id: concurrent or nested backward() calls keep separate entries."""

    def __init__(self):
        self.lock = threading.RLock()
        self.raw_scratch = {}     # (device index, P) -> list of zeroed [P * 10] float buffers, one per view of a batched
        #                           backward; left clean by every backward (b3gs_backward_raw_accumulate)
        self.zero_m2d = {}        # (device index, P) -> zeros [P, 3]: storage behind every render's `viewspace_points` leaf
        self.order_hint = {}      # device index -> the last raw forward: dict(P, key_bits, geom, xyz, words, zkey, stream)
        self.last_raw_ctx = {}    # device index -> weakref of the last DIFFERENTIATED raw forward's node (+ what it rendered)
        self.pending_fwd = {}     # device index -> [_PendingFwd]: differentiated renders whose forward is not launched yet
        self.tasks = {}           # graph-task id -> _TaskState of a backward() in flight
        self.word_ring = {}       # device index -> [zeroed int32 ring, next slot]
        self.busy = {}            # (device index, stream) -> event behind the last forward / backward this module launched there
        self.adapt = {}           # (device index, stream) -> the eager-on-idle rule's memory (see _RasterizeRaw.forward)
        self.lazy_adapt = {}      # device index -> the adaptive batch size of pending forwards (see _LAZY_ADAPT)
        self.stats = {"hinted": 0, "trusted": 0, "deferred": 0, "batched_views": 0, "launches": 0, "lazy_batches": 0,
                      "lazy_views": 0, "shared": 0, "eager_idle": 0}   # (tests / bench read these)


_S = _DropinState()
# (the names tests and tools have always used: the SAME dict objects)
_raw_scratch, _zero_m2d, _order_hint, _last_raw_ctx = _S.raw_scratch, _S.zero_m2d, _S.order_hint, _S.last_raw_ctx
_pending_fwd, _stats = _S.pending_fwd, _S.stats


def _locked(fn):
    import functools

    @functools.wraps(fn)
    def wrapper(*a, **k):
        with _S.lock:
            return fn(*a, **k)
    return wrapper


# ---- private torch entry points, probed once -------------------------------------------------------------------------
# The batched backward and the end-of-backward capacity check lean on three engine hooks that are not public API:
#   torch._C._will_engine_execute_node(node)        "will this backward() run that node?"
#   torch._C._current_graph_task_id()               identity of the backward() in flight
#   Variable._execution_engine.queue_callback(fn)   "call fn as the last act of this backward()"
# and the pending forward on torch._C.DisableTorchFunctionSubclass.  A torch build without one of them gets the plain
# behaviour instead of an AttributeError in the middle of loss.backward(): every node launches its own backward and RETURNS
# its gradients to autograd, the capacity check sits at the entry of the node, every render launches before it returns.
def _probe_engine() -> bool:
    try:
        eng = torch.autograd.Variable._execution_engine
        return (callable(torch._C._will_engine_execute_node) and callable(torch._C._current_graph_task_id)
                and callable(eng.queue_callback))
    except AttributeError:
        return False


def _probe_lazy() -> bool:
    return hasattr(torch._C, "DisableTorchFunctionSubclass") and hasattr(torch.Tensor, "as_subclass")


_ENGINE_HOOKS = _probe_engine()
_LAZY_OK = _probe_lazy()


def _refresh_probes():
    """(tests: after monkeypatching the private entry points away)"""
    global _ENGINE_HOOKS, _LAZY_OK
    _ENGINE_HOOKS, _LAZY_OK = _probe_engine(), _probe_lazy()
    return _ENGINE_HOOKS, _LAZY_OK


def _mark_busy(di, stream_id):
    """An event behind the work this module just queued on the CURRENT stream (id `stream_id`; see the wait decision in
    _RasterizeRaw.forward)."""
    if _LAZY_WHEN_IDLE:
        return              # (only the eager-on-idle rule, B3GS_DROPIN_LAZY_IDLE=0, ever asks)
    key = (di, stream_id)
    ev = _S.busy.get(key)
    if ev is None:
        if len(_S.busy) > 16:
            _S.busy.clear()
        ev = _S.busy[key] = torch.cuda.Event()
    # (Event.record() without an argument means the current stream of the CURRENT device -- name the stream)
    ev.record(torch.cuda.ExternalStream(stream_id, device=torch.device("cuda", di)))


def _stream_obj(p):
    """The torch Stream of a pending forward / a parked backward job (built only when a launch crosses streams)."""
    return torch.cuda.ExternalStream(p.stream_id, device=p.dev)


def _current_task() -> int:
    return torch._C._current_graph_task_id() if _ENGINE_HOOKS else -1


def _will_execute(node) -> bool:
    return bool(_ENGINE_HOOKS and torch._C._will_engine_execute_node(node))


class _PendingFwd:
    """One render() whose forward is still to be launched: everything b3gs_forward_raw_batch needs, already allocated."""
    __slots__ = ("ctx", "view", "geom", "binning", "img", "out", "radii", "words", "cap", "key", "hint", "trusted", "zkey",
                 "stream_id", "stream", "fkey", "xyz_id", "vis", "key_bits", "dev", "P", "ring", "row", "handed", "fresh")


def _meta_funcs():
    """Tensor functions that read no element: they do not make a pending forward run."""
    T = torch.Tensor
    fs = {getattr(T, n) for n in ("size", "dim", "numel", "stride", "is_contiguous", "element_size", "is_floating_point",
                                  "is_complex", "is_signed", "get_device", "storage_offset", "ndimension", "nelement",
                                  "__len__", "__hash__") if hasattr(T, n)}
    for name in ("shape", "dtype", "device", "requires_grad", "grad_fn", "is_cuda", "ndim", "is_leaf", "layout", "names",
                 "is_sparse", "is_quantized", "is_meta", "grad", "output_nr", "_version", "is_nested", "_backward_hooks"):
        g = getattr(getattr(T, name, None), "__get__", None)
        if g is not None:
            fs.add(g)
    return fs


class _LazyOut(torch.Tensor):
    """What render() hands out while the forward of a differentiated render is still pending: the REAL output tensors
    (storage allocated, autograd node attached) under a subclass whose only job is to notice the first use.  train.py:100-128
    looks at nothing between its two render() calls -- it only files the outputs away -- so the input view and its shifted
    partner can be launched as ONE two-view forward (one projection pass over the parameters, batched binning, one blend
    launch) when the second call arrives.  Any torch function, method or operator applied to a pending output (everything but
    shape / dtype / device-style metadata) launches what is pending first; so does the backward of the render's node, a
    render of other parameters, another stream, another image size, and the `B3GS_DROPIN_LAZY_MAX`-th (2nd) pending render.
    Until then the images hold NaN, so that a consumer that bypasses torch's function dispatch (a pybind extension reading
    the raw pointer) cannot go unnoticed.  B3GS_DROPIN_LAZY=0 switches the whole mechanism off."""

    _META = None

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        if _pending_fwd:
            if cls._META is None:
                cls._META = _meta_funcs()
            if func not in cls._META:
                _flush_pending()
        with torch._C.DisableTorchFunctionSubclass():
            return func(*args, **(kwargs or {}))

    def __reduce_ex__(self, proto):
        # (pickling / torch.save of an output that is still pending: launch, then pickle as the plain tensor it is)
        _flush_pending()
        with torch._C.DisableTorchFunctionSubclass():
            return self.as_subclass(torch.Tensor).__reduce_ex__(proto)


def _lazy_max(di) -> int:
    """Renders a pending forward of device `di` waits for (>= _LAZY_MAX: a test or tool that raises _LAZY_MAX still gets it)."""
    if not _LAZY_ADAPT:
        return _LAZY_MAX
    a = _S.lazy_adapt.get(di)
    return _LAZY_MAX if a is None else max(_LAZY_MAX, a["max"])


def _lazy_adapt_state(di):
    a = _S.lazy_adapt.get(di)
    if a is None:
        a = _S.lazy_adapt[di] = {"max": _LAZY_MAX, "renders": 0, "small": 0, "hold": 0, "fails": 0}
    return a


def _lazy_adapt_step():
    """Once per backward() (its final callback): see _LAZY_ADAPT."""
    for a in _S.lazy_adapt.values():
        renders, a["renders"] = a["renders"], 0
        small, a["small"] = a["small"], 0
        if small:
            if small < a["max"]:
                a["max"] = max(_LAZY_MAX, small)
                a["hold"] = min(64, 4 << a["fails"])
                a["fails"] = min(a["fails"] + 1, 4)
        elif a["hold"] > 0:
            a["hold"] -= 1
        elif renders > a["max"]:
            a["max"] = min(8, renders)


def touch_pending(*tensors):
    """For consumers that do NOT go through torch's function dispatch -- the entry points of the compiled `_C` module read
    raw pointers: a pending output among `tensors` launches what is pending first (what `_LazyOut.__torch_function__` does
    for every torch function)."""
    if _pending_fwd:
        for t in tensors:
            if type(t) is _LazyOut:
                _flush_pending()
                return


@_locked
def _flush_pending(di=None, full=False):
    """Launch the pending forwards (of one device, or of all).  full: the batch reached its size (not: somebody needed it)."""
    for d in ([di] if di is not None else list(_pending_fwd)):
        lst = _pending_fwd.pop(d, None)
        if lst:
            if _LAZY_ADAPT and not full and len(lst) < _lazy_max(d):
                a = _lazy_adapt_state(d)
                a["small"] = len(lst) if not a["small"] else min(a["small"], len(lst))
            _launch_forward(lst)


def _launch_forward(lst):
    """ONE b3gs_forward_raw_batch for the pending renders `lst` (same parameters, stream, image size, key width)."""
    p0 = lst[0]
    n = len(lst)
    specs, order_from = [], []
    for k, p in enumerate(lst):
        frm, trusted, hint_geom = -1, 0, None
        if (k > 0 and _ORDER_HINT and _lazy.trust_hints and p.zkey and p.zkey == lst[k - 1].zkey
                and order_from[k - 1] == -1):
            # the host knows that this view's z row is its predecessor's (camera_depth_key): one depth sort for both; the
            # projection compares the keys and a difference drops the step (bit 3) and ends the trust
            frm, trusted = k - 1, 1
            _stats["shared"] += 1
        if p.hint is not None and n == 1:
            hint_geom, trusted = p.hint["geom"], int(p.trusted)
            _stats["hinted"] += 1
            _stats["trusted"] += int(p.trusted)
        order_from.append(frm)
        # (view, geometry, binning, capacity, image, out, radii, words, visible, key bits, order from, hint, trusted, fresh image,
        #  seg1 fraction): csrc/host/raster.cpp raw_forward_launch fills the B3gsForwardView array from these
        specs.append((p.view, p.geom, p.binning, p.cap, p.img, p.out, p.radii, p.words, p.vis, p.key_bits, frm, hint_geom,
                      trusted, p.fresh, 0.0))
    other = raw_stream(p0.dev) != p0.stream_id
    cur = torch.cuda.current_stream(p0.dev) if other else None
    with device_guard(p0.dev), (torch.cuda.stream(_stream_obj(p0)) if other else contextlib.nullcontext()):
        _C.raw_forward_launch(specs, p0.stream_id)
        try:
            if n > 1 and all(q.ring is p0.ring and q.row == p0.row + k for k, q in enumerate(lst)):
                toks = _lazy.track_rows([q.key for q in lst], [q.cap for q in lst], p0.ring[p0.row:p0.row + n])
            else:
                toks = [_lazy.track(q.key, q.cap, q.words[:2]) for q in lst]
        except BaseException:
            # (track*() polls older renders when its ring is full and may raise THEIR overflow: these views are launched
            # all the same -- mark them so, with nothing left to confirm)
            for p in lst:
                p.ctx.lazy_token = [p.key, p.cap, None, None, True]
            raise
        for p, tok in zip(lst, toks):
            p.ctx.lazy_token = tok
        _mark_busy(_dev_index(p0.dev), p0.stream_id)
    if other:
        # (whoever triggered the launch is about to read the outputs on ITS stream -- a consumer on a side stream
        # that ordered itself behind render()'s stream before the first use saw an empty queue there)
        cur.wait_stream(_stream_obj(p0))
    # the outputs handed out as _LazyOut are plain tensors from here on: nothing is pending behind them any more, and a
    # later torch.save() / deepcopy / subclass check sees torch.Tensor, not a private class of this module
    for p in lst:
        for ref in (p.handed or ()):
            t = ref()
            if t is not None:
                try:
                    t.__class__ = torch.Tensor
                except TypeError:
                    pass
        p.handed = None
    _stats["lazy_batches"] += 1
    _stats["lazy_views"] += n
    if _ORDER_HINT:
        # (what a later render may adopt: the order of a view that SORTED -- a view that borrowed its neighbour's order inside
        # this batch has no sorted arrays of its own)
        p = [q for k, q in enumerate(lst) if order_from[k] == -1][-1]
        _order_hint[_dev_index(p.dev)] = dict(P=p.P, key_bits=p.key_bits, geom=p.geom, xyz=p.xyz_id, words=p.words, zkey=p.zkey,
                                              stream=p.stream_id)


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


_word_ring = _S.word_ring


def _dev_index(dev) -> int:
    return dev.index if dev.index is not None else torch.cuda.current_device()


def _zero_words(dev, with_row=False):
    """Four zeroed int32 words on `dev` without a fill kernel per render ([N, overflow word, depth-key mismatch, spare]):
    slots of a ring that is zeroed once per lap (a slot handed out is read by the host long before the ring comes round:
    4096 renders later).  with_row: also (ring, row index) -- consecutive calls get consecutive rows of one tensor."""
    idx = _dev_index(dev)
    ent = _word_ring.get(idx)
    if ent is None or ent[1] >= ent[0].shape[0]:
        ent = _word_ring[idx] = [torch.zeros((4096, 4), dtype=torch.int32, device=dev), 0]
        # (slots are handed to launches on ANY stream: the fill must have run before the first of them -- once per 4096 renders)
        torch.cuda.current_stream(dev).synchronize()
    w = ent[0][ent[1]]
    ent[1] += 1
    return (w, ent[0], ent[1] - 1) if with_row else w


def camera_depth_key(cam):
    """What the HOST knows about the z row of a camera's view matrix (every depth key is a function of it and of the
    positions): bytes that are equal for two cameras exactly when their z rows are the same bits, or False when it cannot
    tell.  camera.Camera.shifted() keeps its parent's row by construction (`same_depth_as`); a camera that carries the
    reference's constructor arguments (scene/cameras.py:17-58: R, T, trans, scale) gets the row re-derived the way that
    constructor derived the matrix (utils/graphics_utils.py:38-49 = camera.world_to_view, bit-equal per golden G3) -- the
    reference's getShiftedCamera (scene/__init__.py:96-115) lands on its input view's row in ~4 of 5 draws, one ulp beside
    it in the others (golden G4).  Cached on the camera object.  Only ever used to decide whether a depth sort is LAUNCHED;
    the device checks the keys themselves."""
    k = getattr(cam, "_b3gs_zkey", None)
    if k is not None:
        return k
    donor = getattr(cam, "same_depth_as", None)
    if donor is not None:
        k = camera_depth_key(donor)
    else:
        try:
            from .camera import world_to_view
            import numpy as np
            k = world_to_view(cam.R, cam.T, getattr(cam, "trans", np.zeros(3)), getattr(cam, "scale", 1.0))[2, :].tobytes()
        except Exception:
            k = False
    try:
        cam._b3gs_zkey = k
    except Exception:
        pass
    return k


def viewspace_leaf(xyz: torch.Tensor) -> torch.Tensor:
    """render()'s `screenspace_points` (gaussian_renderer/__init__.py:26-30: zeros whose .grad receives the screen-space
    gradient): a fresh LEAF per render over one shared block of zeros per (device, P) -- its values are never read or
    written by anybody, so the 12 bytes per Gaussian are not filled again for every render."""
    key = (_dev_index(xyz.device), xyz.shape[0])
    z = _zero_m2d.get(key)
    if z is None:
        if len(_zero_m2d) > 4:
            _zero_m2d.clear()
        z = _zero_m2d[key] = torch.zeros((xyz.shape[0], 3), dtype=torch.float32, device=xyz.device)
    return z.detach().requires_grad_(True)


class _RawJob:
    """One render's backward, ready to be launched: its node, what the node saved, its pixel gradients."""
    __slots__ = ("ctx", "saved", "gc", "gd", "ga", "dev", "stream_id", "needs_m2d", "m2d_leaf", "task")

    def __init__(self, ctx, gc, gd, ga, task):
        self.ctx, self.saved, self.gc, self.gd, self.ga, self.task = ctx, ctx.saved_tensors, gc, gd, ga, task
        self.dev = gc.device
        self.stream_id = raw_stream(gc.device)
        self.needs_m2d = bool(ctx.needs_input_grad[6])
        self.m2d_leaf = ctx.m2d_leaf


def _leaf_accumulates(node, t) -> bool:
    """Will the engine run `t`'s AccumulateGrad in THIS backward, with nothing hooked to the gradient?  (False under
    torch.autograd.grad(), for an input left out of backward(inputs=...), for a non-leaf, for hooked tensors.)"""
    if node is None or type(node).__name__ != "AccumulateGrad" or getattr(node, "variable", None) is not t:
        return False
    try:
        if not _will_execute(node):
            return False
    except RuntimeError:      # "a leaf node was passed ... but we are currently running autograd.grad()"
        return False
    if t._backward_hooks or getattr(t, "_post_accumulate_grad_hooks", None):
        return False
    g = t.grad
    return g is None or (g.dtype == torch.float32 and g.shape == t.shape and g.is_contiguous() and g.device == t.device
                         and g.layout == torch.strided and not g.requires_grad)


def _self_accumulate_ok(ctx, params) -> bool:
    """May this node add its gradients to `.grad` itself (what the six AccumulateGrad nodes behind it would do, minus one
    pass over 92 bytes per Gaussian per render -- and the precondition for launching the backward of several renders as
    one batch)?  Only inside a plain accumulating backward(): every parameter a leaf whose AccumulateGrad the engine is
    going to execute in this graph task, no hooks, no create_graph.  torch.autograd.grad(), backward(inputs=[...]) without
    all six, hooked parameters, double backward: the gradients are RETURNED to autograd like the reference's."""
    if not (_INPLACE_GRADS and _ENGINE_HOOKS) or torch.is_grad_enabled() or not all(ctx.needs_input_grad[:6]):
        return False
    nf = ctx.next_functions
    # (the answer for the six parameters is the same for every node of this graph task that renders them: asked once)
    acc = _task_state().acc
    ok = acc.get(ctx.batch_key)
    if ok is None:
        ok = acc[ctx.batch_key] = all(_leaf_accumulates(nf[i][0], t) for i, t in enumerate(params))
    if not ok:
        return False
    return (not ctx.needs_input_grad[6]) or _leaf_accumulates(nf[6][0], ctx.m2d_leaf)


class _TaskState:
    """What one backward() (one graph task of the engine) has parked with this module."""
    __slots__ = ("streams", "flush", "tokens", "acc", "created")

    def __init__(self):
        self.streams, self.flush, self.tokens = [], [], []
        self.acc = {}          # batch_key -> may the nodes of these parameters accumulate into .grad themselves?
        self.created = []      # tensors whose .grad THIS backward created (set back to None when the step is refused)


def _task_state() -> _TaskState:
    """The entry of the backward() in flight (created on first use, together with its final callback)."""
    task = _current_task()
    st = _S.tasks.get(task)
    if st is None:
        if len(_S.tasks) >= 16:      # entries of backward() calls that raised inside a node (their final callback never ran)
            for old in sorted(_S.tasks)[:len(_S.tasks) - 15]:
                del _S.tasks[old]
        st = _S.tasks[task] = _TaskState()
        if _ENGINE_HOOKS:
            torch.autograd.Variable._execution_engine.queue_callback(lambda: _end_of_backward(task))
    return st


@_locked
def _end_of_backward(task):
    """Final callback of the graph task (runs on the caller's ambient streams, after the engine has synchronised them with
    the streams of the leaves IT accumulated), i.e. the last thing `loss.backward()` does before it returns:
      * launch what is still deferred -- a node that was expected to run in this backward and did not;
      * make the caller's stream wait for the streams this module wrote `.grad` on;
      * check the N of every sync-free forward differentiated in this backward (_LazyN): truncated tile lists raise HERE, out
        of backward() and therefore before `optimizer.step()` of train.py:196-198 -- but behind the backward's launches, so
        the host waits for the forwards' read-backs while the device still has the whole backward queued (checking at the
        entry of the first node drained the queue between forward and backward of every iteration: ~0.2 ms of idle device).
        A refused step leaves no half-made gradients behind: every `.grad` this backward CREATED is set back to None
        (gradients it added into an existing `.grad` cannot be taken out again -- the message says to zero them)."""
    st = _S.tasks.pop(task, None)
    if st is None:
        return
    if _LAZY_ADAPT:
        _lazy_adapt_step()
    for ref in st.flush:
        node = ref()
        jobs = getattr(node, "pending", None) if node is not None else None
        if jobs:
            node.pending = []
            jobs = [j for j in jobs if j.task == task]     # (see _RasterizeRaw.backward)
            if jobs:
                _launch_backward(jobs, True, st)
    for dev, sid in st.streams:
        if raw_stream(dev) != sid:
            torch.cuda.current_stream(dev).wait_stream(torch.cuda.ExternalStream(sid, device=dev))
    err = None
    for tok in st.tokens:
        try:
            _lazy.confirm(tok)
        except _lib.B3gsError as exc:      # (every token is resolved -- capacities grow -- before the first error leaves)
            err = err or exc
    if err is not None:
        for t in st.created:
            t.grad = None
        raise _lib.B3gsError(f"{err} (gradients this backward() created were discarded; call zero_grad() before repeating "
                             f"the step if .grad held earlier contributions)")


def _register_task(flush_ref=None, token=None):
    st = _task_state()
    if flush_ref is not None:
        st.flush.append(flush_ref)
    if token is not None:
        st.tokens.append(token)
    return st


def _launch_backward(jobs, self_acc, task_state=None):
    """Blend backward + per-Gaussian chain rule of `jobs` (renders of the SAME parameter tensors), at most 8 views per
    launch.  self_acc: the gradients are added to (or become) `.grad` of the parameters and of every render's
    `viewspace_points` leaf, nothing is returned; otherwise (one job) fresh tensors are returned for autograd."""
    params = jobs[-1].saved[:6]
    dev, P = params[0].device, params[0].shape[0]
    f32 = dict(dtype=torch.float32, device=dev)
    cur_id = raw_stream(dev)
    skey = (_dev_index(dev), P, cur_id)       # (per stream: two streams' backwards may be in flight together)
    pool = _raw_scratch.get(skey)
    if pool is None:
        if len(_raw_scratch) > 4:
            _raw_scratch.clear()
        pool = _raw_scratch[skey] = []
    for j in jobs:
        if j.stream_id != cur_id:           # pixel gradients produced on another stream (deferred from another node)
            torch.cuda.current_stream(dev).wait_stream(_stream_obj(j))
    if self_acc:
        missing = [t.grad is None for t in params]
        overwrite = all(missing)
        grads = []
        tstate = task_state if task_state is not None else _register_task()
        if (dev, cur_id) not in tstate.streams:
            tstate.streams.append((dev, cur_id))
        for t, m in zip(params, missing):
            if m:
                t.grad = torch.empty_like(t) if overwrite else torch.zeros_like(t)
                tstate.created.append(t)
            grads.append(t.grad)
    else:
        assert len(jobs) == 1
        overwrite = True
        grads = [torch.empty_like(t) for t in params]
    while len(pool) < min(len(jobs), 8):      # (one zeroed scratch block per view of a batched launch; left zero by it)
        pool.append(torch.zeros((max(_C.backward_scratch_floats(P), 1),), **f32))
    # blend backward + chain rule, at most 8 views per launch pair (csrc/host/raster.cpp raw_backward_launch).  overwrite
    # mode: every row of every gradient tensor is stored (zeros for Gaussians without a contribution); accumulate mode:
    # only the rows that received something are touched
    m2d_out = _C.raw_backward_launch(
        [(j.ctx.view, j.saved[6], j.saved[7], j.saved[8], j.saved[9], j.gc, j.gd, j.ga, j.needs_m2d, j.ctx.cap) for j in jobs],
        pool, grads, overwrite, cur_id)
    _stats["launches"] += (len(jobs) + 7) // 8
    _stats["batched_views"] += len(jobs)
    _mark_busy(_dev_index(dev), cur_id)
    # the chain rule uses view 0's depth-sort key arrays as scratch -- a later forward must not adopt "sorted
    # keys" from a geometry buffer whose backward has run (include/b3gs_raster.h, depth_order_hint)
    di = _dev_index(dev)
    hint = _order_hint.get(di)
    if hint is not None and any(hint["geom"] is j.saved[7] for j in jobs):
        _order_hint.pop(di, None)
    if self_acc:
        for j, g in zip(jobs, m2d_out):
            if g is not None:
                leaf = j.m2d_leaf
                if leaf.grad is None:
                    leaf.grad = g
                    tstate.created.append(leaf)
                else:
                    leaf.grad += g
        return None
    return grads, m2d_out[0]


class _RasterizeRaw(torch.autograd.Function):
    """gaussian_renderer/__init__.py:53-93 as ONE node: the accessors (get_scaling / get_rotation / get_opacity /
    get_features: ~30 PyTorch kernels with their autograd per render) run inside the projection and chain-rule kernels
    (b3gs_forward_raw_batch / b3gs_backward_raw_accumulate), the depth sort takes three passes, and the Gaussians are
    binned into the tiles their alpha >= 1/255 footprint reaches (same images; `_C.rasterize_gaussians` keeps the
    reference's binning rule and bit-exact lists).

    What an unchanged train.py:100,128 gets on top of that (the input view and its shifted partner are two render()
    calls of one iteration):
      * forward -- by default the first render WAITS for the second (its outputs are _LazyOut: they launch what is pending
        at their first use) and both run as one two-view b3gs_forward_raw_batch with one depth sort when the host knows the
        z rows are equal (keys compared on the device, ABI 8).  When every render launches by itself (something consumed
        the first result in between, B3GS_DROPIN_LAZY=0) the second render adopts the depth order of the first: its view matrix has the same z row (the shift
        is along the camera x axis, scene/__init__.py:96-115), so every depth key is equal; the library CHECKS that on the
        device (B3gsForwardView::depth_order_hint) and sorts itself when a key differs, so the answer never depends on it;
      * backward -- inside a plain accumulating `loss.backward()` the nodes of renders of the same parameters hand their
        work down the chain (the engine runs the later render first) and the earliest one launches ONE blend backward and
        ONE chain-rule pass for all of them, adding into `.grad` what the AccumulateGrad nodes would have added.  Under
        torch.autograd.grad(), backward(inputs=...), hooks, create_graph the gradients are returned per node."""

    @staticmethod
    @_locked
    def forward(ctx, xyz, f_dc, f_rest, scaling, rotation, opacity, means2D, cfg):
        dev, P = xyz.device, xyz.shape[0]
        W, H = cfg["W"], cfg["H"]
        K = f_dc.shape[1] + f_rest.shape[1]
        di = _dev_index(dev)
        key = ("raw", di, P, W, H)
        lazy = bool(cfg["differentiated"]) and _lazy.enabled and not cfg["debug"]
        if lazy:
            _lazy.poll()
        stream_id = _stream(dev)
        # the chain of this iteration's renders of these parameters (see backward)
        batch_key = (di, P, K, int(cfg["sh_degree"]), float(cfg["scale_modifier"]), bool(cfg["debug"]),
                     tuple(t.data_ptr() for t in (xyz, f_dc, f_rest, scaling, rotation, opacity)))
        # may this render's forward WAIT for its partner (see _LazyOut)?  Only when render() hands out its outputs as
        # _LazyOut, the render will be differentiated and the capacity of its shape is known (sync-free N)
        fkey = (batch_key, stream_id, W, H, _lazy.key_bits, xyz._version)
        wait = bool(_LAZY_FWD and _LAZY_OK and cfg.get("lazy_outputs") and lazy and key in _lazy.capacity and P > 0)
        pend = _pending_fwd.get(di)
        if wait and not _LAZY_WHEN_IDLE:
            # Waiting for the partner pays while the device still works on what this module queued before (the previous
            # pair of a loop that renders several pairs per iteration) and whenever the loop is bound by the HOST (a shared
            # launch is ~20 kernel launches less).  When the device is idle AND the scene is large enough for it to matter
            # -- train.py's own loop reads the loss / indexes with a mask between two iterations, so every iteration starts
            # on an empty queue -- the device would sit idle while the host prepares the partner's render and builds its
            # camera (train.py:124-127): start this view now; its partner adopts the depth order.  "Large enough" is
            # observed, not guessed: when the partner of an eagerly started view arrives, was that view's forward still
            # running?  Yes -> the overlap was real, stay eager; no (twice in a row) -> wait again, and probe once in a while.
            am = _S.adapt.setdefault((di, stream_id), {"eager": True, "miss": 0, "probe": 0, "check": None, "key": None})
            if not pend:
                if am["check"] is not None and am["key"] == batch_key:      # this is the partner of an eager view: sample
                    still_running = not am["check"].query()
                    am["miss"] = 0 if still_running else am["miss"] + 1
                    if still_running:
                        am["eager"] = True
                    elif am["miss"] >= 2:
                        am["eager"], am["probe"] = False, 64
                    am["check"] = None
                else:
                    ev = _S.busy.get((di, stream_id))
                    if ev is None or ev.query():                            # nothing of ours is executing
                        if am["eager"] or am["probe"] <= 0:
                            wait = False
                            am["key"], am["want_check"] = batch_key, True
                            _stats["eager_idle"] += 1
                        else:
                            am["probe"] -= 1
        if pend and not (wait and pend[0].fkey == fkey):
            _flush_pending(di)        # something else is rendered first: what is pending goes now
            pend = None
        cap = _lazy.capacity.get(key) or max(1 << 20, 12 * P)
        # everything one view needs, in ONE call into the compiled module (csrc/host/raster.cpp raw_prepare): the camera
        # half of B3gsScene + the six parameter pointers (`view`), geometry / image state / binning buffers, the outputs
        # ([5,H,W]: colour | depth | alpha) and render()'s `visibility_filter` = radii > 0 (written by the projection,
        # B3gsForwardView::visible).  The image state is recycled allocator memory: `fresh_image` tells the library to read
        # nothing from it.  A render that will wait for its partner gets NaN images (see _LazyOut) unless it completes the batch
        lazy_max = _lazy_max(di)
        if _LAZY_ADAPT and cfg["differentiated"]:
            _lazy_adapt_state(di)["renders"] += 1
        nan_fill = wait and len(pend or ()) + 1 < lazy_max
        view, geom, img, out, color, depth, alpha, radii, vis, binning = _C.raw_prepare(
            xyz, f_dc, f_rest, scaling, rotation, opacity, cfg["bg"], cfg["viewmatrix"], cfg["projmatrix"], cfg["campos"], W, H,
            cfg["tanfovx"], cfg["tanfovy"], cfg["scale_modifier"], int(cfg["sh_degree"]), bool(cfg["debug"]), cap,
            bool(cfg.get("lazy_outputs")), nan_fill)
        cfg["vis"] = vis
        # the previous raw forward of the same position tensor (same storage, same version counter): probably the same
        # Gaussians -- its depth order is offered to the library, which verifies key by key
        hint = _order_hint.get(di) if _ORDER_HINT else None
        if hint is not None and not (hint["P"] == P and hint["key_bits"] == _lazy.key_bits and hint["stream"] == stream_id and
                                     hint["xyz"] == (xyz.data_ptr(), xyz._version) and hint["geom"].numel() == geom.numel()):
            hint = None
        # ... and when the host knows both z rows (camera_depth_key) the sort is either not launched at all (same row:
        # `hint_trusted`, still verified key by key -- a wrong guess drops the step) or launched without a hint
        zkey, trusted = cfg.get("zkey", False), False
        if hint is not None and zkey and hint["zkey"] and _lazy.trust_hints:
            if zkey == hint["zkey"]:
                trusted = True
            else:
                hint = None
        ctx.lazy_token = None
        if wait:
            p = _PendingFwd()
            p.ctx, p.view, p.geom, p.img, p.out, p.radii, p.cap, p.key = ctx, view, geom, img, out, radii, cap, key
            p.binning, p.fresh = binning, 1
            p.words, p.ring, p.row = _zero_words(dev, with_row=True)
            p.hint, p.trusted, p.zkey, p.key_bits = hint, trusted, zkey, _lazy.key_bits
            p.stream_id, p.stream, p.fkey, p.dev, p.P = stream_id, None, fkey, dev, P
            p.xyz_id, p.vis = (xyz.data_ptr(), xyz._version), vis
            p.handed = None
            cfg["pending"] = p
            lst = _pending_fwd.setdefault(di, [])
            lst.append(p)
            if len(lst) >= lazy_max:
                _flush_pending(di, full=True)
        else:
            first = True
            while True:
                if not first:
                    binning = _C.raw_binning(view, cap)
                first = False
                words = _zero_words(dev)                                   # [N, overflow word, key mismatch, spare], zero
                key_bits = _lazy.key_bits
                use_hint = hint is not None and P > 0
                if use_hint:
                    _stats["hinted"] += 1
                    _stats["trusted"] += int(trusted)
                _C.raw_forward_launch([(view, geom, binning, cap, img, out, radii, words, vis, key_bits, -1,
                                        hint["geom"] if use_hint else None, int(trusted) if use_hint else 0, 1, 0.0)], stream_id)
                if lazy and key in _lazy.capacity:
                    ctx.lazy_token = _lazy.track(key, cap, words[:2])
                    break
                n, flag = (int(v) for v in words[:2].tolist())            # exact render: one read-back
                _lazy.note(key, n)
                if flag & 2:
                    _lazy.key_bits = 0
                    hint = None
                if flag & 8:
                    _lazy.trust_hints, trusted = False, False
                ctx.lazy_token = None
                if n <= cap and not (flag & 10):
                    break
                cap = max(cap, _lazy.capacity[key])                       # repeat with what it needs
            if lazy:
                _mark_busy(di, stream_id)
                am = _S.adapt.get((di, stream_id))
                if am is not None and am.pop("want_check", False):
                    ck = am.get("ck_event")
                    if ck is None:
                        ck = am["ck_event"] = torch.cuda.Event()
                    ck.record(torch.cuda.ExternalStream(stream_id, device=dev))
                    am["check"] = ck
            if _ORDER_HINT:
                _order_hint[di] = dict(P=P, key_bits=key_bits, geom=geom, xyz=(xyz.data_ptr(), xyz._version),
                                       words=words, zkey=zkey, stream=stream_id)
        ctx.cfg, ctx.view = cfg, view
        ctx.save_for_backward(xyz, f_dc, f_rest, scaling, rotation, opacity, radii, geom, binning, img)
        ctx.cap = cap
        ctx.m2d_leaf = means2D
        if wait and ctx.lazy_token is None:
            # still pending: the outputs leave as _LazyOut (made here, in the forward's no-grad mode: as_subclass() on a tensor
            # that already carries a grad_fn would put an alias node between it and this one)
            color, radii, depth, alpha = (t.as_subclass(_LazyOut) for t in (color, radii, depth, alpha))
            cfg["pending"].handed = [weakref.ref(t) for t in (color, radii, depth, alpha)]
        ctx.mark_non_differentiable(radii)
        # an output nobody differentiates (depth / alpha of the shifted render, train.py:128-129) arrives as None in the
        # backward, not as an image of zeros the blend backward would have to read
        ctx.set_materialize_grads(False)
        ctx.batch_key = batch_key
        ctx.partner = None
        ctx.pending = []
        ctx.done_task = None
        if cfg["differentiated"]:
            prev = _last_raw_ctx.get(di)
            if prev is not None and prev[1] == ctx.batch_key and prev[0]() is not None:
                ctx.partner = prev[0]
            _last_raw_ctx[di] = (weakref.ref(ctx), ctx.batch_key)
        return color, radii, depth, alpha

    @staticmethod
    @_locked
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        if _pending_fwd:
            _flush_pending()          # (a render whose outputs nobody touched before backward(): its forward runs now)
        if ctx.lazy_token is not None and not ctx.lazy_token[4]:
            if _ENGINE_HOOKS:
                _register_task(token=ctx.lazy_token)    # truncated lists / key span: raised at the end of THIS backward()
            else:
                _lazy.confirm(ctx.lazy_token)           # (no final callback available: at the entry of the node)
        saved = ctx.saved_tensors
        params = saved[:6]
        dev = params[0].device
        cfg = ctx.cfg
        if grad_color is None:
            grad_color = torch.zeros((3, cfg["H"], cfg["W"]), dtype=torch.float32, device=dev)
        task = _current_task()
        job = _RawJob(ctx, _dev_f32(grad_color, "dL_dout_color"),
                      None if grad_depth is None else _dev_f32(grad_depth, "dL_dout_depth"),
                      None if grad_alpha is None else _dev_f32(grad_alpha, "dL_dout_alpha"), task)
        ctx.done_task = task
        # jobs parked here by later renders of THIS backward(); anything older was left by a backward() that raised before
        # this node ran (retain_graph=True and a second attempt): its gradients belong to nobody
        pending, ctx.pending = [j for j in ctx.pending if j.task == task], []
        needs = ctx.needs_input_grad
        if not _self_accumulate_ok(ctx, params):
            if pending:                      # (deferred by nodes for which self-accumulation was fine: launch them so)
                _launch_backward(pending, True)
            grads, g_m2d = _launch_backward([job], False)
            return (*[g if needs[i] else None for i, g in enumerate(grads)], g_m2d if needs[6] else None, None)
        jobs = pending + [job]
        partner = ctx.partner() if ctx.partner is not None else None
        if (partner is not None and partner.done_task != task and partner.batch_key == ctx.batch_key
                and not cfg["debug"] and _will_execute(partner)):
            # an earlier render of the same parameters runs its backward later in THIS graph task: it launches ours with its
            # own (one blend backward, one chain-rule pass for all of them)
            partner.pending = partner.pending + jobs
            _register_task(flush_ref=weakref.ref(partner))
            _stats["deferred"] += len(jobs)
            return (None,) * 8
        _launch_backward(jobs, True)
        return (None,) * 8


def rasterize_raw(pc, means2D, raster_settings, camera=None, lazy_outputs=False):
    """render()'s fast path: `pc` = a model raw_model_ok() accepts.  -> (color, radii, depth, alpha).  `camera` (optional):
    the object the matrices of `raster_settings` came from, for camera_depth_key().  lazy_outputs: the caller hands the
    outputs out as _LazyOut (render() does): -> (color, radii, depth, alpha, visibility) where the forward may still be
    pending."""
    rs = raster_settings
    # (the compiled module reads these four through their data pointers: a float32, contiguous device tensor of the right
    # size goes as it is -- the usual case, a camera's own matrices -- anything else is converted first)
    t = [x if (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()) else _dev_f32(x, n)
         for x, n in ((rs.bg, "bg"), (rs.viewmatrix, "viewmatrix"), (rs.projmatrix, "projmatrix"), (rs.campos, "campos"))]
    if t[0].numel() != 3 or t[1].numel() != 16 or t[2].numel() != 16 or t[3].numel() != 3:
        raise ValueError("bg/campos must have 3 and viewmatrix/projmatrix 16 elements")
    params = (pc._xyz, pc._features_dc, pc._features_rest, pc._scaling, pc._rotation, pc._opacity)
    cfg = dict(W=int(rs.image_width), H=int(rs.image_height), tanfovx=rs.tanfovx, tanfovy=rs.tanfovy, bg=t[0],
               scale_modifier=rs.scale_modifier, viewmatrix=t[1], projmatrix=t[2], campos=t[3], sh_degree=rs.sh_degree,
               debug=rs.debug, zkey=camera_depth_key(camera) if (camera is not None and _ORDER_HINT) else False,
               differentiated=torch.is_grad_enabled() and any(p.requires_grad for p in params + (means2D,)))
    if not lazy_outputs:
        return _RasterizeRaw.apply(*params, means2D, cfg)
    cfg["lazy_outputs"] = True
    color, radii, depth, alpha = _RasterizeRaw.apply(*params, means2D, cfg)
    p, vis = cfg.pop("pending", None), cfg.pop("vis")
    if p is None or p.ctx.lazy_token is not None:            # launched already (not eligible, or it completed a batch)
        return color, radii, depth, alpha, vis
    vis = vis.as_subclass(_LazyOut)
    p.handed.append(weakref.ref(vis))
    return color, radii, depth, alpha, vis


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
