"""Two host-side shortcuts every launch of this package goes through (measured on the reference's own iteration shape,
tools/unchanged_host_profile.py: `torch.cuda.current_stream(dev).cuda_stream` is ~5 us a call -- a Stream object is built
and thrown away -- and was asked 18 times per iteration; `with torch.cuda.device(dev)` ~3 us, 15 times per iteration):

    raw_stream(dev)     the current stream's handle as an integer
    device_guard(dev)   a context manager that switches the device only when it is not already current
"""
from __future__ import annotations

import contextlib

import torch

_get_raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)     # (what torch's own compiled kernels call; probed)
_NO_GUARD = contextlib.nullcontext()


def raw_stream(dev) -> int:
    idx = dev.index
    if _get_raw is not None:
        return _get_raw(idx if idx is not None else torch.cuda.current_device())
    return torch.cuda.current_stream(dev).cuda_stream


def device_guard(dev):
    idx = dev.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_GUARD
    return torch.cuda.device(dev)
