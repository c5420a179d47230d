"""usage (GPU box): python tools/adam_time.py -- the one-launch Adam at several sizes / options (HIP events, mean of 50)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from binocular3dgs_amd import synth
from binocular3dgs_amd.step import FusedAdam, ShardedAdam, FlatGradSlab

LRS = [1.6e-4, 2.5e-3, 1.25e-4, 5e-3, 1e-3, 0.05]


def run(P, kind, decay, mask_frac):
    model = synth.synth_model(P, seed=0, device="cuda")
    kw = dict(eps=1e-15, opacity_decay=0.995 if decay else 0.0, opacity_index=5, decay_first=True)
    opt = (ShardedAdam if kind == "sharded" else FusedAdam)(model.parameters(), LRS, **kw)
    slab = FlatGradSlab(model.parameters(), getattr(opt, "padded_numel", 0))
    slab.flat.normal_()
    mask = None
    if mask_frac is not None:
        words = (P + 63) // 64
        bits = (torch.rand(words, 64, device="cuda") < mask_frac).to(torch.int64)
        mask = (bits << torch.arange(64, device="cuda")).sum(1)
    def step():
        if kind == "sharded":
            opt.step(slab, row_mask=mask)
        else:
            opt.step(row_mask=mask) if mask is not None else opt.step()
    for _ in range(5):
        step()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(50):
        step()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    nbytes = 23 * P * 28
    print(f"P={P:8d} {kind:8s} decay={int(decay)} mask={mask_frac}  {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s (28 B per float)")


for P in (100_000, 500_000, 1_000_000):
    for kind in ("fused", "sharded"):
        for decay in (False, True):
            for mf in (None, 0.2):
                run(P, kind, decay, mf)
