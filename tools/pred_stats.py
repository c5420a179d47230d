"""usage (GPU box): python tools/pred_stats.py -- the headline workload with two binning rounds, prediction settled: per
view the number of tiles predicted open and the share of the Gaussians behind segment 1 whose rect reaches one of them
(the ones the scan must gather and the emission must visit)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
sys.argv = sys.argv[:1]
args = bench.parse()
dev = torch.device("cuda", 0)
job = bench.Job(args, dev, 0, 1, False, args.gaussians, args.width, args.height, args.fov, 6, "weak")
for _ in range(6):
    job.eager_step()
torch.cuda.synchronize()
fr = job.fused
W, H, P = args.width, args.height, args.gaussians
gx, gy = (W + 15) // 16, (H + 15) // 16
rw = (gx + 63) // 64
L = __import__("binocular3dgs_amd._lib", fromlist=["lib"]).lib()
img_bytes = L.b3gs_image_bytes(W, H)
al = lambda x: (x + 255) & ~255
A = al((gy + 1) * rw * 8)
F = al(((P + 63) // 64) * 8)
T = (P + 4095) // 4096
TAIL = al(T * 4096 * 4) + 2 * al(T * 4)           # flist | fcount | tsum follow pflag (b3gs_internal.h geometry carve)
K1 = -(-int(fr.seg1_fraction * P + 0.999999) // 4096) * 4096
for k, s in enumerate(fr.slots[:6]):
    # the last three arrays of the image buffer: open_rows | pred_rows | pred_next; pflag | flist | fcount | tsum end the geometry buffer
    pred = s.img[-2 * A:-A].view(torch.int64)[: gy * rw]
    bits = ((pred.view(-1, 1) >> torch.arange(64, device=dev)) & 1).sum().item()
    pf = s.geom[-F - TAIL:-TAIL].view(torch.int64)[: (P + 63) // 64]
    flagged = ((pf.view(-1, 1) >> torch.arange(64, device=dev)) & 1).reshape(-1)[:P].bool()
    fc = s.geom[-2 * al(T * 4):-al(T * 4)].view(torch.int32)[:T]
    hdr = s.img[:64].view(torch.int32)
    vis = (s.radii > 0)
    print("view", k, "N1", int(hdr[0]), "N2", int(hdr[2]), "tiles predicted open", int(bits), "of", gx * gy,
          "| Gaussians flagged %.1f %% of all, %.1f %% of the visible" % (100.0 * flagged.float().mean().item(),
          100.0 * (flagged & vis).float().sum().item() / max(vis.float().sum().item(), 1)),
          "| compacted list entries behind K1:", int(fc[K1 // 4096:].sum()))
print("K1 =", K1, "seg1_fraction", fr.seg1_fraction)
