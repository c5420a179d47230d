import sys, time, torch
sys.path.insert(0, ".")
import bench_ref_schedule as b
dev = torch.device("cuda:0")
def probe(step):
    out = []
    for r in range(14):
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(40):
            step()
        torch.cuda.synchronize(dev)
        out.append(round(40 / (time.perf_counter() - t0), 1))
    return out
for surf in ("unchanged", "render"):
    print(surf, b.run(dev, 500_000, 504, 378, 60.0, surf, probe=probe))
