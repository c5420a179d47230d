"""usage (GPU box): python tools/stream_rates.py -- what pure-write, pure-read and copy streams reach on this part (torch kernels over
2 GiB): the ceilings the streaming kernels of the iteration are held against (the projection is 81 % writes)."""
import torch
dev = torch.device("cuda", 0)
n = 1 << 29                      # 2 GiB of float32
x = torch.empty(n, dtype=torch.float32, device=dev)
y = torch.empty(n, dtype=torch.float32, device=dev)


def timed(fn, bytes_moved, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    return round(bytes_moved / ms / 1e6, 1)      # GB/s


print("pure write (fill_)      GB/s:", timed(lambda: x.fill_(1.0), 4 * n))
print("pure read  (sum)        GB/s:", timed(lambda: x.sum(), 4 * n))
print("copy (read + write)     GB/s:", timed(lambda: y.copy_(x), 8 * n))
print("read-modify-write (add_) GB/s:", timed(lambda: x.add_(1.0), 8 * n))
z = torch.empty(n // 4, dtype=torch.float32, device=dev)
print("4 reads + 1 write (a+b+c+d) GB/s:", timed(lambda: torch.add(x[:n // 4], x[n // 4:n // 2], out=z), 12 * (n // 4)))
