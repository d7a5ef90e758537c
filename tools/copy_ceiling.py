"""Measured HBM streaming ceiling of the box ('report a measured copy-kernel ceiling'): device-to-device copy
and a read-only reduction over 2 GiB, through torch (plumbing only)."""
import time, torch
n = 1 << 28  # 2 GiB of float64
a = torch.empty(n, dtype=torch.float64, device="cuda").normal_()
b = torch.empty_like(a)
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
t = timed(lambda: b.copy_(a))
print("copy   %.2f TB/s (read + write of %.1f GiB)" % (2 * a.numel() * 8 / t / 1e12, a.numel() * 8 / 2**30))
t = timed(lambda: a.sum())
print("reduce %.2f TB/s (read only)" % (a.numel() * 8 / t / 1e12))
