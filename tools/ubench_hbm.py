"""Sustained HBM rate yardstick: device-to-device copy and read-only reduction of 2 GiB (torch ops), wall clock."""
import torch
n = 1 << 29  # 2 GiB of fp32
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
tc = t(lambda: b.copy_(a))
tr = t(lambda: a.sum())
tw = t(lambda: b.fill_(1.0))
print("copy  2 GiB -> 2 GiB: %.2f ms = %.2f TB/s (read + write)" % (tc * 1e3, 2 * 4 * n / tc / 1e12))
print("read  2 GiB (sum)   : %.2f ms = %.2f TB/s" % (tr * 1e3, 4 * n / tr / 1e12))
print("write 2 GiB (fill)  : %.2f ms = %.2f TB/s" % (tw * 1e3, 4 * n / tw / 1e12))
