"""Calibrate what this MI355X box reaches: copy (read+write), read-only reduce, write-only fill."""
import time, torch
dev = torch.device("cuda:0")
n = 1 << 30  # 1 GiB of uint8 = 256M float32... use float32 tensors of 1 GiB
a = torch.empty(n // 4, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
def t(fn, reps=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
tc = t(lambda: b.copy_(a))
tf = t(lambda: b.fill_(1.0))
tr = t(lambda: a.sum())
print(f"copy  {2*n/tc/1e12:.2f} TB/s (read+write)   fill {n/tf/1e12:.2f} TB/s   sum(read) {n/tr/1e12:.2f} TB/s")
