"""Does a re-read of a buffer that fits the 256 MiB Infinity Cache run faster than HBM streaming?"""
import time, torch
dev = torch.device("cuda:0")
def bw(nbytes, reps=50):
    a = torch.empty(nbytes // 4, dtype=torch.float32, device=dev).normal_()
    b = torch.empty_like(a)
    for fn, name, mult in ((lambda: a.sum(), "read", 1), (lambda: b.copy_(a), "copy", 2), (lambda: b.fill_(0.5), "fill", 1)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print(f"{nbytes/2**20:7.0f} MiB {name}: {mult*nbytes/dt/1e12:.2f} TB/s", end="   ")
    print()
for mb in (16, 32, 64, 96, 128, 192, 256, 512, 2048):
    bw(mb * 2**20)
