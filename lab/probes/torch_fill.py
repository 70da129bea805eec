import torch, numpy as np
n = 1072916736 // 8
bufs = [torch.empty(n, dtype=torch.float64, device="cuda") for _ in range(4)]
src = torch.randn(n, dtype=torch.float64, device="cuda")
def timeit(f, reps=10):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for i, b in enumerate(bufs):
    print(i, "zero_ %.1f  fill_(1.5) %.1f  copy_ %.1f  mul_(2) %.1f us" % (timeit(lambda: b.zero_()), timeit(lambda: b.fill_(1.5)), timeit(lambda: b.copy_(src)), timeit(lambda: b.mul_(2.0))))
