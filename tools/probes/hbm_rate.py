"""Streaming rates of the box's HBM as torch sees them: fill (write only), copy (read + write), sum (read only)."""
import torch, time
n = 1 << 30  # 4 GiB of float32
a = torch.empty(n, dtype=torch.float32, device="cuda")
b = torch.empty(n, dtype=torch.float32, device="cuda")
def t(f, reps=5):
    f(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
dt = t(lambda: a.fill_(1.0)); print("fill  %.2f TB/s written" % (4 * n / dt / 1e12))
dt = t(lambda: b.copy_(a)); print("copy  %.2f TB/s read + %.2f TB/s written" % (4 * n / dt / 1e12, 4 * n / dt / 1e12))
dt = t(lambda: a.sum()); print("sum   %.2f TB/s read" % (4 * n / dt / 1e12))
