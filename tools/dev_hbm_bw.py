"""HBM bandwidth sanity numbers (torch fill / copy / read-reduce) to put the streaming kernels' rates in context."""
import torch
x = torch.empty(2 * 1024 ** 3, dtype=torch.float32, device="cuda")      # 8 GB
y = torch.empty_like(x)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
gb = x.numel() * 4 / 1e9
print("fill   (write only) %.2f TB/s" % (gb / t(lambda: x.fill_(1.0)) / 1e3))
print("copy   (read+write) %.2f TB/s" % (2 * gb / t(lambda: y.copy_(x)) / 1e3))
print("sum    (read only)  %.2f TB/s" % (gb / t(lambda: x.sum()) / 1e3))
