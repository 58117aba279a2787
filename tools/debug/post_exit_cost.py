"""Developer: what the early-exit workgroups of the two mask_post_x4 passes cost -- 4096 prompts, k of them kept / above the score cut."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crowdsam_amd import hip
dev = "cuda"
B = 4096
torch.manual_seed(0)
low = torch.randn(B, 4, 256, 256, device=dev)
sel = torch.zeros(B, dtype=torch.int32, device=dev)
store = torch.zeros(1024, 1024, 1024, dtype=torch.uint8, device=dev)
inter = torch.zeros(B, dtype=torch.int32, device=dev); uni = torch.zeros_like(inter); box = torch.zeros(B, 4, dtype=torch.int32, device=dev)


def tm(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for k in (0, 1, 64, 360, 720):
    keep = torch.zeros(B, dtype=torch.uint8, device=dev)
    idx = torch.randperm(B, device=dev)[:k]
    keep[idx] = 1
    slot = torch.zeros(B, dtype=torch.int32, device=dev)
    slot[idx] = torch.arange(k, dtype=torch.int32, device=dev)
    score = keep.float()
    t1 = tm(lambda: hip.mask_write(low, sel, keep, B, (1024, 1024), (1024, 1024), 0.0, store, slot=slot))
    t0 = tm(lambda: hip.mask_post_scored(low, sel, score, 0.5, B, (1024, 1024), (1024, 1024), 0.0, 1.0, inter, uni, box))
    print("kept %4d of %d: statistics pass %.1f us, byte pass %.1f us" % (k, B, t0, t1), flush=True)
