"""Time the two mask post-processing passes (statistics, mask bytes) on B random low-res logit sets."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crowdsam_amd import hip
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dev = "cuda"
torch.manual_seed(0)
low = torch.nn.functional.interpolate(torch.randn(B, 4, 32, 32, device=dev), (256, 256), mode="bilinear") * 4
sel = torch.randint(0, 4, (B,), dtype=torch.int32, device=dev)
score = torch.rand(B, device=dev)
i32 = lambda *s: torch.empty(*s, dtype=torch.int32, device=dev)
inter, uni, box = i32(B), i32(B), i32(B, 4)
keep = (torch.rand(B, device=dev) < 0.45).to(torch.uint8)
slot = (torch.cumsum(keep.int(), 0) - 1).to(torch.int32)
masks = torch.empty(int(keep.sum().item()) + 1, 1024, 1024, dtype=torch.uint8, device=dev)
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
a = t(lambda: hip.mask_post_scored(low, sel, score, 0.0, B, (1024, 1024), (1024, 1024), 0.0, 1.0, inter, uni, box, None))
b = t(lambda: hip.mask_write(low, sel, keep, B, (1024, 1024), (1024, 1024), 0.0, masks, None, slot=slot))
print(f"{os.environ.get('CSAM_LIB', 'default')}: stats pass {a:.0f} us, mask bytes pass {b:.0f} us (B={B}, {int(keep.sum())} kept)  chk {int(inter.sum())} {int(uni.sum())} {int(box.sum())} {int(masks[:-1].sum())}")
