"""developer: csam_flash_attn at DINOv2's shape (T = 5330, 16 heads), one and four images per launch."""
import sys, torch
sys.path.insert(0, ".")
from crowdsam_amd import hip
cuda = torch.device("cuda:0")
T, nH, D = 5330, 16, 1024
for B in (1, 4):
    qkv = (torch.randn(B * T, 3 * D, device=cuda) * 0.5).half()
    out = torch.empty(B * T, D, dtype=torch.float16, device=cuda)
    fn = lambda: hip.flash_attn(qkv, out, T, nH, 0.125, D, q_prescaled=True, n_images=B)
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("B=%d: %.1f us per launch, %.1f us per image, %.0f TFLOP/s" % (B, us, us / B, 4.0 * T * T * 64 * nH * B / us / 1e6))
