"""Developer probe: the vendor's fused attention (torch.nn.functional.scaled_dot_product_attention on ROCm: the
flash / memory-efficient backends) beside csam_flash_attn on the two global-attention shapes of the frame, same box, back to back:
DINOv2 ViT-L/14 [16 heads, 5330 tokens, 64] and the SAM global blocks [16, 4096, 64] (without the decomposed rel-pos bias, which the
vendor kernel cannot take), at one and four images per call.  Prints time, TFLOP/s (4 T^2 d per head) and the max difference."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from crowdsam_amd import hip

dev = torch.device("cuda")


def tm(fn, n=20):
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for T, nH, nimg in [(5330, 16, 1), (5330, 16, 4), (4096, 16, 1), (4096, 16, 4)]:
    D = nH * 64
    g = torch.Generator(device="cpu").manual_seed(T + nimg)
    qkv = torch.randn(nimg * T, 3 * D, generator=g).to(dev).half()
    out = torch.empty(nimg * T, D, device=dev, dtype=torch.float16)
    t_own = tm(lambda: hip.flash_attn(qkv, out, T, nH, 0.125, D, n_images=nimg))
    q, k, v = (qkv.view(nimg, T, 3, nH, 64)[:, :, i].permute(0, 2, 1, 3).contiguous() for i in range(3))   # [n, H, T, 64]
    rows = []
    for name, be in (("flash", "FLASH_ATTENTION"), ("mem_efficient", "EFFICIENT_ATTENTION"), ("math", "MATH")):
        try:
            from torch.nn.attention import SDPBackend, sdpa_kernel
            with sdpa_kernel(getattr(SDPBackend, be)):
                o = F.scaled_dot_product_attention(q, k, v, scale=0.125)
                t = tm(lambda: F.scaled_dot_product_attention(q, k, v, scale=0.125), n=20 if be != "MATH" else 3)
            ref = o.permute(0, 2, 1, 3).reshape(nimg * T, D)
            rows.append((name, t, (ref.float() - out.float()).abs().max().item()))
        except Exception as e:     # backend not built into this torch / not eligible for the shape
            rows.append((name, None, str(e).splitlines()[0][:80]))
    fl = 4.0 * T * T * 64 * nH * nimg / 1e6
    line = f"T={T} heads={nH} images={nimg}: own csam_flash_attn {t_own:8.1f} us {fl / t_own:7.1f} TF/s"
    for name, t, d in rows:
        line += f" | sdpa {name}: " + (f"{t:8.1f} us {fl / t:7.1f} TF/s (own/vendor time {t_own / t:.2f}, max diff {d:.1e})" if t else f"unavailable ({d})")
    print(line, flush=True)
