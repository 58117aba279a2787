"""Developer check: bitwise repeatability of upscale / GEMM / flash / pool on fixed inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crowdsam_amd import hip
torch.manual_seed(0)
dev = "cuda"
def rep(name, fn, n=4):
    outs = []
    for r in range(n):
        outs.append(fn().clone()); torch.cuda.synchronize()
    for r in range(1, n):
        a, b = outs[0].contiguous().view(-1), outs[r].contiguous().view(-1)
        it = torch.int16 if a.element_size() == 2 else torch.int32
        d = int((a.view(it) != b.view(it)).sum())
        print(name, "run", r, "differing", d, "of", a.numel(), flush=True)
# GEMMs
for (M, N, K) in [(4096, 3072, 1024), (4096, 1024, 4096), (5330, 4096, 1024)]:
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    rep(f"gemm {M}x{N}x{K}", lambda: hip.gemm_f16(a, w, out=out))
# flash
T, nH = 5330, 16
qkv = torch.randn(T, 3 * nH * 64, device=dev).half(); fo = torch.empty(T, nH * 64, device=dev, dtype=torch.float16)
rep("flash", lambda: hip.flash_attn(qkv, fo, T, nH, 0.125, nH * 64))
# upscale
B = 256
keys = (torch.randn(B * 4096, 256, device=dev) * 0.5).half()
W1 = (torch.randn(256, 256, device=dev) * 0.05).half(); b1 = torch.randn(256, device=dev)
g = torch.ones(64, device=dev); be = torch.zeros(64, device=dev)
W2 = (torch.randn(128, 64, device=dev) * 0.1).half(); b2 = torch.randn(128, device=dev)
hy = torch.randn(B, 4, 32, device=dev); masks = torch.empty(B, 4, 256, 256, device=dev); stats = torch.empty(B * 4, 2, device=dev)
rep("upscale", lambda: (hip.upscale_fused(keys, W1, b1, g, be, 1e-6, W2, b2, hy, masks, B, stats=stats), masks)[1])
