"""Developer micro-benchmark of csam_gemm_f16 (TFLOP/s at encoder / DINOv2 shapes, with their epilogues)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crowdsam_amd import hip

def bench(M, N, K, mode="plain", iters=20):
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") * 0.05).half()
    bias = torch.randn(N, device="cuda")
    kw = {}
    if mode == "gelu":
        out = torch.empty(M, N, device="cuda", dtype=torch.float16); kw = dict(bias=bias, act=hip.ACT_GELU)
    elif mode == "res32":
        out = torch.randn(M, N, device="cuda"); kw = dict(bias=bias, residual=out)
    elif mode == "bias":
        out = torch.empty(M, N, device="cuda", dtype=torch.float16); kw = dict(bias=bias)
    else:
        out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    for _ in range(3):
        hip.gemm_f16(a, w, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        hip.gemm_f16(a, w, out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"M={M} N={N} K={K} {mode:6s}: {ms*1e3:7.1f} us  {2*M*N*K/ms/1e9:7.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    for s in [(4096, 4096, 4096, "plain"), (4096, 3072, 1024, "bias"), (4096, 1024, 1024, "res32"), (4096, 4096, 1024, "gelu"),
              (4096, 1024, 4096, "res32"), (5330, 3072, 1024, "bias"), (5330, 1024, 1024, "res32"), (5330, 4096, 1024, "gelu"),
              (5330, 1024, 4096, "res32"), (1792, 256, 256, "bias"), (1792, 2048, 256, "bias"), (1792, 256, 2048, "res32"),
              (1024, 256, 5376, "plain")]:
        bench(*s)
