"""Developer: what does the epilogue of the 256x256 ping-pong GEMM cost?  Same shape with / without bias / GELU."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crowdsam_amd import hip

def tm(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for M, N, K in ((4096, 3072, 1024), (4096, 4096, 1024), (5330, 3072, 1024)):
    for wscale in (1.0, 0.05):
        a = torch.randn(M, K, device="cuda").half()
        w = (torch.randn(N, K, device="cuda") * wscale).half()
        bias = torch.randn(N, device="cuda")
        out = torch.empty(M, N, device="cuda", dtype=torch.float16)
        t0 = tm(lambda: hip.gemm_f16(a, w, out=out))
        t1 = tm(lambda: hip.gemm_f16(a, w, out=out, bias=bias))
        t2 = tm(lambda: hip.gemm_f16(a, w, out=out, bias=bias, act=hip.ACT_GELU))
        t3 = tm(lambda: hip.gemm_f16(a, w, out=out))
        print(f"M={M} N={N} K={K} w*{wscale}: plain {t0:.1f} us | bias {t1:.1f} | bias+gelu {t2:.1f} | plain again {t3:.1f}", flush=True)
