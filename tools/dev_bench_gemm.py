"""Developer micro-benchmark of csam_gemm_f16 (TFLOP/s at encoder shapes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crowdsam_amd import hip

def bench(M, N, K, iters=20):
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") * 0.05).half()
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    for _ in range(3):
        hip.gemm_f16(a, w, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        hip.gemm_f16(a, w, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"M={M} N={N} K={K}: {ms*1e3:.1f} us  {2*M*N*K/ms/1e9:.1f} TFLOP/s", flush=True)

if __name__ == "__main__":
    for s in [(4096, 4096, 4096), (8192, 8192, 8192), (4096, 3072, 1024), (4096, 1024, 1024),
              (4096, 4096, 1024), (4096, 1024, 4096), (5330, 4096, 1024), (131072, 128, 256)]:
        bench(*s)
