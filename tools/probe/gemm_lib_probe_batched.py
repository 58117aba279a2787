"""Developer probe: the vendor GEMM (hipBLASLt through torch.mm, fp16, no epilogue) beside csam_gemm_f16 (bias epilogue) on the IMAGE-BATCHED
shapes of the encoders (four images per pass) -- how much head-room a hand-scheduled 256 x 256 kernel would have."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crowdsam_amd import hip
dev = torch.device("cuda")


def tm(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for M, N, K, f32out in [(16384, 3072, 1024, 0), (16384, 4096, 1024, 0), (16384, 1024, 1024, 1), (16384, 1024, 4096, 1),
                        (21320, 3072, 1024, 0), (21320, 4096, 1024, 0), (21320, 1024, 1024, 1), (21320, 1024, 4096, 1)]:
    a = torch.randn(M, K, device=dev).half()
    w = (torch.randn(N, K, device=dev) * 0.05).half()
    bias = torch.randn(N, device=dev)
    out16 = torch.empty(M, N, device=dev, dtype=torch.float16)
    res = torch.randn(M, N, device=dev)
    out32 = torch.empty(M, N, device=dev)
    t_lib = tm(lambda: torch.mm(a, w.t(), out=out16))
    if f32out:
        t_own = tm(lambda: hip.gemm_f16(a, w, out=out32, bias=bias, residual=res))
    else:
        t_own = tm(lambda: hip.gemm_f16(a, w, out=out16, bias=bias))
    fl = 2.0 * M * N * K / 1e6
    print(f"M={M} N={N} K={K}: vendor (fp16 out, no epilogue) {t_lib:7.1f} us {fl / t_lib:7.1f} TF/s | own "
          f"({'fp32 + residual' if f32out else 'fp16 + bias'}) {t_own:7.1f} us {fl / t_own:7.1f} TF/s", flush=True)
