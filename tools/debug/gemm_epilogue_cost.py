"""Developer: what each piece of the 128 x 128 tile kernel's epilogue costs at the image-batched N = 1024 shapes (proj / fc2)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crowdsam_amd import hip
dev = "cuda"


def tm(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (M, N, K) in ((16384, 1024, 1024), (16384, 1024, 4096), (21320, 1024, 1024), (21320, 1024, 4096)):
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
    bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev); o32 = torch.empty(M, N, device=dev)
    o16 = torch.empty(M, N, device=dev, dtype=torch.float16); st = torch.empty(M, N // 128, 2, device=dev)
    ls = torch.rand(N, device=dev)
    rows = [("fp16 out, bias", lambda: hip.gemm_f16(a, w, out=o16, bias=bias)),
            ("fp32 out, bias", lambda: hip.gemm_f16(a, w, out=o32, bias=bias)),
            ("fp32 out, bias, fp32 residual", lambda: hip.gemm_f16(a, w, out=o32, bias=bias, residual=res)),
            ("  ... in place (out = residual)", lambda: hip.gemm_f16(a, w, out=res, bias=bias, residual=res)),
            ("  ... + fp16 copy + row statistics (SAM proj / fc2)", lambda: hip.gemm_f16_ln(a, w, res, bias=bias, residual=res, out16=o16, stats_out=st)),
            ("  ... + LayerScale (DINOv2 proj / fc2)", lambda: hip.gemm_f16_ln(a, w, res, bias=bias, residual=res, out16=o16, stats_out=st, colscale=ls))]
    print("M %d N %d K %d" % (M, N, K))
    for name, fn in rows:
        try:
            print("   %-58s %7.1f us" % (name, tm(fn)), flush=True)
        except Exception as e:
            print("   %-58s failed: %s" % (name, str(e)[:80]))
