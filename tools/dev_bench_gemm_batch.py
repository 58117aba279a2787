"""Developer micro-benchmark: the encoder GEMM shapes at M = B x tokens (B images per launch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crowdsam_amd import hip


def bench(M, N, K, mode, iters=20):
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") * 0.05).half()
    bias = torch.randn(N, device="cuda")
    if mode == "gelu":
        out = torch.empty(M, N, device="cuda", dtype=torch.float16); kw = dict(bias=bias, act=hip.ACT_GELU)
    elif mode == "res32":
        out = torch.randn(M, N, device="cuda"); kw = dict(bias=bias, residual=out)
    else:
        out = torch.empty(M, N, device="cuda", dtype=torch.float16); kw = dict(bias=bias)
    for _ in range(3):
        hip.gemm_f16(a, w, out=out, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        hip.gemm_f16(a, w, out=out, **kw)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms * 1e3, 2 * M * N * K / ms / 1e9


if __name__ == "__main__":
    for T in (4096, 5330):
        for (N, K, mode) in ((3072, 1024, "bias"), (4096, 1024, "gelu"), (1024, 1024, "res32"), (1024, 4096, "res32")):
            row = []
            for B in (1, 2, 3, 4, 6, 8):
                us, tf = bench(T * B, N, K, mode)
                row.append("B=%d %6.1f us/img %5.0f TF" % (B, us / B, tf))
            print("T=%d N=%d K=%d %-5s | " % (T, N, K, mode) + " | ".join(row), flush=True)
