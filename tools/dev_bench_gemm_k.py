"""Developer: time of the 256 x 256 ping-pong GEMM against K at the image-batched shapes -- fixed cost per tile round vs k-loop."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crowdsam_amd import hip


def bench(M, N, K, iters=20):
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") * 0.05).half()
    bias = torch.randn(N, device="cuda")
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    for _ in range(3):
        hip.gemm_f16(a, w, out=out, bias=bias)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        hip.gemm_f16(a, w, out=out, bias=bias)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for (M, N) in ((4096, 4096), (16384, 3072), (16384, 4096)):
    tiles = (M // 256) * (N // 256)
    rounds = -(-tiles // 256)
    row = []
    for K in (128, 256, 512, 1024, 2048, 4096):
        us = bench(M, N, K)
        row.append("K=%d %.1f us (%.0f TF, %.1f us/round)" % (K, us, 2 * M * N * K / us / 1e6, us / rounds))
    print("M=%d N=%d: %d tiles = %d rounds | " % (M, N, tiles, rounds) + " | ".join(row), flush=True)
