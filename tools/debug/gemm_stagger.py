"""Developer: would it pay to de-synchronise the two workgroups that share a CU in the 128 x 128 tile kernel?  Both reach their epilogue
(fp32 residual read + fp32 / fp16 writes, ~16 us per round) at the same time, so neither hides the other's.  Emulation without touching
the kernel: the launch split into two half-M launches on two streams, the second delayed by a fraction of a kernel -- every CU then holds
one workgroup of each, out of phase."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crowdsam_amd import hip
dev = "cuda"
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def run(M, N, K, delay_cycles, iters=20):
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
    bias = torch.randn(N, device=dev); res = torch.randn(M, N, device=dev); out = torch.empty(M, N, device=dev)
    x16 = torch.empty(M, N, device=dev, dtype=torch.float16); st = torch.empty(M, N // 128, 2, device=dev)
    h = M // 2

    def whole():
        hip.gemm_f16_ln(a, w, out, bias=bias, residual=res, out16=x16, stats_out=st)

    def split():
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            hip.gemm_f16_ln(a[:h], w, out[:h], bias=bias, residual=res[:h], out16=x16[:h], stats_out=st[:h])
        with torch.cuda.stream(s2):
            if delay_cycles:
                torch.cuda._sleep(delay_cycles)
            hip.gemm_f16_ln(a[h:], w, out[h:], bias=bias, residual=res[h:], out16=x16[h:], stats_out=st[h:])
        cur.wait_stream(s1); cur.wait_stream(s2)

    def tm(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / iters * 1e3
    return tm(whole), tm(split)


for (M, N, K) in ((16384, 1024, 1024), (16384, 1024, 4096), (32768, 1024, 4096)):
    for d in (0, 20000, 50000, 100000):
        t_w, t_s = run(M, N, K, d)
        print("M %5d N %4d K %4d: one launch %.1f us | two half launches on two streams, second delayed by %6d cycles: %.1f us" % (M, N, K, t_w, d, t_s), flush=True)
