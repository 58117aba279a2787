"""developer: DINOv2 fc1 / qkv at M = 5376 as one launch vs split into 4096 + 1280 rows."""
import sys, torch
sys.path.insert(0, ".")
from crowdsam_amd import hip
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (M, N, K, act) in [(5376, 4096, 1024, hip.ACT_GELU), (5376, 3072, 1024, 0), (4900, 4096, 1024, hip.ACT_GELU), (4900, 3072, 1024, 0)]:
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * 0.05).half()
    bias = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    one = t(lambda: hip.gemm_f16(a, w, out=out, bias=bias, act=act))
    res = [f"M={M} N={N}: one launch {one:.1f} us"]
    for m0 in (4096, 3840, 3584):
        if m0 >= M: continue
        two = t(lambda: (hip.gemm_f16(a[:m0], w, out=out[:m0], bias=bias, act=act), hip.gemm_f16(a[m0:], w, out=out[m0:], bias=bias, act=act)))
        res.append(f"{m0}+{M-m0}: {two:.1f}")
    print("  ".join(res), flush=True)
