"""Developer probe: what the vendor GEMM (hipBLASLt through torch.mm / addmm, fp16) reaches on the path's shapes,
beside csam_gemm_f16 -- sets the target for the hand-written tiles."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crowdsam_amd import hip

shapes = [(4096, 4096, 4096), (4096, 3072, 1024), (4900, 3072, 1024), (4096, 1024, 1024), (4096, 4096, 1024), (4096, 1024, 4096),
          (5330, 3072, 1024), (5330, 1024, 1024), (5330, 4096, 1024), (5330, 1024, 4096), (14336, 256, 2048), (14336, 2048, 256)]
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


for M, N, K in shapes:
    a = torch.randn(M, K, device=dev).half()
    w = torch.randn(N, K, device=dev).half()          # nn.Linear layout: y = a @ w.T
    wt = w.t().contiguous()
    bias = torch.randn(N, device=dev).half()
    out = torch.empty(M, N, device=dev, dtype=torch.float16)
    t_nt = tm(lambda: torch.mm(a, w.t(), out=out))
    t_nn = tm(lambda: torch.mm(a, wt, out=out))
    t_b = tm(lambda: torch.addmm(bias, a, w.t(), out=out))
    t_own = tm(lambda: hip.gemm_f16(a, w, out=out))
    fl = 2.0 * M * N * K / 1e6
    print(f"M={M} N={N} K={K}: lib NT {t_nt:7.1f} us {fl / t_nt:7.1f} TF/s | lib NN {t_nn:7.1f} us {fl / t_nn:7.1f} | "
          f"lib NT+bias {t_b:7.1f} us | own {t_own:7.1f} us {fl / t_own:7.1f} TF/s", flush=True)
