"""Developer: bitwise repeatability of the encoder's GEMM shapes over many launches (an LDS-DMA wait hazard shows as rare mismatches)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crowdsam_amd import hip
torch.manual_seed(0)
dev = "cuda"
bad = 0
for (M, N, K, res) in [(4096, 1024, 4096, True), (4096, 1024, 1024, True), (5330, 1024, 4096, True), (5330, 1024, 1024, True),
                       (4096, 3072, 1024, False), (224, 256, 2048, True), (28672, 256, 2048, True)]:
    a = torch.randn(M, K, device=dev).half(); w = (torch.randn(N, K, device=dev) * 0.05).half()
    r = torch.randn(M, N, device=dev) if res else None
    out = torch.empty(M, N, device=dev, dtype=torch.float32 if res else torch.float16)
    hip.gemm_f16(a, w, out=out, residual=r)
    ref = out.clone()
    refc = torch.matmul(a.float(), w.float().t()) + (r if res else 0)
    err = (ref.float() - refc).abs().max().item() / refc.abs().mean().item()
    n_bad = 0
    for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 300):
        out.zero_()
        hip.gemm_f16(a, w, out=out, residual=r)
        if not torch.equal(out, ref): n_bad += 1
    bad += n_bad
    print("M %5d N %4d K %4d: max err / mean |ref| %.2e, mismatching launches %d" % (M, N, K, err, n_bad))
print("TOTAL mismatches", bad)
