"""A few launches of csam_gemm_f16 for PMC collection (rocprofv3 --pmc ... -- python tools/dev_pmc_gemm.py M N K)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crowdsam_amd import hip
M, N, K = (int(v) for v in sys.argv[1:4])
a = torch.randn(M, K, device="cuda").half()
w = (torch.randn(N, K, device="cuda") * 0.05).half()
out = torch.empty(M, N, device="cuda", dtype=torch.float16)
for _ in range(4):
    hip.gemm_f16(a, w, out=out)
torch.cuda.synchronize()
