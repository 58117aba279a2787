"""Developer (build with CSAM_DEFS_gemm_f16=-DCSAM_GEMM_TS): s_memtime split of the k-loop of gemm_f16_kernel for one wave of one
workgroup: wait for the stage / barrier / issue of the next stage / fragment reads + MFMAs."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crowdsam_amd import hip
L = hip.lib()
ts = torch.zeros(8, dtype=torch.int64, device="cuda")
L.csam_dbg_set_gemm_ts.argtypes = [ctypes.c_void_p]
assert L.csam_dbg_set_gemm_ts(ts.data_ptr()) == 0
for (M, N, K) in [(4096, 1024, 4096), (4096, 1024, 1024), (5330, 1024, 4096)]:
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * 0.05).half()
    r = torch.randn(M, N, device="cuda"); out = torch.empty(M, N, device="cuda")
    big = torch.empty(64 << 20, device="cuda")
    for flush in (False, True):
        for _ in range(3):
            if flush: big.fill_(1.0)                 # operands leave the L2 / MALL
            hip.gemm_f16(a, w, out=out, residual=r)
        torch.cuda.synchronize()
        t = ts.cpu().tolist()
        nk = max(t[5], 1)
        print("M %d N %d K %d %s: per k-step cycles: wait %.0f  barrier %.0f  issue %.0f  reads+MFMA %.0f  loop edge %.0f | k-loop %.0f cycles for %d steps"
              % (M, N, K, "after a 256 MB fill" if flush else "back to back   ", t[0] / nk, t[1] / nk, t[2] / nk, t[3] / nk, t[4] / nk, t[6], nk))
