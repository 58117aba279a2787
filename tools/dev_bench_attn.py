import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crowdsam_amd import hip
def bench(T, nH, bias=False, iters=20):
    D = nH * 64
    qkv = torch.randn(T, 3 * D, device="cuda").half()
    out = torch.empty(T, D, device="cuda", dtype=torch.float16)
    rp = None
    if bias:
        rp = torch.randn(nH, T, 256, device="cuda")
    for _ in range(3): hip.flash_attn(qkv, out, T, nH, 0.125, D, relpos=rp, q_prescaled=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): hip.flash_attn(qkv, out, T, nH, 0.125, D, relpos=rp, q_prescaled=True)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(f"flash T={T} nH={nH} bias={bias}: {ms*1e3:.1f} us  {4*T*T*64*nH/ms/1e9:.1f} TFLOP/s")
bench(5330, 16); bench(4096, 16, True); bench(4096, 16)
# window attention
nH, D = 16, 1024
qkv = torch.randn(4096, 3 * D, device="cuda").half(); b = torch.randn(3 * D, device="cuda")
rh = torch.randn(27, 64, device="cuda"); rw = torch.randn(27, 64, device="cuda"); out = torch.empty(4096, D, device="cuda", dtype=torch.float16)
rc = hip.relcat_window(rh, rw)
for _ in range(3): hip.win_attn(qkv, b, rc, out, D, nH, 0.125)
torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
for _ in range(20): hip.win_attn(qkv, b, rc, out, D, nH, 0.125)
e1.record(); torch.cuda.synchronize(); print(f"win_attn: {e0.elapsed_time(e1)/20*1e3:.1f} us")
