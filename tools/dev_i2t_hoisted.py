"""Time the hoisted-Q form of csam_i2t_stream alone (B prompts)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crowdsam_amd import hip
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 2048, 4096
dev = "cuda"
X = (torch.randn(T, 256, device=dev) * 0.7).half()
Q = (torch.randn(T, 128, device=dev) * 0.9).half()
k = (torch.randn(B * 7, 128, device=dev) * 0.3).half()
v = (torch.randn(B * 7, 128, device=dev) * 0.8).half()
Wo = (torch.randn(256, 128, device=dev) * 0.08).half()
bo, g, be = torch.randn(256, device=dev) * 0.2, torch.ones(256, device=dev), torch.zeros(256, device=dev)
out = torch.empty(B * T, 256, dtype=torch.float16, device=dev)
f = lambda: hip.i2t_stream(X, 0, k, v, Wo, bo, g, be, 1e-5, out, B, T, Q=Q, q_bstride=0)
for _ in range(2): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): f()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"hoisted i2t B={B}: {ms*1e3:.0f} us  write {B*T*512/ms/1e9:.2f} TB/s")
ws = torch.empty(hip.i2t_rank_workspace_bytes(B) // 2, dtype=torch.float16, device=dev)
f2 = lambda: hip.i2t_rank(X, 0, Q, 0, k, v, Wo, bo, g, be, 1e-5, out, B, T, ws)
for _ in range(2): f2()
torch.cuda.synchronize()
e0.record()
for _ in range(5): f2()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print(f"rank-56 hoisted i2t B={B}: {ms*1e3:.0f} us  write {B*T*512/ms/1e9:.2f} TB/s")
