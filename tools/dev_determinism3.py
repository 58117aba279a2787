"""Developer check: bitwise repeatability of single fused kernels on fixed inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crowdsam_amd import hip
torch.manual_seed(0)
B = 512
dev = "cuda"
X = (torch.randn(B * 4096, 256, device=dev) * 0.5).half()
Wkv = (torch.randn(256, 256, device=dev) * 0.05).half()
kpe = torch.randn(4096, 128, device=dev)
bv = torch.randn(128, device=dev)
q = (torch.randn(B * 7, 128, device=dev) * 0.5).half()
ws = torch.empty(hip.attn_t2i_workspace_bytes(B, 8) // 4, dtype=torch.float32, device=dev)
outs = []
for r in range(4):
    out = torch.zeros(B * 7, 128, dtype=torch.float16, device=dev)
    hip.t2i_fused(q, out, B, ws, X=X, Wkv=Wkv, kpe=kpe, bv=bv)
    torch.cuda.synchronize()
    outs.append(out.clone())
for r in range(1, 4):
    d = (outs[0].view(torch.int16) != outs[r].view(torch.int16))
    print("t2i_fused<1> run", r, "differing elements", int(d.sum()), "max abs diff", (outs[0].float() - outs[r].float()).abs().max().item())
# i2t<1>
k = (torch.randn(B * 7, 128, device=dev) * 0.5).half(); v = (torch.randn(B * 7, 128, device=dev) * 0.5).half()
Wq = (torch.randn(128, 256, device=dev) * 0.05).half(); qpe = torch.randn(4096, 128, device=dev)
Wo = (torch.randn(256, 128, device=dev) * 0.05).half(); bo = torch.randn(256, device=dev)
g = torch.ones(256, device=dev); be = torch.zeros(256, device=dev)
outs = []
for r in range(4):
    out = torch.zeros(B * 4096, 256, dtype=torch.float16, device=dev)
    hip.i2t_fused(X, 4096 * 256, k, v, Wo, bo, g, be, 1e-5, out, B, 4096, Wq=Wq, qpe=qpe)
    torch.cuda.synchronize()
    outs.append(out.clone())
for r in range(1, 4):
    d = (outs[0].view(torch.int16) != outs[r].view(torch.int16))
    print("i2t_fused<1> run", r, "differing elements", int(d.sum()), "max abs diff", (outs[0].float() - outs[r].float()).abs().max().item())
# t2i MODE 0 (hoisted K0 / V0T through the same attention + record code)
K0 = (torch.randn(4096, 128, device=dev) * 0.5).half(); V0T = (torch.randn(128, 4096, device=dev) * 0.5).half()
outs = []
for r in range(4):
    out = torch.zeros(B * 7, 128, dtype=torch.float16, device=dev)
    hip.t2i_fused(q, out, B, ws, K0=K0, V0T=V0T)
    torch.cuda.synchronize()
    outs.append(out.clone())
for r in range(1, 4):
    d = (outs[0].view(torch.int16) != outs[r].view(torch.int16))
    print("t2i_fused<0> run", r, "differing elements", int(d.sum()), "max abs diff", (outs[0].float() - outs[r].float()).abs().max().item())
