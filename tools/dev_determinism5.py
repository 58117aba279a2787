import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crowdsam_amd import hip
torch.manual_seed(0)
B = 64
dev = "cuda"
X = (torch.randn(B * 4096, 256, device=dev) * 0.5).half()
Wkv = (torch.randn(256, 256, device=dev) * 0.05).half()
kpe = torch.randn(4096, 128, device=dev); bv = torch.randn(128, device=dev)
q = (torch.randn(B * 7, 128, device=dev) * 0.5).half()
ws = torch.empty(hip.attn_t2i_workspace_bytes(B, 8) // 4, dtype=torch.float32, device=dev)
parts = []
for r in range(6):
    out = torch.zeros(B * 7, 128, dtype=torch.float16, device=dev)
    hip.t2i_fused(q, out, B, ws, X=X, Wkv=Wkv, kpe=kpe, bv=bv)
    torch.cuda.synchronize()
    parts.append(ws[: B * 32 * 56 * 18].clone().view(B, 32, 56, 18))
ref = parts[0]
for r in range(1, 6):
    d = (ref.view(torch.int32) != parts[r].view(torch.int32))
    print("run", r, "differing record floats", int(d.sum()))
    if d.any():
        idx = d.nonzero()
        print("  prompts", sorted(set(idx[:, 0].tolist()))[:20], "tiles", sorted(set(idx[:, 1].tolist())), "hj sample", sorted(set(idx[:, 2].tolist()))[:12], "fields", sorted(set(idx[:, 3].tolist())))
        i0 = idx[0]
        print("  example", i0.tolist(), ref[tuple(i0.tolist())].item(), parts[r][tuple(i0.tolist())].item())
# which one is right?  fp32 reference of the partial-free final output
outs = []
for r in range(3):
    out = torch.zeros(B * 7, 128, dtype=torch.float16, device=dev)
    if r == 0:
        # cold-ish: flush caches with a big memset
        junk = torch.empty(1 << 28, dtype=torch.float32, device=dev).fill_(1.0); torch.cuda.synchronize()
    hip.t2i_fused(q, out, B, ws, X=X, Wkv=Wkv, kpe=kpe, bv=bv)
    torch.cuda.synchronize()
    outs.append(out.float().clone())
Xf = X.float().view(B, 4096, 256)
K = Xf @ Wkv[:128].float().t() + kpe            # [B,4096,128]
V = Xf @ Wkv[128:].float().t() + bv
qf = q.float().view(B, 7, 8, 16).permute(0, 2, 1, 3)         # [B,8,7,16]
Kh = K.half().float().view(B, 4096, 8, 16).permute(0, 2, 1, 3)
Vh = V.half().float().view(B, 4096, 8, 16).permute(0, 2, 1, 3)
S = (qf @ Kh.transpose(-1, -2)) * 0.25
ref = (S.softmax(-1) @ Vh).permute(0, 2, 1, 3).reshape(B * 7, 128)
for r in range(3):
    e = (outs[r] - ref).abs()
    print("out run", r, "max err vs fp32 ref", e.max().item(), "mean", e.mean().item(), "n > 0.01:", int((e > 0.01).sum()))
