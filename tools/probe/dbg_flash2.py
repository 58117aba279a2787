import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crowdsam_amd import hip
cuda = torch.device("cuda")
T, nH, D = 4096, 2, 128
g = torch.Generator().manual_seed(1)
qkv = torch.randn(T, 3 * D, generator=g).to(cuda)
v = torch.zeros(T, nH, 64, device=cuda)
MODE = os.environ.get('VMODE', 'tile')
if MODE == 'tile':
    v[torch.arange(T), :, torch.arange(T) // 64] = 1.0           # v[k][d] = [tile(k) == d]
else:
    v[torch.arange(64), :, torch.arange(64)] = 1.0               # v[k][d] = [k == d], keys of tile 0 only
qkv[:, 2 * D:] = v.view(T, D)
qkv = qkv.half()
q = qkv[:, :D].float().view(T, nH, 64).transpose(0, 1)
k = qkv[:, D:2 * D].float().view(T, nH, 64).transpose(0, 1)
vv = qkv[:, 2 * D:].float().view(T, nH, 64).transpose(0, 1)
s0 = (q * 0.125) @ k.transpose(-1, -2)
ref = (s0.softmax(-1) @ vv).transpose(0, 1).reshape(T, D)
for c in (-3.0, 2.0):
    traw = torch.full((nH, T, 256), c, device=cuda)
    out = torch.zeros(T, D, device=cuda, dtype=torch.float16)
    hip.flash_attn(qkv, out, T, nH, 0.125, D, relpos=traw)
    e = (out.float() - ref)
    rows = (e.abs().max(1)[0] > (2e-3 if MODE == 'tile' else 2e-4)).nonzero().flatten()
    print("bias", c, "max err", e.abs().max().item(), "bad rows", rows.numel(), rows[:24].tolist())
    for r in rows[:3].tolist():
        for h in range(nH):
            eh = e[r, h * 64:(h + 1) * 64]
            if eh.abs().max() > (2e-3 if MODE == 'tile' else 2e-4):
                ratio = (out[r, h * 64:(h + 1) * 64].float() / ref[r, h * 64:(h + 1) * 64])
                print("  row", r, "wave", (r % 128) // 32, "rt", (r % 32) // 16, "head", h, "sum out", out[r, h * 64:(h + 1) * 64].float().sum().item())
                print("   ratio per tile:", [round(x, 3) for x in ratio.tolist()])
