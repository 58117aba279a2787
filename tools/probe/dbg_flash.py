import sys; sys.path.insert(0, "/root/repo")
import torch
from crowdsam_amd import hip
cuda = torch.device("cuda")
T, nH, D = 4096, 2, 128
g = torch.Generator().manual_seed(1)
qkv = torch.randn(T, 3 * D, generator=g).to(cuda).half()
q = qkv[:, :D].float().view(T, nH, 64).transpose(0, 1)
k = qkv[:, D:2 * D].float().view(T, nH, 64).transpose(0, 1)
v = qkv[:, 2 * D:].float().view(T, nH, 64).transpose(0, 1)
s0 = (q * 0.125) @ k.transpose(-1, -2)
def run(traw, bias_full):
    out = torch.zeros(T, D, device=cuda, dtype=torch.float16)
    hip.flash_attn(qkv, out, T, nH, 0.125, D, relpos=traw)
    ref = ((s0 + bias_full).softmax(-1) @ v).transpose(0, 1).reshape(T, D)
    e = (out.float() - ref).abs()
    return e.max().item(), e.mean().item()
z = torch.zeros(nH, T, 256, device=cuda)
print("zeros", run(z, 0))
print("ones", run(z + 1, 0))
th = torch.randn(nH, T, 256, generator=g).to(cuda) * 2
qi = torch.arange(T, device=cuda)
qh, qw = qi >> 6, qi & 63
kh, kw = qi >> 6, qi & 63
# Th[q][kh] = traw[q][qh - kh + 63], Tw[q][kw] = traw[q][128 + qw - kw + 63]
def full(tr, use_h, use_w):
    b = torch.zeros(nH, T, T, device=cuda)
    if use_h:
        idx = (qh[:, None] - kh[None, :] + 63)
        b += torch.gather(tr, 2, idx[None].expand(nH, T, T))
    if use_w:
        idx = (128 + qw[:, None] - kw[None, :] + 63)
        b += torch.gather(tr, 2, idx[None].expand(nH, T, T))
    return b
t_h = th.clone(); t_h[:, :, 128:] = 0
t_w = th.clone(); t_w[:, :, :128] = 0
print("Th only", run(t_h, full(th, True, False)))
print("Tw only", run(t_w, full(th, False, True)))
print("both", run(th, full(th, True, True)))
def where(traw, bias_full, name):
    out = torch.zeros(T, D, device=cuda, dtype=torch.float16)
    hip.flash_attn(qkv, out, T, nH, 0.125, D, relpos=traw)
    ref = ((s0 + bias_full).softmax(-1) @ v).transpose(0, 1).reshape(T, D)
    e = (out.float() - ref).abs()
    rows = (e.max(1)[0] > 1e-3).nonzero().flatten()
    print(name, "bad rows", rows.numel(), rows[:40].tolist())
    if rows.numel():
        r = rows[0].item()
        print(" row", r, "err per head", e[r].view(nH, 64).max(1)[0].tolist(), "nan?", torch.isnan(out[r]).any().item())
        hh = e[r].view(nH, 64).max(1)[0].argmax().item()
        sc = (s0 + bias_full)[hh, r] if torch.is_tensor(bias_full) else s0[hh, r]
        print(" score max", sc.max().item(), "argmax key", sc.argmax().item(), "tile0 max", sc[:64].max().item())
where(z + 1, 0, "ones")
where(t_w, full(th, False, True), "Tw")
print("---- guard band")
big = torch.ones(nH * T * 256 * 3, device=cuda)
mid = big[nH * T * 256: 2 * nH * T * 256].view(nH, T, 256)
print("ones in guard band", run(mid, 0))
big2 = torch.full((nH * T * 256 * 3,), 1000.0, device=cuda)
mid2 = big2[nH * T * 256: 2 * nH * T * 256].view(nH, T, 256)
mid2.fill_(1.0)
print("ones in 1000-band", run(mid2, 0))
print("ones again", run(z + 1, 0), run(z + 1, 0))
print("halves", run(z + 0.5, 0))
print("minus", run(z - 3, 0))
