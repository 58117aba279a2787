import sys, torch
sys.path.insert(0, ".")
from crowdsam_amd import hip
cuda = torch.device("cuda:0")
SC = 0.25 * 1.4426950408889634
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 1, 4096
gen = torch.Generator().manual_seed(1)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(cuda)
k_s, v = (r(B * 7, 128, sc=0.8) * SC).half(), r(B * 7, 128, sc=0.8).half()
Wo, bo = r(256, 128, sc=0.08).half(), r(256, sc=0.2)
g, be = (torch.rand(256, generator=gen) + 0.5).to(cuda), r(256, sc=0.2)
Wk, kpe16 = r(128, 256, sc=0.06).half(), r(T, 128, sc=0.5).half()
qs = (r(B * 7, 128, sc=1.2) * SC).half()
qp = torch.empty(B * 64, 256, dtype=torch.float16, device=cuda)
X0, Q0 = r(T, 256, sc=0.7).half(), r(T, 128, sc=0.9).half()
out = torch.zeros(B * T, 256, dtype=torch.float16, device=cuda)
Y = torch.zeros(B * 7, 2048, dtype=torch.float16, device=cuda)
ws = torch.empty(hip.i2t_t2i_workspace_bytes(B) // 2, dtype=torch.float16, device=cuda)
proj = len(sys.argv) > 2
out2 = torch.full_like(out, float("nan")); Y2 = torch.full_like(Y, float("nan"))
if proj:
    Xs = r(5 * T, 256, sc=0.7).half()
    X = Xs.view(5, T * 256)[torch.arange(B, device=cuda) % 5].contiguous().view(B * T, 256)
    Wq = r(128, 256, sc=0.06).half()
    hip.i2t_rank_proj(X, T * 256, Q0, Wq, k_s, v, Wo, bo, g, be, 1e-5, out, B, T, ws)
    hip.t2i_rank(out, Wk, kpe16, qs, qp, Y, B, T)
    hip.i2t_t2i(X, T * 256, Q0, 0, Wq, k_s, v, Wo, bo, g, be, 1e-5, out2, Wk, kpe16, qs, Y2, B, T, ws)
else:
    hip.i2t_rank(X0, 0, Q0, 0, k_s, v, Wo, bo, g, be, 1e-5, out, B, T, ws)
    hip.t2i_rank(out, Wk, kpe16, qs, qp, Y, B, T)
    hip.i2t_t2i(X0, 0, Q0, 0, None, k_s, v, Wo, bo, g, be, 1e-5, out2, Wk, kpe16, qs, Y2, B, T, ws)
torch.cuda.synchronize()
bad = out.view(torch.int16) != out2.view(torch.int16)
print("out mismatches", int(bad.sum()), "of", bad.numel(), "nan", int(out2.isnan().sum()))
rows = bad.any(1).nonzero().flatten()
print("rows", rows[:40].tolist(), "n rows", rows.numel())
if rows.numel():
    r0 = int(rows[0]); c = bad[r0].nonzero().flatten()
    print("row", r0, "cols", c[:32].tolist(), out[r0, c[:6]].tolist(), out2[r0, c[:6]].tolist())
    print("max abs diff", (out.float() - out2.float()).abs().nan_to_num(9).max().item())
badY = Y.view(torch.int16) != Y2.view(torch.int16)
print("Y mismatches", int(badY.sum()), "of", badY.numel(), (Y.float() - Y2.float()).abs().nan_to_num(9).max().item())

rowsY = badY.view(B, 7, 8, 256).any(-1).any(-1).any(-1).nonzero().flatten()
print("Y bad prompts", rowsY[:40].tolist(), rowsY.numel())
if rowsY.numel():
    b0 = int(rowsY[0]); bb = badY.view(B, 7, 8, 256)[b0]
    print("prompt", b0, "bad (j,head) pairs", bb.any(-1).nonzero()[:20].tolist(), "n bad ch", int(bb.sum()))
    jj, hh = bb.any(-1).nonzero()[0].tolist()
    print(Y.view(B,7,8,256)[b0,jj,hh,:8].tolist(), Y2.view(B,7,8,256)[b0,jj,hh,:8].tolist())
