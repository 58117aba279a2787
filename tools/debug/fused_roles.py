import sys, torch
sys.path.insert(0, ".")
from crowdsam_amd import hip
cuda = torch.device("cuda:0")
SC = 0.25 * 1.4426950408889634
B, T = 4096, 4096
gen = torch.Generator().manual_seed(1)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(cuda)
k_s, v = (r(B * 7, 128, sc=0.8) * SC).half(), r(B * 7, 128, sc=0.8).half()
Wo, bo = r(256, 128, sc=0.08).half(), r(256, sc=0.2)
g, be = (torch.rand(256, generator=gen) + 0.5).to(cuda), r(256, sc=0.2)
Wk, kpe16 = r(128, 256, sc=0.06).half(), r(T, 128, sc=0.5).half()
qs = (r(B * 7, 128, sc=1.2) * SC).half()
Xp = (torch.randn(64 * T, 256, generator=gen) * 0.7).half().to(cuda).repeat(B // 64, 1)
X0, Q0 = r(T, 256, sc=0.7).half(), r(T, 128, sc=0.9).half()
Wq, qpe = r(128, 256, sc=0.06).half(), r(T, 128, sc=0.5).half()
out = torch.zeros(B * T, 256, dtype=torch.float16, device=cuda)
Y = torch.zeros(B * 7, 2048, dtype=torch.float16, device=cuda)
ws = torch.empty(hip.i2t_t2i_workspace_bytes(B) // 2, dtype=torch.float16, device=cuda)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("L0 fold1 %.3f ms" % timeit(lambda: hip.i2t_t2i(X0, 0, Q0, 0, None, k_s, v, Wo, bo, g, be, 1e-5, out, Wk, kpe16, qs, Y, B, T, ws, fold=1)))
print("L1 fold3 %.3f ms" % timeit(lambda: hip.i2t_t2i(Xp, T * 256, qpe, 0, Wq, k_s, v, Wo, bo, g, be, 1e-5, out, Wk, kpe16, qs, Y, B, T, ws, fold=3)))
