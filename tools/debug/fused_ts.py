"""developer: s_memtime phase stamps of one producer tile / reader step of csam_i2t_t2i (library built with -DFUSE_TS)."""
import sys, ctypes, torch
sys.path.insert(0, ".")
from crowdsam_amd import hip
cuda = torch.device("cuda:0")
SC = 0.25 * 1.4426950408889634
B, T = 4096, 4096
proj = len(sys.argv) > 1
gen = torch.Generator().manual_seed(1)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc).to(cuda)
k_s, v = (r(B * 7, 128, sc=0.8) * SC).half(), r(B * 7, 128, sc=0.8).half()
Wo, bo = r(256, 128, sc=0.08).half(), r(256, sc=0.2)
g, be = (torch.rand(256, generator=gen) + 0.5).to(cuda), r(256, sc=0.2)
Wk, kpe16 = r(128, 256, sc=0.06).half(), r(T, 128, sc=0.5).half()
qs = (r(B * 7, 128, sc=1.2) * SC).half()
X0, Q0 = r(T, 256, sc=0.7).half(), r(T, 128, sc=0.9).half()
Wq = r(128, 256, sc=0.06).half()
Xp = (torch.randn(64 * T, 256, generator=gen) * 0.7).half().to(cuda).repeat(B // 64, 1) if proj else None
out = torch.zeros(B * T, 256, dtype=torch.float16, device=cuda)
Y = torch.zeros(B * 7, 2048, dtype=torch.float16, device=cuda)
ws = torch.empty(hip.i2t_t2i_workspace_bytes(B) // 2, dtype=torch.float16, device=cuda)
ts = torch.zeros(32, dtype=torch.int64, device=cuda)
L = hip.lib()
L.csam_dbg_set_fuse_ts.argtypes = [ctypes.c_void_p]
L.csam_dbg_set_fuse_ts(ts.data_ptr())
for _ in range(3):
    if proj:
        hip.i2t_t2i(Xp, T * 256, Q0, 0, Wq, k_s, v, Wo, bo, g, be, 1e-5, out, Wk, kpe16, qs, Y, B, T, ws)
    else:
        hip.i2t_t2i(X0, 0, Q0, 0, None, k_s, v, Wo, bo, g, be, 1e-5, out, Wk, kpe16, qs, Y, B, T, ws)
    torch.cuda.synchronize()
    t = ts.cpu().tolist()
    names = ["start", "scores", "softmax", "acc-init", "P.M", "LN-stats", "normalize", "wait", "stores", "barrier"]
    p = t[:10]
    print("producer:", "  ".join("%s %d" % (names[i], p[i] - p[i - 1]) for i in range(1, 10)), " total", p[9] - p[0])
    q = t[16:23]
    rn = ["start", "S0", "softmax0", "Y0", "S1", "softmax1", "Y1"]
    print("reader:  ", "  ".join("%s %d" % (rn[i], q[i] - q[i - 1]) for i in range(1, 7)), " total", q[6] - q[0])
