"""Soak of the persistent decoder kernels: random batch sizes / seeds, each launch repeated and compared bitwise, and
checked against the tile-per-workgroup kernels.  python tools/dev_stream_soak.py [iterations]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crowdsam_amd import hip
from crowdsam_amd.decoder import _kperm
dev = "cuda"
SC = 0.25 * 1.4426950408889634
T = 4096
rs = np.random.RandomState(1234)
n_it = int(sys.argv[1]) if len(sys.argv) > 1 else 12
worst = dict(i2t=0.0, rank=0.0, t2i=0.0, up=0.0)
same = lambda a, b: bool((a.view(torch.int16) == b.view(torch.int16)).all()) if a.element_size() == 2 else bool((a.view(torch.int32) == b.view(torch.int32)).all())
for it in range(n_it):
    B = int(rs.randint(257, 1400))
    torch.manual_seed(int(rs.randint(1 << 30)))
    r = lambda *s, sc=1.0: torch.randn(*s, device=dev) * sc
    X = r(B * T, 256, sc=0.7).half()
    k, v = r(B * 7, 128, sc=0.8), r(B * 7, 128, sc=0.8).half()
    Wq, qpe = r(128, 256, sc=0.06).half(), r(T, 128, sc=0.5)
    Wo, bo = r(256, 128, sc=0.08).half(), r(256, sc=0.2)
    g, be = torch.rand(256, device=dev) + 0.5, r(256, sc=0.2)
    ks = (k * SC).half()
    o1, o2, o3 = (torch.empty(B * T, 256, dtype=torch.float16, device=dev) for _ in range(3))
    hip.i2t_stream(X, T * 256, ks, v, Wo, bo, g, be, 1e-5, o1, B, T, Wq=Wq, qpe=qpe)
    hip.i2t_stream(X, T * 256, ks, v, Wo, bo, g, be, 1e-5, o2, B, T, Wq=Wq, qpe=qpe)
    assert same(o1, o2), ("i2t_stream not repeatable", it, B)
    hip.i2t_fused(X, T * 256, k.half(), v, Wo[:, _kperm(128)].contiguous(), bo, g, be, 1e-5, o3, B, T, Wq=Wq, qpe=qpe)
    worst["i2t"] = max(worst["i2t"], (o1.float() - o3.float()).abs().max().item())
    # hoisted form: rank-56 vs stream
    Xs, Q = X[:T].contiguous(), r(T, 128, sc=0.9).half()
    ws = torch.empty(hip.i2t_rank_workspace_bytes(B) // 2, dtype=torch.float16, device=dev)
    hip.i2t_rank(Xs, 0, Q, 0, ks, v, Wo, bo, g, be, 1e-5, o1, B, T, ws)
    hip.i2t_rank(Xs, 0, Q, 0, ks, v, Wo, bo, g, be, 1e-5, o2, B, T, ws)
    assert same(o1, o2), ("i2t_rank not repeatable", it, B)
    hip.i2t_stream(Xs, 0, ks, v, Wo, bo, g, be, 1e-5, o3, B, T, Q=Q, q_bstride=0)
    worst["rank"] = max(worst["rank"], (o1.float() - o3.float()).abs().max().item())
    del o1, o2, o3
    # t2i
    Wkv, kpe, bv = r(256, 256, sc=0.06).half(), r(T, 128, sc=0.5), r(128, sc=0.3)
    q = r(B * 7, 128, sc=1.2).half()
    a1, a2, a3 = (torch.empty(B * 7, 128, dtype=torch.float16, device=dev) for _ in range(3))
    hip.t2i_stream(q, a1, B, X, Wkv, kpe, bv, T)
    hip.t2i_stream(q, a2, B, X, Wkv, kpe, bv, T)
    assert same(a1, a2), ("t2i_stream not repeatable", it, B)
    wsp = torch.empty(hip.attn_t2i_workspace_bytes(B, 8) // 4 + B * 32 * 56 * 18, dtype=torch.float32, device=dev)
    hip.t2i_fused(q, a3, B, wsp, X=X, Wkv=Wkv, kpe=kpe, bv=bv)
    worst["t2i"] = max(worst["t2i"], (a1.float() - a3.float()).abs().max().item())
    del wsp
    # upscaler
    W1, b1 = r(256, 256, sc=0.06).half(), r(256, sc=0.3)
    lg, lb = torch.rand(64, device=dev) + 0.5, r(64, sc=0.2)
    W2, b2 = r(128, 64, sc=0.15).half(), r(128, sc=0.3)
    hy = r(B, 4, 32, sc=0.7)
    m1, m2, m3 = (torch.empty(B, 4, 256, 256, device=dev) for _ in range(3))
    s1, s2 = torch.empty(B * 4, 2, device=dev), torch.empty(B * 4, 2, device=dev)
    hip.upscale_stream(X, W1, b1, lg, lb, 1e-6, W2, b2, hy, m1, B, stats=s1)
    hip.upscale_stream(X, W1, b1, lg, lb, 1e-6, W2, b2, hy, m2, B, stats=s2)
    assert same(m1, m2) and same(s1, s2), ("upscale_stream not repeatable", it, B)
    hip.upscale_fused(X, W1, b1, lg, lb, 1e-6, W2, b2, hy, m3, B, stats=s2)
    worst["up"] = max(worst["up"], ((m1 - m3).abs().max() / m3.abs().mean()).item())
    assert torch.allclose(s1[:, 0], m1.view(B * 4, -1).max(1).values)
    del m1, m2, m3
    print(f"it {it} B={B} ok  worst so far {worst}", flush=True)
print("SOAK OK", worst)
