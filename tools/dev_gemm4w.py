"""Developer harness for the hand-scheduled four-wave GEMM (gemm4w_kernel): error against an fp32 reference, an output hash (run
it under two builds with CSAM_LIB=... to check bit-identity), time beside the vendor library on the image-batched encoder shapes.
    python tools/dev_gemm4w.py [--quick]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crowdsam_amd import hip

dev = torch.device("cuda")


def tm(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [(300, 2048, 128), (256, 2048, 256), (1000, 2304, 4096), (4096, 3072, 1024), (4096, 4096, 1024), (5330, 3072, 1024),
          (16384, 3072, 1024), (16384, 4096, 1024), (21320, 3072, 1024), (21320, 4096, 1024), (4096, 4096, 4096), (16384, 4096, 4096)]
if "--quick" in sys.argv:
    shapes = shapes[:4]
AB = "--ab" in sys.argv          # timing only, three shapes: for tools/debug/gemm4w_variants.sh builds
if AB:
    shapes = [(16384, 3072, 1024), (16384, 4096, 1024), (16384, 4096, 4096)]
print("lib:", hip.LIB_PATH)
for M, N, K in shapes:
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to(dev).half()
    w = (torch.randn(N, K, generator=g) * 0.05).to(dev).half()
    bias = torch.randn(N, generator=g).to(dev)
    out = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
    hip.gemm_f16(a, w, out=out, bias=bias)
    torch.cuda.synchronize()
    h = hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12]
    rows = torch.randint(0, M, (64,), generator=g).to(dev)
    ref = a[rows].float() @ w.float().t() + bias
    err = (out[rows].float() - ref).abs().max().item()
    # repeated launches must agree bitwise (an LDS-DMA race would show as flicker)
    flick = 0
    for _ in range(0 if AB else 6):
        o2 = torch.empty_like(out)
        hip.gemm_f16(a, w, out=o2, bias=bias)
        flick += int(not torch.equal(o2, out))
    # the LayerNorm-consumer + GELU epilogue (csam_gemm_f16_ln): hash for the two-build bit-identity check, error vs fp32 LN -> GEMM -> GELU
    h2, err2, t_ln = "-", float("nan"), float("nan")
    if K // 128 <= 10 and (K // 128) % 2 == 0:
        af = a.float().view(M, K // 128, 128)
        stats = torch.stack([af.sum(-1), (af * af).sum(-1)], -1).contiguous()
        colsum = w.float().sum(1).contiguous()
        o3 = torch.empty_like(out)
        hip.gemm_f16_ln(a, w, o3, bias=bias, act=hip.ACT_GELU, stats_in=stats, colsum=colsum, eps=1e-6)
        torch.cuda.synchronize()
        h2 = hashlib.sha1(o3.cpu().numpy().tobytes()).hexdigest()[:12]
        x = a[rows].float()
        xn = (x - x.mean(1, keepdim=True)) * torch.rsqrt(x.var(1, unbiased=False, keepdim=True) + 1e-6)
        ref2 = torch.nn.functional.gelu(xn @ w.float().t() + bias)
        err2 = (o3[rows].float() - ref2).abs().max().item()
        t_ln = tm(lambda: hip.gemm_f16_ln(a, w, o3, bias=bias, act=hip.ACT_GELU, stats_in=stats, colsum=colsum, eps=1e-6))
    t_own = tm(lambda: hip.gemm_f16(a, w, out=out, bias=bias))
    o16 = torch.empty_like(out)
    t_lib = t_own if AB else tm(lambda: torch.mm(a, w.t(), out=o16))
    fl = 2.0 * M * N * K / 1e6
    print(f"M={M:6d} N={N:5d} K={K:5d}: own {t_own:7.1f} us {fl / t_own:7.1f} TF/s | vendor {t_lib:7.1f} us {fl / t_lib:7.1f} TF/s | "
          f"ratio {t_lib / t_own:.2f} | err {err:.2e} finite {bool(torch.isfinite(out).all())} flicker {flick} hash {h} | "
          f"LN+GELU {t_ln:7.1f} us err {err2:.2e} hash {h2}", flush=True)

# ---- the residual projections: fp32 out + fp32 residual (+ LayerScale) + fp16 copy + LayerNorm partials (csam_gemm_f16_ln producer)
if "--f32" in sys.argv or not AB:
    for M, N, K, cs in [(16384, 1024, 1024, 0), (16384, 1024, 4096, 0), (21320, 1024, 1024, 1), (21320, 1024, 4096, 1), (12288, 1024, 1024, 0),
                        (20000, 1024, 4096, 1), (4096, 1024, 1024, 0)]:
        g = torch.Generator(device="cpu").manual_seed(M + N + K)
        a = (torch.randn(M, K, generator=g) * 0.5).to(dev).half()
        w = (torch.randn(N, K, generator=g) * 0.05).to(dev).half()
        bias = torch.randn(N, generator=g).to(dev)
        res = torch.randn(M, N, generator=g).to(dev)
        colscale = (torch.rand(N, generator=g) + 0.5).to(dev) if cs else None
        out = torch.full((M, N), float("nan"), device=dev)
        out16 = torch.full((M, N), float("nan"), device=dev, dtype=torch.float16)
        stats = torch.full((M, N // 128, 2), float("nan"), device=dev)
        run = lambda: hip.gemm_f16_ln(a, w, out, bias=bias, residual=res, colscale=colscale, out16=out16, stats_out=stats)
        run()
        torch.cuda.synchronize()
        hs = [hashlib.sha1(t.cpu().numpy().tobytes()).hexdigest()[:10] for t in (out, out16, stats)]
        rows = torch.randint(0, M, (64,), generator=g).to(dev)
        ref = a[rows].float() @ w.float().t() + bias
        if cs:
            ref = ref * colscale
        ref = ref + res[rows]
        err = (out[rows] - ref).abs().max().item()
        o4 = out.view(M, N // 128, 128)
        serr = max((stats[..., 0] - o4.sum(-1)).abs().max().item(), ((stats[..., 1] - (o4 * o4).sum(-1)).abs() / (o4 * o4).sum(-1)).max().item())
        ok16 = torch.equal(out16, out.half())
        # in place (C == residual), as the encoders call it
        res2 = res.clone()
        hip.gemm_f16_ln(a, w, res2, bias=bias, residual=res2, colscale=colscale, out16=out16, stats_out=stats)
        inpl = torch.equal(res2, out)
        t_own = tm(run)
        o16 = torch.empty(M, N, device=dev, dtype=torch.float16)
        t_lib = t_own if AB else tm(lambda: torch.mm(a, w.t(), out=o16))
        fl = 2.0 * M * N * K / 1e6
        print(f"f32 M={M:6d} N={N:5d} K={K:5d} cs={cs}: own {t_own:7.1f} us {fl / t_own:7.1f} TF/s | vendor (fp16, no epilogue) {t_lib:7.1f} us | ratio {t_lib / t_own:.2f} | "
              f"err {err:.2e} stats err {serr:.2e} fp16 copy ok {ok16} in-place ok {inpl} finite {bool(torch.isfinite(out).all() and torch.isfinite(stats).all())} hash {' '.join(hs)}", flush=True)
