"""ctypes binding of libcsam_hip.so -- the C-ABI boundary declared in include/csam.h.

PyTorch is used only as plumbing here: device allocations (``torch.empty``), the current HIP
stream handle and ``data_ptr()``.  Every hot operator below is a hand-written HIP kernel for
gfx950.  There is NO CPU / eager fallback: if the library is missing or a call fails, a
``RuntimeError`` is raised (the oracle under ``oracle/`` is test infrastructure and is never
imported from here).
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CSAM_LIB") or os.path.join(_HERE, "libcsam_hip.so")

ACT_NONE, ACT_GELU, ACT_RELU = 0, 1, 2
DT_F16, DT_F32 = 0, 1

_c = ctypes
_P, _I, _L, _F, _D = _c.c_void_p, _c.c_int, _c.c_long, _c.c_float, _c.c_double

# name -> argtypes (all return int).  Must match include/csam.h line by line.
SIGNATURES = {
    "csam_gemm_f16": [_P, _P, _L, _P, _L, _P, _L, _I, _P, _P, _P, _L, _I, _I, _I, _I, _I],
    "csam_gemm_f16_ln": [_P, _P, _L, _P, _L, _P, _L, _I, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P, _L, _P, _P, _I, _F, _P],
    "csam_layernorm": [_P, _P, _L, _I, _P, _L, _I, _P, _P, _I, _I, _F],
    "csam_layernorm_cast": [_P, _P, _P, _P, _I, _I, _F, _P, _P, _P, _P],
    "csam_sam_im2col": [_P, _P, _I, _I, _P, _P, _P],
    "csam_dino_im2col": [_P, _P, _I, _I, _I, _P, _P, _P],
    "csam_im2col3x3": [_P, _P, _P, _I],
    "csam_add_cast": [_P, _P, _P, _L, _P, _P, _L, _I],
    "csam_win_attn": [_P, _P, _P, _P, _P, _I, _I, _F],
    "csam_win_attn_batched": [_P, _P, _P, _P, _P, _I, _I, _F, _I],
    "csam_flash_attn": [_P, _P, _L, _I, _I, _I, _P, _P, _L, _I, _I, _F, _P, _L, _I],
    "csam_flash_attn_batched": [_P, _P, _L, _I, _I, _I, _P, _P, _L, _I, _I, _F, _P, _L, _I, _I],
    "csam_flash_attn80": [_P, _P, _L, _I, _I, _I, _P, _P, _L, _I, _I, _F, _P, _L],
    "csam_gemm_f16_resmod": [_P, _P, _L, _P, _L, _P, _L, _I, _P, _P, _L, _I, _I, _I, _I, _I, _I],
    "csam_gemm_f16_batched": [_P, _P, _L, _L, _P, _L, _L, _P, _L, _L, _I, _P, _L, _I, _I, _I, _I, _I],
    "csam_linear_f32": [_P, _P, _L, _P, _L, _P, _P, _L, _P, _L, _I, _I, _I, _I],
    "csam_linear_f32_batched": [_P, _P, _L, _L, _P, _L, _L, _P, _L, _P, _L, _L, _I, _I, _I, _I, _I],
    "csam_point_tokens": [_P, _P, _P, _P, _P, _P, _P, _I],
    "csam_point_tokens_labeled": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I],
    "csam_box_tokens": [_P, _P, _P, _P, _P, _P, _P, _I],
    "csam_pe_points": [_P, _P, _P, _P, _I],
    "csam_token_self_attn": [_P, _P, _P, _P, _I],
    "csam_attn_t2i": [_P, _P, _P, _P, _L, _L, _P, _I, _I, _I, _P, _L],
    "csam_attn_i2t": [_P, _P, _L, _L, _P, _P, _P, _I, _I, _I],
    "csam_ln64_gelu": [_P, _P, _P, _P, _L, _F],
    "csam_hyper_masks": [_P, _P, _P, _P, _I],
    "csam_softmax_stats": [_P, _P, _P, _I],
    "csam_pool_adjoint": [_P, _P, _P, _P, _P, _L, _I],
    "csam_rowscale_bias": [_P, _P, _P, _P, _P, _I, _I],
    "csam_splitk_reduce": [_P, _P, _I, _L, _P, _P, _P, _L, _P, _L, _I, _I],
    "csam_token_block_a": [_P, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _P, _I],
    "csam_token_heads": [_P, _P, _P, _I, _P, _P, _P, _P, _P, _F] + [_P] * 21 + [_I],
    "csam_token_block_b": [_P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P,
                           _P, _I],
    "csam_t2i_fused_parts": [],
    "csam_select_masks": [_P, _P, _P, _I, _P, _P, _P, _P, _I],
    "csam_mask_post": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P, _P],
    "csam_mask_post_scored": [_P, _P, _P, _P, _F, _I, _I, _I, _I, _I, _F, _F, _P, _P, _P, _P],
    "csam_post_finalize": [_P, _P, _P, _P, _P, _F, _F, _F, _P, _P, _P, _I],
    "csam_mask_write": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _P, _P],
    "csam_occupancy_lookup": [_P, _P, _I, _P, _P, _P, _I, _I, _I, _P],
    "csam_eps_select": [_P, _P, _P, _I, _I, _D, _D, _P, _P, _P],
    "csam_occupancy_prune": [_P, _P, _I, _P, _P, _P, _I, _I, _I, _P],
    "csam_post_finalize_compact": [_P, _P, _P, _P, _P, _P, _P, _F, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P],
    "csam_box_nms": [_P, _P, _P, _I, _F, _P, _P, _P, _L],
    "csam_rle_count_box": [_P, _P, _P, _P, _I, _I, _I, _P, _P],
    "csam_rle_write_box": [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P],
    "csam_coco_rle_pack": [_P, _P, _P, _P, _I, _L, _L, _P, _L, _P, _L, _P],
    "csam_mask_nms": [_P, _P, _P, _I, _I, _I, _F, _P, _P, _P, _L],
    "csam_caltech_match": [_P, _P, _P, _P, _P, _P, _I, _I, ctypes.c_double, _P, _P],
    "csam_mask_mean_bilinear": [_P, _P, _I, _I, _I, _P, _I, _I, _I, _P, _P],
    "csam_small_regions": [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _L],
    "csam_small_regions_idx": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _L],
    "csam_mask_window_copy": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I],
    "csam_bilinear_f32": [_P, _P, _I, _I, _I, _P, _I, _I],
    "csam_i2t_fused": [_P, _P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _I, _I],
    "csam_upscale_stream": [_P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _I],
    "csam_i2t_rank": [_P, _P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _F, _P, _I, _I, _P, _L],
    "csam_i2t_rank_proj": [_P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _I, _I, _P, _L],
    "csam_t2i_stream": [_P, _P, _P, _P, _P, _P, _P, _I, _I],
    "csam_t2i_rank": [_P, _P, _P, _P, _P, _P, _L, _P, _I, _I],
    "csam_i2t_t2i": [_P, _P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _I, _I, _P, _L],
    "csam_i2t_t2i_fold": [_P, _P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _I, _I, _P, _L, _I],
    "csam_i2t_stream": [_P, _P, _L, _P, _L, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _I, _I],
    "csam_upscale_fused": [_P, _P, _P, _P, _P, _P, _F, _P, _P, _P, _P, _P, _I],
    "csam_head_gather": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F],
    "csam_softmax_relpos": [_P, _P, _P, _P, _I, _I, _I, _I, _F],
    "csam_head_scatter": [_P, _P, _P, _I, _I, _I, _I, _I, _I],
    "csam_t2i_shared": [_P, _P, _P, _P, _P, _I],
    "csam_pool_adjoint_mfma": [_P, _P, _P, _P, _P, _L, _I],
    "csam_t2i_fused": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, _L],
    "csam_t2i_merge_launch": [_P, _P, _P, _I, _I],
    "csam_preprocess_pad": [_P, _P, _I, _I, _P, _P, _P],
    "csam_sigmoid_max": [_P, _P, _I, _I, _P],
    "csam_resize_linear_u8": [_P, _P, _I, _I, _P, _P, _P, _P, _I, _I, _I, _P, _P],
    "csam_u8hwc_to_f32chw": [_P, _P, _I, _I, _P],
    "csam_pil_resample_u8": [_P, _P, _I, _I, _P, _P, _P, _I, _I, _I, _P, _P],
}
LONG_RETURNS = {
    "csam_coco_rle_string": [_P, _L, _P, _L],
    "csam_coco_rle_strings": [_P, _P, _L, _P, _L, _P],
    "csam_attn_t2i_workspace_bytes": [_I, _I],
    "csam_coco_rle_pack_workspace_bytes": [_I, _L],
    "csam_box_nms_workspace_bytes": [_I],
    "csam_mask_nms_workspace_bytes": [_I],
    "csam_small_regions_workspace_bytes": [_I, _I, _I],
    "csam_small_regions_idx_workspace_bytes": [_I, _I, _I],
    "csam_t2i_fused_workspace_bytes": [_I],
    "csam_flash_attn_workspace_bytes": [_I, _I],
    "csam_i2t_rank_workspace_bytes": [_I],
    "csam_i2t_rank_proj_workspace_bytes": [_I],
    "csam_i2t_t2i_workspace_bytes": [_I],
    "csam_flash_attn80_workspace_bytes": [_I, _I],
}

_lib = None


def lib():
    """Load the library once.  Raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m crowdsam_amd.build` "
                "(the HIP extension is mandatory; there is no CPU fallback)")
        L = ctypes.CDLL(LIB_PATH)
        L.csam_abi_version.restype = _I
        L.csam_last_error.restype = _c.c_char_p
        for name, argt in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argt
            fn.restype = _I
        for name, argt in LONG_RETURNS.items():
            fn = getattr(L, name)
            fn.argtypes = argt
            fn.restype = _L
        L.csam_adj_taps_bytes.restype = _I
        L.csam_adj_mfma_bytes.restype = _I
        _lib = L
    return _lib


def _check(rc, name):
    if rc != 0:
        msg = lib().csam_last_error().decode()
        raise RuntimeError(f"{name} failed (rc={rc}): {msg}")


def _ptr(t):
    return None if t is None else _P(t.data_ptr())


def _stream():
    return _P(torch.cuda.current_stream().cuda_stream)


def _dt(dtype):
    if dtype == torch.float16:
        return DT_F16
    if dtype == torch.float32:
        return DT_F32
    raise ValueError(f"unsupported dtype {dtype}")


class KernelTimer:
    """Optional HIP-event timing of selected entry points on the launch stream (bench.py roofline leg).
    Events are recorded around each call on torch's current stream (the stream the kernels are launched
    on) and only read after the caller synchronises."""

    def __init__(self, names):
        self.names = set(names)
        self.records = []          # (name, start_event, end_event, work)

    @staticmethod
    def gemm_flops(name, args):
        if name == "csam_gemm_f16_batched":
            M, N, K, batch = args[-4:]
            return 2.0 * M * N * K * batch
        if name == "csam_gemm_f16_ln":        # (..., act, M, N, K, C16_out, ldc16, rowstats_out, rowstats_in, n_partials, eps, colsum)
            M, N, K = args[14:17]
            return 2.0 * M * N * K
        M, N, K = args[-3:]
        return 2.0 * M * N * K

    def summary(self):
        out = {}
        for name, e0, e1, work in self.records:
            d = out.setdefault(name, dict(calls=0, ms=0.0, work=0.0))
            d["calls"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["work"] += work
        return out


_timer = None
GRAPHS_ENABLED = os.environ.get("CSAM_GRAPHS", "1") != "0"


class GraphCache:
    """hipGraph capture/replay of a launch sequence that only touches static buffers.  The decoder batch is
    ~100 short kernels: replaying one graph removes the per-launch host cost (Python + ctypes + hipLaunch).
    Disabled while a KernelTimer is active (events cannot be recorded inside a replayed graph).
    Invariant: the first call for a key runs ``fn`` eagerly, captures it and then REPLAYS it, i.e. the sequence
    executes twice on the same buffers -- every captured kernel must therefore be idempotent on its static
    operands (pure functions of their inputs; no accumulating atomics onto un-reinitialised memory)."""

    def __init__(self):
        self.graphs = {}

    def clear(self):
        self.graphs.clear()

    def run(self, key, fn):
        if not GRAPHS_ENABLED or _timer is not None:
            return fn()
        ent = self.graphs.get(key)
        if ent is None:
            fn()                                   # eager warm-up (also sets kernel attributes)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = fn()
            ent = self.graphs[key] = (g, out)
        ent[0].replay()
        return ent[1]



def set_timer(timer):
    global _timer
    _timer = timer


def timer_active():
    return _timer is not None


def call(name, *args):
    if _timer is not None and name in _timer.names:
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(lib(), name)(*args)
        e1.record()
        _timer.records.append((name, e0, e1, KernelTimer.gemm_flops(name, args) if "gemm" in name else 0.0))
    else:
        rc = getattr(lib(), name)(*args)
    _check(rc, name)


# ----------------------------------------------------------------------------------------------
# GEMM
# ----------------------------------------------------------------------------------------------
def gemm_f16(a, w, out=None, bias=None, act=ACT_NONE, residual=None, colscale=None,
             out_dtype=torch.float16, M=None):
    """out[M,N] = act(a[M,K] @ w[N,K]^T + bias) * colscale + residual  (fp16 in, fp32 accumulate)."""
    assert a.dtype == torch.float16 and w.dtype == torch.float16
    assert a.stride(-1) == 1 and w.stride(-1) == 1
    if M is None:
        M = a.shape[0]
    K = a.shape[1]
    N = w.shape[0]
    assert w.shape[1] == K
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    assert out.stride(-1) == 1
    call("csam_gemm_f16", _stream(), _ptr(a), a.stride(0), _ptr(w), w.stride(0),
         _ptr(out), out.stride(0), _dt(out.dtype), _ptr(bias), _ptr(colscale),
         _ptr(residual), 0 if residual is None else residual.stride(0),
         DT_F16 if residual is None else _dt(residual.dtype), act, M, N, K)
    return out


def gemm_f16_ln(a, w, out, bias=None, act=ACT_NONE, residual=None, colscale=None, M=None, out16=None, stats_out=None,
                stats_in=None, eps=1e-6, colsum=None):
    """csam_gemm_f16 with the LayerNorm that follows / precedes it folded in (include/csam.h: csam_gemm_f16_ln).
    Producer: ``out`` fp32, ``out16`` [M,N] fp16 copy, ``stats_out`` f32 [M, N/128, 2].  Consumer: ``stats_in`` f32
    [M, K/128, 2] of the rows of ``a``, ``colsum`` f32 [N] of the gamma-folded fp16 weight ``w``."""
    assert a.dtype == torch.float16 and w.dtype == torch.float16 and a.stride(-1) == 1 and w.stride(-1) == 1
    if M is None:
        M = a.shape[0]
    K, N = a.shape[1], w.shape[0]
    assert w.shape[1] == K and out.stride(-1) == 1
    if stats_out is not None:
        assert stats_out.dtype == torch.float32 and stats_out.is_contiguous() and stats_out.numel() >= M * (N // 128) * 2
    n_part = 0
    if stats_in is not None:
        n_part = K // 128
        assert stats_in.dtype == torch.float32 and stats_in.is_contiguous() and stats_in.numel() >= M * n_part * 2
        assert colsum is not None and colsum.dtype == torch.float32 and colsum.numel() == N
    call("csam_gemm_f16_ln", _stream(), _ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(out), out.stride(0), _dt(out.dtype),
         _ptr(bias), _ptr(colscale), _ptr(residual), 0 if residual is None else residual.stride(0),
         DT_F16 if residual is None else _dt(residual.dtype), act, M, N, K, _ptr(out16),
         0 if out16 is None else out16.stride(0), _ptr(stats_out), _ptr(stats_in), n_part, float(eps), _ptr(colsum))
    return out


def fold_layernorm(w, b, gamma, beta):
    """(W', b', colsum) of LN(x) W^T + b = rstd (x W'^T - mean colsum) + b':  W' = fp16(W * gamma), colsum = row sums of
    the fp16 values actually multiplied (so the mean term cancels what the MFMA accumulated), b' = b + W beta (fp32)."""
    w32 = w.detach().float()
    wf = (w32 * gamma.detach().float()[None, :]).to(torch.float16).contiguous()
    colsum = wf.float().sum(1).contiguous()
    bf = (b.detach().float() + w32 @ beta.detach().float()).contiguous()
    return wf, bf, colsum


# ----------------------------------------------------------------------------------------------
# encoder-side kernels
# ----------------------------------------------------------------------------------------------
def layernorm(x, gamma, beta, eps, out=None, out_dtype=torch.float16, M=None):
    """Row LayerNorm over the last dim (one wave per row)."""
    if M is None:
        M = x.shape[0]
    D = x.shape[-1]
    if out is None:
        out = torch.empty((M, D), dtype=out_dtype, device=x.device)
    call("csam_layernorm", _stream(), _ptr(x), x.stride(0), _dt(x.dtype), _ptr(out), out.stride(0),
         _dt(out.dtype), _ptr(gamma), _ptr(beta), M, D, float(eps))
    return out


def layernorm_cast(x, gamma, beta, eps, out, out16=None, pe=None, outpe16=None):
    """out (fp32) = LayerNorm(x fp32 rows); out16 = fp16(out); outpe16 = fp16(out + pe): LayerNorm + add_cast x 2 in one
    launch, same arithmetic.  All tensors contiguous [M, D]."""
    M, D = x.shape
    assert x.is_contiguous() and out.is_contiguous() and x.dtype == torch.float32 and out.dtype == torch.float32
    for t in (out16, pe, outpe16):
        assert t is None or (t.is_contiguous() and tuple(t.shape) == (M, D))
    call("csam_layernorm_cast", _stream(), _ptr(x), _ptr(gamma), _ptr(beta), M, D, float(eps), _ptr(out), _ptr(out16),
         _ptr(pe), _ptr(outpe16))
    return out


_MEAN = (_F * 3)(123.675, 116.28, 103.53)   # sam.py:38-39 pixel_mean / pixel_std
_STD = (_F * 3)(58.395, 57.12, 57.375)


_ZERO3 = (_F * 3)(0.0, 0.0, 0.0)
_ONE3 = (_F * 3)(1.0, 1.0, 1.0)


def sam_im2col(img_chw, out, normalized=False):
    """Sam.preprocess + 16x16 patch im2col: img f32 [3,h,w] (0..255) -> out f16 [4096,768].
    ``normalized=True``: the input is already (x-mean)/std (API path: image_encoder(preprocessed))."""
    assert img_chw.dtype == torch.float32 and img_chw.is_contiguous()
    call("csam_sam_im2col", _stream(), _ptr(img_chw), img_chw.shape[1], img_chw.shape[2],
         _ZERO3 if normalized else _MEAN, _ONE3 if normalized else _STD, _ptr(out))
    return out


def dino_im2col(img_chw, out, normalized_1022=False):
    """preprocess + bilinear 1024->1022 + 14x14 patch im2col: -> out f16 [5329,640].
    ``normalized_1022=True``: the input is the already normalised/resized [3,1022,1022] tensor (API path)."""
    assert img_chw.dtype == torch.float32 and img_chw.is_contiguous()
    if normalized_1022:
        assert tuple(img_chw.shape) == (3, 1022, 1022)
        call("csam_dino_im2col", _stream(), _ptr(img_chw), 1022, 1022, 1022, _ZERO3, _ONE3, _ptr(out))
    else:
        call("csam_dino_im2col", _stream(), _ptr(img_chw), img_chw.shape[1], img_chw.shape[2], 1024, _MEAN, _STD,
             _ptr(out))
    return out


def im2col3x3(x, out, C):
    call("csam_im2col3x3", _stream(), _ptr(x), _ptr(out), C)
    return out


def add_cast(a, b=None, b_row_stride=0, out16=None, out32=None):
    """out = a (+ b broadcast by row stride); a f32 [M,N]; writes f16 and/or f32."""
    M, N = a.shape
    call("csam_add_cast", _stream(), _ptr(a), _ptr(b), b_row_stride, _ptr(out16), _ptr(out32), M, N)


def relcat_window(rel_h, rel_w):
    """[64, head_dim] fp16 operand of the windowed kernel: rows 0..26 rel_pos_h, 27..53 rel_pos_w, rest zero."""
    r = torch.zeros(64, rel_h.shape[1], dtype=torch.float16, device=rel_h.device)
    r[:27] = rel_h.half()
    r[27:54] = rel_w.half()
    return r


def relcat_global(rel_h, rel_w):
    """[256,64] fp16 GEMM operand of the global blocks: rows 0..126 rel_pos_h, 128..254 rel_pos_w."""
    r = torch.zeros(256, 64, dtype=torch.float16, device=rel_h.device)
    r[:127] = rel_h.half()
    r[128:255] = rel_w.half()
    return r


def win_attn(qkv, qkv_bias, relcat, out, D, nH, scale, n_images=1):
    """qkv f16 [n_images*4096, 3D] -> out f16 [n_images*4096, D]: one launch for all images of an encoder pass."""
    call("csam_win_attn_batched", _stream(), _ptr(qkv), _ptr(qkv_bias), _ptr(relcat), _ptr(out), D, nH, float(scale), n_images)
    return out


def relpos_raw(qkv, relcat_g, out, nH, n_images=1):
    """out f32 [n_images,nH,4096,256] = q_h @ relcat_g^T for every head (one batched MFMA GEMM, K = 64, per image)."""
    for b in range(n_images):
        gemm_f16_batched(qkv[b * 4096:], qkv.stride(0), 64, relcat_g, 64, 0, out[b] if n_images > 1 else out, 256, 4096 * 256,
                         4096, 256, 64, nH)
    return out


_vt_ws = {}


def flash_vt_workspace(T, nH, device, n_images=1):
    """Zero-initialised per-head V^T scratch [n_images,nH,64,Tpad] f16 (cached per shape/device; static for graphs)."""
    key = (T, nH, str(device), n_images)
    if key not in _vt_ws:
        n = lib().csam_flash_attn_workspace_bytes(T, nH) * n_images
        _vt_ws[key] = torch.zeros(n // 2, dtype=torch.float16, device=device)
    return _vt_ws[key]


FLASH_QMUL = 1.4426950408889634     # x scale: what a plan folds into the q rows of its qkv projection (q_prescaled)


def flash_attn(qkv, out, T, nH, scale, D, relpos=None, vt=None, q_prescaled=False, n_images=1):
    """qkv f16 [n_images*T, 3*D] laid out [3][nH][64] per row -> out f16 [n_images*T, D]; relpos = relpos_raw(...) or None.
    q_prescaled: the q columns (and relpos) already carry scale * log2(e).  Images are independent sequences of T rows."""
    if vt is None:
        vt = flash_vt_workspace(T, nH, qkv.device, n_images)
    call("csam_flash_attn_batched", _stream(), _ptr(qkv), qkv.stride(0), 0, D, 2 * D, _ptr(relpos), _ptr(out),
         out.stride(0), T, nH, float(scale), _ptr(vt), vt.numel() * 2, int(bool(q_prescaled)), n_images)
    return out


def flash_attn80(qkv, out, T, nH, scale, D, relpos=None):
    """head_dim 80: qkv f16 [T, 3*D] laid out [3][nH][80] per row -> out f16 [T, D]; relpos = relpos_raw80(...) or None."""
    key = ("hd80", T, nH, str(qkv.device))
    if key not in _vt_ws:
        _vt_ws[key] = torch.zeros(lib().csam_flash_attn80_workspace_bytes(T, nH) // 2, dtype=torch.float16, device=qkv.device)
    vt = _vt_ws[key]
    call("csam_flash_attn80", _stream(), _ptr(qkv), qkv.stride(0), 0, D, 2 * D, _ptr(relpos), _ptr(out), out.stride(0), T, nH,
         float(scale), _ptr(vt), vt.numel() * 2)
    return out


def relcat_global80(rel_h, rel_w):
    """[256,128] fp16 GEMM operand of the global blocks at head_dim 80: rows 0..126 rel_pos_h, 128..254 rel_pos_w, columns
    80..127 zero (the GEMM's K is 128: the 48 extra q columns it reads belong to the next head and meet zeros)."""
    r = torch.zeros(256, 128, dtype=torch.float16, device=rel_h.device)
    r[:127, :80] = rel_h.half()
    r[128:255, :80] = rel_w.half()
    return r


def relpos_raw80(qkv, relcat_g80, out, nH):
    """out f32 [nH,4096,256] = q_h @ relcat^T for every head at head_dim 80 (one batched MFMA GEMM, K = 128 zero-padded)."""
    gemm_f16_batched(qkv, qkv.stride(0), 80, relcat_g80, 128, 0, out, 256, 4096 * 256, 4096, 256, 128, nH)
    return out


def gemm_f16_resmod(a, w, out, bias, residual, res_mod, act=ACT_NONE, M=None):
    """Prompt-stacked GEMM whose residual is a per-image [res_mod, N] constant (row m %% res_mod)."""
    if M is None:
        M = a.shape[0]
    K, N = a.shape[1], w.shape[0]
    call("csam_gemm_f16_resmod", _stream(), _ptr(a), a.stride(0), _ptr(w), w.stride(0), _ptr(out), out.stride(0),
         _dt(out.dtype), _ptr(bias), _ptr(residual), residual.stride(0), _dt(residual.dtype), res_mod, act, M, N, K)
    return out


def gemm_f16_batched(a, lda, sa, w, ldw, sw, out, ldc, sc, M, N, K, batch, bias=None, sbias=0, act=ACT_NONE):
    """batch independent GEMMs (grid.z): element strides sa/sw/sc/sbias between problems."""
    call("csam_gemm_f16_batched", _stream(), _ptr(a), lda, sa, _ptr(w), ldw, sw, _ptr(out), ldc, sc, _dt(out.dtype),
         _ptr(bias), sbias, act, M, N, K, batch)
    return out


def gemm_f16_splitk(a, w, out, splits, scratch, bias=None, residual=None, rowstats=None, M=None):
    """fp32 out[M,N] = (a[M,K] @ w[N,K]^T) * (1 / rowstats[:,1]) + bias + residual with K cut into ``splits`` slices that run as
    one batched launch (csam_gemm_f16_batched) into ``scratch`` (fp32, >= splits * M * N) and are summed in slice order
    (csam_splitk_reduce).  For skinny products (M of a few hundred rows, K in the thousands) whose single-pass form is a serial
    chain of K / 64 steps on a handful of workgroups."""
    if M is None:
        M = a.shape[0]
    K = a.shape[1]
    N = w.shape[0]
    assert a.dtype == torch.float16 and w.dtype == torch.float16 and out.dtype == torch.float32 and scratch.dtype == torch.float32
    assert K % (64 * splits) == 0 and scratch.numel() >= splits * M * N and a.stride(-1) == 1 and w.stride(-1) == 1
    ks = K // splits
    call("csam_gemm_f16_batched", _stream(), _ptr(a), a.stride(0), ks, _ptr(w), w.stride(0), ks, _ptr(scratch), N, M * N,
         _dt(torch.float32), None, 0, ACT_NONE, M, N, ks, splits)
    call("csam_splitk_reduce", _stream(), _ptr(scratch), splits, M * N, _ptr(rowstats), _ptr(bias), _ptr(residual),
         0 if residual is None else residual.stride(0), _ptr(out), out.stride(0), M, N)
    return out


def frag_order(w):
    """fp16 [.., N, K] row-major -> the fragment order csam_token_block_* / csam_token_heads read (include/csam.h)."""
    *lead, N, K = w.shape
    assert N % 16 == 0 and K % 32 == 0 and w.dtype == torch.float16
    return w.reshape(*lead, N // 16, 16, K // 32, 4, 8).permute(*range(len(lead)), len(lead), len(lead) + 2, len(lead) + 3,
                                                                len(lead) + 1, len(lead) + 4).contiguous()


def token_block_a(src_qk, src_v, tokens0, residual, qk_w, qk_b, v_w, v_b, o_w, o_b, norm_g, norm_b, eps, q_w, q_b,
                  queries, q16, qpe16, t2i_q, B):
    """Token self-attention block + norm1 + the q projection of the token->image attention in one launch (small batches;
    csrc/token_block.hip).  ``src_qk`` None: layer 0, both operands are fp16(tokens0)."""
    call("csam_token_block_a", _stream(), _ptr(src_qk), _ptr(src_v), _ptr(tokens0), 1 if src_qk is None else 0, _ptr(residual),
         _ptr(qk_w), _ptr(qk_b), _ptr(v_w), _ptr(v_b), _ptr(o_w), _ptr(o_b), _ptr(norm_g), _ptr(norm_b), eps, _ptr(q_w),
         _ptr(q_b), _ptr(queries), _ptr(q16), _ptr(qpe16), _ptr(t2i_q), B)


def t2i_fused_parts():
    return int(lib().csam_t2i_fused_parts())


def token_block_b(attn_o, partials, queries, tokens0, o_w, o_b, n2_g, n2_b, m1_w, m1_b, m2_w, m2_b, n3_g, n3_b, k_w, k_b, v_w, v_b,
                  eps, q16, qpe16, i2t_k, i2t_v, B, next_q_w=None, next_q_b=None, t2i_q=None):
    """Out projection of the token->image attention + norm2 + MLP + norm3 + the k / v projections of the image->token
    attention (+ the next token->image attention's q projection) in one launch (small batches; csrc/token_block.hip)."""
    call("csam_token_block_b", _stream(), _ptr(attn_o), _ptr(partials), 0 if partials is None else t2i_fused_parts(), _ptr(queries), _ptr(tokens0), _ptr(o_w), _ptr(o_b), _ptr(n2_g), _ptr(n2_b),
         _ptr(m1_w), _ptr(m1_b), _ptr(m2_w), _ptr(m2_b), _ptr(n3_g), _ptr(n3_b), _ptr(k_w), _ptr(k_b), _ptr(v_w), _ptr(v_b),
         _ptr(next_q_w), _ptr(next_q_b), eps, _ptr(q16), _ptr(qpe16), _ptr(i2t_k), _ptr(i2t_v), _ptr(t2i_q), B)


def token_heads(attn_o, partials, queries, o_w, o_b, norm_g, norm_b, eps, hw0, hb0, hw1, hb1, hw2, hb2, iw0, ib0, iw1, ib1, iw2, ib2,
                pw0, pb0, pw1, pb1, pw2, pb2, hyper, iou0, res_iou, B):
    """Out projection + LayerNorm of the final token->image attention, the four hyper-network MLPs, the IoU head and the
    parallel residual IoU head in one launch (small batches; csrc/token_block.hip)."""
    call("csam_token_heads", _stream(), _ptr(attn_o), _ptr(partials), 0 if partials is None else t2i_fused_parts(), _ptr(queries), _ptr(o_w), _ptr(o_b), _ptr(norm_g), _ptr(norm_b), eps,
         _ptr(hw0), _ptr(hb0), _ptr(hw1), _ptr(hb1), _ptr(hw2), _ptr(hb2), _ptr(iw0), _ptr(ib0), _ptr(iw1), _ptr(ib1), _ptr(iw2),
         _ptr(ib2), _ptr(pw0), _ptr(pb0), _ptr(pw1), _ptr(pb1), _ptr(pw2), _ptr(pb2), _ptr(hyper), _ptr(iou0), _ptr(res_iou), B)


def linear_f32(a, w, bias=None, out=None, act=ACT_NONE, residual=None, M=None, lda=None):
    """fp32 out[M,N] = act(a[M,K] @ w[N,K]^T + bias) (+ residual).  ``lda`` allows strided row gathers."""
    N, K = w.shape
    if M is None:
        M = a.shape[0]
    if lda is None:
        lda = a.stride(0)
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=w.device)
    call("csam_linear_f32", _stream(), _ptr(a), lda, _ptr(w), w.stride(0), _ptr(bias), _ptr(residual),
         0 if residual is None else residual.stride(0), _ptr(out), out.stride(0), M, N, K, act)
    return out


def point_tokens(coords, gauss, out_tokens5, point_embed1, not_a_point, tokens):
    B = coords.shape[0]
    call("csam_point_tokens", _stream(), _ptr(coords), _ptr(gauss), _ptr(out_tokens5), _ptr(point_embed1),
         _ptr(not_a_point), _ptr(tokens), B)
    return tokens


def point_tokens_labeled(coords, labels, gauss, out_tokens5, point_embed0, point_embed1, not_a_point, tokens):
    """point_tokens with an int32 label per prompt (1 foreground / 0 background / -1 not a point)."""
    B = coords.shape[0]
    call("csam_point_tokens_labeled", _stream(), _ptr(coords), _ptr(labels), _ptr(gauss), _ptr(out_tokens5), _ptr(point_embed0),
         _ptr(point_embed1), _ptr(not_a_point), _ptr(tokens), B)
    return tokens


def box_tokens(boxes, gauss, out_tokens5, point_embed2, point_embed3, tokens):
    """tokens f32 [B,7,256] of B box prompts (boxes f32 [B,4] XYXY in the input frame)."""
    B = boxes.shape[0]
    call("csam_box_tokens", _stream(), _ptr(boxes), _ptr(gauss), _ptr(out_tokens5), _ptr(point_embed2), _ptr(point_embed3),
         _ptr(tokens), B)
    return tokens


def pe_points(coords, gauss, out):
    call("csam_pe_points", _stream(), _ptr(coords), _ptr(gauss), _ptr(out), coords.shape[0])
    return out


def token_self_attn(qk, v, out, B):
    call("csam_token_self_attn", _stream(), _ptr(qk), _ptr(v), _ptr(out), B)
    return out


def attn_t2i_workspace_bytes(B, nsplit):
    return lib().csam_attn_t2i_workspace_bytes(B, nsplit)


def attn_t2i(q, K, V, ldkv, kv_prompt_stride, out, B, T, nsplit, workspace):
    call("csam_attn_t2i", _stream(), _ptr(q), _ptr(K), _ptr(V), ldkv, kv_prompt_stride, _ptr(out), B, T, nsplit,
         _ptr(workspace), workspace.numel() * workspace.element_size())
    return out


def attn_i2t(Qi, ldq, q_prompt_stride, k, v, out, B, T, nsplit):
    call("csam_attn_i2t", _stream(), _ptr(Qi), ldq, q_prompt_stride, _ptr(k), _ptr(v), _ptr(out), B, T, nsplit)
    return out


def ln64_gelu(x, gamma, beta, rows, eps=1e-6):
    call("csam_ln64_gelu", _stream(), _ptr(x), _ptr(gamma), _ptr(beta), rows, float(eps))


def hyper_masks(up2, hyper, masks, B):
    call("csam_hyper_masks", _stream(), _ptr(up2), _ptr(hyper), _ptr(masks), B)


def softmax_stats(masks, stats, rows):
    call("csam_softmax_stats", _stream(), _ptr(masks), _ptr(stats), rows)


def pool_adjoint(masks, stats, taps, w, rows):
    call("csam_pool_adjoint", _stream(), _ptr(masks), _ptr(stats), _ptr(taps), _ptr(w), w.stride(0), rows)


def rowscale_bias(P, stats, bias, out, rows, N):
    call("csam_rowscale_bias", _stream(), _ptr(P), _ptr(stats), _ptr(bias), _ptr(out), rows, N)


def select_masks(iou, cls, n_class, sel, score, category, fused, B):
    call("csam_select_masks", _stream(), _ptr(iou), _ptr(cls), n_class, _ptr(sel), _ptr(score), _ptr(category),
         _ptr(fused), B)


def mask_post(lowres, sel, B, in_hw, out_hw, thr, off, out_mask, inter, uni, box, tmp=None):
    call("csam_mask_post", _stream(), _ptr(lowres), _ptr(sel), B, in_hw[0], in_hw[1], out_hw[0], out_hw[1],
         float(thr), float(off), _ptr(out_mask), _ptr(inter), _ptr(uni), _ptr(box), _ptr(tmp))


def mask_post_scored(lowres, sel, score, score_thr, B, in_hw, out_hw, thr, off, inter, uni, box, tmp=None):
    """Statistics pass (counts + box) of the selected candidates, skipping prompts with score <= score_thr."""
    call("csam_mask_post_scored", _stream(), _ptr(lowres), _ptr(sel), _ptr(score), float(score_thr), B, in_hw[0], in_hw[1],
         out_hw[0], out_hw[1], float(thr), float(off), _ptr(inter), _ptr(uni), _ptr(box), _ptr(tmp))


def post_finalize(score, inter, uni, box, pred_iou_thresh, stab_thresh, filter_thresh, stability, keep, occ, B):
    call("csam_post_finalize", _stream(), _ptr(score), _ptr(inter), _ptr(uni), _ptr(box), float(pred_iou_thresh),
         float(stab_thresh), float(filter_thresh), _ptr(stability), _ptr(keep), _ptr(occ), B)


def occupancy_lookup(points, masks, occ, B, H, W, out, slot=None):
    call("csam_occupancy_lookup", _stream(), _ptr(points), points.shape[0], _ptr(masks), _ptr(occ), _ptr(slot), B, H, W,
         _ptr(out))


def eps_select(points, alive, B, scale_x, scale_y, out_points, out_coords, counts):
    """Device-resident sampler step: the first min(B, #alive) alive points (list order) -> out_points i32 [B,2] and, scaled
    in float64 like ResizeLongestSide.apply_coords, out_coords f32 [B,2]; flags cleared; counts i32 [2] = (valid, left)."""
    call("csam_eps_select", _stream(), _ptr(points), _ptr(alive), points.shape[0], B, float(scale_x), float(scale_y),
         _ptr(out_points), _ptr(out_coords), _ptr(counts))


def occupancy_prune(points, masks, occ, B, H, W, alive, slot=None):
    call("csam_occupancy_prune", _stream(), _ptr(points), points.shape[0], _ptr(masks), _ptr(occ), _ptr(slot), B, H, W,
         _ptr(alive))


def post_finalize_compact(score, inter, uni, box, category, points, pred_iou_thresh, stab_thresh, filter_thresh, keep, occ,
                          slot, counter, store, B, edge=None, n_valid=None):
    """Filters + in-kernel compaction of the survivors into ``store`` (dict of image-level device arrays).
    ``edge`` = (crop_box, orig_box, downscale, atol) enables the crop-edge filter of crowdsam/utils.py:213-223;
    ``n_valid`` (device i32) = number of slots of the batch that hold a prompt (device-resident sampler)."""
    e10 = None
    if edge is not None:
        e10 = (_F * 10)(*[float(v) for v in edge[0]], *[float(v) for v in edge[1]], float(edge[2]), float(edge[3]))
    call("csam_post_finalize_compact", _stream(), _ptr(score), _ptr(inter), _ptr(uni), _ptr(box), _ptr(category),
         _ptr(points), float(pred_iou_thresh), float(stab_thresh), float(filter_thresh), _ptr(keep), _ptr(occ),
         _ptr(slot), _ptr(counter), _ptr(store["score"]), _ptr(store["stability"]), _ptr(store["boxes"]),
         _ptr(store["category"]), _ptr(store["points"]), B, store["score"].shape[0],
         None if e10 is None else _c.cast(e10, _P), _ptr(n_valid))


def box_nms(boxes, scores, thr):
    """Greedy NMS (torchvision semantics).  boxes f32 [N,4], scores f32 [N] -> kept indices int64 (device)."""
    N = boxes.shape[0]
    if N == 0:
        return torch.zeros(0, dtype=torch.int64, device=boxes.device)
    boxes = boxes.contiguous().float()
    scores = scores.contiguous().float()
    nbytes = lib().csam_box_nms_workspace_bytes(N)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=boxes.device)
    keep = torch.empty(N, dtype=torch.int64, device=boxes.device)
    count = torch.zeros(1, dtype=torch.int32, device=boxes.device)
    call("csam_box_nms", _stream(), _ptr(boxes), _ptr(scores), N, float(thr), _ptr(keep), _ptr(count), _ptr(ws), nbytes)
    return keep[: int(count.item())]


def caltech_match(dt, dt_off, gt, gt_off, gt_npos, thres):
    """Caltech matching of the CrowdHuman evaluator on device (float64).  dt [Nd,5] / gt [Ng,5] float64 device
    tensors in the order csam.h documents, CSR offsets int64, gt_npos int32 -> (label int8 [Nd], pos uint8 [Nd])."""
    n_img = gt_npos.numel()
    label = torch.empty((dt.shape[0],), dtype=torch.int8, device=dt.device)
    pos = torch.empty((dt.shape[0],), dtype=torch.uint8, device=dt.device)
    if n_img == 0 or dt.shape[0] == 0:
        return label, pos
    max_pos = int(gt_npos.max().item())
    call("csam_caltech_match", _stream(), _ptr(dt), _ptr(dt_off), _ptr(gt), _ptr(gt_off), _ptr(gt_npos), n_img, max_pos,
         float(thres), _ptr(label), _ptr(pos))
    return label, pos


def mask_mean_bilinear(masks, sim):
    """Mean over each mask's pixels of `sim` ([fh,fw] f32, may be a strided view of a wider map) resized bilinearly
    to the mask frame; 0 for an empty mask.  masks u8/bool [n,H,W] -> f32 [n]."""
    n, H, W = masks.shape
    if n == 0:
        return torch.zeros((0,), dtype=torch.float32, device=masks.device)
    m8 = (masks.view(torch.uint8) if masks.dtype == torch.bool else masks).contiguous()
    assert sim.dtype == torch.float32 and sim.stride(1) == 1
    s = torch.empty((n,), dtype=torch.float64, device=masks.device)
    c = torch.empty((n,), dtype=torch.int32, device=masks.device)
    call("csam_mask_mean_bilinear", _stream(), _ptr(m8), n, H, W, _ptr(sim), sim.shape[0], sim.shape[1], sim.stride(0),
         _ptr(s), _ptr(c))
    return torch.where(c > 0, s / c.clamp(min=1), torch.zeros_like(s)).float()


def small_regions(masks, min_area):
    """Hole filling then island removal (8-connected) of u8/bool masks [n,H,W] on device.
    -> (edited masks u8, changed int32 [n], boxes f32 [n,4])."""
    n, H, W = masks.shape
    m8 = (masks.view(torch.uint8) if masks.dtype == torch.bool else masks).contiguous()
    out = torch.empty_like(m8)
    changed = torch.empty((n,), dtype=torch.int32, device=m8.device)
    boxes = torch.empty((n, 4), dtype=torch.float32, device=m8.device)
    if n == 0:
        return out, changed, boxes
    nbytes = lib().csam_small_regions_workspace_bytes(n, H, W)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=m8.device)
    call("csam_small_regions", _stream(), _ptr(m8), _ptr(out), _ptr(changed), _ptr(boxes), n, H, W, int(min_area),
         _ptr(ws), nbytes)
    return out, changed, boxes


def small_regions_idx(mask_store, idx, min_area, out_store=None):
    """Compact form (csam_small_regions_idx): cleans up the masks ``mask_store[idx[i]]`` WHERE THEY LIE (``out_store`` None:
    in place; else the same slots of ``out_store``) -- no gather of the NMS survivors, no per-pixel label arrays.
    mask_store u8/bool [cap,H,W]; idx int32 [n] device tensor or None (= the first n = cap masks).
    -> (changed int32 [n], boxes f32 [n,4])."""
    cap, H, W = mask_store.shape
    assert mask_store.is_contiguous()
    m8 = mask_store.view(torch.uint8) if mask_store.dtype == torch.bool else mask_store
    o8 = m8 if out_store is None else (out_store.view(torch.uint8) if out_store.dtype == torch.bool else out_store)
    n = cap if idx is None else int(idx.shape[0])
    changed = torch.empty((n,), dtype=torch.int32, device=m8.device)
    boxes = torch.empty((n, 4), dtype=torch.float32, device=m8.device)
    if n == 0:
        return changed, boxes
    if idx is not None:
        assert idx.dtype == torch.int32 and idx.is_contiguous()
    nbytes = lib().csam_small_regions_idx_workspace_bytes(n, H, W)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=m8.device)
    call("csam_small_regions_idx", _stream(), _ptr(m8), _ptr(idx), _ptr(o8), _ptr(changed), _ptr(boxes), n, H, W,
         int(min_area), _ptr(ws), nbytes)
    return changed, boxes


def mask_window_copy(store, slots, windows, crop, to_store, only=None):
    """Gather (to_store False) the windows ``windows`` int32 [n,6] = (x0, y0, w, h, ox, oy) of the masks store[slots[i]] into the
    dense stack ``crop`` u8 [n, Hc, Wc] at (ox, oy) (zero elsewhere), or scatter them back (``only`` u8 [n]: just those masks)."""
    cap, H, W = store.shape
    n, Hc, Wc = crop.shape
    s8 = store.view(torch.uint8) if store.dtype == torch.bool else store
    assert s8.is_contiguous() and crop.is_contiguous() and crop.dtype == torch.uint8 and windows.dtype == torch.int32
    assert windows.is_contiguous() and tuple(windows.shape) == (n, 6)
    assert slots is None or (slots.dtype == torch.int32 and slots.is_contiguous())
    call("csam_mask_window_copy", _stream(), _ptr(s8), _ptr(slots), _ptr(windows), _ptr(only), _ptr(crop), n, H, W, Hc, Wc,
         1 if to_store else 0)


SMALL_REGIONS_PAD = 16       # ring of background kept around a mask's box in the windowed clean-up


def small_regions_windowed(mask_store, idx, boxes_xyxy, min_area):
    """small_regions_idx restricted to the masks' bounding boxes (round 4).  remove_small_regions (amg.py:267-291) labels the
    whole frame, but every component of the mask and every hole lies inside the mask's box; with a 16-pixel ring of
    background around the box, the clean-up of that WINDOW is the clean-up of the frame: the ring is background, connected to
    everything outside the window exactly as the outside is connected in the frame, at least 16 x 16 pixels (> min_area)
    wherever it is not the frame's own border strip, and no island touches it.  A window cut by a frame edge keeps that edge
    as ITS edge (it is placed flush with the stack's edge on that side: the zero padding never extends the frame).  Masks
    whose window covers at most a quarter of the frame are gathered into one dense stack (csam_mask_window_copy), cleaned
    there and scattered back where they changed; the rest -- frame-filling masks, masks cut by two opposite frame edges,
    empty masks -- take the full-frame call.  ``boxes_xyxy``: the masks' boxes in store coordinates (inclusive maxima,
    batched_mask_to_box), the ones the statistics pass of the mask post-processing produced.  Same return as
    small_regions_idx; bit-identical results (tests/test_regions_gpu.py)."""
    cap, H, W = mask_store.shape
    n = int(idx.shape[0])
    p = SMALL_REGIONS_PAD
    if n < 4 or min_area > p * p:
        return small_regions_idx(mask_store, idx, min_area)
    import numpy as np
    b = boxes_xyxy.detach().to(torch.int64).cpu().numpy()              # one small D2H (n x 4)
    wx0, wy0 = np.maximum(b[:, 0] - p, 0), np.maximum(b[:, 1] - p, 0)
    wx1, wy1 = np.minimum(b[:, 2] + p, W - 1), np.minimum(b[:, 3] + p, H - 1)
    cl, ct, cr, cb = b[:, 0] - p < 0, b[:, 1] - p < 0, b[:, 2] + p > W - 1, b[:, 3] + p > H - 1      # cut by a frame edge
    ww, wh = wx1 - wx0 + 1, wy1 - wy0 + 1
    small = (b[:, 2] >= b[:, 0]) & (b[:, 3] >= b[:, 1]) & (b[:, 2] + b[:, 3] > 0) & ~(cl & cr) & ~(ct & cb) & (ww * wh * 4 <= H * W)
    ns = int(small.sum())
    if ns < 4:
        return small_regions_idx(mask_store, idx, min_area)
    Hc, Wc = int(-(-wh[small].max() // 64) * 64), int(-(-ww[small].max() // 64) * 64)
    if Hc * Wc * 2 > H * W:                                            # nothing to gain
        return small_regions_idx(mask_store, idx, min_area)
    dev = mask_store.device
    changed = torch.empty((n,), dtype=torch.int32, device=dev)
    boxes = torch.empty((n, 4), dtype=torch.float32, device=dev)
    sel_s = torch.as_tensor(np.nonzero(small)[0], device=dev)
    ox = np.where(cr[small], Wc - ww[small], 0)                        # flush right / bottom when cut there (left / top cuts sit
    oy = np.where(cb[small], Hc - wh[small], 0)                        # at the stack's origin anyway)
    win = torch.as_tensor(np.stack([wx0[small], wy0[small], ww[small], wh[small], ox, oy], 1).astype(np.int32)).to(dev)
    slots_s = idx[sel_s].contiguous()
    crop = torch.empty((ns, Hc, Wc), dtype=torch.uint8, device=dev)
    mask_window_copy(mask_store, slots_s, win, crop, False)
    ch_s, bx_s = small_regions_idx(crop, None, min_area)
    # stack -> store coordinates.  A non-empty mask cannot have the box (0, 0, 0, 0) unless it is the single pixel at the stack's
    # origin, which is then the frame's (window cut left and top) origin pixel as well: the shift is 0 - 0 in that case
    nonempty = (bx_s != 0).any(1, keepdim=True)
    shift = (win[:, [0, 1, 0, 1]] - win[:, [4, 5, 4, 5]]).to(torch.float32)
    bx_s = torch.where(nonempty, bx_s + shift, bx_s)
    mask_window_copy(mask_store, slots_s, win, crop, True, only=(ch_s != 0).to(torch.uint8))
    changed[sel_s], boxes[sel_s] = ch_s, bx_s
    if ns < n:
        sel_l = torch.as_tensor(np.nonzero(~small)[0], device=dev)
        ch_l, bx_l = small_regions_idx(mask_store, idx[sel_l].contiguous(), min_area)
        changed[sel_l], boxes[sel_l] = ch_l, bx_l
    return changed, boxes


def mask_nms(masks, scores, thr):
    """Coverage NMS over 150x150 nearest-resampled masks (crowdsam/utils.py mask_iou_nms).  masks u8/bool [N,H,W],
    scores f32 [N] -> kept indices int64 (device) in descending-score order."""
    N, H, W = masks.shape
    if N == 0:
        return torch.zeros((0,), dtype=torch.int64, device=masks.device)
    m8 = (masks.view(torch.uint8) if masks.dtype == torch.bool else masks).contiguous()
    nbytes = lib().csam_mask_nms_workspace_bytes(N)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=masks.device)
    keep = torch.empty((N,), dtype=torch.int64, device=masks.device)
    count = torch.zeros((1,), dtype=torch.int32, device=masks.device)
    call("csam_mask_nms", _stream(), _ptr(m8), _ptr(scores.float().contiguous()), N, H, W, float(thr), _ptr(keep),
         _ptr(count), _ptr(ws), nbytes)
    return keep[: int(count.item())]


_D2H = {"buf": None}
_D2H_CHUNK = 1 << 24


def to_host_numpy(t):
    """Device tensor -> a NEW numpy array through a fixed 16 MB PINNED staging buffer, in chunks (a pageable ``.cpu()`` of a
    100 MB result runs at a few GB/s; a staging buffer that GROWS with the largest result re-pins memory in the middle of a
    stream -- the 20-40 ms stall of single frames that rounds 3-4 chased).  The result owns its memory."""
    import numpy as np
    n = t.numel() * t.element_size()
    np_dtype = torch.empty(0, dtype=t.dtype).numpy().dtype
    if n == 0:
        return np.empty(tuple(t.shape), dtype=np_dtype)
    if _D2H["buf"] is None:
        _D2H["buf"] = torch.empty(_D2H_CHUNK, dtype=torch.uint8).pin_memory()
    src = t.contiguous().view(-1).view(torch.uint8)
    out = np.empty(n, dtype=np.uint8)
    stream = torch.cuda.current_stream(t.device)
    for off in range(0, n, _D2H_CHUNK):
        m = min(_D2H_CHUNK, n - off)
        host = _D2H["buf"][:m]
        host.copy_(src[off:off + m], non_blocking=True)
        stream.synchronize()
        out[off:off + m] = host.numpy()
    return out.view(np_dtype).reshape(tuple(t.shape))


def rle_coco_strings(masks, idx=None, boxes=None):
    """COCO compressed-RLE strings (list of str) of u8 masks without the change positions ever visiting the host:
    csam_rle_count_box -> [D2H of N totals: sizes the position buffer] -> csam_rle_write_box -> csam_coco_rle_pack ->
    D2H of the N + 1 string offsets and of exactly the string bytes.  Arguments as for rle_encode."""
    cap_, H, W = masks.shape
    dev = masks.device
    N = cap_ if idx is None else int(idx.shape[0])
    if N == 0:
        return []
    col = torch.empty((N, W), dtype=torch.int32, device=dev)
    totals = torch.empty(N, dtype=torch.int32, device=dev)
    assert boxes is None or (boxes.dtype == torch.int32 and boxes.is_contiguous() and tuple(boxes.shape) == (N, 4))
    call("csam_rle_count_box", _stream(), _ptr(masks), _ptr(idx), _ptr(boxes), N, H, W, _ptr(col), _ptr(totals))
    pos_off = torch.zeros(N + 1, dtype=torch.int64, device=dev)
    torch.cumsum(totals, 0, out=pos_off[1:])
    # pixel (0, 0) of every mask: a strided view of the store's first column, N bytes gathered on the device
    col0 = masks.view(cap_, -1)[:, 0]
    first = (col0 if idx is None else col0.index_select(0, idx.long())).contiguous()
    n_pos = int(pos_off[-1].item())                      # the one small synchronising D2H before the strings
    pos = torch.empty(max(n_pos, 1), dtype=torch.int32, device=dev)
    call("csam_rle_write_box", _stream(), _ptr(masks), _ptr(idx), _ptr(boxes), N, H, W, _ptr(col), _ptr(pos_off), _ptr(pos))
    max_counts = n_pos + 2 * N
    nws = lib().csam_coco_rle_pack_workspace_bytes(N, max_counts)
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    cap = 5 * max_counts + 16
    while True:
        chars = torch.empty(cap, dtype=torch.uint8, device=dev)
        str_off = torch.empty(N + 1, dtype=torch.int64, device=dev)
        call("csam_coco_rle_pack", _stream(), _ptr(pos), _ptr(pos_off), _ptr(first), N, H * W, max_counts, _ptr(ws), nws,
             _ptr(chars), cap, _ptr(str_off))
        o = to_host_numpy(str_off)
        total = int(o[-1])
        if total <= cap:
            break
        cap = total + 16                               # H * W > 2^24: more than 5 characters per count
    raw = to_host_numpy(chars[:total]).tobytes().decode("ascii")
    o = o.tolist()
    return [raw[o[i]:o[i + 1]] for i in range(N)]


def rle_encode(masks, idx=None, boxes=None):
    """Column-major change positions of u8 masks -> (positions uint32 (device), offsets int64 (host, N+1)).
    ``idx`` None: masks [N,H,W]; else masks is a store [cap,H,W] and idx (int32 device [N]) names the slots to encode.
    ``boxes`` int32 [N,4] (x0, y0, x1, y1, inclusive): the masks' bounding boxes -- the passes then read the boxes only."""
    cap, H, W = masks.shape
    N = cap if idx is None else int(idx.shape[0])
    col = torch.empty((N, W), dtype=torch.int32, device=masks.device)
    totals = torch.empty(N, dtype=torch.int32, device=masks.device)
    assert boxes is None or (boxes.dtype == torch.int32 and boxes.is_contiguous() and tuple(boxes.shape) == (N, 4))
    call("csam_rle_count_box", _stream(), _ptr(masks), _ptr(idx), _ptr(boxes), N, H, W, _ptr(col), _ptr(totals))
    tot = totals.cpu().to(torch.int64)
    offs = torch.zeros(N + 1, dtype=torch.int64)
    offs[1:] = torch.cumsum(tot, 0)
    out = torch.empty(max(int(offs[-1]), 1), dtype=torch.int32, device=masks.device)
    offs_dev = offs[:-1].to(masks.device)
    call("csam_rle_write_box", _stream(), _ptr(masks), _ptr(idx), _ptr(boxes), N, H, W, _ptr(col), _ptr(offs_dev), _ptr(out))
    return out, offs


def bilinear_f32(src, out_hw):
    """fp32 [n, sh, sw] -> [n, H, W], torch F.interpolate(mode='bilinear', align_corners=False) semantics."""
    n, sh, sw = src.shape
    out = torch.empty((n, out_hw[0], out_hw[1]), dtype=torch.float32, device=src.device)
    call("csam_bilinear_f32", _stream(), _ptr(src), n, sh, sw, _ptr(out), out_hw[0], out_hw[1])
    return out


def preprocess_pad(img_chw):
    """Sam.preprocess: raw f32 [3,h,w] -> normalised, zero-padded f32 [3,1024,1024]."""
    out = torch.empty((3, 1024, 1024), dtype=torch.float32, device=img_chw.device)
    call("csam_preprocess_pad", _stream(), _ptr(img_chw), img_chw.shape[1], img_chw.shape[2], _MEAN, _STD, _ptr(out))
    return out


def resize_linear_u8(src_hwc, tables, out_hw, want_u8=True, want_f32chw=True):
    """cv2.resize INTER_LINEAR of a uint8 [h,w,3] device frame (crowdsam/utils.py:149).  ``tables`` from
    crowdsam_amd.resize.cv2_linear_tables_device; returns (uint8 [dh,dw,3] | None, fp32 [3,dh,dw] | None)."""
    sh, sw, c = src_hwc.shape
    assert c == 3 and src_hwc.dtype == torch.uint8 and src_hwc.is_contiguous()
    dh, dw = out_hw
    u8 = torch.empty((dh, dw, 3), dtype=torch.uint8, device=src_hwc.device) if want_u8 else None
    f32 = torch.empty((3, dh, dw), dtype=torch.float32, device=src_hwc.device) if want_f32chw else None
    if tables is None:      # exact 2x decimation
        call("csam_resize_linear_u8", _stream(), _ptr(src_hwc), sh, sw, None, None, None, None, dh, dw, 1, _ptr(u8),
             _ptr(f32))
    else:
        xofs, xcoef, yofs, ycoef = tables
        call("csam_resize_linear_u8", _stream(), _ptr(src_hwc), sh, sw, _ptr(xofs), _ptr(xcoef), _ptr(yofs),
             _ptr(ycoef), dh, dw, 0, _ptr(u8), _ptr(f32))
    return u8, f32


def pil_resize_bilinear_u8(src_hwc, out_hw, tables_x, tables_y, want_f32chw=True):
    """PIL.Image.resize((w, h), BILINEAR) of a uint8 [h,w,3] device frame: horizontal pass, uint8 rounding, vertical pass
    (a pass whose size does not change is skipped).  tables_* from crowdsam_amd.resize.pil_bilinear_tables_device.
    Returns (uint8 [dh,dw,3], fp32 [3,dh,dw] | None)."""
    sh, sw, c = src_hwc.shape
    assert c == 3 and src_hwc.dtype == torch.uint8 and src_hwc.is_contiguous()
    dh, dw = out_hw
    cur, f32 = src_hwc, None
    passes = ([(0, sh, dw, tables_x)] if dw != sw else []) + ([(1, dh, dw, tables_y)] if dh != sh else [])
    for i, (axis, oh, ow, tb) in enumerate(passes):
        out = torch.empty((oh, ow, 3), dtype=torch.uint8, device=src_hwc.device)
        last = i == len(passes) - 1
        if last and want_f32chw:
            f32 = torch.empty((3, oh, ow), dtype=torch.float32, device=src_hwc.device)
        call("csam_pil_resample_u8", _stream(), _ptr(cur), cur.shape[0], cur.shape[1], _ptr(tb[0]), _ptr(tb[1]), _ptr(tb[2]),
             oh, ow, axis, _ptr(out), _ptr(f32) if last else None)
        cur = out
    if not passes:
        cur = src_hwc.clone()
    if want_f32chw and f32 is None:
        f32 = u8hwc_to_f32chw(cur)
    return cur, f32


def u8hwc_to_f32chw(src_hwc):
    """uint8 [h,w,3] -> fp32 [3,h,w] (values 0..255)."""
    h, w, c = src_hwc.shape
    assert c == 3 and src_hwc.dtype == torch.uint8 and src_hwc.is_contiguous()
    out = torch.empty((3, h, w), dtype=torch.float32, device=src_hwc.device)
    call("csam_u8hwc_to_f32chw", _stream(), _ptr(src_hwc), h, w, _ptr(out))
    return out


def sigmoid_max(x):
    """x f32 [C, N] -> max_c sigmoid(x) f32 [N]."""
    C, N = x.shape
    out = torch.empty(N, dtype=torch.float32, device=x.device)
    call("csam_sigmoid_max", _stream(), _ptr(x), C, N, _ptr(out))
    return out


def i2t_fused(X, x_bstride, k, v, Wo_perm, bo, gamma, beta, eps, out, B, T, Q=None, q_bstride=0, Wq=None, qpe=None):
    """Fused image->token half-block: [Q-proj] -> 7-key attention -> out-proj -> +residual -> LayerNorm."""
    call("csam_i2t_fused", _stream(), _ptr(X), x_bstride, _ptr(Q), q_bstride, _ptr(Wq), _ptr(qpe), _ptr(k), _ptr(v),
         _ptr(Wo_perm), _ptr(bo), _ptr(gamma), _ptr(beta), float(eps), _ptr(out), B, T)
    return out


def i2t_stream(X, x_bstride, k_scaled, v, Wo, bo, gamma, beta, eps, out, B, T, Q=None, q_bstride=0, Wq=None, qpe=None):
    """Persistent weight-stationary form of ``i2t_fused`` (one 8-wave workgroup per CU streaming 128-token tiles).
    ``k_scaled`` is the token-side k projection times 0.25*log2(e); ``Wo`` is the plain [256,128] out-proj weight."""
    call("csam_i2t_stream", _stream(), _ptr(X), x_bstride, _ptr(Q), q_bstride, _ptr(Wq), _ptr(qpe), _ptr(k_scaled),
         _ptr(v), _ptr(Wo), _ptr(bo), _ptr(gamma), _ptr(beta), float(eps), _ptr(out), B, T)
    return out


def upscale_fused(keys, W1, b1, ln_g, ln_b, eps, W2_perm, b2, hyper, masks, B, stats=None):
    """Fused ConvT -> LN2d -> GELU -> ConvT -> GELU -> hyper product: keys f16 [B*4096,256] -> masks f32 [B,4,256,256].
    ``stats`` f32 [B*4,2]: column 0 receives the per-plane max of the logits (for the PWD-Net softmax)."""
    call("csam_upscale_fused", _stream(), _ptr(keys), _ptr(W1), _ptr(b1), _ptr(ln_g), _ptr(ln_b), float(eps),
         _ptr(W2_perm), _ptr(b2), _ptr(hyper), _ptr(masks), _ptr(stats), B)
    return masks


def pool_adjoint_mfma(masks, stats, tables, w, rows):
    call("csam_pool_adjoint_mfma", _stream(), _ptr(masks), _ptr(stats), _ptr(tables), _ptr(w), w.stride(0), rows)


def t2i_fused(q, out, B, workspace, X=None, Wkv=None, kpe=None, bv=None, K0=None, V0T=None):
    """Fused token->image attention: K/V projections of the key state + softmax + PV + partial merge (``out`` None: the
    partial records stay in ``workspace`` for a consumer that merges them, csam_token_block_b / csam_token_heads)."""
    call("csam_t2i_fused", _stream(), _ptr(X), _ptr(Wkv), _ptr(kpe), _ptr(bv), _ptr(K0), _ptr(V0T), _ptr(q), _ptr(out),
         B, _ptr(workspace), workspace.numel() * workspace.element_size())
    return out


def i2t_rank_workspace_bytes(B):
    return lib().csam_i2t_rank_workspace_bytes(B)


def i2t_rank(X, x_bstride, Q, q_bstride, k_scaled, v, Wo, bo, gamma, beta, eps, out, B, T, workspace):
    """Hoisted-Q image->token half-block in its rank-56 form (wave-local, barrier-free inside a prompt)."""
    call("csam_i2t_rank", _stream(), _ptr(X), x_bstride, _ptr(Q), q_bstride, _ptr(k_scaled), _ptr(v), _ptr(Wo), _ptr(bo),
         _ptr(gamma), _ptr(beta), float(eps), _ptr(out), B, T, _ptr(workspace),
         workspace.numel() * workspace.element_size())
    return out


def i2t_rank_proj_workspace_bytes(B):
    return lib().csam_i2t_rank_proj_workspace_bytes(B)


def i2t_rank_proj(X, x_bstride, qpe16, Wq, k_scaled, v, Wo, bo, gamma, beta, eps, out, B, T, workspace):
    """csam_i2t_rank for per-prompt keys (layer 1): qpe16 f16 [T,128] = pe Wq^T + bq, Wq f16 [128,256]; the q projection of
    the image tokens is replaced by 56 back-projected token keys per prompt."""
    call("csam_i2t_rank_proj", _stream(), _ptr(X), x_bstride, _ptr(qpe16), _ptr(Wq), _ptr(k_scaled), _ptr(v), _ptr(Wo),
         _ptr(bo), _ptr(gamma), _ptr(beta), float(eps), _ptr(out), B, T, _ptr(workspace),
         workspace.numel() * workspace.element_size())
    return out


def upscale_stream(keys, W1, b1, ln_g, ln_b, eps, W2_perm, b2, hyper, masks, B, stats=None):
    """Persistent weight-stationary form of ``upscale_fused`` (whole prompts per workgroup, W1 slices in registers)."""
    call("csam_upscale_stream", _stream(), _ptr(keys), _ptr(W1), _ptr(b1), _ptr(ln_g), _ptr(ln_b), float(eps),
         _ptr(W2_perm), _ptr(b2), _ptr(hyper), _ptr(masks), _ptr(stats), B)
    return masks


def t2i_stream(q, out, B, X, Wkv, kpe, bv, T=4096):
    """Persistent weight-stationary token->image attention (K/V projections fused, online softmax, no partials)."""
    call("csam_t2i_stream", _stream(), _ptr(X), _ptr(Wkv), _ptr(kpe), _ptr(bv), _ptr(q), _ptr(out), B, T)
    return out


def t2i_rank(X, Wk, kpe16, q_scaled, Qp_ws, Y, B, T=4096):
    """Rank-56 token->image attention: X f16 [B*T,256]; Wk f16 [128,256]; kpe16 f16 [T,128] = pe Wk^T; q_scaled f16 [B*7,128]
    (already x 0.25 log2 e); Qp_ws f16 >= [B*64,256] scratch; Y f16 [B*7, 8*256] <- softmax-weighted key sums per (query, head)."""
    call("csam_t2i_rank", _stream(), _ptr(X), _ptr(Wk), _ptr(kpe16), _ptr(q_scaled), _ptr(Qp_ws), Qp_ws.numel() * 2, _ptr(Y), B, T)
    return Y


def i2t_t2i_workspace_bytes(B):
    return lib().csam_i2t_t2i_workspace_bytes(B)


def i2t_t2i(X, x_bstride, Q, q_bstride, Wq, k_scaled, v, Wo, bo, gamma, beta, eps, out, t2i_Wk, t2i_kpe16, t2i_q_scaled, Y,
            B, T, workspace, fold=0):
    """``i2t_rank`` (Wq None: Q = hoisted image-side queries) / ``i2t_rank_proj`` (Q = qpe16) writing ``out``, with the NEXT
    block's ``t2i_rank`` folded in: Y f16 [B*7, 8*256] comes from the new keys while they are still in LDS.
    ``fold`` (csam_i2t_t2i_fold): 1 = out-projection bias in M_b; 3 = also plain normalised keys (gamma / beta folded into
    the consumers by the caller; ``t2i_Wk`` must then be Wk (.) gamma)."""
    args = (_stream(), _ptr(X), x_bstride, _ptr(Q), q_bstride, _ptr(Wq) if Wq is not None else None,
            _ptr(k_scaled), _ptr(v), _ptr(Wo), _ptr(bo), _ptr(gamma), _ptr(beta), float(eps), _ptr(out), _ptr(t2i_Wk),
            _ptr(t2i_kpe16), _ptr(t2i_q_scaled), _ptr(Y), B, T, _ptr(workspace), workspace.numel() * workspace.element_size())
    if fold:
        call("csam_i2t_t2i_fold", *args, int(fold))
    else:
        call("csam_i2t_t2i", *args)
    return out, Y


def head_gather(qkv, qkv_bias, Qs, K, VT, D, nH, hd, Tp, T_valid, window, scale):
    call("csam_head_gather", _stream(), _ptr(qkv), _ptr(qkv_bias), _ptr(Qs), _ptr(K), _ptr(VT), D, nH, hd, Tp, T_valid,
         int(window), float(scale))


def softmax_relpos(S, traw, P, G, Tp, T_valid, side, inv_scale):
    call("csam_softmax_relpos", _stream(), _ptr(S), _ptr(traw), _ptr(P), G, Tp, T_valid, side, float(inv_scale))


def head_scatter(O, out, D, nH, hd, Tp, T_valid, window):
    call("csam_head_scatter", _stream(), _ptr(O), _ptr(out), D, nH, hd, Tp, T_valid, int(window))


def t2i_shared(q, Kh, Vh, out, B):
    """Layer-0 token->image attention of the whole prompt batch over the per-image K / V tiles."""
    call("csam_t2i_shared", _stream(), _ptr(q), _ptr(Kh), _ptr(Vh), _ptr(out), B)
    return out


def linear_f32_batched(a, lda, sa, w, ldw, sw, bias, sbias, out, ldc, sc, M, N, K, batch, act=ACT_NONE):
    call("csam_linear_f32_batched", _stream(), _ptr(a), lda, sa, _ptr(w), ldw, sw, _ptr(bias), sbias, _ptr(out), ldc, sc,
         M, N, K, act, batch)
    return out


def mask_write(lowres, sel, keep, B, in_hw, out_hw, thr, out_mask, tmp=None, slot=None):
    """Second pass: mask bytes of the prompts with keep[b] != 0, at out_mask[b] or out_mask[slot[b]]."""
    call("csam_mask_write", _stream(), _ptr(lowres), _ptr(sel), _ptr(keep), _ptr(slot), B, in_hw[0], in_hw[1], out_hw[0],
         out_hw[1], float(thr), _ptr(out_mask), _ptr(tmp))


def coco_rle_strings(counts_list):
    """Host helper: COCO compressed-RLE strings of many masks in ONE C call (run-length arrays -> list of str)."""
    import numpy as np
    n = len(counts_list)
    if n == 0:
        return []
    sizes = np.fromiter((len(c) for c in counts_list), dtype=np.int64, count=n)
    offs = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(sizes, out=offs[1:])
    allc = np.ascontiguousarray(np.concatenate([np.asarray(c, dtype=np.int64) for c in counts_list]), dtype=np.int64) \
        if offs[-1] else np.zeros(1, np.int64)
    cap = int(offs[-1]) * 13 + 1
    buf = ctypes.create_string_buffer(cap)
    out_offs = np.zeros(n + 1, dtype=np.int64)
    tot = lib().csam_coco_rle_strings(allc.ctypes.data_as(_P), offs.ctypes.data_as(_P), n, buf, cap,
                                      out_offs.ctypes.data_as(_P))
    if tot < 0:
        raise RuntimeError("csam_coco_rle_strings: buffer too small")
    raw = buf.raw[:tot].decode("ascii")
    o = out_offs.tolist()
    return [raw[o[i]:o[i + 1]] for i in range(n)]


def coco_rle_string(counts):
    """Host helper: COCO compressed-RLE string for a list / array of run lengths."""
    import numpy as np
    c = np.ascontiguousarray(counts, dtype=np.int64)
    if c.size == 0:
        return ""
    buf = ctypes.create_string_buffer(int(c.size) * 13 + 1)
    n = lib().csam_coco_rle_string(c.ctypes.data_as(_P), int(c.size), buf, len(buf))
    if n < 0:
        raise RuntimeError("csam_coco_rle_string: buffer too small")
    return buf.raw[:n].decode("ascii")
