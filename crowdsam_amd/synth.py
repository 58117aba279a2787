"""Portable synthetic weights and frames (there is no network: no checkpoints, no CrowdHuman).

State dicts use the reference's parameter names and shapes exactly (SURVEY.md §5 "checkpoint";
segment_anything_cs/build_sam.py:104-158, modeling/*.py constructors; DINOv2 ViT-L/14 keys per
SURVEY.md Appendix C), generated from ``np.random.RandomState`` (legacy MT19937 stream, frozen
across numpy versions) so the authoring container and the GPU box build bit-identical weights.
"""
import math
from collections import OrderedDict

import numpy as np
import torch

SAM_CONFIGS = {
    # name: (embed_dim, depth, heads, global_attn_indexes)   build_sam.py:14-45
    "vit_b": (768, 12, 12, (2, 5, 8, 11)),
    "vit_l": (1024, 24, 16, (5, 11, 17, 23)),
    "vit_h": (1280, 32, 16, (7, 15, 23, 31)),
    # narrow test encoder with the real token geometry (SURVEY.md §8c golden recipe)
    "vit_tiny_test": (64, 2, 2, (1,)),
    "vit_test128": (128, 4, 2, (1, 3)),
}


def _lin(out, name, n_out, n_in, bias=True):
    out.append((name + ".weight", (n_out, n_in), "w", n_in))
    if bias:
        out.append((name + ".bias", (n_out,), "b", 0))


def _ln(out, name, n):
    out.append((name + ".weight", (n,), "g", 0))
    out.append((name + ".bias", (n,), "b", 0))


def sam_param_specs(embed_dim, depth, heads, global_idx, n_class=1):
    """Ordered (name, shape, kind, fan_in) list == keys of the reference's Sam.state_dict()."""
    s = []
    D = embed_dim
    hd = D // heads
    E = "image_encoder."
    s.append((E + "pos_embed", (1, 64, 64, D), "p", 0))
    s.append((E + "patch_embed.proj.weight", (D, 3, 16, 16), "w", 768))
    s.append((E + "patch_embed.proj.bias", (D,), "b", 0))
    for i in range(depth):
        B = f"{E}blocks.{i}."
        _ln(s, B + "norm1", D)
        L = 127 if i in global_idx else 27
        s.append((B + "attn.rel_pos_h", (L, hd), "r", 0))
        s.append((B + "attn.rel_pos_w", (L, hd), "r", 0))
        _lin(s, B + "attn.qkv", 3 * D, D)
        _lin(s, B + "attn.proj", D, D)
        _ln(s, B + "norm2", D)
        _lin(s, B + "mlp.lin1", 4 * D, D)
        _lin(s, B + "mlp.lin2", D, 4 * D)
    s.append((E + "neck.0.weight", (256, D, 1, 1), "w", D))
    _ln(s, E + "neck.1", 256)
    s.append((E + "neck.2.weight", (256, 256, 3, 3), "w", 2304))
    _ln(s, E + "neck.3", 256)
    P = "prompt_encoder."
    s.append((P + "pe_layer.positional_encoding_gaussian_matrix", (2, 128), "n", 0))
    for i in range(4):
        s.append((P + f"point_embeddings.{i}.weight", (1, 256), "e", 0))
    s.append((P + "not_a_point_embed.weight", (1, 256), "e", 0))
    s.append((P + "mask_downscaling.0.weight", (4, 1, 2, 2), "w", 4))
    s.append((P + "mask_downscaling.0.bias", (4,), "b", 0))
    _ln(s, P + "mask_downscaling.1", 4)
    s.append((P + "mask_downscaling.3.weight", (16, 4, 2, 2), "w", 16))
    s.append((P + "mask_downscaling.3.bias", (16,), "b", 0))
    _ln(s, P + "mask_downscaling.4", 16)
    s.append((P + "mask_downscaling.6.weight", (256, 16, 1, 1), "w", 16))
    s.append((P + "mask_downscaling.6.bias", (256,), "b", 0))
    s.append((P + "no_mask_embed.weight", (1, 256), "e", 0))
    M = "mask_decoder."
    for i in range(2):
        T = f"{M}transformer.layers.{i}."
        for a, internal in (("self_attn", 256), ("cross_attn_token_to_image", 128),
                            ("cross_attn_image_to_token", 128)):
            for pr in ("q_proj", "k_proj", "v_proj"):
                _lin(s, T + a + "." + pr, internal, 256)
            _lin(s, T + a + ".out_proj", 256, internal)
        for n in ("norm1", "norm2", "norm3", "norm4"):
            _ln(s, T + n, 256)
        _lin(s, T + "mlp.lin1", 2048, 256)
        _lin(s, T + "mlp.lin2", 256, 2048)
    T = M + "transformer.final_attn_token_to_image."
    for pr in ("q_proj", "k_proj", "v_proj"):
        _lin(s, T + pr, 128, 256)
    _lin(s, T + "out_proj", 256, 128)
    _ln(s, M + "transformer.norm_final_attn", 256)
    s.append((M + "iou_token.weight", (1, 256), "e", 0))
    s.append((M + "mask_tokens.weight", (4, 256), "e", 0))
    s.append((M + "output_upscaling.0.weight", (256, 64, 2, 2), "w", 256))
    s.append((M + "output_upscaling.0.bias", (64,), "b", 0))
    _ln(s, M + "output_upscaling.1", 64)
    s.append((M + "output_upscaling.3.weight", (64, 32, 2, 2), "w", 64))
    s.append((M + "output_upscaling.3.bias", (32,), "b", 0))
    for i in range(5):  # trap 5: five hyper-MLPs exist, index 4 is never used
        H = f"{M}output_hypernetworks_mlps.{i}.layers."
        _lin(s, H + "0", 256, 256)
        _lin(s, H + "1", 256, 256)
        _lin(s, H + "2", 32, 256)
    H = M + "iou_prediction_head.layers."
    _lin(s, H + "0", 256, 256)
    _lin(s, H + "1", 256, 256)
    _lin(s, H + "2", 4, 256)
    _lin(s, M + "dino_proj", 256, 1024)
    H = M + "parallel_iou_head.layers."
    _lin(s, H + "0", 256, 512)
    _lin(s, H + "1", 256, 256)
    _lin(s, H + "2", 1, 256)
    H = M + "point_classifier.layers."
    _lin(s, H + "0", 256, 256)
    _lin(s, H + "1", n_class, 256)
    return s


def dino_param_specs(embed_dim=1024, depth=24, patch=14, grid=37):
    """facebookresearch/dinov2 ViT-L/14 checkpoint keys."""
    s = []
    D = embed_dim
    s.append(("cls_token", (1, 1, D), "e", 0))
    s.append(("pos_embed", (1, 1 + grid * grid, D), "p", 0))
    s.append(("mask_token", (1, D), "e", 0))
    s.append(("patch_embed.proj.weight", (D, 3, patch, patch), "w", 3 * patch * patch))
    s.append(("patch_embed.proj.bias", (D,), "b", 0))
    for i in range(depth):
        B = f"blocks.{i}."
        _ln(s, B + "norm1", D)
        _lin(s, B + "attn.qkv", 3 * D, D)
        _lin(s, B + "attn.proj", D, D)
        s.append((B + "ls1.gamma", (D,), "s", 0))
        _ln(s, B + "norm2", D)
        _lin(s, B + "mlp.fc1", 4 * D, D)
        _lin(s, B + "mlp.fc2", D, 4 * D)
        s.append((B + "ls2.gamma", (D,), "s", 0))
    _ln(s, "norm", D)
    return s


def _draw(rs, shape, kind, fan_in):
    n = int(np.prod(shape))
    x = rs.standard_normal(n).astype(np.float32).reshape(shape)
    if kind == "w":
        return x * np.float32(1.0 / math.sqrt(fan_in))
    if kind == "b":
        return x * np.float32(0.1)
    if kind == "g":
        return np.float32(1.0) + x * np.float32(0.1)
    if kind == "p":
        return x * np.float32(0.5)
    if kind == "r":
        return x * np.float32(0.25)
    if kind == "e":
        return x * np.float32(0.5)
    if kind == "s":  # LayerScale gamma
        return np.float32(0.5) + x * np.float32(0.1)
    if kind == "n":
        return x
    raise ValueError(kind)


def make_state_dict(specs, seed):
    rs = np.random.RandomState(seed)
    sd = OrderedDict()
    for name, shape, kind, fan_in in specs:
        sd[name] = torch.from_numpy(_draw(rs, shape, kind, fan_in))
    return sd


def make_sam_state_dict(arch="vit_l", n_class=1, seed=0):
    D, depth, heads, gidx = SAM_CONFIGS[arch]
    return make_state_dict(sam_param_specs(D, depth, heads, gidx, n_class), seed)


def shipped_scale_heads(sd, logit_gain=30.0, iou_shift=-0.2, cls_shift=2.0):
    """Seeded random weights whose decision values straddle the SHIPPED thresholds of configs/crowdhuman.yaml:43-58.
    With plain random weights the shipped test block is degenerate (stability 0.001-0.4 against 0.8: no mask survives;
    fused scores 0.3-0.6 against filter_thresh 0.7: nothing prunes).  Three head constants are moved IN PLACE so that every
    shipped threshold is an active decision boundary on the same architecture: the last layer of the four used
    hyper-MLPs x logit_gain (mask logits x 30 -> stability 0.80-0.97), the IoU head's output bias + iou_shift and the
    point classifier's output bias + cls_shift (fused scores 0.45-0.95 around 0.7; the FG prior moves with the
    classifier, 0.28-0.99 around pos_sim_thresh 0.5).  Used by the shipped-scale EPS golden
    (oracle/make_goldens.py::golden_pipeline_eps_shipped) and its GPU test; returns sd."""
    M = "mask_decoder."
    for i in range(4):
        sd[f"{M}output_hypernetworks_mlps.{i}.layers.2.weight"] *= logit_gain
        sd[f"{M}output_hypernetworks_mlps.{i}.layers.2.bias"] *= logit_gain
    sd[M + "iou_prediction_head.layers.2.bias"] += iou_shift
    sd[M + "point_classifier.layers.1.bias"] += cls_shift
    return sd


def make_dino_state_dict(embed_dim=1024, depth=24, seed=1):
    return make_state_dict(dino_param_specs(embed_dim, depth), seed)


def synthetic_crowd_frame(index, size=1024, n_ellipses=150):
    """Synthetic crowded frame: grey gradient + random filled ellipses."""
    rs = np.random.RandomState(index)
    h = w = size
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.empty((h, w, 3), np.float32)
    base = 64.0 + 96.0 * (xx + yy) / float(h + w)
    img[:] = base[..., None]
    for _ in range(n_ellipses):
        cx, cy = rs.uniform(0, w), rs.uniform(0, h)
        ax, ay = rs.uniform(15, 60), rs.uniform(15, 60)
        col = rs.uniform(0, 255, size=3).astype(np.float32)
        x0, x1 = int(max(0, cx - ax)), int(min(w, cx + ax + 1))
        y0, y1 = int(max(0, cy - ay)), int(min(h, cy + ay + 1))
        if x1 <= x0 or y1 <= y0:
            continue
        sub = ((xx[y0:y1, x0:x1] - cx) / ax) ** 2 + ((yy[y0:y1, x0:x1] - cy) / ay) ** 2 <= 1.0
        img[y0:y1, x0:x1][sub] = col
    return np.clip(img, 0, 255).astype(np.uint8)


def blob_heads(sd, sel_gain=0.5, key_gain=1.5, point_const=8.0, level=(3.4, 3.0, 2.6, 2.2), steep=25.0, pe_scale=2.0, noise=0.1, iou_shift=-0.6):
    """Synthetic decoder weights whose masks are BLOBS AROUND THE PROMPT POINT instead of the noise-like fields random
    weights give -- same architecture, same kernels, same FLOPs; only the values move, IN PLACE:

      * layer 0's image->token attention (transformer.py:186-190) scores every key against the point token on 64 of the
        128 random-Fourier frequencies of the positional encoding (prompt_encoder.py:189-196): q / k projections select the
        matching sin / cos channels, so the logit is sum cos(2 pi g . (x - p)) over the 64 highest frequencies -- a kernel peaked at the prompt -- and the
        other tokens score 0 (their embeddings and the queries' LayerNorm are zero on those channels);
      * its value projection reads ONE channel that only the point token carries (point_embeddings.1), its output
        projection writes the attention mass into ONE key channel that the neck leaves empty: keys[x, 254] grows
        with the kernel;
      * the upscaler's first transposed convolution copies that key channel into channel 0, the second one thresholds it
        (bias -1) beside a constant channel 1, and the hyper-networks' last layers put a positive weight on channel 0 and a
        negative one on channel 1: mask m = a (blob) - b_m, four nested blobs per prompt;
      * the token-side paths (token->image attention, MLP, layer-1 attentions) keep their random weights at a tenth of
        their scale: they still perturb, they no longer drown the construction.
    Prompts 16 px apart give heavily overlapping blobs, so the shipped box NMS (0.65) thins them the way it thins
    detections of a crowd, and stability scores sit in 0.85-0.99: the shipped thresholds are live decision boundaries.
    Returns sd."""
    M, T, P = "mask_decoder.", "mask_decoder.transformer.", "prompt_encoder."
    # The positional encoding's frequencies are doubled (a buffer of the model like any other: the kernel's main lobe becomes
    # person-sized, ~45 px at 1024) and the kernel uses a middle band of 64 of the 128: the lowest ones would put a pedestal as
    # wide as the image under it, the very highest add only ripple
    gm = sd[P + "pe_layer.positional_encoding_gaussian_matrix"]
    gm *= pe_scale
    freq = torch.argsort((gm * gm).sum(0), descending=True)[20:84].tolist()
    S = freq + [128 + f for f in freq]                      # their sin / cos channels
    spare = [f for f in range(128) if f not in freq]
    E, Dk = 128 + spare[0], 128 + spare[1]                  # point-marker channel, blob key channel (outside the kernel's band)
    z = lambda name: sd[name].zero_()
    # neck: the image embedding leaves the kernel channels nearly empty and the blob channel empty
    sd["image_encoder.neck.3.weight"][S] = 0.05
    sd["image_encoder.neck.3.bias"][S] = 0.0
    sd["image_encoder.neck.3.weight"][Dk] = 0.0
    sd["image_encoder.neck.3.bias"][Dk] = 0.0
    for name in (P + "no_mask_embed.weight", P + "not_a_point_embed.weight", M + "iou_token.weight", M + "mask_tokens.weight",
                 P + "point_embeddings.1.weight"):
        sd[name][:, S] = 0.0
        sd[name][:, E] = 0.0
        sd[name][:, Dk] = 0.0
    sd[P + "point_embeddings.1.weight"][0, E] = point_const
    eye = torch.eye(256)
    for i in range(2):
        L = f"{T}layers.{i}."
        if i == 0:      # layer 0's self-attention REPLACES the tokens: make it (nearly) the identity
            for pr, g in (("q_proj", 5.0), ("k_proj", 5.0), ("v_proj", 1.0), ("out_proj", 1.0)):
                sd[L + f"self_attn.{pr}.weight"].copy_(eye * g)
                z(L + f"self_attn.{pr}.bias")
        else:
            sd[L + "self_attn.out_proj.weight"] *= 0.1
            sd[L + "self_attn.out_proj.bias"] *= 0.1
        for nm in ("cross_attn_token_to_image.out_proj", "mlp.lin2"):
            sd[L + nm + ".weight"] *= 0.1
            sd[L + nm + ".bias"] *= 0.1
        for nm in ("cross_attn_token_to_image.out_proj", "mlp.lin2") + (("self_attn.out_proj",) if i else ()):
            sd[L + nm + ".weight"][E] = 0.0                 # nothing but the identity path writes the marker channel
            sd[L + nm + ".bias"][E] = 0.0
        for n in ("norm1", "norm2", "norm3", "norm4"):
            sd[L + n + ".weight"].fill_(1.0)
            z(L + n + ".bias")
        sd[L + "norm3.weight"][S] = 0.0                     # the queries carry nothing on the kernel channels: only query_pe does
        A = L + "cross_attn_image_to_token."
        if i == 0:
            for pr in ("q_proj", "k_proj"):
                w = torch.zeros(128, 256)
                for h in range(8):                          # head h: eight of the frequencies (every 8th by rank), sin then cos
                    for t in range(8):
                        f = freq[t * 8 + h]
                        w[h * 16 + t, f] = sel_gain * 4.0 ** 0.5
                        w[h * 16 + 8 + t, 128 + f] = sel_gain * 4.0 ** 0.5
                sd[A + pr + ".weight"].copy_(w)
                z(A + pr + ".bias")
            w = torch.zeros(128, 256)
            w[:, E] = 1.0 / point_const                     # every value lane = (marker channel) / its nominal size
            sd[A + "v_proj.weight"].copy_(w)
            z(A + "v_proj.bias")
            w = torch.zeros(256, 128)
            w[Dk, :] = key_gain / 128.0 * 8.0               # sum over heads of the point token's attention mass (0 .. 8)
            sd[A + "out_proj.weight"].copy_(w)
            z(A + "out_proj.bias")
            sd[A + "out_proj.bias"][Dk] = -key_gain * 8.0 / 7.0 * 1.0   # the mass of a uniform softmax over 7 tokens
        else:
            sd[A + "out_proj.weight"] *= 0.1
            sd[A + "out_proj.bias"] *= 0.1
    sd[T + "final_attn_token_to_image.out_proj.weight"] *= 0.1
    sd[T + "final_attn_token_to_image.out_proj.bias"] *= 0.1
    U = M + "output_upscaling."
    sd[U + "0.weight"][:, 0] = 0.0                          # ConvTranspose2d weight [in 256, out 64, 2, 2]
    sd[U + "0.weight"][Dk, 0] = 1.0
    sd[U + "0.bias"][0] = 0.0
    sd[U + "1.weight"][0] = 1.0
    sd[U + "1.bias"][0] = 0.0
    sd[U + "3.weight"][:, 0] = 0.0                          # [in 64, out 32, 2, 2]
    sd[U + "3.weight"][0, 0] = 1.0
    sd[U + "3.bias"][0] = -1.0
    sd[U + "3.weight"][:, 1] = 0.0
    sd[U + "3.bias"][1] = 2.0
    for m in range(4):
        H = f"{M}output_hypernetworks_mlps.{m}.layers.2."
        sd[H + "weight"] *= noise
        sd[H + "bias"] *= noise
        sd[H + "weight"][0:2] = 0.0
        sd[H + "bias"][0] = steep
        sd[H + "bias"][1] = -steep * level[m] / 1.954       # GELU(2) = 1.954: mask m = steep * (blob - level[m]) + noise
    # the predicted-IoU head's output bias: about a quarter of the prompts then clear the shipped pred_iou_thresh (0.1) -- a dense grid
    # over a crowd has most of its points on background or on an already-covered person
    sd[M + "iou_prediction_head.layers.2.bias"] += iou_shift
    return sd
