"""ORACLE (test infrastructure, NOT product code): fp32 PyTorch-CPU restatement of the reference's
SAM model graph for Crowd-SAM's dense-prompt path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the
product path (crowdsam_amd/, segment_anything_cs/, crowdsam/) never does.

Functional style over a flat state dict that uses the reference's parameter names.  Every
function cites the reference lines it restates (paths relative to /root/reference).  Pinned
against the imported reference in the authoring container by oracle/make_goldens.py and
tests/test_oracle_vs_golden.py (golden vectors under tests/golden/).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------
# small helpers
# ------------------------------------------------------------------------------------------------
def linear(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd.get(name + ".bias"))


def layer_norm(sd, name, x, eps):
    return F.layer_norm(x, (x.shape[-1],), sd[name + ".weight"], sd[name + ".bias"], eps)


def layer_norm_2d(sd, name, x, eps=1e-6):
    """segment_anything_cs/modeling/common.py:38-43 (per-pixel LN over channels of NCHW)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    x = (x - u) / torch.sqrt(s + eps)
    return sd[name + ".weight"][:, None, None] * x + sd[name + ".bias"][:, None, None]


def mlp_relu(sd, name, x, n_layers):
    """mask_decoder.py:204-254 MLP / DropMLP in eval mode (dropout inactive)."""
    for i in range(n_layers):
        x = linear(sd, f"{name}.layers.{i}", x)
        if i < n_layers - 1:
            x = F.relu(x)
    return x


# ------------------------------------------------------------------------------------------------
# image encoder (image_encoder.py)
# ------------------------------------------------------------------------------------------------
def _rel_table(rel_pos, size):
    """image_encoder.py:292-322 get_rel_pos for q_size == k_size == size (no interpolation)."""
    assert rel_pos.shape[0] == 2 * size - 1, "rel-pos interpolation is not needed at 1024^2"
    idx = torch.arange(size)[:, None] - torch.arange(size)[None, :] + (size - 1)
    return rel_pos[idx]  # [q, k, hd]


def encoder_attention(sd, name, x, heads):
    """image_encoder.py:224-240 + add_decomposed_rel_pos :325-361.  x: [B, H, W, C]."""
    B, H, W, C = x.shape
    hd = C // heads
    qkv = linear(sd, name + "qkv", x).reshape(B, H * W, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]  # [B, heads, HW, hd]
    out = torch.empty(B, heads, H * W, hd, dtype=x.dtype)
    Rh = _rel_table(sd[name + "rel_pos_h"], H)
    Rw = _rel_table(sd[name + "rel_pos_w"], W)
    scale = hd ** -0.5
    for b in range(B):          # loop to bound the 4096^2 score tensor to one head at a time
        for h in range(heads):
            qq = q[b, h]
            attn = (qq * scale) @ k[b, h].transpose(0, 1)
            rq = qq.reshape(H, W, hd)
            rel_h = torch.einsum("hwc,hkc->hwk", rq, Rh)   # unscaled q (trap: :349-359)
            rel_w = torch.einsum("hwc,wkc->hwk", rq, Rw)
            attn = (attn.view(H, W, H, W) + rel_h[:, :, :, None] + rel_w[:, :, None, :]).view(H * W, H * W)
            out[b, h] = attn.softmax(-1) @ v[b, h]
    x = out.permute(0, 2, 1, 3).reshape(B, H, W, C)
    return linear(sd, name + "proj", x)


def encoder_block(sd, name, x, heads, window):
    """image_encoder.py:166-182 Block.forward with window_partition/unpartition :243-289."""
    shortcut = x
    x = layer_norm(sd, name + "norm1", x, 1e-6)
    if window > 0:
        B, H, W, C = x.shape
        ph = (window - H % window) % window
        pw = (window - W % window) % window
        x = F.pad(x, (0, 0, 0, pw, 0, ph))      # zero pad AFTER the LayerNorm (trap 4)
        Hp, Wp = H + ph, W + pw
        x = x.view(B, Hp // window, window, Wp // window, window, C).permute(0, 1, 3, 2, 4, 5)
        x = x.reshape(-1, window, window, C)
        x = encoder_attention(sd, name + "attn.", x, heads)
        x = x.view(B, Hp // window, Wp // window, window, window, C).permute(0, 1, 3, 2, 4, 5)
        x = x.reshape(B, Hp, Wp, C)[:, :H, :W, :]
    else:
        x = encoder_attention(sd, name + "attn.", x, heads)
    x = shortcut + x
    h = layer_norm(sd, name + "norm2", x, 1e-6)
    h = linear(sd, name + "mlp.lin2", F.gelu(linear(sd, name + "mlp.lin1", h)))
    return x + h


def image_encoder(sd, x, depth, heads, global_idx, window=14, prefix="image_encoder."):
    """image_encoder.py:106-116 ImageEncoderViT.forward.  x: [B,3,1024,1024] -> [B,256,64,64]."""
    x = F.conv2d(x, sd[prefix + "patch_embed.proj.weight"], sd[prefix + "patch_embed.proj.bias"],
                 stride=16).permute(0, 2, 3, 1)
    x = x + sd[prefix + "pos_embed"]
    for i in range(depth):
        x = encoder_block(sd, f"{prefix}blocks.{i}.", x, heads, 0 if i in global_idx else window)
    x = x.permute(0, 3, 1, 2)
    x = F.conv2d(x, sd[prefix + "neck.0.weight"])
    x = layer_norm_2d(sd, prefix + "neck.1", x)
    x = F.conv2d(x, sd[prefix + "neck.2.weight"], padding=1)
    x = layer_norm_2d(sd, prefix + "neck.3", x)
    return x


def preprocess(x, pixel_mean=(123.675, 116.28, 103.53), pixel_std=(58.395, 57.12, 57.375), img_size=1024):
    """sam.py:163-173 Sam.preprocess: normalise, zero-pad bottom/right to img_size."""
    mean = torch.tensor(pixel_mean).view(-1, 1, 1)
    std = torch.tensor(pixel_std).view(-1, 1, 1)
    x = (x - mean) / std
    h, w = x.shape[-2:]
    return F.pad(x, (0, img_size - w, 0, img_size - h))


# ------------------------------------------------------------------------------------------------
# prompt encoder (prompt_encoder.py)
# ------------------------------------------------------------------------------------------------
def _pe_encoding(sd, coords01):
    """prompt_encoder.py:189-196: coords in [0,1] -> random-Fourier features [.., 256]."""
    G = sd["prompt_encoder.pe_layer.positional_encoding_gaussian_matrix"]
    c = 2 * coords01 - 1
    c = c @ G
    c = 2 * np.pi * c
    return torch.cat([torch.sin(c), torch.cos(c)], dim=-1)


def dense_pe(sd, size=64):
    """prompt_encoder.py:64-73,198-209 get_dense_pe -> [1,256,size,size]."""
    grid = torch.ones((size, size), dtype=torch.float32)
    y = (grid.cumsum(dim=0) - 0.5) / size
    x = (grid.cumsum(dim=1) - 0.5) / size
    pe = _pe_encoding(sd, torch.stack([x, y], dim=-1))
    return pe.permute(2, 0, 1).unsqueeze(0)


def embed_points(sd, coords, labels, img_size=1024):
    """prompt_encoder.py:75-93 _embed_points(pad=True) + :211-218 forward_with_coords.

    coords [B,N,2] (x,y) in the 1024-frame, float64 or float32 (trap 6: the +0.5 and /1024
    happen in the input dtype, the cast to f32 comes last); labels [B,N] in {1,0,-1}.
    Returns sparse embeddings [B,N+1,256].
    """
    pts = coords + 0.5
    pad_pt = torch.zeros((pts.shape[0], 1, 2), dtype=pts.dtype)
    pad_lb = -torch.ones((labels.shape[0], 1), dtype=labels.dtype)
    pts = torch.cat([pts, pad_pt], dim=1)
    lbs = torch.cat([labels, pad_lb], dim=1)
    c = pts.clone()
    c[:, :, 0] = c[:, :, 0] / img_size
    c[:, :, 1] = c[:, :, 1] / img_size
    emb = _pe_encoding(sd, c.to(torch.float))
    emb[lbs == -1] = 0.0
    emb[lbs == -1] += sd["prompt_encoder.not_a_point_embed.weight"]
    emb[lbs == 0] += sd["prompt_encoder.point_embeddings.0.weight"]
    emb[lbs == 1] += sd["prompt_encoder.point_embeddings.1.weight"]
    return emb


def embed_boxes(sd, boxes, img_size=1024):
    """prompt_encoder.py:95-102 _embed_boxes + :211-218 forward_with_coords: boxes [B,4] XYXY in the 1024 frame -> the two
    corner embeddings [B,2,256] (with points == None the sparse prompt is exactly these: no padding point, :152-163)."""
    c = (boxes + 0.5).reshape(-1, 2, 2).clone()
    c[:, :, 0] = c[:, :, 0] / img_size
    c[:, :, 1] = c[:, :, 1] / img_size
    emb = _pe_encoding(sd, c.to(torch.float))
    emb[:, 0, :] += sd["prompt_encoder.point_embeddings.2.weight"][0]
    emb[:, 1, :] += sd["prompt_encoder.point_embeddings.3.weight"][0]
    return emb


# ------------------------------------------------------------------------------------------------
# two-way transformer + mask decoder (transformer.py, mask_decoder.py)
# ------------------------------------------------------------------------------------------------
def _dec_attention(sd, name, q, k, v, heads=8):
    """transformer.py:228-254 Attention.forward (attn_sim is None on this path)."""
    q = linear(sd, name + "q_proj", q)
    k = linear(sd, name + "k_proj", k)
    v = linear(sd, name + "v_proj", v)

    def split(t):
        b, n, c = t.shape
        return t.reshape(b, n, heads, c // heads).transpose(1, 2)

    q, k, v = split(q), split(k), split(v)
    ch = q.shape[-1]
    attn = (q @ k.permute(0, 1, 3, 2)) / math.sqrt(ch)
    attn = torch.softmax(attn, dim=-1)
    out = attn @ v
    b, h, n, c = out.shape
    out = out.transpose(1, 2).reshape(b, n, h * c)
    return linear(sd, name + "out_proj", out)


def two_way_transformer(sd, src, pos, tokens, prefix="mask_decoder.transformer."):
    """transformer.py:62-114 + TwoWayAttentionBlock.forward :160-192.

    src, pos: [B,256,64,64]; tokens [B,7,256].  Returns (queries [B,7,256], keys [B,4096,256]).
    LayerNorm eps is the nn.LayerNorm default 1e-5 here.
    """
    keys = src.flatten(2).permute(0, 2, 1)
    key_pe = pos.flatten(2).permute(0, 2, 1)
    queries = tokens
    query_pe = tokens
    for i in range(2):
        L = f"{prefix}layers.{i}."
        if i == 0:   # skip_first_layer_pe: replaces, no residual (:164-165)
            queries = _dec_attention(sd, L + "self_attn.", queries, queries, queries)
        else:
            q = queries + query_pe
            queries = queries + _dec_attention(sd, L + "self_attn.", q, q, queries)
        queries = layer_norm(sd, L + "norm1", queries, 1e-5)
        q = queries + query_pe
        k = keys + key_pe
        queries = queries + _dec_attention(sd, L + "cross_attn_token_to_image.", q, k, keys)
        queries = layer_norm(sd, L + "norm2", queries, 1e-5)
        m = linear(sd, L + "mlp.lin2", F.relu(linear(sd, L + "mlp.lin1", queries)))
        queries = layer_norm(sd, L + "norm3", queries + m, 1e-5)
        q = queries + query_pe
        k = keys + key_pe
        keys = keys + _dec_attention(sd, L + "cross_attn_image_to_token.", k, q, queries)
        keys = layer_norm(sd, L + "norm4", keys, 1e-5)
    q = queries + query_pe
    k = keys + key_pe
    queries = queries + _dec_attention(sd, prefix + "final_attn_token_to_image.", q, k, keys)
    queries = layer_norm(sd, prefix + "norm_final_attn", queries, 1e-5)
    return queries, keys


def upscale_hyper(sd, src, hyper, prefix="mask_decoder."):
    """mask_decoder.py:172-181: ConvT(256->64,k2,s2) -> LayerNorm2d -> GELU(erf) -> ConvT(64->32,k2,s2) -> GELU(erf), then
    masks[b,l] = sum_c hyper[b,l,c] * up[b,c].  src [B,256,64,64], hyper [B,4,32] -> [B,4,256,256]."""
    U = prefix + "output_upscaling."
    up = F.conv_transpose2d(src, sd[U + "0.weight"], sd[U + "0.bias"], stride=2)
    up = F.gelu(layer_norm_2d(sd, U + "1", up))
    up = F.gelu(F.conv_transpose2d(up, sd[U + "3.weight"], sd[U + "3.bias"], stride=2))
    b, c, h, w = up.shape
    return (hyper @ up.view(b, c, h * w)).view(b, -1, h, w)


def mask_decoder(sd, image_embeddings, image_pe, sparse, dino_feats, prefix="mask_decoder."):
    """mask_decoder.py:138-199 predict_masks (multimask_output=True keeps all 4, :129-135).

    image_embeddings [1,256,64,64]; image_pe [1,256,64,64]; sparse [B,2,256];
    dino_feats [1,73,73,1024].  Returns masks [B,4,256,256], iou [B,4], cls [B,4,n_class].
    """
    B = sparse.shape[0]
    out_tok = torch.cat([sd[prefix + "iou_token.weight"], sd[prefix + "mask_tokens.weight"]], dim=0)
    tokens = torch.cat([out_tok.unsqueeze(0).expand(B, -1, -1), sparse], dim=1)
    dense = sd["prompt_encoder.no_mask_embed.weight"].reshape(1, -1, 1, 1)   # prompt_encoder.py:168-170
    src = torch.repeat_interleave(image_embeddings, B, dim=0) + dense
    pos = torch.repeat_interleave(image_pe, B, dim=0)
    b, c, h, w = src.shape
    hs, keys = two_way_transformer(sd, src, pos, tokens, prefix + "transformer.")
    iou_tok = hs[:, 0, :]
    mask_toks = hs[:, 1:5, :]
    src = keys.transpose(1, 2).view(b, c, h, w)
    hyper = torch.stack([mlp_relu(sd, f"{prefix}output_hypernetworks_mlps.{i}", mask_toks[:, i, :], 3)
                         for i in range(4)], dim=1)
    masks = upscale_hyper(sd, src, hyper, prefix)
    iou = mlp_relu(sd, prefix + "iou_prediction_head", iou_tok, 3)
    # PWD-Net heads (:186-198)
    d = linear(sd, prefix + "dino_proj", dino_feats)
    d = F.interpolate(d.permute(0, 3, 1, 2), (256, 256), mode="bilinear")
    wgt = masks.flatten(2).softmax(-1).reshape(b, 4, 256, 256)
    pooled = torch.einsum("blhw,chw->blc", wgt, d[0])
    cls = mlp_relu(sd, prefix + "point_classifier", pooled, 2)
    fused = torch.cat([iou_tok.unsqueeze(1).repeat(1, 4, 1), mask_toks], dim=-1)
    res = mlp_relu(sd, prefix + "parallel_iou_head", fused, 3).squeeze(2)
    return masks, iou + res, cls


def postprocess_masks(masks, input_size, original_size, img_size=1024):
    """sam.py:132-161."""
    masks = F.interpolate(masks, (img_size, img_size), mode="bilinear", align_corners=False)
    masks = masks[..., : input_size[0], : input_size[1]]
    return F.interpolate(masks, original_size, mode="bilinear", align_corners=False)


def predict_fg_map(sd, dino_feats, prefix="mask_decoder."):
    """predictor.py:113-121: dino_proj -> point_classifier -> [1,n_class,256,256] logits."""
    d = linear(sd, prefix + "dino_proj", dino_feats)
    cls = mlp_relu(sd, prefix + "point_classifier", d, 2).permute(0, 3, 1, 2)
    return F.interpolate(cls, (256, 256), mode="bilinear")


# ------------------------------------------------------------------------------------------------
# DINOv2 ViT-L/14 (external dependency of the reference: facebookresearch/dinov2, un-vendored,
# unpinned -- "parity unpinned").  Restated from its published
# architecture: patch 14, cls token, learned pos-embed (37x37 @518) bicubic-resized to the token
# grid, 24 pre-LN blocks with LayerScale, GELU MLP, final LN; x_norm_patchtokens = norm(x)[:,1:].
# ------------------------------------------------------------------------------------------------
def dino_interp_pos_embed(pos_embed, gh, gw, offset=0.1):
    """dinov2/models/vision_transformer.py interpolate_pos_encoding (Aug-2024 upstream default:
    scale_factor with interpolate_offset=0.1; offset=None selects the size= form)."""
    N = pos_embed.shape[1] - 1
    M = int(math.sqrt(N))
    cls_pe = pos_embed[:, :1]
    patch_pe = pos_embed[:, 1:].reshape(1, M, M, -1).permute(0, 3, 1, 2)
    if offset is not None:
        sx = float(gw + offset) / M
        sy = float(gh + offset) / M
        patch_pe = F.interpolate(patch_pe, scale_factor=(sy, sx), mode="bicubic", antialias=False)
    else:
        patch_pe = F.interpolate(patch_pe, size=(gh, gw), mode="bicubic", antialias=False)
    assert patch_pe.shape[-2:] == (gh, gw)
    patch_pe = patch_pe.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)
    return torch.cat([cls_pe, patch_pe], dim=1)


def dinov2_forward(sd, x, depth=24, heads=16, patch=14, pos_offset=0.1):
    """forward_features(x)['x_norm_patchtokens']: x [1,3,H,W] -> [1, (H/14)*(W/14), D]."""
    B, _, H, W = x.shape
    gh, gw = H // patch, W // patch
    t = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], stride=patch)
    t = t.flatten(2).transpose(1, 2)
    t = torch.cat([sd["cls_token"].expand(B, -1, -1), t], dim=1)
    t = t + dino_interp_pos_embed(sd["pos_embed"], gh, gw, pos_offset)
    D = t.shape[-1]
    hd = D // heads
    for i in range(depth):
        p = f"blocks.{i}."
        h = layer_norm(sd, p + "norm1", t, 1e-6)
        qkv = linear(sd, p + "attn.qkv", h).reshape(B, -1, 3, heads, hd).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]
        a = ((q * hd ** -0.5) @ k.transpose(-2, -1)).softmax(-1) @ v
        a = a.transpose(1, 2).reshape(B, -1, D)
        t = t + sd[p + "ls1.gamma"] * linear(sd, p + "attn.proj", a)
        h = layer_norm(sd, p + "norm2", t, 1e-6)
        h = linear(sd, p + "mlp.fc2", F.gelu(linear(sd, p + "mlp.fc1", h)))
        t = t + sd[p + "ls2.gamma"] * h
    t = layer_norm(sd, "norm", t, 1e-6)
    return t[:, 1:]
