"""Host orchestration of the SAM ViTDet image encoder on the HIP kernels.

Mirrors segment_anything_cs/modeling/image_encoder.py:106-116 (ImageEncoderViT.forward) of the
reference; all arithmetic runs in libcsam_hip.so.  Data layout in HBM: token-major [4096, C] for
the whole encoder (NHWC of the 64x64 grid) -- no permutes, no window-partition copies.
Residual stream fp32, GEMM operands fp16, fp32 MFMA accumulation.
"""
import os

import torch

from . import hip


class EncoderPlan:
    """Device-resident fp16/fp32 operand copies + static activation workspace for one encoder."""

    def __init__(self, sd, prefix, embed_dim, depth, heads, global_idx, device):
        D = embed_dim
        assert D % heads == 0
        self.hd = hd = D // heads
        # head_dim 64 (ViT-B/L) runs the fused flash / window kernels; any other head_dim <= 128 (ViT-H: 80) runs the
        # materialised gather / batched-GEMM / softmax route of csrc/attn_generic.hip
        self.fused_attn = hd == 64
        # round 3: the windowed kernel also exists for head_dim 80 (28 of ViT-H's 32 blocks); its global blocks stay generic
        self.fused_win = hd in (64, 80) and os.environ.get("CSAM_WIN_HD80", "1") != "0"
        # round 4: the 2 x depth LayerNorm launches are folded into the GEMMs around them (CSAM_LN_FOLD=0: separate kernels)
        self.ln_fold = os.environ.get("CSAM_LN_FOLD", "1") != "0"
        assert hd <= 128 and hd % 8 == 0, "head_dim must be a multiple of 8, at most 128"
        assert D % 128 == 0, "GEMM tiles need embed_dim % 128 == 0"
        self.D, self.depth, self.heads, self.global_idx = D, depth, heads, tuple(global_idx)
        self.device = device
        f16 = lambda t: t.detach().to(device=device, dtype=torch.float16).contiguous()
        f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
        P = prefix
        self.patch_w = f16(sd[P + "patch_embed.proj.weight"].reshape(D, 768))
        self.patch_b = f32(sd[P + "patch_embed.proj.bias"])
        self.pos = f32(sd[P + "pos_embed"].reshape(4096, D))
        self.blocks = []
        for i in range(depth):
            B = f"{P}blocks.{i}."
            qkv_w, qkv_b = sd[B + "attn.qkv.weight"].detach().float().cpu(), sd[B + "attn.qkv.bias"].detach().float().cpu()
            if self.fused_attn and i in self.global_idx:
                # global blocks run csam_flash_attn with q_prescaled: softmax scale and the base-2 conversion live in
                # the q rows of the projection (the rel-pos tables, computed from that q, are rescaled in the kernel)
                qkv_w, qkv_b = qkv_w.clone(), qkv_b.clone()
                qkv_w[:D] *= (hd ** -0.5) * hip.FLASH_QMUL
                qkv_b[:D] *= (hd ** -0.5) * hip.FLASH_QMUL
            self.blocks.append(dict(
                ln1_g=f32(sd[B + "norm1.weight"]), ln1_b=f32(sd[B + "norm1.bias"]),
                qkv_w=f16(qkv_w), qkv_b=f32(qkv_b),
                rel_h=f32(sd[B + "attn.rel_pos_h"]), rel_w=f32(sd[B + "attn.rel_pos_w"]),
                proj_w=f16(sd[B + "attn.proj.weight"]), proj_b=f32(sd[B + "attn.proj.bias"]),
                ln2_g=f32(sd[B + "norm2.weight"]), ln2_b=f32(sd[B + "norm2.bias"]),
                lin1_w=f16(sd[B + "mlp.lin1.weight"]), lin1_b=f32(sd[B + "mlp.lin1.bias"]),
                lin2_w=f16(sd[B + "mlp.lin2.weight"]), lin2_b=f32(sd[B + "mlp.lin2.bias"]),
                is_global=i in self.global_idx))
            bl = self.blocks[-1]
            if self.ln_fold:
                # LayerNorm folded into the projection that consumes it (hip.fold_layernorm / csam_gemm_f16_ln); the
                # windowed kernel keeps the ORIGINAL qkv bias for its pad tokens (their LayerNorm output is 0, not beta)
                bl["qkv_wf"], bl["qkv_bf"], bl["qkv_cs"] = hip.fold_layernorm(f32(qkv_w), bl["qkv_b"], bl["ln1_g"], bl["ln1_b"])
                bl["lin1_wf"], bl["lin1_bf"], bl["lin1_cs"] = hip.fold_layernorm(f32(sd[B + "mlp.lin1.weight"]), bl["lin1_b"],
                                                                                 bl["ln2_g"], bl["ln2_b"])
            L = 127 if i in self.global_idx else 27
            assert tuple(sd[B + "attn.rel_pos_h"].shape) == (L, hd), "rel-pos table must be (2S-1, head_dim) at 1024^2"
            if self.fused_attn:
                bl["relcat"] = (hip.relcat_global if bl["is_global"] else hip.relcat_window)(bl["rel_h"], bl["rel_w"])
            elif self.fused_win and not bl["is_global"]:
                bl["relcat"] = hip.relcat_window(bl["rel_h"], bl["rel_w"])
            elif self.fused_win and hd == 80:
                bl["relcat"] = hip.relcat_global80(bl["rel_h"], bl["rel_w"])
            else:   # [256, 128]: rows 0..L-1 = rel_h, rows 128..128+L-1 = rel_w, head dims zero-padded to 128
                rc = torch.zeros(256, 128, dtype=torch.float16, device=device)
                rc[:L, :hd] = bl["rel_h"].to(torch.float16)
                rc[128:128 + L, :hd] = bl["rel_w"].to(torch.float16)
                bl["relcat"] = rc
        self.neck0_w = f16(sd[P + "neck.0.weight"].reshape(256, D))
        self.neck1_g, self.neck1_b = f32(sd[P + "neck.1.weight"]), f32(sd[P + "neck.1.bias"])
        # [co, ci, ky, kx] -> [co, (ky*3+kx)*256 + ci] to match csam_im2col3x3
        self.neck2_w = f16(sd[P + "neck.2.weight"].permute(0, 2, 3, 1).reshape(256, 2304))
        self.neck3_g, self.neck3_b = f32(sd[P + "neck.3.weight"]), f32(sd[P + "neck.3.bias"])
        e = lambda *s, dt=torch.float16: torch.empty(*s, dtype=dt, device=device)
        self.ws = dict(
            col=e(4096, 768), x=e(4096, D, dt=torch.float32), h=e(4096, D), qkv=e(4096, 3 * D),
            attn=e(4096, D), mlp=e(4096, 4 * D), traw=e(heads, 4096, 256, dt=torch.float32),
            n0=e(4096, 256, dt=torch.float32), n1=e(4096, 256),
            col3=e(4096, 2304), n2=e(4096, 256, dt=torch.float32),
            img=e(3 * 1024 * 1024, dt=torch.float32), feat=e(4096, 256, dt=torch.float32),
            x16=e(4096, D), st=e(4096, D // 128, 2, dt=torch.float32))
        # workspaces of the materialised route: only when a block can actually take it (head_dim 80 runs the two
        # head_dim-80 kernels unless CSAM_WIN_HD80=0; ~1.6 GB for ViT-H otherwise never touched)
        if not self.fused_attn and not (self.fused_win and hd == 80):
            gmax = max(heads * 25 * 256, heads * 4096 if self.global_idx else 0)        # group rows, windowed vs global
            smax = max(heads * 25 * 256 * 256, heads * 4096 * 4096 if self.global_idx else 0)
            self.ws.update(gq=e(gmax, 128), gk=e(gmax, 128), gvt=e(gmax * 128), go=e(gmax, 128),
                           gt=e(gmax, 256, dt=torch.float32), gs=e(smax, dt=torch.float32), gp=e(smax))
        self.graphs = hip.GraphCache()

    def forward_static(self, img_chw_f32):
        """Graph-replayed forward: the raw frame is copied into a static buffer, features land in a static
        buffer (valid until the next image).  One graph per frame shape (h, w)."""
        _, h, w = img_chw_f32.shape
        buf = self.ws["img"][: 3 * h * w].view(3, h, w)
        buf.copy_(img_chw_f32)
        return self.graphs.run((h, w), lambda: self.forward(buf, out=self.ws["feat"]))

    def forward(self, img_chw_f32, out=None, skip_im2col=False):
        """img f32 [3,h,w] raw 0..255 (long side <= 1024) -> features f32 [4096,256] token-major.
        ``skip_im2col``: ws['col'] was already filled (API path with a pre-normalised tensor)."""
        D, nH = self.D, self.heads
        ws = self.ws
        scale = self.hd ** -0.5
        if not skip_im2col:
            hip.sam_im2col(img_chw_f32, ws["col"])
        fold = self.ln_fold
        x16, st = ws["x16"], ws["st"]
        if fold:     # every projection that writes the residual stream also leaves its fp16 copy + row statistics
            x = hip.gemm_f16_ln(ws["col"], self.patch_w, ws["x"], bias=self.patch_b, residual=self.pos, out16=x16, stats_out=st)
        else:
            x = hip.gemm_f16(ws["col"], self.patch_w, out=ws["x"], bias=self.patch_b, residual=self.pos)
        for b in self.blocks:
            if fold:
                hip.gemm_f16_ln(x16, b["qkv_wf"], ws["qkv"], bias=b["qkv_bf"], stats_in=st, colsum=b["qkv_cs"], eps=1e-6)
            else:
                hip.layernorm(x, b["ln1_g"], b["ln1_b"], 1e-6, out=ws["h"])
                hip.gemm_f16(ws["h"], b["qkv_w"], out=ws["qkv"], bias=b["qkv_b"])
            if not self.fused_attn and self.fused_win and not b["is_global"]:
                hip.win_attn(ws["qkv"], b["qkv_b"], b["relcat"], ws["attn"], D, nH, scale)
            elif not self.fused_attn and self.fused_win and self.hd == 80:
                hip.relpos_raw80(ws["qkv"], b["relcat"], ws["traw"], nH)
                hip.flash_attn80(ws["qkv"], ws["attn"], 4096, nH, scale, D, relpos=ws["traw"])
            elif not self.fused_attn:
                self._attn_generic(b)
            elif b["is_global"]:
                hip.relpos_raw(ws["qkv"], b["relcat"], ws["traw"], nH)
                hip.flash_attn(ws["qkv"], ws["attn"], 4096, nH, scale, D, relpos=ws["traw"], q_prescaled=True)
            else:
                hip.win_attn(ws["qkv"], b["qkv_b"], b["relcat"], ws["attn"], D, nH, scale)
            if fold:
                hip.gemm_f16_ln(ws["attn"], b["proj_w"], x, bias=b["proj_b"], residual=x, out16=x16, stats_out=st)
                hip.gemm_f16_ln(x16, b["lin1_wf"], ws["mlp"], bias=b["lin1_bf"], act=hip.ACT_GELU, stats_in=st,
                                colsum=b["lin1_cs"], eps=1e-6)
                hip.gemm_f16_ln(ws["mlp"], b["lin2_w"], x, bias=b["lin2_b"], residual=x, out16=x16, stats_out=st)
            else:
                hip.gemm_f16(ws["attn"], b["proj_w"], out=x, bias=b["proj_b"], residual=x)
                hip.layernorm(x, b["ln2_g"], b["ln2_b"], 1e-6, out=ws["h"])
                hip.gemm_f16(ws["h"], b["lin1_w"], out=ws["mlp"], bias=b["lin1_b"], act=hip.ACT_GELU)
                hip.gemm_f16(ws["mlp"], b["lin2_w"], out=x, bias=b["lin2_b"], residual=x)
        if fold:
            hip.gemm_f16(x16, self.neck0_w, out=ws["n0"])          # the last block's fp16 copy IS the neck's operand
        else:
            hip.add_cast(x, out16=ws["h"])
            hip.gemm_f16(ws["h"], self.neck0_w, out=ws["n0"])
        hip.layernorm(ws["n0"], self.neck1_g, self.neck1_b, 1e-6, out=ws["n1"])
        hip.im2col3x3(ws["n1"], ws["col3"], 256)
        hip.gemm_f16(ws["col3"], self.neck2_w, out=ws["n2"])
        if out is None:
            out = torch.empty(4096, 256, dtype=torch.float32, device=self.device)
        hip.layernorm(ws["n2"], self.neck3_g, self.neck3_b, 1e-6, out=out)
        return out

    def _attn_generic(self, b):
        """ws['qkv'] -> ws['attn'] for any head_dim <= 128: per (window, head) group, S = (q*scale) k^T and the
        decomposed rel-pos tables by batched GEMM, softmax+bias kernel, O = P v by batched GEMM
        (image_encoder.py:224-240, 243-289, 325-361)."""
        D, nH, hd, ws = self.D, self.heads, self.hd, self.ws
        glob = b["is_global"]
        G, Tp, Tv, side = (nH, 4096, 4096, 64) if glob else (nH * 25, 256, 196, 14)
        scale = hd ** -0.5
        hip.head_gather(ws["qkv"], b["qkv_b"], ws["gq"], ws["gk"], ws["gvt"], D, nH, hd, Tp, Tv, not glob, scale)
        hip.gemm_f16_batched(ws["gq"], 128, Tp * 128, b["relcat"], 128, 0, ws["gt"], 256, Tp * 256, Tp, 256, 128, G)
        hip.gemm_f16_batched(ws["gq"], 128, Tp * 128, ws["gk"], 128, Tp * 128, ws["gs"], Tp, Tp * Tp, Tp, Tp, 128, G)
        hip.softmax_relpos(ws["gs"], ws["gt"], ws["gp"], G, Tp, Tv, side, 1.0 / scale)
        hip.gemm_f16_batched(ws["gp"], Tp, Tp * Tp, ws["gvt"], Tp, 128 * Tp, ws["go"], 128, Tp * 128, Tp, 128, Tp, G)
        hip.head_scatter(ws["go"], ws["attn"], D, nH, hd, Tp, Tv, not glob)

    def flops(self):
        """Required FLOPs per image (pad tokens of edge windows are not multiplied: their q/k/v are
        the bias, so QKV/proj run on the 4096 real tokens only)."""
        D, nH = self.D, self.heads
        f = 2 * 4096 * 768 * D
        for b in self.blocks:
            f += 2 * 4096 * D * (3 * D + D + 8 * D)
            if b["is_global"]:
                f += 4 * 4096 * 4096 * self.hd * nH + 2 * 4096 * 128 * self.hd * nH
            else:
                f += 25 * nH * (4 * 196 * 196 * self.hd + 2 * 196 * 28 * self.hd)
        f += 2 * 4096 * D * 256 + 2 * 4096 * 2304 * 256
        return f
