"""Host orchestration of the SAM ViTDet image encoder on the HIP kernels.

Mirrors segment_anything_cs/modeling/image_encoder.py:106-116 (ImageEncoderViT.forward) of the
reference; all arithmetic runs in libcsam_hip.so.  Data layout in HBM: token-major [4096, C] for
the whole encoder (NHWC of the 64x64 grid) -- no permutes, no window-partition copies.
Residual stream fp32, GEMM operands fp16, fp32 MFMA accumulation.
"""
import torch

from . import hip


class EncoderPlan:
    """Device-resident fp16/fp32 operand copies + static activation workspace for one encoder."""

    def __init__(self, sd, prefix, embed_dim, depth, heads, global_idx, device, fused_win=True, ln_fold=True):
        """``fused_win`` False: a head_dim-80 encoder (ViT-H) takes the materialised attention route instead of its two fused
        kernels (parity tests of that route).  ``ln_fold`` False: separate LayerNorm launches instead of csam_gemm_f16_ln."""
        D = embed_dim
        assert D % heads == 0
        self.hd = hd = D // heads
        # head_dim 64 (ViT-B/L) runs the fused flash / window kernels; any other head_dim <= 128 (ViT-H: 80) runs the
        # materialised gather / batched-GEMM / softmax route of csrc/attn_generic.hip
        self.fused_attn = hd == 64
        # round 3: the windowed kernel also exists for head_dim 80 (28 of ViT-H's 32 blocks); its global blocks stay generic
        self.fused_win = hd in (64, 80) and bool(fused_win)
        # round 4: the 2 x depth LayerNorm launches are folded into the GEMMs around them
        self.ln_fold = bool(ln_fold)
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
        self.cap = 0
        self.ws = {}
        self._alloc(1)
        self.graphs = hip.GraphCache()

    def _alloc(self, cap):
        """Static activation workspaces for passes of up to ``cap`` images (token matrices [cap * 4096, .]; rows of image b are
        b * 4096 .. b * 4096 + 4095).  288 GB of HBM: ~0.6 GB per image of capacity for ViT-L."""
        if cap <= self.cap:
            return
        D, heads, hd, device = self.D, self.heads, self.hd, self.device
        e = lambda *s, dt=torch.float16: torch.empty(*s, dtype=dt, device=device)
        T = cap * 4096
        self.ws = dict(
            col=e(T, 768), x=e(T, D, dt=torch.float32), h=e(T, D), qkv=e(T, 3 * D),
            attn=e(T, D), mlp=e(T, 4 * D), traw=e(cap, heads, 4096, 256, dt=torch.float32),
            n0=e(T, 256, dt=torch.float32), n1=e(T, 256),
            col3=e(T, 2304), n2=e(T, 256, dt=torch.float32),
            img=e(cap, 3 * 1024 * 1024, dt=torch.float32), feat=e(cap, 4096, 256, dt=torch.float32),
            x16=e(T, D), st=e(T, D // 128, 2, dt=torch.float32))
        # the position embedding as the patch projection's residual, one copy per image of the pass
        self.pos_b = self.pos.repeat(cap, 1).contiguous()
        # workspaces of the materialised route: only when a block can actually take it (head_dim 80 runs the two
        # head_dim-80 kernels unless fused_win is off; ~1.6 GB for ViT-H otherwise never touched); one image at a time
        if not self.fused_attn and not (self.fused_win and hd == 80):
            gmax = max(heads * 25 * 256, heads * 4096 if self.global_idx else 0)        # group rows, windowed vs global
            smax = max(heads * 25 * 256 * 256, heads * 4096 * 4096 if self.global_idx else 0)
            self.ws.update(gq=e(gmax, 128), gk=e(gmax, 128), gvt=e(gmax * 128), go=e(gmax, 128),
                           gt=e(gmax, 256, dt=torch.float32), gs=e(smax, dt=torch.float32), gp=e(smax))
        self.cap = cap
        if hasattr(self, "graphs"):
            self.graphs.clear()                    # captured graphs hold the old buffers' addresses

    def forward_static(self, img_chw_f32):
        """Graph-replayed forward: the raw frame is copied into a static buffer, features land in a static
        buffer (valid until the next image).  One graph per frame shape (h, w)."""
        _, h, w = img_chw_f32.shape
        buf = self.ws["img"][0, : 3 * h * w].view(3, h, w)
        buf.copy_(img_chw_f32)
        return self.graphs.run((h, w), lambda: self.forward(buf, out=self.ws["feat"][0]))

    def forward(self, img_chw_f32, out=None, skip_im2col=False):
        """img f32 [3,h,w] raw 0..255 (long side <= 1024) -> features f32 [4096,256] token-major.
        ``skip_im2col``: ws['col'] was already filled (API path with a pre-normalised tensor)."""
        self.embed(None if skip_im2col else [img_chw_f32], 1)
        self.run_blocks(0, self.depth, 1)
        if out is None:
            out = torch.empty(4096, 256, dtype=torch.float32, device=self.device)
        return self.neck(out.view(1, 4096, 256), 1)[0]

    # ---- image-batched pass (image_encoder.py:106-116 with B > 1): B frames -- the crops of one image, or the look-ahead
    # frames of a stream -- go through every projection as ONE [B * 4096, D] token matrix.  Rows are independent in every
    # kernel (same K order per output element whatever the tile shape), so each image's features are what a pass of its own
    # produces.  The pass can be cut at block boundaries (embed / run_blocks(lo, hi) / neck): crowdsam.model queues a quarter
    # of the next group's pass beside each frame's tail.
    def load_images(self, imgs):
        """Copy B raw frames f32 [3,h,w] into the static input buffers; returns the views a captured pass reads."""
        self._alloc(len(imgs))
        views = []
        for b, im in enumerate(imgs):
            _, h, w = im.shape
            v = self.ws["img"][b, : 3 * h * w].view(3, h, w)
            v.copy_(im)
            views.append(v)
        return views

    def forward_batch_static(self, imgs, out=None):
        """B raw frames -> features f32 [B,4096,256] in a static buffer (valid until the next pass); one graph per
        (B, frame shapes)."""
        B = len(imgs)
        views = self.load_images(imgs)
        out = self.ws["feat"][:B] if out is None else out
        key = ("batch", B, tuple(tuple(v.shape[1:]) for v in views), out.data_ptr())

        def run():
            self.embed(views, B)
            self.run_blocks(0, self.depth, B)
            return self.neck(out, B)
        return self.graphs.run(key, run)

    def embed(self, imgs, B):
        """Sam.preprocess + patch embedding + position embedding of B frames -> the residual stream ws['x'][:B*4096]."""
        self._alloc(B)
        ws, T = self.ws, B * 4096
        if imgs is not None:
            for b, im in enumerate(imgs):
                hip.sam_im2col(im, ws["col"][b * 4096:(b + 1) * 4096])
        if self.ln_fold:  # every projection that writes the residual stream also leaves its fp16 copy + row statistics
            hip.gemm_f16_ln(ws["col"][:T], self.patch_w, ws["x"][:T], bias=self.patch_b, residual=self.pos_b[:T],
                            out16=ws["x16"][:T], stats_out=ws["st"][:T])
        else:
            hip.gemm_f16(ws["col"][:T], self.patch_w, out=ws["x"][:T], bias=self.patch_b, residual=self.pos_b[:T])

    def run_blocks(self, lo, hi, B):
        """Transformer blocks lo .. hi-1 on the residual stream of B images."""
        D, nH = self.D, self.heads
        T = B * 4096
        ws = dict(self.ws)
        for k in ("x", "h", "qkv", "attn", "mlp", "x16", "st"):
            ws[k] = self.ws[k][:T]
        scale = self.hd ** -0.5
        fold = self.ln_fold
        x, x16, st = ws["x"], ws["x16"], ws["st"]
        per_image = lambda t, b: t[b * 4096:(b + 1) * 4096]
        for b in self.blocks[lo:hi]:
            if fold:
                hip.gemm_f16_ln(x16, b["qkv_wf"], ws["qkv"], bias=b["qkv_bf"], stats_in=st, colsum=b["qkv_cs"], eps=1e-6)
            else:
                hip.layernorm(x, b["ln1_g"], b["ln1_b"], 1e-6, out=ws["h"])
                hip.gemm_f16(ws["h"], b["qkv_w"], out=ws["qkv"], bias=b["qkv_b"])
            if not self.fused_attn and self.fused_win and not b["is_global"]:
                hip.win_attn(ws["qkv"], b["qkv_b"], b["relcat"], ws["attn"], D, nH, scale, n_images=B)
            elif not self.fused_attn and self.fused_win and self.hd == 80:
                for i in range(B):          # head_dim 80 global blocks: one image per launch
                    hip.relpos_raw80(per_image(ws["qkv"], i), b["relcat"], ws["traw"][0], nH)
                    hip.flash_attn80(per_image(ws["qkv"], i), per_image(ws["attn"], i), 4096, nH, scale, D, relpos=ws["traw"][0])
            elif not self.fused_attn:
                for i in range(B):
                    self._attn_generic(b, per_image(ws["qkv"], i), per_image(ws["attn"], i))
            elif b["is_global"]:
                hip.relpos_raw(ws["qkv"], b["relcat"], ws["traw"], nH, n_images=B)
                hip.flash_attn(ws["qkv"], ws["attn"], 4096, nH, scale, D, relpos=ws["traw"], q_prescaled=True, n_images=B)
            else:
                hip.win_attn(ws["qkv"], b["qkv_b"], b["relcat"], ws["attn"], D, nH, scale, n_images=B)
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

    def neck(self, out, B):
        """1x1 conv, LayerNorm2d, 3x3 conv, LayerNorm2d (image_encoder.py:88-104) -> out f32 [B,4096,256]."""
        T = B * 4096
        ws = self.ws
        if self.ln_fold:
            hip.gemm_f16(ws["x16"][:T], self.neck0_w, out=ws["n0"][:T])   # the last block's fp16 copy IS the neck's operand
        else:
            hip.add_cast(ws["x"][:T], out16=ws["h"][:T])
            hip.gemm_f16(ws["h"][:T], self.neck0_w, out=ws["n0"][:T])
        hip.layernorm(ws["n0"][:T], self.neck1_g, self.neck1_b, 1e-6, out=ws["n1"][:T])
        for b in range(B):                      # the 3x3 window stops at each image's border
            hip.im2col3x3(ws["n1"][b * 4096:(b + 1) * 4096], ws["col3"][b * 4096:(b + 1) * 4096], 256)
        hip.gemm_f16(ws["col3"][:T], self.neck2_w, out=ws["n2"][:T])
        hip.layernorm(ws["n2"][:T], self.neck3_g, self.neck3_b, 1e-6, out=out.view(T, 256))
        return out

    def _attn_generic(self, b, qkv, attn):
        """qkv -> attn (one image) for any head_dim <= 128: per (window, head) group, S = (q*scale) k^T and the
        decomposed rel-pos tables by batched GEMM, softmax+bias kernel, O = P v by batched GEMM
        (image_encoder.py:224-240, 243-289, 325-361)."""
        D, nH, hd, ws = self.D, self.heads, self.hd, self.ws
        glob = b["is_global"]
        G, Tp, Tv, side = (nH, 4096, 4096, 64) if glob else (nH * 25, 256, 196, 14)
        scale = hd ** -0.5
        hip.head_gather(qkv, b["qkv_b"], ws["gq"], ws["gk"], ws["gvt"], D, nH, hd, Tp, Tv, not glob, scale)
        hip.gemm_f16_batched(ws["gq"], 128, Tp * 128, b["relcat"], 128, 0, ws["gt"], 256, Tp * 256, Tp, 256, 128, G)
        hip.gemm_f16_batched(ws["gq"], 128, Tp * 128, ws["gk"], 128, Tp * 128, ws["gs"], Tp, Tp * Tp, Tp, Tp, 128, G)
        hip.softmax_relpos(ws["gs"], ws["gt"], ws["gp"], G, Tp, Tv, side, 1.0 / scale)
        hip.gemm_f16_batched(ws["gp"], Tp, Tp * Tp, ws["gvt"], Tp, 128 * Tp, ws["go"], 128, Tp * 128, Tp, 128, Tp, G)
        hip.head_scatter(ws["go"], attn, D, nH, hd, Tp, Tv, not glob)

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
