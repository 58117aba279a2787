"""DINOv2 ViT-L/14 forward on the HIP kernels (external dependency of the reference:
facebookresearch/dinov2, loaded at crowdsam/model.py:33-36 and used only as
``forward_features(x)['x_norm_patchtokens']`` on a 1022x1022 tensor, predictor.py:104-106).

Architecture restated from the published model ("parity unpinned"): patch 14,
cls token, learned 37x37 pos-embed bicubic-resized to 73x73, 24 pre-LN blocks with LayerScale,
GELU MLP, final LayerNorm.  Sequence = 1 + 5329 tokens, rows padded to 5376 in HBM.
"""
import math

import torch
import torch.nn.functional as F

from . import hip

T_DINO = 5330
N_PATCH = 5329
GRID = 73


def interpolate_pos_embed(pos_embed, gh, gw, offset=0.1):
    """Weight preprocessing (once per model, host): bicubic resize of the patch pos-embed.
    ``offset=0.1`` is the Aug-2024 upstream default (scale_factor form); ``None`` selects size=."""
    N = pos_embed.shape[1] - 1
    M = int(math.sqrt(N))
    cls_pe = pos_embed[:, :1].float()
    patch_pe = pos_embed[:, 1:].float().reshape(1, M, M, -1).permute(0, 3, 1, 2)
    if offset is not None:
        patch_pe = F.interpolate(patch_pe, scale_factor=(float(gh + offset) / M, float(gw + offset) / M),
                                 mode="bicubic", antialias=False)
    else:
        patch_pe = F.interpolate(patch_pe, size=(gh, gw), mode="bicubic", antialias=False)
    assert patch_pe.shape[-2:] == (gh, gw)
    return torch.cat([cls_pe, patch_pe.permute(0, 2, 3, 1).reshape(1, gh * gw, -1)], dim=1)


class DinoPlan:
    def __init__(self, sd, device, depth=24, heads=16, pos_offset=0.1, ln_fold=True):
        D = sd["cls_token"].shape[-1]
        assert D // heads == 64 and D % 128 == 0
        self.D, self.depth, self.heads, self.device = D, depth, heads, device
        f16 = lambda t: t.detach().to(device=device, dtype=torch.float16).contiguous()
        f32 = lambda t: t.detach().to(device=device, dtype=torch.float32).contiguous()
        w = sd["patch_embed.proj.weight"].reshape(D, 588).float()
        self.patch_w = f16(F.pad(w, (0, 640 - 588)))          # K padded to 640 with zeros
        self.patch_b = f32(sd["patch_embed.proj.bias"])
        pos = interpolate_pos_embed(sd["pos_embed"].cpu(), GRID, GRID, pos_offset)[0]   # [5330, D]
        self.pos = f32(pos)
        self.cls_row = f32(sd["cls_token"].reshape(1, D).float().cpu() + pos[:1])
        self.blocks = []
        # softmax scale and the base-2 conversion live in the q rows of the qkv projection (csam_flash_attn q_prescaled)
        qfold = torch.ones(3 * D, 1)
        qfold[:D] = ((D // heads) ** -0.5) * hip.FLASH_QMUL     # the same head_dim forward() passes as `scale`
        for i in range(depth):
            B = f"blocks.{i}."
            self.blocks.append(dict(
                ln1_g=f32(sd[B + "norm1.weight"]), ln1_b=f32(sd[B + "norm1.bias"]),
                qkv_w=f16(sd[B + "attn.qkv.weight"].detach().float().cpu() * qfold),
                qkv_b=f32(sd[B + "attn.qkv.bias"].detach().float().cpu() * qfold[:, 0]),
                proj_w=f16(sd[B + "attn.proj.weight"]), proj_b=f32(sd[B + "attn.proj.bias"]),
                ls1=f32(sd[B + "ls1.gamma"]),
                ln2_g=f32(sd[B + "norm2.weight"]), ln2_b=f32(sd[B + "norm2.bias"]),
                fc1_w=f16(sd[B + "mlp.fc1.weight"]), fc1_b=f32(sd[B + "mlp.fc1.bias"]),
                fc2_w=f16(sd[B + "mlp.fc2.weight"]), fc2_b=f32(sd[B + "mlp.fc2.bias"]),
                ls2=f32(sd[B + "ls2.gamma"])))
        # round 4: LayerNorm folded into the qkv / fc1 projections (csam_gemm_f16_ln); block 0's first LayerNorm stays a
        # kernel (its input's cls row comes from a copy, not from a projection)
        self.ln_fold = bool(ln_fold)
        if self.ln_fold:
            for i, bl in enumerate(self.blocks):
                B = f"blocks.{i}."
                bl["qkv_wf"], bl["qkv_bf"], bl["qkv_cs"] = hip.fold_layernorm(
                    f32(sd[B + "attn.qkv.weight"].detach().float().cpu() * qfold), bl["qkv_b"], bl["ln1_g"], bl["ln1_b"])
                bl["fc1_wf"], bl["fc1_bf"], bl["fc1_cs"] = hip.fold_layernorm(f32(sd[B + "mlp.fc1.weight"]), bl["fc1_b"],
                                                                               bl["ln2_g"], bl["ln2_b"])
        self.norm_g, self.norm_b = f32(sd["norm.weight"]), f32(sd["norm.bias"])
        self.cap = 0
        self.ws = {}
        self._alloc(1)
        self.graphs = hip.GraphCache()

    def _alloc(self, cap):
        """Static workspaces for passes of up to ``cap`` images: token matrices [cap * 5330 (+ slack), .], image b = rows
        b * 5330 .. b * 5330 + 5329 (cls token first)."""
        if cap <= self.cap:
            return
        D, device = self.D, self.device
        e = lambda *s, dt=torch.float16: torch.empty(*s, dtype=dt, device=device)
        TP = cap * T_DINO + 46                 # one image: 5376 rows as before
        self.ws = dict(col=e(cap * N_PATCH, 640), x=e(TP, D, dt=torch.float32), h=e(TP, D), qkv=e(TP, 3 * D),
                       attn=e(TP, D), mlp=e(TP, 4 * D), img=e(cap, 3 * 1024 * 1024, dt=torch.float32),
                       x16=e(TP, D), st=e(TP, D // 128, 2, dt=torch.float32))
        self.cap = cap
        if hasattr(self, "graphs"):
            self.graphs.clear()

    def forward_static(self, img_chw_f32, out):
        """Graph-replayed forward into the caller's static ``out`` buffer (one graph per (h, w, out))."""
        _, h, w = img_chw_f32.shape
        buf = self.ws["img"][0, : 3 * h * w].view(3, h, w)
        buf.copy_(img_chw_f32)
        return self.graphs.run((h, w, out.data_ptr()), lambda: self.forward(buf, out=out))

    def forward(self, img_chw_f32, out=None, normalized_1022=False):
        """raw f32 [3,h,w] image (0..255) -> x_norm_patchtokens f16 [5329, D]."""
        self.embed([img_chw_f32], 1, normalized_1022)
        self.run_blocks(0, self.depth, 1)
        if out is None:
            out = torch.empty(N_PATCH, self.D, dtype=torch.float16, device=self.device)
        return self.final_norm([out], 1)[0]

    # ---- image-batched pass: B frames as ONE [B * 5330, D] token matrix through every projection, attention per image
    # (csam_flash_attn_batched); cut at block boundaries like EncoderPlan's
    def load_images(self, imgs):
        self._alloc(len(imgs))
        views = []
        for b, im in enumerate(imgs):
            _, h, w = im.shape
            v = self.ws["img"][b, : 3 * h * w].view(3, h, w)
            v.copy_(im)
            views.append(v)
        return views

    def forward_batch_static(self, imgs, outs):
        """B raw frames -> x_norm_patchtokens f16 into outs[b][:5329] (static buffers of the caller); one graph per
        (B, frame shapes, outs)."""
        B = len(imgs)
        views = self.load_images(imgs)
        key = ("batch", B, tuple(tuple(v.shape[1:]) for v in views), tuple(o.data_ptr() for o in outs))

        def run():
            self.embed(views, B)
            self.run_blocks(0, self.depth, B)
            return self.final_norm(outs, B)
        return self.graphs.run(key, run)

    def embed(self, imgs, B, normalized_1022=False):
        """preprocess + 1024 -> 1022 bilinear + patch embedding + cls token + position embedding of B frames."""
        self._alloc(B)
        ws, x = self.ws, self.ws["x"]
        for b, im in enumerate(imgs):
            col = ws["col"][b * N_PATCH:(b + 1) * N_PATCH]
            hip.dino_im2col(im, col, normalized_1022)
            x[b * T_DINO:b * T_DINO + 1].copy_(self.cls_row)
            hip.gemm_f16(col, self.patch_w, out=x[b * T_DINO + 1:], bias=self.patch_b, residual=self.pos[1:], M=N_PATCH)

    def run_blocks(self, lo, hi, B):
        D, nH = self.D, self.heads
        T = B * T_DINO
        ws = self.ws
        x, x16, st = ws["x"], ws["x16"], ws["st"]
        scale = (D // nH) ** -0.5
        fold = self.ln_fold
        for i in range(lo, hi):
            b = self.blocks[i]
            if fold and i > 0:
                hip.gemm_f16_ln(x16, b["qkv_wf"], ws["qkv"], bias=b["qkv_bf"], M=T, stats_in=st, colsum=b["qkv_cs"], eps=1e-6)
            else:
                hip.layernorm(x, b["ln1_g"], b["ln1_b"], 1e-6, out=ws["h"], M=T)
                hip.gemm_f16(ws["h"], b["qkv_w"], out=ws["qkv"], bias=b["qkv_b"], M=T)
            hip.flash_attn(ws["qkv"], ws["attn"], T_DINO, nH, scale, D, q_prescaled=True, n_images=B)
            if fold:
                hip.gemm_f16_ln(ws["attn"], b["proj_w"], x, bias=b["proj_b"], colscale=b["ls1"], residual=x, M=T, out16=x16,
                                stats_out=st)
                # ONE image: fc1 in two row ranges.  4096 rows are exactly one round of the 256 x 256 four-wave kernel (gemm4w; 16 x 16
                # tiles on 256 CUs, as in the SAM encoder); all 5330 rows would be 336 tiles = 1.3 rounds and fall back to the
                # 128 x 128 kernel (77 us at 580 TFLOP/s alone, 127 us beside the SAM encoder's stream); the remaining 1234 rows
                # are 80 more tiles of the same kernel: -1.4 ms of GEMM time per frame (profiles/r04_dino_fc1_split.txt).
                # B >= 2 images: 672 / 1008 / 1344 tiles fill their last round to >= 87 % -- one launch.
                S = 4096 if B == 1 else T
                hip.gemm_f16_ln(x16[:S], b["fc1_wf"], ws["mlp"][:S], bias=b["fc1_bf"], act=hip.ACT_GELU, M=S, stats_in=st[:S],
                                colsum=b["fc1_cs"], eps=1e-6)
                if T > S:
                    hip.gemm_f16_ln(x16[S:], b["fc1_wf"], ws["mlp"][S:], bias=b["fc1_bf"], act=hip.ACT_GELU, M=T - S,
                                    stats_in=st[S:], colsum=b["fc1_cs"], eps=1e-6)
                hip.gemm_f16_ln(ws["mlp"], b["fc2_w"], x, bias=b["fc2_b"], colscale=b["ls2"], residual=x, M=T, out16=x16,
                                stats_out=st)
            else:
                hip.gemm_f16(ws["attn"], b["proj_w"], out=x, bias=b["proj_b"], colscale=b["ls1"], residual=x, M=T)
                hip.layernorm(x, b["ln2_g"], b["ln2_b"], 1e-6, out=ws["h"], M=T)
                hip.gemm_f16(ws["h"], b["fc1_w"], out=ws["mlp"], bias=b["fc1_b"], act=hip.ACT_GELU, M=T)
                hip.gemm_f16(ws["mlp"], b["fc2_w"], out=x, bias=b["fc2_b"], colscale=b["ls2"], residual=x, M=T)

    def final_norm(self, outs, B):
        """LayerNorm of the patch tokens of image b -> outs[b][:5329] (f16)."""
        x = self.ws["x"]
        for b in range(B):
            hip.layernorm(x[b * T_DINO + 1:], self.norm_g, self.norm_b, 1e-6, out=outs[b], M=N_PATCH)
        return outs

    def flops(self):
        D, T = self.D, T_DINO
        return self.depth * (2 * T * D * 12 * D + 4 * T * T * D) + 2 * 5329 * 588 * D


class DinoV2(torch.nn.Module):
    """Parameter container with the facebookresearch/dinov2 ViT-L/14 checkpoint keys and a HIP
    forward.  Stands where the reference calls ``torch.hub.load(dino_repo, 'dinov2_vitl14', ...)``
    (crowdsam/model.py:33-36): same ``load_state_dict`` / ``.to`` / ``forward_features`` surface."""

    def __init__(self, embed_dim=1024, depth=24, num_heads=16, pos_offset=0.1):
        super().__init__()
        from . import synth
        self.embed_dim, self.depth, self.num_heads, self.pos_offset = embed_dim, depth, num_heads, pos_offset
        for name, shape, _k, _f in synth.dino_param_specs(embed_dim, depth):
            parts = name.split(".")
            node = self
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, torch.nn.Module())
                node = node._modules[p]
            node.register_parameter(parts[-1], torch.nn.Parameter(torch.zeros(shape), requires_grad=False))
        self._plan = None

    def load_state_dict(self, *a, **k):
        self._plan = None
        return super().load_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._plan = None
        return super()._apply(fn, *a, **k)

    def plan(self):
        if self._plan is None or self._plan.ln_fold != bool(getattr(self, "ln_fold", True)):
            dev = next(self.parameters()).device
            if dev.type != "cuda":
                raise RuntimeError("crowdsam_amd runs on MI355X only: move DinoV2 to 'cuda' first")
            self._plan = DinoPlan(self.state_dict(), dev, self.depth, self.num_heads, self.pos_offset,
                                  ln_fold=getattr(self, "ln_fold", True))
        return self._plan

    @torch.no_grad()
    def patch_tokens16(self, raw_chw_f32, out):
        """Fast path: raw image -> x_norm_patchtokens fp16 written into out[:5329]."""
        return self.plan().forward_static(raw_chw_f32, out[:N_PATCH])

    @torch.no_grad()
    def forward_features(self, x):
        """API path (predictor.py:105): x [1,3,1022,1022] normalised -> {'x_norm_patchtokens': [1,5329,D]}."""
        assert x.shape[0] == 1 and tuple(x.shape[1:]) == (3, 1022, 1022)
        t = self.plan().forward(x[0].float().contiguous(), normalized_1022=True)
        return {"x_norm_patchtokens": t.float().unsqueeze(0)}
