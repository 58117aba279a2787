"""``Sam`` and its three sub-models as parameter containers with HIP-backed forwards.

The modules register parameters/buffers under exactly the reference's state-dict keys
(reference: segment_anything_cs/modeling/{sam,image_encoder,prompt_encoder,mask_decoder}.py), so
``load_state_dict`` / ``state_dict`` / ``.to(device)`` behave as for the reference checkpoints
(sam_vit_l_0b3195.pth, adapter 10_shot.pth with its unused 5th hyper-MLP, SURVEY.md trap 5).
No PyTorch operator runs in a forward: the compute lives in crowdsam_amd (libcsam_hip.so) and is
(re)planned lazily whenever parameters change or move.
"""
import torch
import torch.nn as nn

from crowdsam_amd import hip, synth
from crowdsam_amd.decoder import DecoderPlan
from crowdsam_amd.encoder import EncoderPlan


class _ParamTree(nn.Module):
    """Nested module whose parameters mirror a list of dotted names (state-dict compatible)."""

    def __init__(self, specs, buffers=()):
        super().__init__()
        for name, shape, _kind, _fan in specs:
            parts = name.split(".")
            node = self
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, _ParamTree([]))
                node = node._modules[p]
            if name in buffers:
                node.register_buffer(parts[-1], torch.zeros(shape))
            else:
                node.register_parameter(parts[-1], nn.Parameter(torch.zeros(shape), requires_grad=False))


class _Planned(_ParamTree):
    """Invalidate the device plan whenever weights are replaced or moved."""

    def __init__(self, specs, buffers=()):
        super().__init__(specs, buffers)
        self._plan = None

    def load_state_dict(self, *a, **k):
        self._plan = None
        return super().load_state_dict(*a, **k)

    def _load_from_state_dict(self, *a, **k):
        # Sam.load_state_dict recurses through the children with _load_from_state_dict, never through their
        # load_state_dict override: invalidate here too, or a resident EncoderPlan keeps the old fp16 weight copies
        self._plan = None
        return super()._load_from_state_dict(*a, **k)

    def _apply(self, fn, *a, **k):
        self._plan = None
        return super()._apply(fn, *a, **k)

    @property
    def device(self):
        return next(self.parameters()).device

    def _require_gpu(self):
        if self.device.type != "cuda":
            raise RuntimeError("crowdsam_amd runs on MI355X only: move the model to 'cuda' first "
                               "(there is no CPU fallback in the product path)")


def _sub(specs, prefix):
    n = len(prefix)
    return [(name[n:], shape, kind, fan) for name, shape, kind, fan in specs if name.startswith(prefix)]


class ImageEncoderViT(_Planned):
    """SAM ViTDet encoder (reference image_encoder.py:17-116); forward runs EncoderPlan on HIP."""

    def __init__(self, specs, embed_dim, depth, num_heads, global_attn_indexes, img_size=1024):
        super().__init__(specs)
        self.img_size = img_size
        self.embed_dim, self.depth, self.num_heads = embed_dim, depth, num_heads
        self.global_attn_indexes = tuple(global_attn_indexes)
        self.ln_fold = True      # config model.ln_fold (crowdsam/model.py): LayerNorms folded into the projections around them

    def plan(self):
        if self._plan is None or self._plan.ln_fold != bool(self.ln_fold):
            self._require_gpu()
            self._plan = EncoderPlan(self.state_dict(), "", self.embed_dim, self.depth, self.num_heads,
                                     self.global_attn_indexes, self.device, ln_fold=self.ln_fold)
        return self._plan

    def forward_tokens(self, raw_chw_f32):
        """Fast path: raw 0..255 image f32 [3,h,w] -> features f32 [4096,256] (token-major)."""
        return self.plan().forward_static(raw_chw_f32)

    @torch.no_grad()
    def forward(self, x):
        """API path: x = Sam.preprocess output [1,3,1024,1024] -> [1,256,64,64]."""
        assert x.shape == (1, 3, 1024, 1024), "HIP encoder is batch-of-one at 1024^2 (reference usage)"
        p = self.plan()
        hip.sam_im2col(x[0].float().contiguous(), p.ws["col"], normalized=True)
        feat = p.forward(None, skip_im2col=True)
        return feat.view(64, 64, 256).permute(2, 0, 1).unsqueeze(0)


class PromptEncoder(_ParamTree):
    """Parameter container (reference prompt_encoder.py); the point branch + dense PE run inside
    DecoderPlan.  Box / mask prompts are not on Crowd-SAM's inference path."""

    def __init__(self, specs):
        super().__init__(specs, buffers=("pe_layer.positional_encoding_gaussian_matrix",))
        self.embed_dim = 256
        self.image_embedding_size = (64, 64)
        self.input_image_size = (1024, 1024)


class MaskDecoder(_Planned):
    """Two-way decoder + PWD-Net heads (reference mask_decoder.py); parameters only, see DecoderPlan."""

    def __init__(self, specs, n_class):
        super().__init__(specs)
        self.n_class = n_class
        self.num_mask_tokens = 4


class Sam(nn.Module):
    mask_threshold: float = 0.0
    image_format: str = "RGB"

    def __init__(self, embed_dim, depth, num_heads, global_attn_indexes, n_class=1,
                 pixel_mean=(123.675, 116.28, 103.53), pixel_std=(58.395, 57.12, 57.375)):
        super().__init__()
        specs = synth.sam_param_specs(embed_dim, depth, num_heads, tuple(global_attn_indexes), n_class)
        self.image_encoder = ImageEncoderViT(_sub(specs, "image_encoder."), embed_dim, depth, num_heads,
                                             global_attn_indexes)
        self.prompt_encoder = PromptEncoder(_sub(specs, "prompt_encoder."))
        self.mask_decoder = MaskDecoder(_sub(specs, "mask_decoder."), n_class)
        self.n_class = n_class
        assert tuple(pixel_mean) == (123.675, 116.28, 103.53) and tuple(pixel_std) == (58.395, 57.12, 57.375), \
            "the HIP preprocess kernels are built for SAM's pixel statistics"
        self.register_buffer("pixel_mean", torch.tensor(pixel_mean).view(-1, 1, 1), False)
        self.register_buffer("pixel_std", torch.tensor(pixel_std).view(-1, 1, 1), False)
        self._dec_plan = None
        self._dec_key = None

    @property
    def device(self):
        return self.pixel_mean.device

    def decoder_plan(self, max_batch=64):
        """DecoderPlan over prompt_encoder + mask_decoder weights (rebuilt when either changes)."""
        key = (self.prompt_encoder.no_mask_embed.weight.data_ptr(), self.mask_decoder.iou_token.weight.data_ptr(),
               self.mask_decoder.iou_token.weight._version, self.mask_decoder._plan is None)
        if self._dec_plan is None or self._dec_key != key or self.mask_decoder._plan is None:
            if self.device.type != "cuda":
                raise RuntimeError("crowdsam_amd runs on MI355X only: move the model to 'cuda' first")
            sd = {"prompt_encoder." + k: v for k, v in self.prompt_encoder.state_dict().items()}
            sd.update({"mask_decoder." + k: v for k, v in self.mask_decoder.state_dict().items()})
            self._dec_plan = DecoderPlan(sd, self.device, self.n_class, max_batch)
            self.mask_decoder._plan = self._dec_plan
            self._dec_key = (key[0], key[1], key[2], False)
        return self._dec_plan

    @torch.no_grad()
    def preprocess(self, x):
        """sam.py:163-173: normalise + zero-pad a [3,h,w] (or [1,3,h,w]) image to 1024^2."""
        squeeze = x.dim() == 3
        img = (x if squeeze else x[0]).float().contiguous()
        out = hip.preprocess_pad(img)
        return out if squeeze else out.unsqueeze(0)

    @torch.no_grad()
    def postprocess_masks(self, masks, input_size, original_size):
        """sam.py:132-161 (API-compatible full up-sampling of every candidate; the driver uses the
        fused csam_mask_post on the selected candidate instead)."""
        B, C = masks.shape[:2]
        m = hip.bilinear_f32(masks.reshape(B * C, masks.shape[2], masks.shape[3]).float().contiguous(),
                             (self.image_encoder.img_size, self.image_encoder.img_size))
        m = m[:, : input_size[0], : input_size[1]].contiguous()
        if tuple(original_size) != tuple(input_size):
            m = hip.bilinear_f32(m, tuple(original_size))
        return m.view(B, C, original_size[0], original_size[1])

    def forward(self, *a, **k):
        raise NotImplementedError("Sam.forward is broken in the reference (SURVEY.md trap 3); use SamPredictor")
