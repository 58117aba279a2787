"""Model builders / registry (reference: segment_anything_cs/build_sam.py:14-158).

Differences from the reference, on purpose: every registry entry accepts ``n_class`` (the
reference's vit_b / vit_h / default entries raise TypeError and vit_t raises NameError, SURVEY.md
trap 1), and the returned ``Sam`` computes on MI355X HIP kernels.  Fixed hyper-parameters are those
of ``_build_sam``: 1024^2 input, patch 16, window 14, decoder dim 256 / 8 heads / mlp 2048 / depth 2.
"""
import torch

from .modeling import Sam


def _build_sam(encoder_embed_dim, encoder_depth, encoder_num_heads, n_class, encoder_global_attn_indexes,
               checkpoint=None):
    sam = Sam(encoder_embed_dim, encoder_depth, encoder_num_heads, tuple(encoder_global_attn_indexes), n_class)
    sam.eval()
    if checkpoint is not None:
        with open(checkpoint, "rb") as f:
            state_dict = torch.load(f, map_location="cpu")
        sam.load_state_dict(state_dict, strict=False)   # adapter heads are absent from SAM checkpoints
    return sam


def build_sam_vit_h(checkpoint=None, n_class=1):
    return _build_sam(1280, 32, 16, n_class, (7, 15, 23, 31), checkpoint)


def build_sam_vit_l(checkpoint=None, n_class=1):
    return _build_sam(1024, 24, 16, n_class, (5, 11, 17, 23), checkpoint)


def build_sam_vit_b(checkpoint=None, n_class=1):
    return _build_sam(768, 12, 12, n_class, (2, 5, 8, 11), checkpoint)


build_sam = build_sam_vit_h

sam_model_registry = {
    "default": build_sam_vit_h,
    "vit_h": build_sam_vit_h,
    "vit_l": build_sam_vit_l,
    "vit_b": build_sam_vit_b,
}


def register_sam_arch(name, embed_dim, depth, num_heads, global_attn_indexes):
    """Add an encoder geometry to the registry (narrow test encoders, SURVEY.md §8c golden recipe)."""
    def _builder(checkpoint=None, n_class=1):
        return _build_sam(embed_dim, depth, num_heads, n_class, tuple(global_attn_indexes), checkpoint)
    sam_model_registry[name] = _builder
    return _builder


register_sam_arch("vit_test128", 128, 4, 2, (1, 3))
