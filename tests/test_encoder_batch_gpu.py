"""Image-batched encoder passes (ImageEncoderViT.forward with B > 1, image_encoder.py:106-116; DINOv2 blocks with B > 1):
B frames go through every projection as ONE [B * tokens, D] matrix and through attention per image.  Every output element
is the same chain of MFMAs in the same K order whatever the tile shape the launcher picks for the taller matrix, so each
image's features must be BIT-IDENTICAL to a pass of its own -- which is what pins the batched route to all the
single-image parity tests (reference goldens, oracle) at once."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _frames(cuda, n):
    from crowdsam_amd import synth
    out = []
    for i in range(n):
        f = synth.synthetic_crowd_frame(11 + i, 1024, 60)
        f = (f, f[:768], f[:, :684], f[:1000])[i % 4]               # full, landscape, portrait, a 1000-row frame
        out.append(torch.from_numpy(np.ascontiguousarray(f)).permute(2, 0, 1).float().contiguous().to(cuda))
    return out


@pytest.mark.parametrize("arch,B", [("vit_l", 2), ("vit_l", 4), ("vit_l", 5), ("vit_b", 3), ("vit_h", 2)])
def test_sam_encoder_batch_is_bitwise_the_single_image_pass(cuda, arch, B):
    from crowdsam_amd import synth
    from crowdsam_amd.encoder import EncoderPlan
    D, depth, heads, gidx = synth.SAM_CONFIGS[arch]
    sd = synth.make_sam_state_dict(arch)
    plan = EncoderPlan(sd, "image_encoder.", D, depth, heads, gidx, cuda)
    imgs = _frames(cuda, B)
    single = [plan.forward_static(im).clone() for im in imgs]
    batch = plan.forward_batch_static(imgs).clone()
    again = plan.forward_batch_static(imgs).clone()                 # graph replay
    for b in range(B):
        assert torch.equal(batch[b], single[b]), (arch, B, b, (batch[b] - single[b]).abs().max().item())
    assert torch.equal(batch, again)
    # the pass cut at block boundaries (what the look-ahead pipeline queues beside each frame's tail)
    views = plan.load_images(imgs)
    plan.embed(views, B)
    cuts = [0, depth // 3, depth // 3 + 1, depth]
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        plan.run_blocks(lo, hi, B)
    out = torch.empty(B, 4096, 256, dtype=torch.float32, device=cuda)
    plan.neck(out, B)
    assert torch.equal(out, batch)
    # a single-image pass after a batched one still works on the grown workspaces
    assert torch.equal(plan.forward_static(imgs[0]), single[0])


@pytest.mark.parametrize("B", [2, 3, 4])
def test_dino_batch_is_bitwise_the_single_image_pass(cuda, B):
    from crowdsam_amd import synth
    from crowdsam_amd.dino import DinoPlan, N_PATCH
    sd = synth.make_dino_state_dict()
    plan = DinoPlan(sd, cuda)
    imgs = _frames(cuda, B)
    single = []
    for im in imgs:
        o = torch.zeros(5376, 1024, dtype=torch.float16, device=cuda)
        plan.forward_static(im, o[:N_PATCH])
        single.append(o.clone())
    outs = [torch.zeros(5376, 1024, dtype=torch.float16, device=cuda) for _ in range(B)]
    plan.forward_batch_static(imgs, [o[:N_PATCH] for o in outs])
    for b in range(B):
        assert torch.equal(outs[b], single[b]), (B, b, (outs[b].float() - single[b].float()).abs().max().item())
    plan.forward_batch_static(imgs, [o[:N_PATCH] for o in outs])   # graph replay
    for b in range(B):
        assert torch.equal(outs[b], single[b])
