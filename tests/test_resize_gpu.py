"""Row a1 on the GPU: csam_resize_linear_u8 (cv2.resize INTER_LINEAR restated) bit-exact vs the oracle restatement,
incl. up-scaling (configs[0]: 512 -> 1024), 1500 -> 1024 (configs[4]), the exact-2x route and a 1023-side case; and the
driver path (CrowdSAM.crop_image -> predictor.set_image) through it."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(512, 512), (1500, 1500), (445, 640), (1080, 1920), (2048, 1536), (700, 1366),
                                   (333, 1024), (90, 64)])
def test_resize_kernel_bit_exact_vs_oracle(shape):
    import crowdsam.utils as cu
    from oracle import resize_oracle as ro
    h, w = shape
    img = np.random.RandomState(h + w).randint(0, 256, (h, w, 3)).astype(np.uint8)
    ref, r_ref = ro.resize_image(img, 1024)
    u8, f32, r = cu.resize_frame_device(img, 1024, torch.device("cuda:0"))
    assert r == r_ref and tuple(u8.shape) == ref.shape
    assert np.array_equal(u8.cpu().numpy(), ref)
    if f32 is not None:
        assert np.array_equal(f32.cpu().numpy(), ref.transpose(2, 0, 1).astype(np.float32))
    out, r2 = cu.resize_image(img, 1024)            # the reference's ndarray API
    assert r2 == r and np.array_equal(out, ref)


def test_u8_to_chw():
    from crowdsam_amd import hip
    img = torch.from_numpy(np.random.RandomState(0).randint(0, 256, (37, 53, 3)).astype(np.uint8)).cuda()
    assert torch.equal(hip.u8hwc_to_f32chw(img), img.permute(2, 0, 1).float())


@pytest.mark.parametrize("shape", [(682, 1023), (1023, 700), (1023, 1023), (600, 900), (445, 640)])
def test_pil_bilinear_kernel_bit_exact_vs_pillow(shape):
    """ResizeLongestSide.apply_image (PIL bilinear through torchvision in the reference, transforms.py:26-31) as the
    device kernel csam_pil_resample_u8, against Pillow itself (present on the GPU box); incl. the 1023 -> 1024 case of
    SURVEY.md trap 9.  And the predictor path that uses it."""
    from PIL import Image
    from crowdsam_amd import hip
    from crowdsam_amd.resize import pil_bilinear_tables_device
    from segment_anything_cs.utils.transforms import ResizeLongestSide
    h, w = shape
    img = np.random.RandomState(h * 7 + w).randint(0, 256, (h, w, 3)).astype(np.uint8)
    th, tw = ResizeLongestSide.get_preprocess_shape(h, w, 1024)
    ref = np.array(Image.fromarray(img).resize((tw, th), Image.BILINEAR))
    d = torch.from_numpy(img).cuda()
    u8, f32 = hip.pil_resize_bilinear_u8(d, (th, tw), pil_bilinear_tables_device(w, tw, "cuda:0"),
                                         pil_bilinear_tables_device(h, th, "cuda:0"))
    assert np.array_equal(u8.cpu().numpy(), ref)
    assert np.array_equal(f32.cpu().numpy(), ref.transpose(2, 0, 1).astype(np.float32))
