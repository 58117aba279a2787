"""Mask coverage NMS: oracle vs goldens captured from the reference's crowdsam/utils.py (CPU),
csam_mask_nms vs oracle and goldens (GPU)."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def _golden():
    g = np.load(os.path.join(HERE, "golden", "mask_nms.npz"))
    shape = tuple(g["shape"])
    masks = np.unpackbits(g["masks_packed"])[: int(np.prod(shape))].reshape(shape).astype(bool)
    return g, masks


def test_oracle_matches_reference_golden():
    from oracle import pipeline_oracle as po
    g, masks = _golden()
    mt = torch.from_numpy(masks)
    for thr in (0.3, 0.5, 0.8):
        keep = po.mask_iou_nms(None, g["scores"], mt, thr)
        assert np.array_equal(keep, g["keep_%02d" % int(thr * 100)])
    a, b = mt[:8].unsqueeze(1), mt[None, 8:20]
    np.testing.assert_array_equal(po.coverage(a, b).numpy(), g["coverage"])
    np.testing.assert_array_equal(po.mask_iou(a, b).numpy(), g["mask_iou"])


@pytest.mark.gpu
def test_device_matches_golden(cuda):
    from crowdsam import utils
    g, masks = _golden()
    for thr in (0.3, 0.5, 0.8):
        keep = utils.mask_iou_nms(None, g["scores"], torch.from_numpy(masks).to(cuda), thr)
        assert np.array_equal(keep, g["keep_%02d" % int(thr * 100)])
    a, b = torch.from_numpy(masks[:8]).to(cuda).unsqueeze(1), torch.from_numpy(masks[None, 8:20]).to(cuda)
    np.testing.assert_array_equal(utils.coverage(a, b).cpu().numpy(), g["coverage"])
    np.testing.assert_array_equal(utils.mask_iou(a, b).cpu().numpy(), g["mask_iou"])


@pytest.mark.gpu
@pytest.mark.parametrize("n,hw", [(1, (40, 40)), (63, (150, 150)), (200, (683, 1024)), (530, (301, 77))])
def test_device_matches_oracle_random(cuda, n, hw):
    from crowdsam_amd import hip
    from oracle import pipeline_oracle as po
    rs = np.random.RandomState(n)
    H, W = hw
    yy, xx = np.mgrid[0:H, 0:W]
    masks = np.zeros((n, H, W), bool)
    for i in range(n):
        cy, cx = rs.uniform(0, H), rs.uniform(0, W)
        ry, rx = rs.uniform(2, H / 3), rs.uniform(2, W / 3)
        masks[i] = ((yy - cy) / ry) ** 2 + ((xx - cx) / rx) ** 2 <= 1
    if n > 5:
        masks[3] = masks[2]                    # exact duplicate: coverage 1
        masks[5] = False                       # empty: NaN coverage, never suppressed nor suppressing
    scores = torch.from_numpy(rs.permutation(n).astype(np.float32))
    for thr in (0.2, 0.6):
        keep = hip.mask_nms(torch.from_numpy(masks).to(cuda), scores.to(cuda), thr).cpu().numpy()
        ref = po.mask_iou_nms(None, scores.numpy(), torch.from_numpy(masks), thr)
        assert np.array_equal(keep, ref)
