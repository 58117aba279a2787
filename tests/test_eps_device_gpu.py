"""Device-resident Efficient Prompt Sampler (csam_eps_select / csam_occupancy_prune) against the host loop it replaces
(crowdsam/model.py:233-249 of the reference: points[:batch_size], then points[~occupy_mask[y, x]])."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_eps_select_matches_list_slicing(cuda):
    """Rounds of select + random pruning against alive[:B] / alive[~occ] on the host: same points in the same order,
    float64-scaled coordinates equal to ResizeLongestSide.apply_coords, valid / left counts, (0, 0) filler slots."""
    from crowdsam_amd import hip
    from segment_anything_cs.utils.transforms import ResizeLongestSide
    rs = np.random.RandomState(0)
    for P, B, size in ((36864, 32, (683, 1024)), (70, 32, (1000, 750)), (5, 8, (333, 500)), (1, 1, (1024, 1024))):
        pts = rs.randint(0, 1024, (P, 2)).astype(np.int64)
        t = ResizeLongestSide(1024)
        nh, nw = t.get_preprocess_shape(size[0], size[1], 1024)
        d_pts = torch.as_tensor(pts, dtype=torch.int32).to(cuda)
        alive_d = torch.ones(P, dtype=torch.uint8, device=cuda)
        alive = np.arange(P)
        o_pts = torch.empty(B, 2, dtype=torch.int32, device=cuda)
        o_xy = torch.empty(B, 2, dtype=torch.float32, device=cuda)
        counts = torch.zeros(2, dtype=torch.int32, device=cuda)
        for rnd in range(12):
            hip.eps_select(d_pts, alive_d, B, nw / size[1], nh / size[0], o_pts, o_xy, counts)
            sel, alive = alive[:B], alive[B:]
            nv, left = counts.tolist()
            assert nv == len(sel) and left == len(alive), (P, rnd)
            np.testing.assert_array_equal(o_pts[:nv].cpu().numpy(), pts[sel])
            want = torch.as_tensor(t.apply_coords(pts[sel], size)).to(torch.float32).numpy()
            np.testing.assert_array_equal(o_xy[:nv].cpu().numpy(), want)               # bit-equal: float64 then fp32 cast
            assert not o_pts[nv:].any() and not o_xy[nv:].any()
            flags = np.zeros(P, np.uint8)
            flags[alive] = 1
            np.testing.assert_array_equal(alive_d.cpu().numpy(), flags)
            # prune: two random rectangles as this round's masks, one of them without the occupancy flag
            H, W = 1024, 1024
            masks = torch.zeros(2, H, W, dtype=torch.uint8, device=cuda)
            x0, y0 = rs.randint(0, 700, 2)
            masks[0, y0:y0 + 300, x0:x0 + 200] = 1
            masks[1, 100:900, 100:900] = 1
            occ = torch.tensor([1, 0], dtype=torch.uint8, device=cuda)
            hip.occupancy_prune(d_pts, masks, occ, 2, H, W, alive_d)
            m0 = masks[0].cpu().numpy().astype(bool)
            alive = alive[~m0[pts[alive, 1], pts[alive, 0]]]


@pytest.mark.parametrize("ppb,maxp", [(8, 24), (8, 64), (24, 64), (5, 500)])
def test_generate_device_sampler_equals_host_loop(cuda, ppb, maxp):
    """Same frame, same seed, sampler on the device vs on the host: the same prompts survive in the same order.  (8, 64)
    and (5, 500) run until the list is dry; (24, 64) ends on a partial round that the device path runs at full width."""
    from crowdsam.model import CrowdSAM
    from crowdsam_amd import synth
    from oracle.make_goldens import PIPE_CFG, pipeline_image
    from tests.test_pipeline_gpu import ARCH, GpuStandInDino, _config
    cfg = dict(PIPE_CFG)
    cfg.update(points_per_batch=ppb, max_prompts=maxp)
    m = CrowdSAM(_config(cfg), sam_state_dict=synth.make_sam_state_dict(ARCH), dino_model=GpuStandInDino(cuda))
    outs = []
    for on_device in (False, True):
        m.eps_on_device = on_device
        np.random.seed(5)
        outs.append(m.generate(pipeline_image()))
    a, b = outs
    assert len(a["boxes"]) > 0 and a["boxes"].shape == b["boxes"].shape, (a["boxes"].shape, b["boxes"].shape)
    np.testing.assert_array_equal(a["points"], b["points"])
    np.testing.assert_array_equal(a["categories"], b["categories"])
    np.testing.assert_allclose(a["scores"], b["scores"], rtol=0, atol=2e-3)      # a partial round decodes at another width
    assert np.abs(a["boxes"] - b["boxes"]).max() <= 2
