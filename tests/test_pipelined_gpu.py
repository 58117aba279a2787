"""Depth-2 pipeline of the per-image loop (tools/test.py:62-82 of the reference is a serial loop): generate(image,
next_image=...) starts the next frame's upload + encoders beside the current frame's tail.  Results must be bit-identical
to serial calls -- no random number is drawn before sample_prompts, and the prefetch only touches buffers the queued sweep
has finished with."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _same(a, b):
    assert a["boxes"].shape == b["boxes"].shape
    for k in ("boxes", "scores", "points", "categories", "stability_score"):
        np.testing.assert_array_equal(a[k], b[k], err_msg=k)
    assert [r["counts"] for r in a["rles"]] == [r["counts"] for r in b["rles"]]


def test_pipelined_generate_equals_serial_small_model(cuda):
    """vit_test128 + stand-in DINO, EPS with pruning (the shipped loop shape), frames of two different shapes in one
    stream (the prefetch must cope with a change of frame geometry) and a look-ahead frame that is then NOT the next one."""
    from crowdsam.model import CrowdSAM
    from crowdsam_amd import synth
    from oracle.make_goldens import PIPE_CFG
    from tests.test_pipeline_gpu import ARCH, GpuStandInDino, _config
    m = CrowdSAM(_config(dict(PIPE_CFG)), sam_state_dict=synth.make_sam_state_dict(ARCH), dino_model=GpuStandInDino(cuda))
    frames = [synth.synthetic_crowd_frame(i, 1024, 60)[:768] for i in range(3)] + [synth.synthetic_crowd_frame(7, 1024, 60)] + \
             [synth.synthetic_crowd_frame(8, 1366, 80)[:700]]
    np.random.seed(11)
    serial = [m.generate(f) for f in frames]
    # depth-2 pipeline; image-batched look-ahead in groups of 2 / 4 (mixed frame shapes in a group), the stream starting with
    # groups of 1, 2, 4 frames (group_ramp, the default) or with a full group
    for batch, ramp in ((1, True), (2, True), (4, True), (4, False)):
        np.random.seed(11)
        m.group_ramp = ramp
        piped = list(m.generate_stream(frames, batch=batch))
        assert len(piped) == len(serial) and any(len(o["boxes"]) for o in serial)
        for a, b in zip(serial, piped):
            _same(a, b)
    m.group_ramp = True
    # a look-ahead that does not come true: the prefetched state must be dropped, not used
    np.random.seed(11)
    a = m.generate(frames[0], next_image=frames[3])
    b = m.generate(frames[1])
    _same(a, serial[0])
    _same(b, serial[1])


def test_work_stream_and_caller_streams_give_the_same_results(cuda):
    """generate() runs the frame on its own high-priority stream (CSAM_WORK_STREAM, round 4): the results must equal those on the
    caller's stream bit for bit -- from the default stream, from a caller-side stream context, and with look-ahead."""
    import crowdsam.model as cm
    from crowdsam.model import CrowdSAM
    from crowdsam_amd import synth
    from oracle.make_goldens import PIPE_CFG
    from tests.test_pipeline_gpu import ARCH, GpuStandInDino, _config
    m = CrowdSAM(_config(dict(PIPE_CFG)), sam_state_dict=synth.make_sam_state_dict(ARCH), dino_model=GpuStandInDino(cuda))
    frames = [synth.synthetic_crowd_frame(i, 1024, 60)[:768] for i in range(3)]
    assert cm._WORK_STREAM
    try:
        cm._WORK_STREAM = False
        np.random.seed(5)
        plain = [m.generate(f) for f in frames]
        cm._WORK_STREAM = True
        np.random.seed(5)
        own = [m.generate(f) for f in frames]
        np.random.seed(5)
        with torch.cuda.stream(torch.cuda.Stream(device=cuda)):
            inside = list(m.generate_stream(frames))
        torch.cuda.synchronize()
    finally:
        cm._WORK_STREAM = True
    assert any(len(o["boxes"]) for o in plain)
    for a, b, c in zip(plain, own, inside):
        _same(a, b)
        _same(a, c)


def test_pipelined_generate_equals_serial_full_model(cuda):
    """The bench's model (ViT-L + DINOv2-L x 24, 64 x 64 dense sweep, crowded thresholds): 7 frames, pipelined == serial --
    the depth-2 pipeline (batch 1) and the image-batched look-ahead in groups of 2, 3 and 4 frames (last group ragged)."""
    from crowdsam.model import CrowdSAM
    from crowdsam.utils import DEFAULT_TEST_CONFIG
    from crowdsam_amd import synth
    t = dict(DEFAULT_TEST_CONFIG)
    t.update(grid_size=64, points_per_batch=4096, stability_score_thresh=0.25, pos_sim_thresh=-float("inf"),
             filter_thresh=float("inf"), max_prompts=4096, box_nms_thresh=1.0, crop_nms_thresh=1.0, pred_iou_thresh=0.889)
    cfg = {"environ": {"device": "cuda"}, "model": {"sam_model": "vit_l", "sam_arch": "crowdsam", "n_class": 1,
                                                     "trainfree": False}, "test": t}
    m = CrowdSAM(cfg, sam_state_dict=synth.make_sam_state_dict("vit_l"), dino_state_dict=synth.make_dino_state_dict())
    frames = [synth.synthetic_crowd_frame(i, 1024, 150) for i in range(7)]
    np.random.seed(3)
    serial = [m.generate(f) for f in frames]
    assert sum(len(o["boxes"]) for o in serial) > 200
    # the second (4, True) stream REPLAYS the chunk graphs the first one captured (groups of 1, 2, 4 frames on two buffer sets
    # that must not move between streams); (4, False) starts with a full group
    for batch, ramp in ((1, True), (2, True), (3, True), (4, True), (4, True), (4, False), (2, True)):
        np.random.seed(3)
        m.group_ramp = ramp
        piped = list(m.generate_stream(frames, batch=batch))
        assert len(piped) == len(serial)
        for a, b in zip(serial, piped):
            _same(a, b)
    torch.cuda.synchronize()


def test_pil_look_ahead_is_adopted(cuda):
    """generate() takes PIL images (crowdsam/model.py:133-140): the look-ahead record remembers the caller's OBJECT, so a PIL
    next_image is adopted by the next call (the ndarray identity check threw its encode away)."""
    from PIL import Image
    from crowdsam.model import CrowdSAM
    from crowdsam_amd import synth
    from oracle.make_goldens import PIPE_CFG
    from tests.test_pipeline_gpu import ARCH, GpuStandInDino, _config
    m = CrowdSAM(_config(dict(PIPE_CFG)), sam_state_dict=synth.make_sam_state_dict(ARCH), dino_model=GpuStandInDino(cuda))
    frames = [Image.fromarray(synth.synthetic_crowd_frame(i, 1024, 60)[:768]) for i in range(2)]
    np.random.seed(2)
    serial = [m.generate(np.array(f)) for f in frames]
    np.random.seed(2)
    a = m.generate(frames[0], next_image=frames[1])
    assert m._prefetched is not None and m._prefetched["src"] is frames[1]
    adopted = []
    orig = m.predictor.adopt_prefetched
    m.predictor.adopt_prefetched = lambda b: (adopted.append(1), orig(b))[1]
    b = m.generate(frames[1])
    assert adopted == [1]
    _same(a, serial[0])
    _same(b, serial[1])


def test_multi_crop_through_one_batched_pass_equals_crop_by_crop(cuda):
    """Multi-crop mode (crowdsam/model.py:151-178: 1 + 4 crops, each resized to max_size and encoded) with the real backbones' plans
    (SAM vit_test128 + DINOv2 at depth 2): the five crops as ONE image-batched pass of both encoders (CrowdSAM._encode_crops)
    against the crop-by-crop route -- identical results, field for field."""
    from crowdsam.model import CrowdSAM
    from crowdsam_amd import synth
    from oracle.make_goldens import PIPE_CFG
    from tests.test_pipeline_gpu import ARCH, _config
    cfg = dict(PIPE_CFG)
    cfg.update(crop_n_layers=1)
    m = CrowdSAM(_config(cfg), sam_state_dict=synth.make_sam_state_dict(ARCH), dino_state_dict=synth.make_dino_state_dict(depth=2),
                 dino_depth=2)
    img = synth.synthetic_crowd_frame(4, 1366, 90)[:700]
    calls = []
    orig = m.predictor.group_chunk
    m.predictor.group_chunk = lambda g, c, n, **kw: (calls.append(g["B"]), orig(g, c, n, **kw))[1]
    np.random.seed(9)
    batched = m.generate(img)
    assert calls == [5], calls                                  # one pass over the five crops
    m._encode_crops = lambda image, boxes: None                 # crop by crop: set_image per crop
    np.random.seed(9)
    serial = m.generate(img)
    assert calls == [5]
    assert len(batched["boxes"]) == len(serial["boxes"])
    _same(batched, serial)
    torch.cuda.synchronize()
