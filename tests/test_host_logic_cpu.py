"""CPU tests of the host-side mirror (MaskData, crop boxes, RLE packing, config overrides, resize
arithmetic) against the reference-generated goldens and the oracle."""
import os

import numpy as np
import pytest
import torch

import crowdsam.utils as cu
from oracle import pipeline_oracle as po
from segment_anything_cs.utils import amg
from segment_anything_cs.utils.transforms import ResizeLongestSide

G = os.path.join(os.path.dirname(__file__), "golden")


def _logits():
    rs = np.random.RandomState(5)
    x = torch.from_numpy((rs.standard_normal((6, 40, 56)) * 2).astype(np.float32))
    x = torch.nn.functional.avg_pool2d(x[None], 5, 1, 2)[0] * 3
    x[4] = -5.0
    x[5] = 5.0
    return x


def test_amg_mirror_matches_reference_golden():
    g = np.load(os.path.join(G, "amg.npz"), allow_pickle=True)
    x = _logits()
    stab = amg.calculate_stability_score(x, 0.0, 1.0).numpy()
    np.testing.assert_array_equal(np.nan_to_num(stab, nan=-1), np.nan_to_num(g["stab"], nan=-1))
    masks = x > 0
    np.testing.assert_array_equal(amg.batched_mask_to_box(masks).numpy(), g["boxes"])
    for i, (r, c) in enumerate(zip(amg.mask_to_rle_pytorch(masks), g["rle_counts"])):
        assert r["counts"] == list(c) and r["size"] == [40, 56]
        np.testing.assert_array_equal(amg.rle_to_mask(r), masks[i].numpy())
    cb, cl = amg.generate_crop_boxes((445, 640), 2, 0.341)
    np.testing.assert_array_equal(np.array(cb), g["crop_boxes"])
    np.testing.assert_array_equal(np.array(cl), g["crop_layers"])
    md = amg.MaskData(a=torch.arange(6), b=np.arange(6) * 2, c=list("abcdef"))
    md.filter(torch.tensor([True, False, True, True, False, True]))
    np.testing.assert_array_equal(md["a"].numpy(), g["md_a"])
    np.testing.assert_array_equal(md["b"], g["md_b"])
    assert md["c"] == list(g["md_c"])


def test_maskdata_cat_and_index_filter():
    a = amg.MaskData(x=torch.arange(3), y=["p", "q", "r"], z=np.ones((3, 2)))
    b = amg.MaskData(x=torch.arange(3, 5), y=["s", "t"], z=np.zeros((2, 2)))
    a.cat(b)
    assert a["x"].tolist() == [0, 1, 2, 3, 4] and a["y"] == list("pqrst") and a["z"].shape == (5, 2)
    a.filter(torch.tensor([4, 0]))
    assert a["x"].tolist() == [4, 0] and a["y"] == ["t", "p"]
    e = amg.MaskData()
    e.cat(a)
    assert e["x"].tolist() == [4, 0]
    a.to_numpy()
    assert isinstance(a["x"], np.ndarray)


def test_rle_roundtrip_and_coco_string():
    rs = np.random.RandomState(3)
    m = torch.from_numpy(rs.rand(3, 37, 53) > 0.6)
    rles = amg.mask_to_rle_pytorch(m)
    for i, r in enumerate(rles):
        np.testing.assert_array_equal(amg.rle_to_mask(r), m[i].numpy())
        assert amg.area_from_rle(r) == int(m[i].sum())
        assert amg.coco_encode_rle(r)["counts"] == po.coco_rle_string(r["counts"])


def test_remove_small_regions_matches_oracle():
    rs = np.random.RandomState(8)
    for _ in range(5):
        m = torch.nn.functional.avg_pool2d(torch.from_numpy(rs.standard_normal((1, 1, 64, 80)).astype(np.float32)), 5, 1, 2)[0, 0].numpy() > 0
        for mode in ("holes", "islands"):
            a, ca = amg.remove_small_regions(m, 12, mode)
            b, cb = po.remove_small_regions(m, 12, mode)
            assert ca == cb
            np.testing.assert_array_equal(a, b)
    tiny = np.zeros((10, 10), bool)
    tiny[2, 2] = True
    out, changed = amg.remove_small_regions(tiny, 100, "islands")   # every region small -> keep largest
    assert changed and out.sum() == 1


def test_transforms_and_resize_arithmetic():
    t = ResizeLongestSide(1024)
    assert t.get_preprocess_shape(682, 1023, 1024) == (683, 1024)
    pts = np.array([[10, 20], [1000, 500]])
    np.testing.assert_allclose(t.apply_coords(pts, (682, 1023)), po.apply_coords(pts, (682, 1023)))
    assert t.apply_coords(pts, (768, 1024)).dtype == np.float64
    # trap 9: int(r*w) lands on 1023 for ~12% of widths
    n1023 = sum(1 for w in range(300, 4001) if max(cu.resize_shape(int(w * 0.66), w, 1024)[:2]) == 1023)
    assert 300 < n1023 < 700
    img = np.zeros((512, 1024, 3), np.uint8)
    out, r = cu.resize_image(img, 1024)
    assert out.shape == (512, 1024, 3) and r == 1.0 and out is not img


def test_config_overrides():
    cfg = {"test": {"grid_size": 192, "pos_sim_thresh": 0.5, "output_rles": True}, "model": {"sam_model": "vit_l"}}
    cu.modify_config(cfg, ["test.grid_size", "64", "test.pos_sim_thresh", "-1.5", "test.output_rles", "False",
                           "model.sam_model", "vit_b"])
    assert cfg["test"] == {"grid_size": 64, "pos_sim_thresh": -1.5, "output_rles": False}
    assert cfg["model"]["sam_model"] == "vit_b"


def test_box_edge_filter_and_uncrop():
    boxes = torch.tensor([[0, 0, 100, 100], [400, 10, 512, 300]])
    near = cu.is_box_near_crop_edge(boxes, [0, 0, 512, 512], [0, 0, 1024, 1024], 1.0)
    assert near.tolist() == [False, True]
    np.testing.assert_array_equal(near.numpy(), po.is_box_near_crop_edge(boxes, [0, 0, 512, 512], [0, 0, 1024, 1024], 1.0).numpy())
    out = cu.uncrop_boxes_xyxy(boxes.float(), [10, 20, 0, 0], 2.0)
    assert out[1].tolist() == [210.0, 25.0, 266.0, 170.0]


def test_state_dict_layout_and_adapter_load():
    """Reference key layout incl. the unused 5th hyper-MLP (trap 5); adapter loads with strict=False."""
    import segment_anything_cs as sa
    from crowdsam_amd import synth
    sam = sa.sam_model_registry["vit_test128"](n_class=1)
    keys = set(sam.state_dict())
    assert "mask_decoder.output_hypernetworks_mlps.4.layers.2.weight" in keys
    assert "image_encoder.blocks.1.attn.rel_pos_h" in keys and "pixel_mean" not in keys
    sd = synth.make_sam_state_dict("vit_test128")
    assert set(sd) == keys
    adapter = {k[len("mask_decoder."):]: v for k, v in sd.items() if k.startswith("mask_decoder.")}
    res = sam.mask_decoder.load_state_dict(adapter, strict=False)
    assert not res.missing_keys and not res.unexpected_keys


def test_batch_eval_convert_to_coco_schema():
    """tools/batch_eval.py:31-58: ids = file_name[:-4], xyxy -> xywh, area from xyxy, running annotation ids."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location(
        "own_batch_eval", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "batch_eval.py"))
    be = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(be)
    gt = {"images": [{"file_name": "a,1.jpg", "id": 7, "width": 10, "height": 20},
                     {"file_name": "b,2.jpg", "id": 9, "width": 30, "height": 40}], "categories": [{"id": 1}]}
    rows = np.array([[1, 2, 3, 12, 23, 0.5], [0, 1, 1, 5, 9, 0.9], [1, 0, 0, 4, 4, 0.25]], np.float32)
    res = be.rows_to_results(rows, 2)
    assert res[0]["boxes"] == [[1.0, 1.0, 5.0, 9.0]] and res[1]["scores"] == [0.5, 0.25]
    coco = be.convert_to_coco(res, gt)
    assert [im["id"] for im in coco["images"]] == ["a,1", "b,2"]
    a = coco["annotations"]
    assert [x["id"] for x in a] == [0, 1, 2] and [x["image_id"] for x in a] == ["a,1", "b,2", "b,2"]
    assert a[0]["bbox"] == [1.0, 1.0, 4.0, 8.0] and a[0]["area"] == 32.0 and a[1]["bbox"] == [2.0, 3.0, 10.0, 20.0]
    assert coco["categories"] == [{"id": 1}]


def test_index_shuffle_equals_row_shuffle():
    """CrowdSAM._process_crop shuffles an index vector instead of the point rows (crowdsam/model.py:231 of the reference
    calls np.random.shuffle(points)): same permutation, same global RNG state afterwards, for every length."""
    for n in (0, 1, 2, 3, 17, 500, 4096):
        pts = np.random.RandomState(n).randint(0, 1024, (n, 2))
        a = pts.copy()
        np.random.seed(1234 + n)
        np.random.shuffle(a)
        nxt_a = np.random.randint(0, 1 << 30)
        np.random.seed(1234 + n)
        perm = np.arange(n)
        np.random.shuffle(perm)
        nxt_b = np.random.randint(0, 1 << 30)
        assert np.array_equal(a, pts[perm]) and nxt_a == nxt_b, n


def test_batched_coco_strings_equal_per_mask_strings():
    """amg.coco_encode_rles (one call into the C packer for all masks of a frame) against coco_encode_rle per mask and the
    oracle's pure-Python packer; empty list, an all-zero mask (a single run), a first-pixel-set mask (leading zero run)."""
    rs = np.random.RandomState(11)
    m = torch.from_numpy(rs.rand(7, 29, 41) > 0.55)
    m[2] = False
    m[3] = True
    m[4, 0, 0] = True
    rles = amg.mask_to_rle_arrays(m)
    batch = amg.coco_encode_rles(rles)
    assert len(batch) == 7 and amg.coco_encode_rles([]) == []
    for r, b in zip(rles, batch):
        assert b == amg.coco_encode_rle(r)
        assert b["counts"] == po.coco_rle_string([int(c) for c in r["counts"]])


def test_roctx_ranges_are_balanced_and_off_by_default():
    """crowdsam_amd.trace (the reference has no tracing): a no-op until enable(); ranges nest, unwind()
    closes what an early return left open; the roctx library of the ROCm install loads without a profiler attached."""
    from crowdsam_amd import trace
    assert not trace.enabled and trace.depth == 0
    trace.push("x"); trace.pop(); trace.mark("m")
    assert trace.depth == 0
    if not trace.enable():
        pytest.skip("no roctx library in this image")
    try:
        with trace.range("generate"):
            trace.push("set_image")
            trace.push("sam_encoder")
            assert trace.depth == 3
            trace.unwind(1)
            assert trace.depth == 1
            trace.mark("eps.batch")
        assert trace.depth == 0
        trace.pop()                         # an unmatched pop is ignored
        assert trace.depth == 0
    finally:
        trace.disable()
    assert not trace.enabled


def test_settled_passes_results_through_and_freezes_once(monkeypatch):
    """crowdsam.model.settled (round 6): a transparent wrapper around a result stream that calls settle_host() exactly once,
    after the `after`-th result (plans and graphs are built lazily by the first frames)."""
    import crowdsam.model as cm
    calls = []
    monkeypatch.setattr(cm, "settle_host", lambda: calls.append(1) or 0)
    assert list(cm.settled(iter(range(5)), after=2)) == [0, 1, 2, 3, 4] and calls == [1]
    calls.clear()
    assert list(cm.settled(iter(range(1)), after=2)) == [0] and calls == []          # shorter than the warm-up: no second freeze
    assert list(zip("ab", cm.settled(iter([10, 20, 30])))) == [("a", 10), ("b", 20)]  # zip may stop early: nothing breaks
