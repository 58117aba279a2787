"""GPU parity (bit-exact) of csam_small_regions -- device connected components for hole filling and island
removal -- against the oracle restatement of amg.py:267-291 / crowdsam/model.py:394-443."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _blobs(n, H, W, seed, k=21, thr=0.0):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, 1, H, W, generator=g)
    x = torch.nn.functional.avg_pool2d(x, k, 1, k // 2)
    return (x[:, 0] > thr * x.std()).numpy()


def _oracle(masks, min_area):
    from oracle import pipeline_oracle as po
    outs, changed = [], []
    for m in masks:
        m1, c1 = po.remove_small_regions(m, min_area, "holes")
        m2, c2 = po.remove_small_regions(m1, min_area, "islands")
        outs.append(np.asarray(m2, dtype=bool))
        changed.append(int(c1 or c2))
    outs = np.stack(outs)
    boxes = po.batched_mask_to_box(torch.as_tensor(outs)).float().numpy()
    return outs, np.array(changed), boxes


def _check(cuda, masks, min_area):
    """Both device forms against the oracle: the two-level form with per-pixel labels (csam_small_regions) and the compact
    ring-forest form the driver uses (csam_small_regions_idx), the latter in place, through an index list that picks the
    masks out of a larger store in shuffled order, and once more into a separate output store."""
    from crowdsam_amd import hip
    ro, rc, rb = _oracle(masks, min_area)
    out, changed, boxes = hip.small_regions(torch.as_tensor(masks).to(cuda), min_area)
    assert np.array_equal(changed.cpu().numpy(), rc)
    assert np.array_equal(out.cpu().numpy().astype(bool), ro)
    assert np.array_equal(boxes.cpu().numpy(), rb)
    n = len(masks)
    rng = np.random.RandomState(n)
    slots = rng.permutation(n + 3)[:n].astype(np.int32)              # store of n + 3 masks, survivors scattered
    store = torch.zeros((n + 3,) + masks.shape[1:], dtype=torch.uint8, device=cuda)
    store[torch.as_tensor(slots).long().to(cuda)] = torch.as_tensor(masks).to(cuda).to(torch.uint8)
    untouched = store.clone()
    idx = torch.as_tensor(slots).to(cuda)
    out2 = torch.full_like(store, 7)
    ch2, bx2 = hip.small_regions_idx(store, idx, min_area, out_store=out2)
    assert torch.equal(store, untouched)                              # input store not written
    assert np.array_equal(ch2.cpu().numpy(), rc) and np.array_equal(bx2.cpu().numpy(), rb)
    assert np.array_equal(out2[idx.long()].cpu().numpy().astype(bool), ro)
    ch3, bx3 = hip.small_regions_idx(store, idx, min_area)            # in place
    assert np.array_equal(ch3.cpu().numpy(), rc) and np.array_equal(bx3.cpu().numpy(), rb)
    assert np.array_equal(store[idx.long()].cpu().numpy().astype(bool), ro)
    rest = np.setdiff1d(np.arange(n + 3), slots)
    assert int(store[torch.as_tensor(rest).long().to(cuda)].sum()) == 0      # slots outside the list untouched


@pytest.mark.parametrize("hw", [(683, 1024), (1024, 683), (97, 130), (64, 64), (5, 200)])
def test_blobs(cuda, hw):
    H, W = hw
    masks = np.concatenate([_blobs(3, H, W, 1, 21), _blobs(3, H, W, 2, 7, 0.5), _blobs(2, H, W, 3, 3, 1.0)])
    _check(cuda, masks, 100)


def test_noise_and_degenerate(cuda):
    H, W = 333, 517
    rng = np.random.RandomState(0)
    noise = rng.rand(4, H, W) > np.array([0.5, 0.3, 0.7, 0.95])[:, None, None]
    empty = np.zeros((1, H, W), bool)
    full = np.ones((1, H, W), bool)
    one = np.zeros((1, H, W), bool)
    one[0, 100, 200] = True                       # single small island: kept as the arg-max fallback
    ties = np.zeros((1, H, W), bool)              # all-small islands with equal areas: first in raster order
    ties[0, 10:13, 300:303] = True
    ties[0, 10:13, 20:23] = True
    ties[0, 200:203, 5:8] = True
    spiral = np.zeros((1, H, W), bool)            # long serpentine component (deep union-find chains)
    for y in range(0, H - 2, 4):
        spiral[0, y, :] = True
        spiral[0, y:y + 4, (W - 1) if (y // 4) % 2 == 0 else 0] = True
    diag = np.zeros((1, H, W), bool)              # 8-connectivity only links
    idx = np.arange(min(H, W))
    diag[0, idx, idx] = True
    diag[0, idx[:-1], idx[:-1] + 2] = True
    masks = np.concatenate([noise, empty, full, one, ties, spiral, diag])
    for min_area in (100, 1, 5000):
        _check(cuda, masks, min_area)


def test_in_place_and_many(cuda):
    from crowdsam_amd import hip
    masks = _blobs(70, 200, 300, 5, 9, 0.3)
    ro, rc, rb = _oracle(masks, 60)
    out, changed, boxes = hip.small_regions(torch.as_tensor(masks).to(cuda), 60)
    assert np.array_equal(out.cpu().numpy().astype(bool), ro)
    assert np.array_equal(changed.cpu().numpy(), rc)
    assert np.array_equal(boxes.cpu().numpy(), rb)


def test_tile_edges(cuda):
    """Patterns aimed at the 64 x 64 tile decomposition of the labelling (tile-local union-find + cross-edge links with one
    union per contact run): components that live only on tile edges and corners, anti-diagonals through tile corners, thin
    chains of single-pixel tile components, a frame-spanning background, sizes that are not multiples of 64."""
    H, W = 200, 330
    m = []
    a = np.zeros((H, W), bool)                    # lines ON the tile edges (rows / columns 63, 64, 127, 128 ...)
    a[63::64, :] = True
    a[:, 64::64] = True
    m.append(a)
    b = np.zeros((H, W), bool)                    # anti-diagonal: crosses tile corners NE <-> SW (8-connectivity only)
    idx = np.arange(min(H, W))
    b[idx, W - 1 - idx] = True
    b[idx[:-1], W - 3 - idx[:-1]] = True
    m.append(b)
    c = np.zeros((H, W), bool)                    # isolated pixels in the four corners of every tile + a staircase
    for dy in (0, 63):
        for dx in (0, 63):
            c[dy::64, dx::64] = True
    for k in range(0, min(H, W) - 1):
        c[k, k] = True
        c[k, k + 1] = True
    m.append(c)
    d = np.ones((H, W), bool)                     # background-like: everything set except small holes on the edges
    d[60:68, 60:68] = False
    d[126:130, 10:300:7] = False
    d[5:195:9, 127:129] = False
    m.append(d)
    rng = np.random.RandomState(3)
    e = rng.rand(H, W) > 0.45                     # noise with a band of single-pixel-wide vertical strokes over an edge
    e[:, 62:66] = False
    e[::2, 63] = True
    e[1::2, 64] = True
    m.append(e)
    masks = np.stack(m)
    for min_area in (100, 3, 40000):
        _check(cuda, masks, min_area)
    big = np.stack([np.kron(rng.rand(16, 16) > 0.5, np.ones((64, 64), bool)),       # whole tiles on / off, 1024 x 1024
                    rng.rand(1024, 1024) > 0.5])
    _check(cuda, big, 100)


def test_trivial_tiles(cuda):
    """The fast paths of the compact form: tiles without a work pixel and tiles made of nothing else skip the labelling.
    Person-like masks (most tiles empty in the islands pass, full in the holes pass), tile-aligned 64 x 64 islands and
    holes whose area (4096) sits on either side of min_area, full tiles glued to ragged neighbours, frame sizes whose
    last tile column / row is partial (such a tile is never 'full')."""
    H, W = 512, 640
    yy, xx = np.mgrid[:H, :W]
    m = []
    a = ((yy - 250) / 180.0) ** 2 + ((xx - 300) / 90.0) ** 2 < 1.0           # a person-sized ellipse ...
    a[240:246, 290:300] = False                                               # ... with a small hole
    a[10:14, 600:606] = True                                                  # ... and a far-away speck
    m.append(a)
    b = np.zeros((H, W), bool)                                                # aligned 64 x 64 island next to a bigger body
    b[64:128, 128:192] = True
    b[300:500, 100:400] = True
    m.append(b)
    c = np.ones((H, W), bool)                                                 # aligned 64 x 64 hole and a ragged one
    c[128:192, 256:320] = False
    c[320:390, 330:397] = False
    m.append(c)
    d = np.zeros((H, W), bool)                                                # only full tiles: the largest is kept by the fallback
    d[0:64, 0:64] = True
    d[192:256, 320:448] = True
    m.append(d)
    e = np.zeros((H, W), bool)                                                # full tiles with a one-pixel bridge through a corner
    e[64:128, 64:128] = True
    e[128:192, 128:192] = True
    e[256:320, 576:640] = True                                                # last (full) column of tiles
    e[448:512, 0:64] = True
    m.append(e)
    masks = np.stack(m)
    for min_area in (100, 4096, 4097, 9000):
        _check(cuda, masks, min_area)
    Hp, Wp = 200, 330                                                          # partial last tiles
    f = np.ones((2, Hp, Wp), bool)
    f[1, 192:200, :] = False
    f[0, 100:164, 300:330] = False
    g = np.zeros((2, Hp, Wp), bool)
    g[0, 128:200, 256:330] = True
    g[1, 0:64, 320:330] = True
    for min_area in (100, 3000):
        _check(cuda, np.concatenate([f, g]), min_area)


def _persons(n, H, W, seed):
    """Person-like masks: filled ellipses with pinholes, bigger holes, specks of every size around min_area inside the box,
    some touching the frame border, one empty, one frame-filling noise mask."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    masks = np.zeros((n, H, W), bool)
    for i in range(n):
        if i == 3:
            continue                                                       # empty mask
        if i == 5:
            masks[i] = _blobs(1, H, W, seed + 1, k=9)[0]                  # frame-filling noise
            continue
        near_edge = i % 7 == 0
        cy = rs.uniform(0, 40) if near_edge else rs.uniform(120, H - 120)
        cx = rs.uniform(W - 30, W) if (near_edge and i % 14 == 0) else rs.uniform(60, W - 60)
        ay, ax = rs.uniform(25, 110), rs.uniform(10, 45)
        m = ((yy - cy) / ay) ** 2 + ((xx - cx) / ax) ** 2 <= 1.0
        for _ in range(6):                                                  # holes of 1 .. ~200 pixels
            hy, hx, s = int(cy + rs.uniform(-0.6, 0.6) * ay), int(cx + rs.uniform(-0.5, 0.5) * ax), int(rs.randint(1, 15))
            m[max(hy, 0):hy + s, max(hx, 0):hx + s] = False
        if m.any():                                                         # specks INSIDE the box corners (islands)
            ys, xs = np.nonzero(m)
            y0, y1, x0, x1 = ys.min(), ys.max(), xs.min(), xs.max()
            for _ in range(4):
                s = int(rs.randint(1, 13))
                sy, sx = int(rs.randint(y0, max(y0 + 1, y1 - s))), int(rs.choice([x0, max(x0, x1 - s)]))
                m[sy:sy + s, sx:sx + s] = True
        masks[i] = m
    return masks


@pytest.mark.parametrize("hw,min_area", [((768, 1024), 100), ((1024, 1024), 100), ((500, 700), 30), ((768, 1024), 250)])
def test_windowed_cleanup_equals_full_frame_cleanup(cuda, hw, min_area):
    """hip.small_regions_windowed (round 4): clean-up inside the masks' padded bounding boxes must give the masks, changed
    flags and boxes of the full-frame clean-up -- and of the oracle (amg.py:267-291 via crowdsam/model.py:394-443) -- bit
    for bit, for person-sized masks, masks cut by the frame border, an empty mask and a frame-filling one."""
    from crowdsam_amd import hip
    from oracle import pipeline_oracle as po
    H, W = hw
    n = 24
    masks = _persons(n, H, W, seed=H + min_area)
    ro, rc, rb = _oracle(masks[:8], min_area)                               # the oracle is slow: 8 masks against it ...
    slots = np.random.RandomState(1).permutation(n + 2)[:n].astype(np.int32)
    idx = torch.as_tensor(slots).to(cuda)
    base = torch.zeros((n + 2, H, W), dtype=torch.uint8, device=cuda)
    base[idx.long()] = torch.as_tensor(masks).to(cuda).to(torch.uint8)
    boxes_in = po.batched_mask_to_box(torch.as_tensor(masks)).to(cuda)
    a = base.clone()
    ch_a, bx_a = hip.small_regions_idx(a, idx, min_area)                    # ... all of them against the full-frame kernel
    b = base.clone()
    ch_b, bx_b = hip.small_regions_windowed(b, idx, boxes_in, min_area)
    assert torch.equal(a, b)
    assert torch.equal(ch_a, ch_b) and torch.equal(bx_a, bx_b)
    assert int(ch_a.sum()) >= 8                                             # the edits are not vacuous
    assert np.array_equal(b[idx.long()[:8]].cpu().numpy().astype(bool), ro)
    assert np.array_equal(ch_b.cpu().numpy()[:8], rc) and np.array_equal(bx_b.cpu().numpy()[:8], rb)


def _edge_cases(n, H, W, seed):
    """Rectangles and ellipses pushed against every frame edge and corner (0 .. 20 pixels away, so cut, partly cut and uncut
    windows all occur), with notches open to the frame edge, holes and specks of 1 .. 200 pixels right at the edge."""
    rs = np.random.RandomState(seed)
    masks = np.zeros((n, H, W), bool)
    for i in range(n):
        h, w = int(rs.randint(20, 200)), int(rs.randint(12, 120))
        gap_y, gap_x = int(rs.choice([0, 0, 1, 3, 15, 16, 17, 20])), int(rs.choice([0, 0, 1, 3, 15, 16, 17, 20]))
        side = i % 8                                                      # N, S, W, E, NW, NE, SW, SE
        y0 = gap_y if side in (0, 4, 5) else (H - h - gap_y if side in (1, 6, 7) else int(rs.randint(40, H - h - 40)))
        x0 = gap_x if side in (2, 4, 6) else (W - w - gap_x if side in (3, 5, 7) else int(rs.randint(40, W - w - 40)))
        m = masks[i]
        m[y0:y0 + h, x0:x0 + w] = True
        for _ in range(10):                                               # holes / notches hugging the rectangle's outline
            s, t = int(rs.randint(1, min(15, h // 2))), int(rs.randint(1, min(15, w // 2)))
            ey = int(rs.choice([y0, y0 + h - s, rs.randint(y0, y0 + h - s + 1)]))
            ex = int(rs.choice([x0, x0 + w - t, rs.randint(x0, x0 + w - t + 1)]))
            m[ey:ey + s, ex:ex + t] = False
            if rs.rand() < 0.5:                                           # ... closed again by a 1-pixel wall on the outline
                m[ey, ex:ex + t] = True
                m[ey:ey + s, ex] = True
        ys, xs = np.nonzero(m)
        by0, by1, bx0, bx1 = ys.min(), ys.max(), xs.min(), xs.max()
        for _ in range(6):                                                # specks inside the box, some on the frame border
            s = int(rs.randint(1, 13))
            sy = int(rs.choice([by0, max(by0, by1 - s + 1), rs.randint(by0, max(by0 + 1, by1 - s + 1))]))
            sx = int(rs.choice([bx0, max(bx0, bx1 - s + 1)]))
            m[sy:min(sy + s, by1 + 1), sx:min(sx + s, bx1 + 1)] = True
    return masks


@pytest.mark.parametrize("hw,min_area,seed", [((768, 1024), 100, 0), ((600, 800), 100, 1), ((333, 517), 256, 2),
                                              ((1024, 1024), 30, 3)])
def test_windowed_cleanup_at_the_frame_border(cuda, hw, min_area, seed):
    """Windows cut by a frame edge keep that edge (they are placed flush with the stack's edge on that side): holes, notches and
    specks that touch the frame border must be treated exactly as the full-frame labelling treats them (amg.py:267-291 counts
    a component that touches the border like any other)."""
    from crowdsam_amd import hip
    from oracle import pipeline_oracle as po
    H, W = hw
    n = 48
    masks = _edge_cases(n, H, W, seed)
    ro, rc, rb = _oracle(masks[:8], min_area)
    idx = torch.arange(n, dtype=torch.int32, device=cuda)
    base = torch.as_tensor(masks).to(cuda).to(torch.uint8)
    boxes_in = po.batched_mask_to_box(torch.as_tensor(masks)).to(cuda)
    a = base.clone()
    ch_a, bx_a = hip.small_regions_idx(a, idx, min_area)
    b = base.clone()
    ch_b, bx_b = hip.small_regions_windowed(b, idx, boxes_in, min_area)
    bad = [i for i in range(n) if not torch.equal(a[i], b[i])]
    assert not bad, bad
    assert torch.equal(ch_a, ch_b) and torch.equal(bx_a, bx_b)
    assert int(ch_a.sum()) >= 24
    assert np.array_equal(b[:8].cpu().numpy().astype(bool), ro)
    assert np.array_equal(ch_b.cpu().numpy()[:8], rc) and np.array_equal(bx_b.cpu().numpy()[:8], rb)
