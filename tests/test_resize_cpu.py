"""Row a1: the frame resizers.  cv2.resize (INTER_LINEAR, uint8; /root/reference/crowdsam/utils.py:149) is third-party
and absent from this image -> the oracle restatement (oracle/resize_oracle.py) is checked against HAND-DERIVED vectors
of OpenCV's published fixed-point algorithm; Pillow's bilinear (transforms.py:26-31 via torchvision) IS in the image
-> its restatement is pinned bit-exactly against Pillow itself.  The product's coefficient tables
(crowdsam_amd/resize.py) must equal the oracle's."""
import os
import sys

import numpy as np
from PIL import Image

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crowdsam_amd import resize as pr  # noqa: E402
from oracle import resize_oracle as ro  # noqa: E402


def _row(vals):
    return np.array(vals, np.uint8)[None, :, None].repeat(3, 2)


def test_cv2_upscale_2x_hand_vector():
    # scale 0.5: dx=0 -> fx=-0.25 -> clamped (sx=0, f=0); dx=1 -> f=.25 -> (1536, 512): (0*1536+100*512)>>4 = 3200,
    # (2048*3200)>>16 = 100, (100+2)>>2 = 25; dx=2 -> (512,1536): 9600 -> 300 -> 75; dx=3 -> sx=1=w-1 -> 100
    out = ro.cv2_resize_linear_u8(_row([0, 100]), (4, 1))
    assert out[0, :, 0].tolist() == [0, 25, 75, 100] and np.array_equal(out[..., 0], out[..., 2])


def test_cv2_downscale_4_to_3_hand_vector():
    # scale 4/3.  dx=0: f=1/6 -> coefficients (1707, 341): 10*1707+20*341 = 23890 -> >>4 = 1493 -> (2048*1493)>>16 = 46
    # -> (46+2)>>2 = 12;  dx=1: f=.5 -> (1024,1024): 51200 -> 3200 -> 100 -> 25;  dx=2: f=5/6 -> (341, 1707):
    # 30*341+250*1707 = 436980 -> 27311 -> 853 -> 213
    out = ro.cv2_resize_linear_u8(_row([10, 20, 30, 250]), (3, 1))
    assert out[0, :, 1].tolist() == [12, 25, 213]
    xs, xa = ro.cv2_linear_tables(4, 3)
    assert xs.tolist() == [0, 1, 2] and xa.tolist() == [[1707, 341], [1024, 1024], [341, 1707]]


def test_cv2_vertical_pass_and_truncation_bias():
    col = _row([0, 100]).transpose(1, 0, 2)                      # 2 rows x 1 col
    # dy=0: fy=-0.25 -> rows clamp to (0,0) with beta (1-0.75.., ..) kept: both rows 0 -> 0; dy=1: beta (1536,512):
    # ((512*(204800>>4))>>16) = 100 -> (100+2)>>2 = 25; dy=3: rows (1,1): 300 + 100 -> 100
    assert ro.cv2_resize_linear_u8(col, (1, 4))[:, 0, 0].tolist() == [0, 25, 75, 100]
    # both passes fractional: horizontal 5120; vertical (1536*320)>>16 = 7, (512*320)>>16 = 2 -> (9+2)>>2 = 2 (exact 2.5)
    g = np.array([[0, 10], [0, 10]], np.uint8)[:, :, None].repeat(3, 2)
    assert ro.cv2_resize_linear_u8(g, (4, 4))[1, 1, 0] == 2


def test_cv2_exact_2x_decimation_is_area_and_same_size_is_copy():
    a = np.array([[1, 2, 9, 9], [3, 5, 9, 10]], np.uint8)[:, :, None].repeat(3, 2)
    assert ro.cv2_resize_linear_u8(a, (2, 1))[0, :, 0].tolist() == [(1 + 2 + 3 + 5 + 2) >> 2, (9 + 9 + 9 + 10 + 2) >> 2]
    b = ro.cv2_resize_linear_u8(a, (4, 2))
    assert np.array_equal(a, b) and b is not a
    c = np.full((17, 23, 3), 201, np.uint8)
    assert np.unique(ro.cv2_resize_linear_u8(c, (64, 40))).tolist() == [201]       # constants are preserved


def test_resize_image_trap9_shape_and_range():
    # int(r*w) lands on 1023 for ~12 % of widths (SURVEY.md trap 9): w = 1366, h = 768 -> r = 1024/1366
    rs = np.random.RandomState(0)
    hits = [w for w in range(1025, 2200) if int(min(1024 / w, 1024 / 700) * w) == 1023]
    assert hits
    w = hits[0]
    img = rs.randint(0, 256, (700, w, 3)).astype(np.uint8)
    out, r = ro.resize_image(img, 1024)
    assert out.shape == (int(r * 700), 1023, 3)
    big = np.array(Image.fromarray(img).resize((1023, out.shape[0]), Image.BILINEAR)).astype(int)
    assert np.abs(out.astype(int) - big).mean() < 40       # same picture (PIL antialiases when shrinking; loose)


def test_pil_restatement_is_bit_exact_vs_pillow():
    rs = np.random.RandomState(1)
    for (h, w), (nh, nw) in [((682, 1023), (683, 1024)), ((100, 150), (37, 80)), ((64, 48), (128, 96)),
                             ((445, 640), (712, 1024)), ((31, 17), (31, 40))]:
        img = rs.randint(0, 256, (h, w, 3)).astype(np.uint8)
        ref = np.array(Image.fromarray(img).resize((nw, nh), Image.BILINEAR))
        assert np.array_equal(ro.pil_resize_bilinear_u8(img, (nh, nw)), ref), ((h, w), (nh, nw))


def test_product_tables_equal_oracle_tables():
    for (sh, sw), (dh, dw) in [((1500, 1500), (1024, 1024)), ((512, 512), (1024, 1024)), ((445, 640), (712, 1024)),
                               ((1080, 1920), (576, 1024)), ((700, 1366), (524, 1023))]:
        xofs, xcoef, yofs, ycoef = pr.cv2_linear_tables(sh, sw, dh, dw)
        xs, xa = ro.cv2_linear_tables(sw, dw)
        r0, r1, yb = ro.cv2_linear_tables_y(sh, dh)
        assert np.array_equal(xofs, xs) and np.array_equal(xcoef, xa)
        assert np.array_equal(yofs, np.stack([r0, r1], 1)) and np.array_equal(ycoef, yb)
    assert pr.cv2_linear_tables(2048, 1536, 1024, 768) is None


def test_product_pil_tables_equal_oracle_tables():
    for a, b in [(1023, 1024), (682, 683), (100, 150), (640, 1024), (512, 1024), (900, 1024)]:
        xm, nt, co = pr.pil_bilinear_tables(a, b)
        for i, (x0, n, k) in enumerate(ro.pil_bilinear_coeffs(a, b)):
            assert xm[i] == x0 and nt[i] == n and np.array_equal(co[i, :n], k), (a, b, i)
