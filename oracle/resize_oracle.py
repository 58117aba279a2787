"""ORACLE (test infrastructure, NOT product code): the two uint8 frame resizers of the reference path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.

* ``cv2_resize_linear_u8`` -- what ``cv2.resize(image, (w, h))`` computes for uint8 HWC images (the call at
  /root/reference/crowdsam/utils.py:149, default interpolation INTER_LINEAR).  OpenCV is a third-party wheel that is
  NOT in this image and is unpinned by the reference (requirements.txt:1-10) => **parity unpinned**: this is a
  restatement of the published OpenCV 4.x algorithm (modules/imgproc/src/resize.cpp, generic path; the IPP path is
  disabled for 8u-linear unless useIPP_NotExact()):
    - half-pixel centres: fx = float((dx + 0.5) * scale - 0.5), scale = 1.0 / (dst / src) in double;
      sx = floor(fx); fx -= sx; left clamp (sx < 0 -> sx = 0, fx = 0); right clamp (sx >= w-1 -> sx = w-1, fx = 0);
    - coefficients as 11-bit fixed point: short(rint(c * 2048)) for (1 - f, f), round-half-even;
    - horizontal pass into int32: S[sx] * a0 + S[sx+1] * a1;
    - vertical pass on rows clamped to [0, h-1] with the unclamped beta pair:
      dst = (((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
    - exact 2x2 decimation (both scales exactly 2) is re-routed to INTER_AREA: (a + b + c + d + 2) >> 2;
    - same size: a copy.
  Pinned here only by hand-derived vectors (tests/test_resize_cpu.py), not by OpenCV itself.
* ``pil_resize_bilinear_u8`` -- Pillow's ``Image.resize(size, BILINEAR)`` (two-pass, support widened when shrinking,
  22-bit fixed-point coefficients), which torchvision's ``resize(to_pil_image(image), size)`` reaches at
  /root/reference/segment_anything_cs/utils/transforms.py:26-31.  Pillow IS in this image, so this restatement is
  pinned bit-exactly against it in tests/test_resize_cpu.py.
"""
import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS


def cv2_linear_tables(src, dst):
    """(ofs int32 [dst], coef int16 [dst, 2]) of one axis, exactly as cv::resize builds them for INTER_LINEAR."""
    inv_scale = np.float64(dst) / np.float64(src)
    scale = np.float64(1.0) / inv_scale
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo], s[lo] = 0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0, src - 1
    c0 = np.rint((np.float32(1.0) - f) * np.float32(COEF_SCALE)).astype(np.int16)   # np.rint: round-half-even
    c1 = np.rint(f * np.float32(COEF_SCALE)).astype(np.int16)
    return s, np.stack([c0, c1], 1)


def cv2_linear_tables_y(src, dst):
    """Row tables: the fractional part is NOT reset at the borders; the row indices are clamped instead."""
    inv_scale = np.float64(dst) / np.float64(src)
    scale = np.float64(1.0) / inv_scale
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = (f - s.astype(np.float32)).astype(np.float32)
    c0 = np.rint((np.float32(1.0) - f) * np.float32(COEF_SCALE)).astype(np.int16)
    c1 = np.rint(f * np.float32(COEF_SCALE)).astype(np.int16)
    r0 = np.clip(s, 0, src - 1)
    r1 = np.clip(s + 1, 0, src - 1)
    return r0, r1, np.stack([c0, c1], 1)


def cv2_resize_linear_u8(img, dsize_wh):
    """img uint8 [h, w, c] (or [h, w]); dsize_wh = (width, height) as cv2.resize takes it."""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    squeeze = img.ndim == 2
    if squeeze:
        img = img[:, :, None]
    h, w = img.shape[:2]
    dw, dh = int(dsize_wh[0]), int(dsize_wh[1])
    if (dw, dh) == (w, h):
        out = img.copy()
    elif w == 2 * dw and h == 2 * dh:
        s = img.astype(np.int32)
        out = ((s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2).astype(np.uint8)
    else:
        xs, xa = cv2_linear_tables(w, dw)
        r0, r1, yb = cv2_linear_tables_y(h, dh)
        x1 = np.minimum(xs + 1, w - 1)
        s = img.astype(np.int32)
        hor = s[:, xs, :] * xa[None, :, 0, None].astype(np.int32) + s[:, x1, :] * xa[None, :, 1, None].astype(np.int32)
        b0 = yb[:, 0].astype(np.int32)[:, None, None]
        b1 = yb[:, 1].astype(np.int32)[:, None, None]
        v = (((b0 * (hor[r0] >> 4)) >> 16) + ((b1 * (hor[r1] >> 4)) >> 16) + 2) >> 2
        out = v.astype(np.uint8)
    return out[:, :, 0] if squeeze else out


def resize_image(image, max_size):
    """/root/reference/crowdsam/utils.py:141-149 for ndarray images."""
    h, w = image.shape[:2]
    r = min(max_size / w, max_size / h)
    nh, nw = int(r * h), int(r * w)
    return cv2_resize_linear_u8(image, (nw, nh)), r


# ------------------------------------------------------------------------------------------------
# Pillow ImagingResample, BILINEAR, 8 bits per channel
# ------------------------------------------------------------------------------------------------
PIL_PRECISION_BITS = 32 - 8 - 2


def pil_bilinear_coeffs(in_size, out_size):
    """Per output index: (xmin, n, int32 coefficients[n]) as Resample.c::precompute_coeffs + normalize_coeffs_8bpc."""
    scale = filterscale = np.float64(in_size) / np.float64(out_size)
    if filterscale < 1.0:
        filterscale = np.float64(1.0)
    support = 1.0 * filterscale
    ss = 1.0 / filterscale
    out = []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        n = xmax - xmin
        xs = (np.arange(n, dtype=np.float64) + xmin - center + 0.5) * ss
        wgt = np.where(np.abs(xs) < 1.0, 1.0 - np.abs(xs), 0.0)
        ww = wgt.sum()
        if ww != 0.0:
            wgt = wgt / ww
        k = np.where(wgt < 0, (-0.5 + wgt * (1 << PIL_PRECISION_BITS)).astype(np.int32),
                     (0.5 + wgt * (1 << PIL_PRECISION_BITS)).astype(np.int32))
        out.append((xmin, n, k))
    return out


def _pil_pass(a, coeffs, axis):
    a = np.moveaxis(a, axis, 0)
    res = np.empty((len(coeffs),) + a.shape[1:], np.uint8)
    for o, (xmin, n, k) in enumerate(coeffs):
        acc = np.full(a.shape[1:], 1 << (PIL_PRECISION_BITS - 1), np.int64)
        for j in range(n):
            acc += a[xmin + j].astype(np.int64) * int(k[j])
        res[o] = np.clip(acc >> PIL_PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(res, 0, axis)


def pil_resize_bilinear_u8(img, size_hw):
    """PIL.Image.fromarray(img).resize((w, h), BILINEAR) for uint8 [h, w, c]: horizontal pass, then vertical pass,
    each rounding to uint8 (Resample.c::ImagingResampleInner); a pass whose size does not change is skipped."""
    img = np.asarray(img)
    assert img.dtype == np.uint8
    h, w = img.shape[:2]
    nh, nw = size_hw
    out = img
    if nw != w:
        out = _pil_pass(out, pil_bilinear_coeffs(w, nw), 1)
    if nh != h:
        out = _pil_pass(out, pil_bilinear_coeffs(h, nh), 0)
    return out.copy() if out is img else out
