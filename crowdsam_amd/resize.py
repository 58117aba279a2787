"""Host side of the device frame resize: the coefficient tables of ``cv2.resize(image, (w, h))`` (INTER_LINEAR, uint8)
as OpenCV 4.x's generic path builds them (reference call: crowdsam/utils.py:149).  OpenCV is an un-vendored,
unpinned dependency of the reference and is not in this image: the algorithm is restated from the published source
(modules/imgproc/src/resize.cpp) -- half-pixel centres in double -> float, 11-bit round-half-even coefficients,
left / right clamps that zero the fraction on the x axis, row clamping with the fraction kept on the y axis.
The kernel (csam_resize_linear_u8) applies them; tests/test_resize_*.py hold hand-derived vectors."""
import functools

import numpy as np
import torch

COEF_SCALE = np.float32(2048.0)


def _axis(src, dst):
    scale = np.float64(1.0) / (np.float64(dst) / np.float64(src))
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int32)
    f = (f - s.astype(np.float32)).astype(np.float32)
    return s, f


def _coef(f):
    return np.stack([np.rint((np.float32(1.0) - f) * COEF_SCALE), np.rint(f * COEF_SCALE)], 1).astype(np.int16)


def cv2_linear_tables(sh, sw, dh, dw):
    """numpy (xofs [dw] i32, xcoef [dw,2] i16, yofs [dh,2] i32, ycoef [dh,2] i16); None for the exact-2x route."""
    if sh == 2 * dh and sw == 2 * dw:
        return None
    xs, fx = _axis(sw, dw)
    lo, hi = xs < 0, xs >= sw - 1
    fx[lo], xs[lo] = 0, 0
    fx[hi], xs[hi] = 0, sw - 1
    ys, fy = _axis(sh, dh)
    yofs = np.stack([np.clip(ys, 0, sh - 1), np.clip(ys + 1, 0, sh - 1)], 1).astype(np.int32)
    return xs.astype(np.int32), _coef(fx), yofs, _coef(fy)


@functools.lru_cache(maxsize=64)
def cv2_linear_tables_device(sh, sw, dh, dw, device):
    t = cv2_linear_tables(sh, sw, dh, dw)
    if t is None:
        return None
    return tuple(torch.from_numpy(np.ascontiguousarray(a)).to(device) for a in t)


# ---- Pillow ImagingResample (BILINEAR, 8 bits per channel): Resample.c precompute_coeffs + normalize_coeffs_8bpc
PIL_PRECISION_BITS = 32 - 8 - 2
PIL_MAXTAPS = 8


def pil_bilinear_tables(in_size, out_size):
    """numpy (xmin [out] i32, ntap [out] i32, coef [out, 8] i32).  support = max(in/out, 1): 2-3 taps when enlarging
    (the 1023 -> 1024 case of SURVEY.md trap 9), up to 2*scale+1 when shrinking (<= 8 taps: scale <= 3.5)."""
    scale = filterscale = np.float64(in_size) / np.float64(out_size)
    if filterscale < 1.0:
        filterscale = np.float64(1.0)
    support = 1.0 * filterscale
    ss = 1.0 / filterscale
    xmin = np.zeros(out_size, np.int32)
    ntap = np.zeros(out_size, np.int32)
    coef = np.zeros((out_size, PIL_MAXTAPS), np.int32)
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        lo = max(int(center - support + 0.5), 0)
        hi = min(int(center + support + 0.5), in_size)
        n = hi - lo
        if n > PIL_MAXTAPS:
            raise ValueError("pil_bilinear_tables: more than %d taps (shrink factor %.2f)" % (PIL_MAXTAPS, scale))
        w = 1.0 - np.abs((np.arange(n, dtype=np.float64) + lo - center + 0.5) * ss)
        w = np.where(w > 0.0, w, 0.0)
        tot = w.sum()
        if tot != 0.0:
            w = w / tot
        k = np.where(w < 0, -0.5 + w * (1 << PIL_PRECISION_BITS), 0.5 + w * (1 << PIL_PRECISION_BITS)).astype(np.int32)
        xmin[xx], ntap[xx] = lo, n
        coef[xx, :n] = k
    return xmin, ntap, coef


@functools.lru_cache(maxsize=64)
def pil_bilinear_tables_device(in_size, out_size, device):
    return tuple(torch.from_numpy(a).to(device) for a in pil_bilinear_tables(in_size, out_size))
