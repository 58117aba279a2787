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
