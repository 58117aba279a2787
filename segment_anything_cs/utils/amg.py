"""Mask-data utilities of the reference's AMG module that Crowd-SAM uses
(reference: segment_anything_cs/utils/amg.py; imported at crowdsam/model.py:13-22).

``MaskData`` is the API type ``CrowdSAM.generate`` returns.  The tensor reductions
(stability score, mask->box, RLE) run as HIP kernels when handed CUDA tensors; the driver's hot loop
does not even call them -- it uses the fused csam_mask_post kernel -- they exist for API parity.
Host-side pieces (connected components, COCO string packing) are host code in the reference too
(cv2 / pycocotools there; scipy / an own encoder here).
"""
import math
from copy import deepcopy
from itertools import product

import numpy as np
import torch

from crowdsam_amd import hip


class MaskData:
    """Dict of per-mask fields (lists, ndarrays or tensors) with batched filter / cat."""

    def __init__(self, **kwargs):
        for v in kwargs.values():
            self._check(v)
        self._stats = dict(**kwargs)

    @staticmethod
    def _check(v):
        assert isinstance(v, (list, np.ndarray, torch.Tensor)), \
            "MaskData only supports list, numpy arrays, and torch tensors."

    def __setitem__(self, key, item):
        self._check(item)
        self._stats[key] = item

    def __delitem__(self, key):
        del self._stats[key]

    def __getitem__(self, key):
        return self._stats[key]

    def __contains__(self, key):
        return key in self._stats

    def items(self):
        return self._stats.items()

    def filter(self, keep):
        for k, v in self._stats.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                self._stats[k] = v[torch.as_tensor(keep, device=v.device)]
            elif isinstance(v, np.ndarray):
                self._stats[k] = v[keep.detach().cpu().numpy()]
            elif isinstance(v, list):
                if keep.dtype == torch.bool:
                    self._stats[k] = [a for a, f in zip(v, keep.tolist()) if f]
                else:
                    self._stats[k] = [v[i] for i in keep.tolist()]
            else:
                raise TypeError(f"MaskData key {k} has an unsupported type {type(v)}.")

    @staticmethod
    def _copy(v):
        """deepcopy of amg.py:72-83, except that the per-mask RLE dicts of a list are copied one level deep: their run-length
        arrays are never modified in place, and deep-copying several hundred of them per crowded frame was 3 ms of host time"""
        if isinstance(v, list) and v and all(isinstance(r, dict) for r in v):
            return [dict(r) for r in v]
        return deepcopy(v)

    def cat(self, new_stats):
        for k, v in new_stats.items():
            if k not in self._stats or self._stats[k] is None:
                self._stats[k] = self._copy(v)
            elif isinstance(v, torch.Tensor):
                self._stats[k] = torch.cat([self._stats[k], v], dim=0)
            elif isinstance(v, np.ndarray):
                self._stats[k] = np.concatenate([self._stats[k], v], axis=0)
            elif isinstance(v, list):
                self._stats[k] = self._stats[k] + self._copy(v)
            else:
                raise TypeError(f"MaskData key {k} has an unsupported type {type(v)}.")

    def to_numpy(self):
        for k, v in self._stats.items():
            if isinstance(v, torch.Tensor):
                self._stats[k] = v.detach().cpu().numpy()


def batch_iterator(batch_size, *args):
    assert len(args) > 0 and all(len(a) == len(args[0]) for a in args), \
        "Batched iteration must have inputs of all the same size."
    for start in range(0, len(args[0]), batch_size):
        yield [a[start:start + batch_size] for a in args]


def generate_crop_boxes(im_size, n_layers, overlap_ratio):
    """Layer i has (2**i)**2 crops; layer 0 is the whole image (amg.py:200-234)."""
    im_h, im_w = im_size
    short_side = min(im_h, im_w)
    crop_boxes, layer_idxs = [[0, 0, im_w, im_h]], [0]

    def span(length, n, overlap):
        return int(math.ceil((overlap * (n - 1) + length) / n))

    for layer in range(n_layers):
        n = 2 ** (layer + 1)
        overlap = int(overlap_ratio * short_side * (2 / n))
        cw, ch = span(im_w, n, overlap), span(im_h, n, overlap)
        x0s = [int((cw - overlap) * i) for i in range(n)]
        y0s = [int((ch - overlap) * i) for i in range(n)]
        for x0, y0 in product(x0s, y0s):
            crop_boxes.append([x0, y0, min(x0 + cw, im_w), min(y0 + ch, im_h)])
            layer_idxs.append(layer + 1)
    return crop_boxes, layer_idxs


def calculate_stability_score(masks, mask_threshold, threshold_offset):
    """IoU of the masks thresholded at +-offset (amg.py:156-176); int32 counts -> fp32 ratio."""
    hi = (masks > (mask_threshold + threshold_offset)).flatten(-2).sum(-1, dtype=torch.int32)
    lo = (masks > (mask_threshold - threshold_offset)).flatten(-2).sum(-1, dtype=torch.int32)
    return hi / lo


def batched_mask_to_box(masks):
    """XYXY boxes (inclusive max index) around boolean masks [..., H, W]; empty -> zeros."""
    if masks.numel() == 0:
        return torch.zeros(*masks.shape[:-2], 4, device=masks.device)
    shape = masks.shape
    h, w = shape[-2:]
    m = masks.reshape(-1, h, w).bool()
    rows, cols = m.any(-1), m.any(-2)
    ar_h = torch.arange(h, device=m.device)
    ar_w = torch.arange(w, device=m.device)
    bottom = (rows * ar_h).amax(-1)
    top = torch.where(rows, ar_h, h).amin(-1)
    right = (cols * ar_w).amax(-1)
    left = torch.where(cols, ar_w, w).amin(-1)
    out = torch.stack([left, top, right, bottom], -1)
    out = out * ~((right < left) | (bottom < top)).unsqueeze(-1)
    return out.reshape(*shape[:-2], 4)


def _runs_from_positions(pos, first_set, hw):
    """Run lengths (int64 array) from the sorted change positions of a column-major flattening."""
    idx = np.concatenate([[0], pos, [hw]]).astype(np.int64)
    runs = np.diff(idx)
    return np.concatenate([[0], runs]) if first_set else runs


def mask_to_rle_arrays(tensor, idx=None, boxes=None):
    """mask_to_rle_pytorch with the run lengths kept as int64 ndarrays (what the driver feeds straight into
    the C string packer: a crowded frame has 1e5+ runs per mask, Python lists would dominate the tail).
    ``idx`` (device int32 [b]): ``tensor`` is a (cap, h, w) mask store and only the slots named by idx are encoded, in
    that order, where they lie."""
    b, h, w = tensor.shape
    if idx is not None:
        b = int(idx.shape[0])
    if b == 0:
        return []
    if tensor.is_cuda:
        # the kernels require strict 0 / 1 bytes (include/csam.h): a bool tensor is that already, as is the driver's uint8
        # mask store (written by csam_mask_write); anything else is normalised with != 0 first
        if tensor.dtype == torch.bool:
            m8 = tensor.view(torch.uint8).contiguous()
        elif tensor.dtype == torch.uint8 and idx is not None:
            m8 = tensor.contiguous()
        else:
            m8 = (tensor != 0).view(torch.uint8).contiguous()
        # ``boxes`` (build extension): XYXY boxes of the masks (inclusive maxima, batched_mask_to_box) -- the encoder then reads the
        # boxes instead of the frames
        pos, offs = hip.rle_encode(m8, idx, None if boxes is None else boxes.to(torch.int32).contiguous())
        pos = hip.to_host_numpy(pos).astype(np.int64)
        # first pixel of every mask: a strided view of the store's first column, then n bytes gathered (pinned staging buffer)
        col0 = m8.view(m8.shape[0], -1)[:, 0]
        first = hip.to_host_numpy(col0 if idx is None else col0.index_select(0, idx.long())).astype(bool)
        # run lengths of ALL masks in one pass (a crowded frame keeps hundreds of masks: per-mask numpy calls were the
        # tail's largest host cost): per mask the sequence [0 if the first pixel is set] 0 pos... hw, differenced
        offs = np.asarray(offs, dtype=np.int64)
        npos = offs[1:] - offs[:-1]
        seg = npos + 2 + first                        # entries of mask i in the boundary array
        so = np.zeros(b + 1, dtype=np.int64)
        np.cumsum(seg, out=so[1:])
        bounds = np.empty(int(so[-1]), dtype=np.int64)
        lead = so[:-1] + first                        # index of the "0" that precedes the positions
        bounds[so[:-1]] = 0                           # (for first-set masks: the extra leading 0 -> a zero-length run)
        bounds[lead] = 0
        bounds[so[1:] - 1] = h * w
        keep = np.ones(int(so[-1]), dtype=bool)
        keep[so[:-1]] = False
        keep[lead] = False
        keep[so[1:] - 1] = False
        bounds[keep] = pos[: int(offs[-1])]
        runs = np.diff(bounds)                        # the difference across a mask boundary lands on a dropped slot
        runs.setflags(write=False)                    # every mask's counts is a VIEW of this array (MaskData.cat copies shallowly)
        out = []
        for i in range(b):
            out.append({"size": [h, w], "counts": runs[so[i]: so[i + 1] - 1]})
        return out
    arr = tensor.numpy().astype(bool)
    out = []
    for i in range(b):
        flat = arr[i].T.reshape(-1)
        pos = np.flatnonzero(flat[1:] != flat[:-1]) + 1
        out.append({"size": [h, w], "counts": _runs_from_positions(pos, bool(flat[0]), h * w)})
    return out


def mask_to_rle_pytorch(tensor):
    """Uncompressed column-major RLE per mask (amg.py:107-135), counts as Python lists like the reference.
    CUDA input: HIP csam_rle_* kernels with one D2H of the change positions (the reference syncs per mask)."""
    return [{"size": r["size"], "counts": r["counts"].tolist()} for r in mask_to_rle_arrays(tensor)]


def rle_to_mask(rle):
    h, w = rle["size"]
    flat = np.zeros(h * w, dtype=bool)
    idx, val = 0, False
    for c in rle["counts"]:
        if val:
            flat[idx:idx + c] = True
        idx += c
        val = not val
    return flat.reshape(w, h).T


def area_from_rle(rle):
    return sum(rle["counts"][1::2])


def remove_small_regions(mask, area_thresh, mode):
    """Fill holes / drop islands smaller than area_thresh, 8-connectivity (amg.py:267-291; host
    connected components: scipy here, cv2 in the reference).  Returns (mask, modified)."""
    from scipy import ndimage
    assert mode in ["holes", "islands"]
    holes = mode == "holes"
    work = np.logical_xor(holes, mask)
    labels, n = ndimage.label(work, structure=np.ones((3, 3), dtype=np.uint8))
    sizes = np.bincount(labels.ravel(), minlength=n + 1)[1:]
    small = np.flatnonzero(sizes < area_thresh) + 1
    if small.size == 0:
        return mask, False
    if holes:
        fill = np.concatenate([[0], small])
    else:
        fill = np.setdiff1d(np.arange(1, n + 1), small)
        if fill.size == 0:
            fill = np.array([int(np.argmax(sizes)) + 1])
    lut = np.zeros(n + 1, dtype=bool)        # label -> keep (a table lookup; np.isin is slow for many labels)
    lut[fill] = True
    return lut[labels], True


def coco_rle_string(counts):
    """COCO compressed-RLE string of run lengths (pycocotools rleToString: 5 data bits + continuation per char
    offset by 48, runs after the third delta-coded against counts[i-2]).  A crowded frame carries 1e5+ runs, so
    the byte loop is the C host helper csam_coco_rle_string (pycocotools is C in the reference as well)."""
    return hip.coco_rle_string(counts)


def coco_encode_rles(uncompressed_rles):
    """coco_encode_rle for a list of masks with one call into the C string packer; entries whose counts are a string already
    (mask_to_coco_rles: packed on the device) pass through."""
    todo = [i for i, r in enumerate(uncompressed_rles) if not isinstance(r["counts"], str)]
    strings = hip.coco_rle_strings([uncompressed_rles[i]["counts"] for i in todo])
    out = [{"size": list(r["size"]), "counts": r["counts"]} for r in uncompressed_rles]
    for i, st in zip(todo, strings):
        out[i]["counts"] = st
    return out


def mask_to_coco_rles(tensor, idx=None, boxes=None):
    """mask_to_rle_pytorch + coco_encode_rle (amg.py:107-135, 294-300) in one device pass: COCO compressed-RLE dicts of the
    masks tensor[idx] (a (cap, h, w) uint8 mask store of strict 0 / 1 bytes, or bool masks) with the run-length arithmetic and
    the string packing on the GPU (csam_coco_rle_pack) -- one D2H of the string bytes per image instead of the change
    positions.  ``boxes``: XYXY boxes of the masks (inclusive maxima): the scan then reads the boxes instead of the frames."""
    b, h, w = tensor.shape
    if idx is not None:
        b = int(idx.shape[0])
    if b == 0:
        return []
    assert tensor.is_cuda
    if tensor.dtype == torch.bool:
        m8 = tensor.view(torch.uint8).contiguous()
    elif tensor.dtype == torch.uint8 and idx is not None:
        m8 = tensor.contiguous()
    else:
        m8 = (tensor != 0).view(torch.uint8).contiguous()
    strings = hip.rle_coco_strings(m8, idx, None if boxes is None else boxes.to(torch.int32).contiguous())
    return [{"size": [h, w], "counts": st} for st in strings]


def coco_encode_rle(uncompressed_rle):
    h, w = uncompressed_rle["size"]
    return {"size": [h, w], "counts": coco_rle_string(uncompressed_rle["counts"])}
