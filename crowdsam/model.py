"""CrowdSAM driver (reference: crowdsam/model.py:24-450) on the MI355X HIP path.

Same constructor / ``generate`` contract and the same ``MaskData`` result fields as the reference
(``boxes, scores, categories, rles, rles_info, points, stability_score, crop_boxes, fboxes``) for
``sam_arch == 'crowdsam'``, ``trainfree == False``.  Control flow restates the Efficient Prompt
Sampler exactly (global NumPy RNG shuffle, ``astype('int')`` truncation, occupancy REPLACED per
batch, permanently shrinking batch size -- SURVEY.md traps 7/8), but the data movement is
re-designed for the GPU:

* per image ONE H2D (the uint8 frame); per EPS batch one small H2D (<=B point coordinates) and one
  small D2H (scores / counts / boxes / keep flags and the occupancy bits of the REMAINING POINTS --
  never the masks);
* only the PWD-Net-selected candidate of each prompt is up-sampled, thresholded, counted and boxed,
  in one fused kernel (csam_mask_post) -- the (B,4,H,W) fp32 tensors of the reference never exist;
* when pruning is impossible (``filter_thresh`` = inf: the dense-sweep benchmark mode) the whole
  sweep is queued without any host synchronisation.
"""
import contextlib
import logging
import math
import os
import time

import numpy as np
import torch

import crowdsam.utils as utils
from crowdsam_amd import hip, trace
from crowdsam_amd.dino import DinoV2
from segment_anything_cs.utils.amg import (MaskData, coco_encode_rle, coco_encode_rles, generate_crop_boxes,
                                           mask_to_coco_rles, mask_to_rle_arrays)


def box_area(b):
    return (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])


def batched_nms(boxes, scores, idxs, iou_threshold):
    """torchvision.ops.batched_nms for the single-category use of the reference (all idxs zero)."""
    return hip.box_nms(boxes, scores, iou_threshold)


_TIMING = os.environ.get("CSAM_TIMING", "0") == "1"
_STAGES = ("set_image", "sample_prompts", "eps_sweep", "gather", "nms", "small_regions", "rle")   # _process_crop, in order


def settle_host():
    """Host hygiene for a per-image loop (tools/test.py, tools/batch_eval.py, bench.py call it once the model is built and
    warm): everything alive at this point -- modules, weights, plans, captured graphs -- is long-lived, so it is moved out of
    the garbage collector's generations (gc.freeze).  CPython otherwise walks all of it in the full collection it starts
    every few thousand container allocations: one 100 ms frame in ~90 on the bench's loop (profiles/r05_gc_stall.txt).
    Returns the number of objects frozen."""
    import gc
    gc.collect()
    gc.freeze()
    return gc.get_freeze_count()


def settled(results, after=2):
    """Wrap a stream of results (CrowdSAM.generate_stream): once ``after`` frames are through -- plans, workspaces and graphs are
    created lazily by the first frames -- settle_host() runs again, so what the warm-up allocated is frozen too (the
    call right after construction froze a cold model).  gc.freeze() is cumulative and permanent for the process: cycles among
    frozen objects are never collected (a model deleted later keeps its memory until exit); gc.unfreeze() undoes it."""
    for n, r in enumerate(results, 1):
        yield r
        if n == after:
            settle_host()


def profile(on=True, ranges=True):
    """tools/test.py --profile: per-stage times into CrowdSAM.timings (device-synchronised, so the stages no longer overlap)
    and roctx ranges for rocprofv3 --marker-trace (the reference has neither).  ``ranges`` alone
    (on=False) keeps the loop asynchronous."""
    global _TIMING
    _TIMING = bool(on)
    if ranges:
        trace.enable()
    else:
        trace.disable()
_WORK_STREAM = True          # generate() on its own high-priority stream (False: the caller's stream; tests toggle it)
_WINDOWED_REGIONS = True      # small-region clean-up and RLE inside the masks' boxes (False: whole frames)


class CrowdSAM:
    vis_img_id = 0

    def _tick(self, name, t0):
        """End of stage ``name``.  Per-stage wall time (ms) into self.timings when CSAM_TIMING=1 / profile(True) (adds device
        syncs: diagnostics only); with roctx ranges on (crowdsam_amd.trace) the stage's range closes and the next one opens."""
        if _TIMING:
            torch.cuda.synchronize()
            self.timings[name] = self.timings.get(name, 0.0) + (time.perf_counter() - t0) * 1e3
        if trace.enabled:
            if name in _STAGES:
                trace.pop()
                k = _STAGES.index(name) + 1
                if k < len(_STAGES):
                    trace.push(_STAGES[k])
            else:
                trace.mark(name)
        return time.perf_counter()

    def __init__(self, config, logger=None, sam_state_dict=None, dino_state_dict=None, dino_depth=24,
                 dino_model=None):
        """``config`` is the reference's YAML dict.  Checkpoints are read from the config paths as in
        the reference (crowdsam/model.py:33-42, 88-115); ``sam_state_dict`` / ``dino_state_dict``
        override them (synthetic weights: there are no checkpoints on the benchmark box)."""
        self.logger = logger or logging.getLogger("crowdsam")
        self.device = torch.device(config["environ"]["device"])
        self.train_free = False
        if config["model"].get("trainfree", False):
            raise NotImplementedError("train-free branch is out of scope (SURVEY.md §2 #11)")
        m = config["model"]
        if dino_model is None:
            # model.dino_depth: build extension for reduced-depth test checkpoints (DINOv2 ViT-L/14 has 24 blocks)
            # model.dino_pos_offset: how the 37x37 learned position embedding reaches 73x73.  0.1 = the hub models'
            # `interpolate_offset` form (F.interpolate(scale_factor=(73.1/37)), what torch.hub.load(dino_repo,
            # 'dinov2_vitl14') of the reference's era builds -- the shipped default); null = the `size=(73,73)` form of
            # newer upstream commits / transformers.  The submodule commit is not recorded: unpinned.
            dino_model = DinoV2(depth=int(m.get("dino_depth", dino_depth)), pos_offset=m.get("dino_pos_offset", 0.1))
            if dino_state_dict is None:
                dino_state_dict = torch.load(m["dino_checkpoint"], map_location="cpu")
            dino_model.load_state_dict(dino_state_dict)
            dino_model = dino_model.to(self.device)
        self.predictor = self.load_sam_model(m["sam_model"], m.get("sam_arch", "crowdsam"), m.get("sam_checkpoint"),
                                             m.get("sam_adapter_checkpoint"), dino_model, m["n_class"],
                                             sam_state_dict)
        # model.ln_fold (build extension, default true): the encoders' LayerNorms are folded into the projections around them
        # (csam_gemm_f16_ln: the projection multiplies the fp16 copy of the UNNORMALISED residual stream and applies mean /
        # rstd in its epilogue).  false = separate LayerNorm launches whose fp16 output feeds a plain GEMM: ~4 % slower
        # encoders, and the operand rounding is relative to |x - mean| instead of |x| -- the conservative choice for a
        # checkpoint whose residual rows sit far from zero (tests/test_encoder_gpu.py::test_ln_fold_on_vit_l_outlier_profile)
        ln_fold = bool(m.get("ln_fold", True))
        self.predictor.model.image_encoder.ln_fold = ln_fold
        if hasattr(dino_model, "plan"):
            dino_model.ln_fold = ln_fold
        t = config["test"]
        self.mask_selection = t["mask_selection"]
        self.apply_box_offsets = t["apply_box_offsets"]
        self.max_prompts = t["max_prompts"]
        self.filter_thresh = t["filter_thresh"]
        self.max_size = t["max_size"]
        self.grid_size = t["grid_size"]
        self.pred_iou_thresh = t["pred_iou_thresh"]
        self.fuse_simmap = t["fuse_simmap"]
        self.stability_score_thresh = t["stability_score_thresh"]
        self.stability_score_offset = t["stability_score_offset"]
        self.box_nms_thresh = t["box_nms_thresh"]
        self.points_per_batch = t["points_per_batch"]
        # Efficient Prompt Sampler state on the device (no host round trip per batch); False keeps the point list on the host
        # and synchronises after every batch, as the reference does (parity tests compare the two)
        self.eps_on_device = True
        self.eps_trace = None       # set to a list to collect (prompt points, n_valid) of every EPS round (tests, debugging)
        self.eps_trace_status = []  # with eps_trace set: (score, survivor flag, feeder flag) of every prompt of every round
        self.crop_n_layers = t["crop_n_layers"]
        self.crop_nms_thresh = t["crop_nms_thresh"]
        self.crop_overlap_ratio = t["crop_overlap_ratio"]
        self.min_mask_region_area = t["min_mask_region_area"]
        self.pos_sim_thresh = t["pos_sim_thresh"]
        self.output_rles = t["output_rles"]
        self.mask_nms_thresh = float(t.get("mask_nms_thresh", 0.0))     # build's opt-in knob, see _process_crop
        if self.mask_selection == "all":
            # the reference's own 'all' branch returns a float tensor that is then used as an index
            # (crowdsam/model.py:326-327,355) and raises IndexError: there is no behaviour to reproduce
            raise NotImplementedError("mask_selection='all' is broken in the reference (float tensor used as an index)")
        if self.mask_selection not in ("max_iou", "max_area", "min_area"):
            raise NotImplementedError(f"unknown mask_selection '{self.mask_selection}'")
        if self.max_size > 4096:
            # free knob as in the reference (crowdsam/utils.py:141-156) up to the RLE column scan's 4096-pixel row
            raise NotImplementedError("test.max_size > 4096 is not supported (csam_rle_count scans rows of <= 4096 pixels)")
        if self.apply_box_offsets:
            raise NotImplementedError("apply_box_offsets is off in the shipped config (the decoder has no offset head)")
        # build extension: frames per image-batched encoder pass of generate_stream (1 = the depth-2 pipeline of round 4)
        self.encoder_batch = int(t.get("encoder_batch", 4))
        self.timings = {}
        self.last_candidates = 0
        self.last_prompts = 0
        self._next_image = None      # depth-2 pipeline (generate(next_image=...)): see _prefetch
        self._prefetched = None
        self.group_ramp = True       # generate_stream starts with groups of 1, 2, 4, .. frames up to encoder_batch
        self.inline_ahead = False    # look-ahead work on the frame's own stream instead of the side stream (no overlap)
        self._look = None            # image-batched look-ahead (generate_stream(batch=B)): see _lookahead_step
        self._cur_group = None
        self._next_group = None
        self._group_parity = 0
        self._pf_stream = None
        self._hi_stream = None
        self._work_stream = None

    def load_sam_model(self, sam_model, sam_arch, sam_checkpoint, sam_adapter_checkpoint, dino_model, n_class,
                       sam_state_dict=None):
        if sam_arch != "crowdsam":
            raise NotImplementedError(f"sam_arch '{sam_arch}' imports packages outside the reference tree")
        from segment_anything_cs import SamPredictor, sam_model_registry
        sam = sam_model_registry[sam_model](checkpoint=None if sam_state_dict is not None else sam_checkpoint,
                                            n_class=n_class)
        if sam_state_dict is not None:
            sam.load_state_dict(sam_state_dict, strict=False)
        elif sam_adapter_checkpoint:
            sam.mask_decoder.load_state_dict(torch.load(sam_adapter_checkpoint, map_location="cpu"), strict=False)
        sam = sam.to(self.device)
        return SamPredictor(sam, dino_model)

    # ------------------------------------------------------------------------------------------
    def crop_image(self, image, crop_box, sim_map=None):
        x0, y0, x1, y1 = crop_box
        if not isinstance(image, np.ndarray):
            image = np.array(image, dtype=np.uint8)
        self.orig_image = image
        # utils.resize_image of the reference (cv2.resize) on the GPU: the crop is uploaded ONCE (the one H2D per
        # crop) and resized there; ``self.image`` (the reference's ndarray attribute) is materialised lazily
        crop = image[y0:y1, x0:x1, :]
        self._frame_u8, self._frame_f32, r = utils.resize_frame_device(crop, self.max_size, self.device)
        self._image_np = np.ascontiguousarray(crop) if self._frame_f32 is None else None
        self.image_hw = tuple(self._frame_u8.shape[:2])
        self.downscale = r

    @property
    def image(self):
        """The resized crop as an ndarray (reference attribute, crowdsam/model.py:130); one D2H on first access."""
        if self._image_np is None:
            self._image_np = self._frame_u8.cpu().numpy()
        return self._image_np

    @torch.no_grad()
    def generate(self, image, next_image=None):
        """image: HWC uint8 RGB ndarray (or PIL image) -> MaskData of numpy fields (crowdsam/model.py:133-190).

        ``next_image`` (build extension, the per-image loop of tools/test.py:62-82 as a depth-2 pipeline): the frame the
        caller will pass to the NEXT generate() call.  Its upload, resize, SAM encoder || DINOv2 and per-image decoder
        constants are enqueued on a side stream as soon as this frame's prompt sweep is queued, so they run beside this
        frame's tail (NMS, connected components, RLE -- latency chains that under-fill the chip) and its D2H / string
        packing.  No random number is drawn before sample_prompts, so results are bit-identical to serial calls
        (tests/test_pipelined_gpu.py).  Ignored with crop_n_layers > 0 (the crops of one frame share the buffers)."""
        self._next_image = next_image if self.crop_n_layers == 0 else None
        d0 = trace.depth
        trace.push("generate")
        try:
            if not _WORK_STREAM or self.device.type != "cuda":
                return self._generate_masks(image)
            # the whole frame on a non-default stream of its own: torch's default stream is HIP's legacy NULL stream, whose
            # implicit synchronisation with every blocking stream costs the pipelined modes 1.3 ms per frame
            # (profiles/r04_eps_pipeline_overlap.txt).  Callers on other streams are ordered before / after it.
            if self._work_stream is None:
                self._work_stream = torch.cuda.Stream(device=self.device, priority=-1)
            cur = torch.cuda.current_stream(self.device)
            if cur == self._work_stream:
                return self._generate_masks(image)
            self._work_stream.wait_stream(cur)
            with torch.cuda.stream(self._work_stream):
                out = self._generate_masks(image)
            cur.wait_stream(self._work_stream)
            return out
        finally:
            self._next_image = None
            trace.unwind(d0)

    def generate_stream(self, images, batch=None):
        """Iterator over generate(image) for an iterable of frames, pipelined (build extension; the per-image loop of
        tools/test.py:62-82).

        ``batch`` (default ``test.encoder_batch`` = 4) frames ahead are encoded TOGETHER: SAM's encoder and DINOv2 run one
        image-batched pass per group of B frames ([B * 4096, D] / [B * 5330, D] token matrices: their GEMMs leave the
        launch-shaped batch-of-one regime), and that pass is cut into len(group) chunks, one queued beside the tail of each
        frame of the group before it.  Every frame's features are bit-identical to a pass of its own
        (tests/test_encoder_batch_gpu.py), no random number is drawn before sample_prompts, so results equal serial
        generate() calls (tests/test_pipelined_gpu.py).  The first group is encoded cold; it is ONE frame and the groups double
        up to B (``group_ramp``), so the first result is not held back by B frames' encoders.  batch <= 1, multi-crop configs
        and non-CUDA devices: the depth-2 pipeline of generate(image, next_image=...)."""
        B = self.encoder_batch if batch is None else int(batch)
        it = iter(images)
        # EPS sweeps (pruning, or small prompt batches) are latency chains of ~30 short launches per batch beside which the
        # look-ahead runs for the WHOLE sweep: the longer launches of a batched pass hold the chain up more than they save
        # (shipped configuration on MI355X: 19.1 ms per frame at one frame ahead, 20.5 / 20.8 at groups of 2 / 4), so the
        # groups are for dense sweeps; an explicit ``batch`` argument is honoured either way
        if batch is None and (math.isfinite(self.filter_thresh) or self.points_per_batch < 256):
            B = 1
        if B <= 1 or self.crop_n_layers > 0 or self.device.type != "cuda":
            try:
                cur = next(it)
            except StopIteration:
                return
            for nxt in it:
                yield self.generate(cur, next_image=nxt)
                cur = nxt
            yield self.generate(cur)
            return
        import itertools
        # The stream STARTS with groups of 1, 2, 4, .. frames (group_ramp): the first result waits for one frame's encoders
        # instead of B frames', and the pipeline fills while it already delivers.  Results do not depend on the grouping.
        size = 1 if self.group_ramp else B
        self.predictor.group_reserve(B)
        cur = list(itertools.islice(it, size))
        first = True
        self._cur_group = self._next_group = None
        self._group_parity = 0        # the groups' buffer sets alternate from the same start: same hipGraph keys every stream
        try:
            while cur:
                size = min(2 * size, B)
                nxt = list(itertools.islice(it, size))
                for j, img in enumerate(cur):
                    self._look = dict(cur=cur, j=j, nxt=nxt, cold=first and j == 0)
                    try:
                        out = self.generate(img)
                    finally:
                        self._look = None
                    yield out
                first = False
                cur = nxt
        finally:
            self._cur_group = self._next_group = None
            self._look = None

    def _generate_masks(self, image):
        img_size = np.array(image).shape[:2]
        crop_boxes, _ = generate_crop_boxes(img_size, self.crop_n_layers, self.crop_overlap_ratio)
        data = MaskData()
        crops = self._encode_crops(image, crop_boxes) if len(crop_boxes) > 1 else None
        for i, crop_box in enumerate(crop_boxes):
            if crops is not None:
                # the crop's decoder constants go into the inactive slot right before its sweep (the slots alternate)
                self._prefetched = dict(src=image, crop_box=list(crop_box), state=crops["states"][i],
                                        bundle=self.predictor.group_bundle(crops["group"], i))
            crop_data = self._process_crop(image, crop_box)
            if crop_data is not None:
                data.cat(crop_data)
        if len(crop_boxes) > 1 and "crop_boxes" in data and len(data["crop_boxes"]) > 0:
            scores = (1 / box_area(data["crop_boxes"])).to(data["boxes"].device)   # prefer small crops
            keep = batched_nms(data["boxes"].float(), scores, None, self.crop_nms_thresh)
            rles_info = data["rles_info"]
            del data["rles_info"]            # per-crop list: not a per-mask field (reference crashes here)
            data.filter(keep)
            data["rles_info"] = rles_info
            del data["crop_boxes"]
        if len(data._stats.keys()) > 0:
            del data["iou_preds"]
        else:
            # no detection at all (crowdsam/model.py:181-183 sets boxes / scores only; tools/test.py:71 copes by
            # filtering on the keys present).  The remaining per-mask fields are set empty as well so that callers
            # indexing them directly do not raise.
            data["boxes"] = torch.zeros(0, 4)
            data["scores"] = torch.zeros(0, 4)
            data["categories"] = torch.zeros(0, dtype=torch.long)
            data["points"] = torch.zeros(0, 2)
            data["stability_score"] = torch.zeros(0)
            data["fboxes"] = torch.zeros(0, 4)
        data["rles"] = coco_encode_rles(data["rles"]) if "rles" in data else []
        data.to_numpy()
        return data

    # ------------------------------------------------------------------------------------------
    def sample_prompts(self, on_device=False):
        """FG prior -> grid -> threshold -> pixel coordinates (crowdsam/model.py:196-223)."""
        h, w = self.image_hw
        g = self.grid_size
        img_size = torch.tensor([h, w])
        feat_size = (img_size * min(g / img_size)).int()
        sim = self.predictor.predict_fg_map(img_size)                    # [1,C,256,256]
        sim = hip.bilinear_f32(sim[0], (g, g))                           # [C,g,g]
        sim = hip.sigmoid_max(sim.view(sim.shape[0], g * g)).view(g, g)
        self.sim_map = sim
        self.sim_feat_size = (int(feat_size[0]), int(feat_size[1]))
        sim_c = sim[: int(feat_size[0]), : int(feat_size[1])]
        inv_factor = torch.tensor([feat_size[1] / w, feat_size[0] / h])
        if on_device:
            # same arithmetic where the map already is: nonzero() in row-major order, fp32 division, truncation
            # (:230's astype("int")) -- the map never goes D2H and only the point COUNT reaches the host
            coords = (sim_c > self.pos_sim_thresh).nonzero()[:, [1, 0]]
            # fp32 quotient via float64 (exactly the correctly rounded fp32 division of the host path, whatever the
            # device's fp32 divide does: 53 >= 2 * 24 + 2 bits make the double rounding innocuous)
            q = (coords.to(torch.float64) / inv_factor.to(coords.device, torch.float64)).to(torch.float32)
            return q.to(torch.int32).contiguous()
        sim_c = sim_c.cpu()                                              # one small D2H
        coords = (sim_c > self.pos_sim_thresh).nonzero()[:, [1, 0]]
        return (coords / inv_factor).numpy()

    def _result_store(self, H, W):
        """Image-level device store the survivors of every EPS batch are compacted into by the GPU itself
        (csam_post_finalize_compact + csam_mask_write): MaskData.cat of the reference (crowdsam/model.py:247)
        without host synchronisation or gather copies.  Static per frame shape, reused across images."""
        cap = int(self.max_prompts) + int(self.points_per_batch)
        key = (H, W, cap)
        if getattr(self, "_store_key", None) != key:
            dev = self.device
            e = lambda *s, dt: torch.empty(*s, dtype=dt, device=dev)
            self._store = dict(masks=e(cap, H, W, dt=torch.uint8), score=e(cap, dt=torch.float32),
                               stability=e(cap, dt=torch.float32), boxes=e(cap, 4, dt=torch.int32),
                               category=e(cap, dt=torch.int32), points=e(cap, 2, dt=torch.int32),
                               counter=torch.zeros(1, dtype=torch.int32, device=dev))
            self._store_key = key
        return self._store

    _CROP_STATE = ("orig_image", "_frame_u8", "_frame_f32", "_image_np", "image_hw", "downscale")

    def _upload_frames(self, frames, boxes=None):
        """crop_image (the one H2D + device resize) of several frames / crops on the current stream -> their crop states,
        with the current frame's state put back afterwards (its tail still needs ``downscale`` & co)."""
        saved = {k: getattr(self, k, None) for k in self._CROP_STATE}
        states = []
        for i, f in enumerate(frames):
            arr = f if isinstance(f, np.ndarray) else np.array(f, dtype=np.uint8)
            h, w = arr.shape[:2]
            self.crop_image(arr, [0, 0, w, h] if boxes is None else boxes[i])
            states.append({k: getattr(self, k) for k in self._CROP_STATE})
        for k, v in saved.items():
            setattr(self, k, v)
        return states

    @staticmethod
    def _device_frame(state):
        return state["_frame_u8"] if state["_frame_f32"] is None else (state["_frame_u8"], state["_frame_f32"])

    def _group_upload(self, frames, boxes=None, states=None):
        """Upload a group of frames (unless ``states`` holds their crop states already) and hand them to the predictor's
        image-batched encoder pass (SamPredictor.group_begin: copies into the encoders' static input buffers, so the stream
        must be ordered behind the previous pass).  None when a frame needs the general set_image route."""
        if states is None:
            states = self._upload_frames(frames, boxes)
        self._group_parity ^= 1
        g = self.predictor.group_begin([self._device_frame(st) for st in states], bufset=self._group_parity)
        if g is None:
            return None
        return dict(group=g, states=states, frames=list(frames))

    def _encode_crops(self, image, crop_boxes):
        """Multi-crop mode (crowdsam/model.py:151-178: every crop is resized to max_size and encoded): the crops of one image
        as ONE image-batched pass of both backbones instead of len(crop_boxes) passes of one."""
        if self.device.type != "cuda":
            return None
        arr = image if isinstance(image, np.ndarray) else np.array(image, dtype=np.uint8)
        if self._pf_stream is not None:
            torch.cuda.current_stream().wait_stream(self._pf_stream)
        rec = self._group_upload([arr] * len(crop_boxes), boxes=[list(b) for b in crop_boxes])
        if rec is None:
            return None
        self.predictor.group_chunk(rec["group"], 0, 1)
        return rec

    def _mark_cross_stream(self, states, main):
        for st in states:
            for t in (st["_frame_u8"], st["_frame_f32"]):
                if torch.is_tensor(t):
                    t.record_stream(main)           # allocated on the side stream, read on the main one later

    def _lookahead_step(self, look, early=False):
        """Image-batched look-ahead, called once per frame where _prefetch would be (EPS sweeps: before the sweep is queued;
        dense sweeps: after): on the side stream, (1) at the first frame of a group the NEXT group's frames go up and into the
        encoders' input buffers, (2) chunk j of len(group) of the next group's encoder passes, (3) the decoder constants of the
        frame the caller processes next (frame j + 1 of this group, or frame 0 of the next one -- its pass has just ended) into
        the decoder plan's inactive slot."""
        main = torch.cuda.current_stream()
        if self._pf_stream is None:
            self._pf_stream = torch.cuda.Stream(device=self.device)
        side = main if self.inline_ahead else self._pf_stream      # inline_ahead: measurement aid (bench.py's per-launch timing leg)
        cur, j, nxt = look["cur"], look["j"], look["nxt"]
        with torch.cuda.stream(side):
            states = None
            if j == 0 and nxt:
                states = self._upload_frames(nxt)                   # fresh tensors only: may start before the sweep ends
                self._mark_cross_stream(states, main)
            side.wait_stream(main)
            if states is not None:
                self._next_group = self._group_upload(nxt, states=states)
            if self._next_group is not None:
                # beside an EPS sweep's latency chain one stream of full-chip launches disturbs it less than two (as in _prefetch)
                self.predictor.group_chunk(self._next_group["group"], j, len(cur), two_streams=not early and not self.inline_ahead)
            rec, b = (self._cur_group, j + 1) if j + 1 < len(cur) else (self._next_group, 0)
            if rec is not None:
                self._prefetched = dict(src=rec["frames"][b], crop_box=None, state=rec["states"][b],
                                        bundle=self.predictor.group_bundle(rec["group"], b))
        if j + 1 == len(cur):
            self._cur_group, self._next_group = self._next_group, None

    def _run_ahead(self, look, early):
        with trace.range("lookahead"):
            if look is not None:
                self._lookahead_step(look, early=early)
            elif self._next_image is not None:
                self._prefetch(self._next_image, early=early)
        self._next_image = None

    def _prefetch(self, image, early):
        """Depth-2 pipeline: upload + resize + SAM encoder || DINOv2 + the decoder's per-image constants of the NEXT frame on
        a side stream, into the decoder plan's inactive slot (SamPredictor.prefetch_image), without touching anything the
        current frame still reads.  ``early`` (EPS sweeps: 16 small, latency-bound prompt batches that leave the GPU mostly
        idle): called BEFORE the sweep is queued, so the encoders run beside the whole sweep; otherwise (dense sweep: the
        persistent decoder kernels own every CU) AFTER the sweep is queued, beside the tail only.  The crop state of the
        current frame is put back afterwards: its tail still needs ``downscale`` & co.
        The record remembers the caller's OBJECT (``src``): the next generate() adopts it only for that very object, which
        therefore must not be mutated in between (a PIL image is converted once, here)."""
        main = torch.cuda.current_stream()
        if self._pf_stream is None:
            self._pf_stream = torch.cuda.Stream(device=self.device)
        side = main if self.inline_ahead else self._pf_stream
        with torch.cuda.stream(side):
            # the one H2D of the frame and its resize write fresh tensors only: they go up first ...
            state = self._upload_frames([image])[0]
            # ... the encoders' workspaces were last used by this stream (the previous prefetch) or by the main stream's
            # set_image before this point; the inactive decoder slot was last read by the frame before the current one
            side.wait_stream(main)
            bundle = self.predictor.prefetch_image(self._device_frame(state), two_streams=not early)
        self._mark_cross_stream([state], main)
        self._prefetched = None if bundle is None else dict(src=image, crop_box=None, state=state, bundle=bundle)

    def _process_crop(self, image, crop_box):
        d0 = trace.depth
        try:
            return self._process_crop_stages(image, crop_box)
        finally:
            trace.unwind(d0)             # an early return (no prompts, no candidates) leaves its stage's range open

    def _process_crop_stages(self, image, crop_box):
        t0 = time.perf_counter()
        trace.push(_STAGES[0])
        pf, self._prefetched = self._prefetched, None
        if self._pf_stream is not None:
            torch.cuda.current_stream().wait_stream(self._pf_stream)     # a prefetch (used or not) owns the shared buffers
        look = self._look
        if look is not None and look["cold"] and pf is None:
            # first frame of a stream: its whole group is encoded now, in one image-batched pass (nothing to overlap it with)
            self._cur_group = self._group_upload(look["cur"])
            if self._cur_group is not None:
                self.predictor.group_chunk(self._cur_group["group"], 0, 1, two_streams=not self.inline_ahead)
                pf = dict(src=image, crop_box=None, state=self._cur_group["states"][0],
                          bundle=self.predictor.group_bundle(self._cur_group["group"], 0))
        if pf is not None and pf["src"] is image and (pf["crop_box"] is None or pf["crop_box"] == list(crop_box)):
            for k, v in pf["state"].items():                             # the frame was prefetched: adopt its state
                setattr(self, k, v)
            self.predictor.adopt_prefetched(pf["bundle"])
        else:
            self.crop_image(image, crop_box)
            self.predictor.set_image(self._frame_u8 if self._frame_f32 is None else (self._frame_u8, self._frame_f32))
        t0 = self._tick("set_image", t0)
        H, W = self.image_hw
        downscale = self.downscale
        orig_h, orig_w = self.orig_image.shape[:2]
        dev = self.device
        prune = math.isfinite(self.filter_thresh)
        # the device-resident sampler also serves the sweep without pruning (filter_thresh = inf): the point list never
        # visits the host, the rounds simply skip csam_occupancy_prune
        dev_sampler = self.eps_on_device
        points_for_image = self.sample_prompts(on_device=dev_sampler)
        t0 = self._tick("sample_prompts", t0)
        store = self._result_store(*self.predictor.original_size)
        store["counter"].zero_()
        ahead = self._next_image is not None or look is not None
        early = ahead and (prune or self.points_per_batch < 256)
        if early:
            # EPS sweep: small prompt batches with the GPU mostly idle -> the next frame's encoders run beside the whole sweep
            self._run_ahead(look, early=True)

        # :230 truncation (already done, on the device, for the device-resident sampler)
        points = points_for_image if dev_sampler else points_for_image.astype("int")
        # :231 np.random.shuffle(points) on the global RNG (seeded by the harness).  NumPy shuffles the rows of a 2-D
        # array one Python-level swap at a time (38 ms for the 36 864 points of the shipped grid_size 192); shuffling an
        # index vector draws the SAME random_interval sequence (Fisher-Yates over n items either way), so the
        # permutation and the RNG state afterwards are identical (tests/test_host_logic_cpu.py) at 3 ms
        perm = np.arange(len(points))
        np.random.shuffle(perm)
        if dev_sampler:
            # the list stays where it was computed; only the permutation (4 bytes per point) goes up
            points = points[torch.as_tensor(perm.astype(np.int32)).to(dev).long()] if len(perm) else points
        else:
            points = points[perm]
        # The shuffled point list goes to the GPU ONCE; the sampler then works on indices into it.  (Uploading the
        # remaining points every batch -- 36 864 of them at the shipped grid_size 192 -- was a pageable H2D of ~300 KB
        # per batch, which ROCm pins on the fly: 0.7-14 ms each, several times the 1 ms the batch itself takes.)
        count = 0
        n_batches = 0
        batch_size = self.points_per_batch
        # With the next frame's encoders running beside it (early prefetch), the EPS sweep -- a chain of ~60 short kernels per
        # batch, each waiting for the one before -- goes onto a HIGH-priority stream: its workgroups are dispatched ahead of the
        # encoder GEMMs' whenever a CU frees up, so the chain's latency stays close to what it is on an idle GPU
        sweep_ctx = contextlib.ExitStack()
        if early and self._work_stream is not None and torch.cuda.current_stream() == self._work_stream:
            early_hi = False                        # the frame's own stream already has the high priority
        else:
            early_hi = early
        if early_hi:
            if self._hi_stream is None:
                self._hi_stream = torch.cuda.Stream(device=dev, priority=-1)
            self._hi_stream.wait_stream(torch.cuda.current_stream())
            main_stream = torch.cuda.current_stream()
            sweep_ctx.enter_context(torch.cuda.stream(self._hi_stream))
        try:           # the side-stream state is restored even when a batch raises (CSAM error, OOM): ADVICE r4
            if dev_sampler and len(points) > 0:
                # Device-resident sampler (csam_eps_select / csam_occupancy_prune): the list and one alive flag per point stay
                # on the GPU; a round takes the first batch_size alive points in list order -- the reference's
                # points[:batch_size] -- and the pruning clears flags, so the sweep is a queue of kernels without a host round
                # trip per batch.  The host does not see len(points): it runs the rounds count < max_prompts allows (a round
                # whose list has run dry has zero valid slots and leaves no trace) and looks at the device's alive count every
                # 4th round to stop early.  The reference's last round shrinks to the points that are left; here it keeps its
                # width and reports how many slots are real (n_valid).
                P = len(points)
                all_pts_dev = points.contiguous()
                alive_dev = torch.ones(P, dtype=torch.uint8, device=dev)
                B = min(batch_size, P)
                pts_b = torch.empty(B, 2, dtype=torch.int32, device=dev)
                coords_b = torch.empty(B, 2, dtype=torch.float32, device=dev)
                counts = torch.zeros(2, dtype=torch.int32, device=dev)
                old_h, old_w = self.predictor.original_size
                new_h, new_w = self.predictor.transform.get_preprocess_shape(old_h, old_w, self.predictor.transform.target_length)
                max_rounds = -(-P // B)
                while count < self.max_prompts and n_batches < max_rounds:
                    tb = time.perf_counter()
                    hip.eps_select(all_pts_dev, alive_dev, B, new_w / old_w, new_h / old_h, pts_b, coords_b, counts)
                    if self.eps_trace is not None:          # debugging / parity aid: the prompts of every round (async copies)
                        self.eps_trace.append((pts_b.clone(), counts[:1].clone()))
                    with trace.range("decoder_batch"):
                        bd = self._process_batch(None, self.predictor.original_size, crop_box, store,
                                                 device_batch=(pts_b, coords_b, counts[:1]))
                    if self.eps_trace is not None:          # ... and (fused score, survivor flag, feeder flag) of its prompts
                        self.eps_trace_status.append((bd["score"].clone(), bd["keep"].clone(), bd["occ"].clone()))
                    if prune:
                        hip.occupancy_prune(all_pts_dev, store["masks"], bd["occ"], B, H, W, alive_dev, slot=bd["slot"])
                    count += B
                    n_batches += 1
                    self._tick("eps.batch", tb)
                    if prune and n_batches % 4 == 0 and count < self.max_prompts and n_batches < max_rounds:
                        # the alive flags AFTER this round's pruning (counts[1] is the selection's view, taken before it: a
                        # prune that empties the list would cost up to four more full-width empty rounds)
                        if int(alive_dev.count_nonzero().item()) == 0:
                            break
            else:
                all_pts_dev = None
                alive = np.arange(len(points))               # indices of the points still in play, in shuffled order
                while len(alive) > 0 and count < self.max_prompts:
                    batch_size = min(len(alive), batch_size)
                    sel_idx, alive = alive[:batch_size], alive[batch_size:]
                    tb = time.perf_counter()
                    if self.eps_trace is not None:
                        self.eps_trace.append((points[sel_idx].copy(), len(sel_idx)))
                    with trace.range("decoder_batch"):
                        bd = self._process_batch(points[sel_idx], self.predictor.original_size, crop_box, store)
                    if self.eps_trace is not None:
                        self.eps_trace_status.append((bd["score"].clone(), bd["keep"].clone(), bd["occ"].clone()))
                    tb = self._tick("eps.batch", tb)
                    if prune and len(alive) > 0:
                        if all_pts_dev is None:
                            all_pts_dev = torch.as_tensor(np.ascontiguousarray(points), dtype=torch.int32).to(dev)
                            occupy_bits = torch.empty(len(points), dtype=torch.uint8, device=dev)
                            tb = self._tick("eps.upload_points", tb)
                        # occupancy of EVERY point under this batch's masks (the mask is replaced per batch, :246); only the
                        # flags of the live points are looked at -- same pruning as points[~occupy_mask[y, x]] (:238-239)
                        hip.occupancy_lookup(all_pts_dev, store["masks"], bd["occ"], batch_size, H, W, occupy_bits,
                                             slot=bd["slot"])
                        occ = occupy_bits.cpu().numpy().astype(bool)                  # the per-batch sync
                        alive = alive[~occ[alive]]
                        self._tick("eps.prune", tb)
                    count += batch_size
                    n_batches += 1
        finally:
            sweep_ctx.close()
            if early_hi:
                main_stream.wait_stream(self._hi_stream)
        self.predictor.reset_image()
        if ahead and not early:
            # dense sweep: it is queued and nothing below reads the predictor -- the next frame's encoders start now, beside
            # this frame's tail
            self._run_ahead(look, early=False)
        t0 = self._tick("eps_sweep", t0)
        self.last_prompts = count                    # prompt slots decoded for this crop (rounds x batch width; bench.py reports it)
        if n_batches == 0:
            return None
        n = int(store["counter"].item())             # the one sync of a dense sweep
        self.last_candidates = n
        if n == 0:
            return None
        # The masks STAY in their store slots: the record carries the slot numbers (``mask_slots``) where the reference
        # carries a (n, H, W) bool tensor, every filter permutes 4-byte slot numbers instead of gathering megabyte masks,
        # and the small-region clean-up, the optional mask NMS / prior fusion and the run-length encoder address
        # store["masks"][slot] directly (a crowded frame keeps hundreds of masks: two gathers of ~0.3 GB each before).
        mstore = store["masks"]
        data = MaskData(mask_slots=torch.arange(n, dtype=torch.int32, device=dev), iou_preds=store["score"][:n],
                        points=store["points"][:n].long(), categories=store["category"][:n].long(),
                        stability_score=store["stability"][:n], boxes=store["boxes"][:n].long())
        t0 = self._tick("gather", t0)

        keep = batched_nms(data["boxes"].float(), data["iou_preds"], None, self.box_nms_thresh)
        data.filter(keep)
        if self.mask_nms_thresh > 0 and len(data["mask_slots"]) > 0:
            # opt-in (test.mask_nms_thresh, off in the shipped config): the coverage NMS on 150x150 masks the
            # reference defines but never calls (crowdsam/utils.py:422-467), on device
            data.filter(hip.mask_nms(mstore[data["mask_slots"].long()], data["iou_preds"], self.mask_nms_thresh))
        t0 = self._tick("nms", t0)
        if self.min_mask_region_area > 0:
            data = self.postprocess_small_regions(data, self.min_mask_region_area,
                                                  max(self.box_nms_thresh, self.crop_nms_thresh), mask_store=mstore)
        t0 = self._tick("small_regions", t0)
        if self.fuse_simmap:
            # crowdsam/model.py:273-286: score = sqrt(iou) * sqrt(clamp(mean of the resized prior over the mask + 0.5))
            fh, fw = self.sim_feat_size
            cls = torch.clamp(hip.mask_mean_bilinear(mstore[data["mask_slots"].long()], self.sim_map[:fh, :fw]) + 0.5, 0, 1)
            data["scores"] = data["iou_preds"] ** 0.5 * cls ** 0.5
        else:
            data["scores"] = data["iou_preds"]
        # run lengths -> C string packer; the passes read the masks' boxes (exact for the final masks: the statistics pass's for
        # untouched masks, the clean-up's for edited ones), not the frames
        data["rles"] = mask_to_coco_rles(mstore, idx=data["mask_slots"].contiguous(),
                                         boxes=data["boxes"] if _WINDOWED_REGIONS else None)
        t0 = self._tick("rle", t0)
        data["rles_info"] = [crop_box, [orig_h, orig_w]]
        del data["mask_slots"]
        data["boxes"] = utils.uncrop_boxes_xyxy(data["boxes"], crop_box, downscale)
        data["points"] = utils.uncrop_points(data["points"], crop_box, downscale)
        data["crop_boxes"] = torch.tensor([crop_box for _ in range(len(data["boxes"]))]).reshape(-1, 4)
        # build extension: the crop box of every mask survives the cross-crop NMS (the reference keeps only the
        # 2-entries-per-crop ``rles_info`` list, which cannot tell which crop frame an RLE lives in)
        data["rles_crop"] = data["crop_boxes"].clone()
        data["fboxes"] = data["boxes"]
        return data

    def select_mask(self, masks, iou_preds):
        """API mirror of crowdsam/model.py:318-331 for materialised [B,4,H,W] logits (the fused path selects inside
        _process_batch without materialising them)."""
        bin_masks = masks > self.predictor.model.mask_threshold
        if self.mask_selection == "max_area":
            ind = bin_masks.sum(dim=[-1, -2]).max(dim=-1)[1]
        elif self.mask_selection == "min_area":
            ind = bin_masks.sum(dim=[-1, -2]).min(dim=-1)[1]
        elif self.mask_selection == "max_iou":
            ind = iou_preds.max(dim=-1)[1]
        else:
            raise NotImplementedError
        return torch.arange(len(masks)), ind

    def _process_batch(self, points, im_size, crop_box, store, device_batch=None):
        """One EPS batch, entirely asynchronous: decode B prompts, PWD-Net selection, statistics pass, filters +
        in-kernel compaction of the survivors into ``store``, mask bytes of the survivors.
        Returns the per-batch occupancy flags and store slots (device) for the pruning lookup.
        ``device_batch`` = (points i32 [B,2], coords f32 [B,2], n_valid i32 [1]) on the device, from csam_eps_select
        (``points`` is then ignored): nothing of the batch touches the host."""
        p = self.predictor
        dev = self.device
        H, W = p.original_size
        if device_batch is None:
            B = len(points)
            tp = p.transform.apply_coords(points, im_size)                       # float64 on the host (trap 6)
            in_points = torch.as_tensor(tp)[:, None, :]
            low, iou, cls = p.decode_points(in_points, None)
            pts_dev, n_valid = None, None
        else:
            pts_dev, coords_dev, n_valid = device_batch
            B = pts_dev.shape[0]
            low, iou, cls = p.decode_coords_device(coords_dev)
        i32 = lambda *s: torch.empty(*s, dtype=torch.int32, device=dev)
        sel, category = i32(B), i32(B)
        score = torch.empty(B, dtype=torch.float32, device=dev)
        fused4 = torch.empty(B, 4, dtype=torch.float32, device=dev) if self.mask_selection != "max_iou" else None
        hip.select_masks(iou, cls, cls.shape[-1], sel, score, category, fused4, B)
        inter, uni, box = i32(B), i32(B), i32(B, 4)
        tmp = None
        if tuple(p.input_size) != (H, W):
            tmp = torch.empty(B, p.input_size[0], p.input_size[1], dtype=torch.float32, device=dev)
        if fused4 is not None:
            # mask_selection max_area / min_area (crowdsam/model.py:320-323; not the shipped config): pixel counts of
            # the four thresholded candidates at the output resolution, one statistics pass per candidate
            areas = torch.empty(4, B, dtype=torch.int32, device=dev)
            for c in range(4):
                hip.mask_post_scored(low, torch.full((B,), c, dtype=torch.int32, device=dev), score, 0.0, B, p.input_size,
                                     (H, W), p.model.mask_threshold, 0.0, inter, areas[c], box, tmp)
            a = areas.t().contiguous().long()
            ind = a.max(dim=-1)[1] if self.mask_selection == "max_area" else a.min(dim=-1)[1]
            sel.copy_(ind)
            score.copy_(fused4.gather(1, ind[:, None])[:, 0])
            category.copy_(cls.max(dim=-1)[1].gather(1, ind[:, None])[:, 0])
        # pass 1: statistics of the selected candidates that pass the predicted-IoU filter (the reference filters on
        # it before it computes stability, crowdsam/model.py:371-376 there); no mask bytes yet
        hip.mask_post_scored(low, sel, score, self.pred_iou_thresh, B, p.input_size, (H, W), p.model.mask_threshold,
                             self.stability_score_offset, inter, uni, box, tmp)
        keep = torch.empty(B, dtype=torch.uint8, device=dev)
        occ = torch.empty(B, dtype=torch.uint8, device=dev)
        slot = i32(B)
        if pts_dev is None:
            pts_dev = torch.as_tensor(np.ascontiguousarray(points), dtype=torch.int32).to(dev)
        edge = None
        if self.crop_n_layers > 0:      # per batch, before the occupancy flags (crowdsam/model.py:386-389)
            orig_h, orig_w = self.orig_image.shape[:2]
            edge = (crop_box, [0, 0, orig_w, orig_h], self.downscale, 20.0)
        hip.post_finalize_compact(score, inter, uni, box, category, pts_dev, self.pred_iou_thresh,
                                  self.stability_score_thresh,
                                  self.filter_thresh if math.isfinite(self.filter_thresh) else 3.0e38,
                                  keep, occ, slot, store["counter"], store, B, edge=edge, n_valid=n_valid)
        # pass 2: mask bytes of the survivors, straight into their store slots
        hip.mask_write(low, sel, keep, B, p.input_size, (H, W), p.model.mask_threshold, store["masks"], tmp, slot=slot)
        # keep / score / sel ride along for tests and tracing (device tensors that exist anyway: no extra work)
        return dict(occ=occ, slot=slot, keep=keep, score=score, sel=sel, inter=inter, union=uni)

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def postprocess_small_regions(mask_data, min_area, nms_thresh, mask_store=None):
        """Hole filling / island removal and a second NMS that prefers untouched masks
        (crowdsam/model.py:394-443).  The reference labels components on the host one mask at a time; here
        csam_small_regions does all masks of the frame on device and also returns the post-edit boxes.
        ``mask_store`` (driver path): the record holds ``mask_slots`` into this (cap, H, W) store instead of ``masks``; the
        clean-up then runs IN PLACE on those slots (csam_small_regions_idx) -- an unchanged mask is written back
        unchanged, an edited one that survives the NMS below is the edited one the reference would write back, and the
        rest is discarded either way."""
        if mask_store is not None and "mask_slots" in mask_data:
            if len(mask_data["mask_slots"]) == 0:
                return mask_data
            # bounding-box-restricted form (round 4): person-sized masks are cleaned up inside their padded boxes, frame-filling
            # ones as before; identical results (hip.small_regions_windowed)
            if _WINDOWED_REGIONS and "boxes" in mask_data:
                changed, boxes = hip.small_regions_windowed(mask_store, mask_data["mask_slots"].contiguous(),
                                                            mask_data["boxes"], min_area)
            else:
                changed, boxes = hip.small_regions_idx(mask_store, mask_data["mask_slots"].contiguous(), min_area)
            scores = (changed == 0).float()
            keep = batched_nms(boxes, scores, None, nms_thresh)
            edited = keep[scores[keep] == 0]
            if edited.numel():
                mask_data["boxes"][edited] = boxes[edited].to(mask_data["boxes"].dtype)
            mask_data.filter(keep)
            return mask_data
        if len(mask_data["masks"]) == 0:
            return mask_data
        masks = mask_data["masks"]
        new_masks, changed, boxes = hip.small_regions(masks, min_area)
        scores = (changed == 0).float()
        keep = batched_nms(boxes, scores, None, nms_thresh)
        edited = keep[scores[keep] == 0]
        if edited.numel():
            mask_data["boxes"][edited] = boxes[edited].to(mask_data["boxes"].dtype)
            # the cleaned-up stack as a whole instead of a gather + scatter of the edited masks (hundreds of MB on a crowded
            # frame): a mask the clean-up did not change is returned unchanged, and only `keep` survives the filter below
            mask_data["masks"] = new_masks.view(masks.dtype) if new_masks.element_size() == masks.element_size() \
                else new_masks.to(masks.dtype)
        mask_data.filter(keep)
        return mask_data

    def match_ref(self, sim_map, pos_sim_thresh):
        return (sim_map > pos_sim_thresh).nonzero()[:, [1, 0]]
