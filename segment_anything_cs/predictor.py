"""SamPredictor facade (reference: segment_anything_cs/predictor.py:14-318) on the HIP path.

Same constructor, methods, attributes and errors as the reference for the Crowd-SAM usage:
``set_image / set_torch_image / predict_fg_map / predict / predict_torch / get_image_embedding /
reset_image`` and ``model, dino_model, transform, features, dino_feats, original_size, input_size,
is_image_set, device``.  Per image: ONE H2D copy of the uint8 frame; the encoder, DINOv2 and the
hoisted decoder constants stay resident in HBM until ``reset_image``.
"""
import numpy as np
import torch

from crowdsam_amd import hip, trace
from crowdsam_amd.decoder import N_DINO, N_DINO_PAD
from crowdsam_amd.dino import DinoV2

from .utils.transforms import ResizeLongestSide


_TWO_STREAMS = True      # DINOv2 on a side stream beside the SAM encoder (bench.py's per-launch timing leg switches it off)


class SamPredictor:
    def __init__(self, sam_model, dino_model):
        self.model = sam_model
        self.dino_model = dino_model
        self.transform = ResizeLongestSide(sam_model.image_encoder.img_size)
        self._dtok16 = None
        self._side_stream = None
        self._group_bufs = {}
        self._group_graphs = {}
        self.group_two_streams = True       # DINOv2 share of a chunk on a second stream (35.9 -> 35.3 ms per frame; False: one stream)
        self.reset_image()

    # ------------------------------------------------------------------------------------------
    def set_image(self, image, mask=None, image_format="RGB", cal_image=True):
        assert image_format in ["RGB", "BGR"], f"image_format must be in ['RGB', 'BGR'], is {image_format}."
        # build extension (the reference takes ndarrays only): a frame that is already resident on the GPU --
        # uint8 HWC tensor, optionally with its fp32 CHW form from the device resize (crowdsam/utils.py
        # resize_frame_device) -- skips the host round trip when it needs no further resize
        f32 = None
        if isinstance(image, tuple):
            image, f32 = image
        if torch.is_tensor(image):
            if image_format != self.model.image_format:
                image, f32 = image.flip(-1).contiguous(), None
            h, w = image.shape[:2]
            size = self.model.image_encoder.img_size
            if mask is None and cal_image and max(h, w) == size:
                if f32 is None:
                    f32 = hip.u8hwc_to_f32chw(image)
                return self.set_torch_image(f32[None], (h, w))
            th, tw = self.transform.get_preprocess_shape(h, w, size)
            if mask is None and cal_image and max(h, w) <= 3.5 * size:
                # ResizeLongestSide.apply_image (PIL bilinear via torchvision, transforms.py:26-31) on the device: the
                # 1023-pixel long side of SURVEY.md trap 9, any other enlargement, and the shrink of a test.max_size >
                # 1024 frame (Pillow widens the filter support to the shrink factor: <= 8 taps up to 3.5x) without a
                # host round trip
                from crowdsam_amd.resize import pil_bilinear_tables_device
                dev = str(image.device)
                _, f32 = hip.pil_resize_bilinear_u8(image.contiguous(), (th, tw), pil_bilinear_tables_device(w, tw, dev),
                                                    pil_bilinear_tables_device(h, th, dev))
                return self.set_torch_image(f32[None], (h, w))
            image = image.cpu().numpy()          # anything else: PIL resize on the host
        elif image_format != self.model.image_format:
            image = image[..., ::-1]
        input_image = self.transform.apply_image(image)
        t = torch.as_tensor(np.ascontiguousarray(input_image)).to(self.device, non_blocking=True)
        t = t.permute(2, 0, 1).contiguous()[None, :, :, :]
        tm = None
        if mask is not None:
            m = torch.as_tensor(np.ascontiguousarray(self.transform.apply_image(mask)), device=self.device)
            tm = m.permute(2, 0, 1).contiguous()[None, :, :, :]
        return self.set_torch_image(t, image.shape[:2], transformed_mask=tm, cal_image=cal_image)

    @torch.no_grad()
    def set_torch_image(self, transformed_image, original_image_size, transformed_mask=None, cal_image=True):
        size = self.model.image_encoder.img_size
        assert (len(transformed_image.shape) == 4 and transformed_image.shape[1] == 3
                and max(*transformed_image.shape[2:]) == size), \
            f"set_torch_image input must be BCHW with long side {size}."
        if cal_image:
            self.reset_image()
            self._adopt(self._encode(transformed_image, original_image_size, prefetch=False))
        if transformed_mask is not None:
            return self.model.preprocess(transformed_mask)

    def _encode(self, transformed_image, original_image_size, prefetch, two_streams=True):
        """SAM encoder || DINOv2 + the decoder's per-image constants for one frame -> a bundle for _adopt().
        ``prefetch``: the constants go into the decoder plan's INACTIVE slot and nothing of the predictor's current image
        is touched (the look-ahead frame of CrowdSAM's depth-2 pipeline: the current frame's prompt batches keep decoding
        against the active slot while this runs on another stream)."""
        raw = transformed_image[0].to(self.device).float().contiguous()      # [3,h,w], 0..255
        if self._dtok16 is None or self._dtok16.device != raw.device:
            self._dtok16 = torch.zeros(N_DINO_PAD, 1024, dtype=torch.float16, device=raw.device)
        if isinstance(self.dino_model, DinoV2) and not (_TWO_STREAMS and two_streams):
            with trace.range("sam_encoder"):
                feat = self.model.image_encoder.forward_tokens(raw)
            with trace.range("dinov2"):
                self.dino_model.patch_tokens16(raw, self._dtok16)
        elif isinstance(self.dino_model, DinoV2):
            # the two backbones are independent until the decoder: DINOv2 runs on a side stream next to the SAM
            # encoder, so one's under-filled launches (N = 1024 GEMMs, LayerNorms, the 2.6-workgroups-per-CU
            # attention grid) fill the other's idle CUs
            main = torch.cuda.current_stream()
            if self._side_stream is None:
                self._side_stream = torch.cuda.Stream(device=raw.device)
            side = self._side_stream
            side.wait_stream(main)
            with torch.cuda.stream(side), trace.range("dinov2"):
                self.dino_model.patch_tokens16(raw, self._dtok16)
            with trace.range("sam_encoder"):
                feat = self.model.image_encoder.forward_tokens(raw)
            main.wait_stream(side)
        else:   # third-party DINO object: feed it the reference's tensor (predictor.py:104-106)
            feat = self.model.image_encoder.forward_tokens(raw)
            x = hip.bilinear_f32(hip.preprocess_pad(raw), (1022, 1022))
            tok = self.dino_model.forward_features(x.unsqueeze(0))["x_norm_patchtokens"]
            self._dtok16[:N_DINO].copy_(tok.reshape(N_DINO, -1))
        plan = self.model.decoder_plan()
        slot = (1 - plan.slot) if prefetch else plan.slot
        with trace.range("decoder_constants"):
            plan.set_image(feat, self._dtok16, slot=slot, activate=not prefetch)  # copies both into the slot's own buffers
        # the API views (features / dino_feats) are served from the slot's OWN copies: the encoders' output buffers are
        # overwritten by the next look-ahead frame while this one is still the current image
        return dict(original_size=tuple(original_image_size), input_size=tuple(transformed_image.shape[-2:]),
                    feat=plan.states[slot]["feat"], plan=plan, slot=slot)

    def _adopt(self, b):
        """Make an _encode() bundle the predictor's current image."""
        self.original_size, self.input_size = b["original_size"], b["input_size"]
        self._feat_tok, self._plan = b["feat"], b["plan"]
        self._plan.activate(b["slot"])
        self.is_image_set = True

    def _frame_to_input(self, image):
        """A frame that is already on the device -- uint8 HWC tensor or the (uint8, fp32 CHW) pair of
        crowdsam.utils.resize_frame_device -- as the encoders' input: (fp32 CHW with long side == img_size, (h, w)), or None
        when it needs the general set_image route (host resize)."""
        f32 = None
        if isinstance(image, tuple):
            image, f32 = image
        if not torch.is_tensor(image) or not image.is_cuda:
            return None
        h, w = image.shape[:2]
        size = self.model.image_encoder.img_size
        if max(h, w) == size:
            if f32 is None:
                f32 = hip.u8hwc_to_f32chw(image)
        elif max(h, w) <= 3.5 * size:
            from crowdsam_amd.resize import pil_bilinear_tables_device
            th, tw = self.transform.get_preprocess_shape(h, w, size)
            dev = str(image.device)
            _, f32 = hip.pil_resize_bilinear_u8(image.contiguous(), (th, tw), pil_bilinear_tables_device(w, tw, dev),
                                                pil_bilinear_tables_device(h, th, dev))
        else:
            return None
        return f32, (h, w)

    @torch.no_grad()
    def prefetch_image(self, image, two_streams=True):
        """Build extension (depth-2 image pipeline): encode a frame that is already on the device -- uint8 HWC tensor or the
        (uint8, fp32 CHW) pair of crowdsam.utils.resize_frame_device -- WITHOUT making it the current image.  Returns a
        bundle for adopt_prefetched(), or None when the frame needs the general set_image route (then nothing is done).
        ``two_streams`` False: SAM encoder, then DINOv2, on the calling stream -- beside the
        latency chain of an EPS sweep one stream of full-chip launches disturbs it less than two (round 4: 20.7-21.3 -> 19.5-19.8
        ms per frame in the shipped configuration; the dense sweep prefers two: 36.6 vs 38.1 ms)."""
        inp = self._frame_to_input(image)
        if inp is None:
            return None
        f32, (h, w) = inp
        return self._encode(f32[None], (h, w), prefetch=True, two_streams=two_streams)

    # ---- image-batched look-ahead (round 5): B frames -- the next frames of a stream, or the crops of one image -- go
    # through SAM's encoder and DINOv2 as ONE pass each (EncoderPlan / DinoPlan: [B * tokens, D] token matrices), cut into
    # chunks at block boundaries so that crowdsam.model can queue one chunk beside each frame's tail.
    @torch.no_grad()
    def group_reserve(self, B):
        """Size the encoder plans' workspaces for passes of up to B frames NOW (crowdsam.model.generate_stream calls this before
        its first group): a stream whose groups grow 1, 2, 4 would otherwise re-allocate them -- and re-capture every chunk
        graph, whose key carries the capacity -- twice on the way."""
        self.model.image_encoder.plan()._alloc(B)
        if isinstance(self.dino_model, DinoV2):
            self.dino_model.plan()._alloc(B)

    @torch.no_grad()
    def group_begin(self, images, bufset=0):
        """images: device frames as for prefetch_image.  Copies them into the encoders' static input buffers and returns the
        group record, or None when a frame needs the general route.  A third-party DINO object (anything but
        crowdsam_amd.dino.DinoV2) is fed one frame at a time through its own forward_features at the end of the pass."""
        inputs = [self._frame_to_input(im) for im in images]
        if any(i is None for i in inputs):
            return None
        B = len(inputs)
        enc = self.model.image_encoder.plan()
        dino = self.dino_model.plan() if isinstance(self.dino_model, DinoV2) else None
        raws = [f.contiguous() for f, _ in inputs]
        dev = raws[0].device
        key = (bufset, str(dev))
        bufs = self._group_bufs.get(key)
        if bufs is None or bufs["feat"].shape[0] < B:
            # sized for the encoder plans' capacity from the start: groups of every size (a stream's first groups, its
            # remainder) share ONE allocation per buffer set, and the chunk graphs captured on it stay valid.  Should it
            # still have to grow, the graphs that hold the old addresses go with it.
            cap = max(B, enc.cap)
            bufs = dict(feat=torch.empty(cap, 4096, 256, dtype=torch.float32, device=dev),
                        dtok=[torch.zeros(N_DINO_PAD, 1024, dtype=torch.float16, device=dev) for _ in range(cap)])
            self._group_bufs[key] = bufs
            self._group_graphs = {k: v for k, v in self._group_graphs.items() if k[4] != bufset}
        return dict(B=B, sizes=[hw for _, hw in inputs], input_sizes=[tuple(f.shape[-2:]) for f in raws],
                    sam_views=enc.load_images(raws), dino_views=dino.load_images(raws) if dino is not None else raws,
                    feat=bufs["feat"], dtok=bufs["dtok"], bufset=bufset, enc=enc, dino=dino)

    @torch.no_grad()
    def group_chunk(self, g, c, n, two_streams=True):
        """Chunk c of n of the group's encoder passes on the current stream: blocks [c * depth / n, (c + 1) * depth / n) of
        both backbones; chunk 0 starts with the patch embeddings, chunk n - 1 ends with SAM's neck and DINOv2's final norm
        (features into the group's buffers).  One hipGraph per (group shape, c, n).  A chunk continues the residual streams
        the previous chunk left in the plans' workspaces, so it is NOT idempotent: the first call for a key executes eagerly
        and is then captured WITHOUT the replay hip.GraphCache.run would add."""
        enc, dino, B = g["enc"], g["dino"], g["B"]
        lo_s, hi_s = c * enc.depth // n, (c + 1) * enc.depth // n
        if dino is not None:
            lo_d, hi_d = c * dino.depth // n, (c + 1) * dino.depth // n

        def run_dino():
            if c == 0:
                dino.embed(g["dino_views"], B)
            dino.run_blocks(lo_d, hi_d, B)
            if c == n - 1:
                dino.final_norm([t[:N_DINO] for t in g["dtok"][:B]], B)

        def run():
            two = dino is not None and self.group_two_streams and two_streams
            if two:       # DINOv2's share of the chunk on a second stream (fork / join; inside a capture: two graph branches)
                main = torch.cuda.current_stream()
                if self._side_stream is None:
                    self._side_stream = torch.cuda.Stream(device=g["feat"].device)
                self._side_stream.wait_stream(main)
                with torch.cuda.stream(self._side_stream):
                    run_dino()
            if c == 0:
                enc.embed(g["sam_views"], B)
            enc.run_blocks(lo_s, hi_s, B)
            if c == n - 1:
                enc.neck(g["feat"][:B], B)
            if two:
                main.wait_stream(self._side_stream)
            elif dino is not None:
                run_dino()
            if c == n - 1:
                if dino is not None:
                    pass
                else:   # third-party DINO object: the reference's tensor, one frame at a time (predictor.py:104-106)
                    for b, raw in enumerate(g["dino_views"]):
                        x = hip.bilinear_f32(hip.preprocess_pad(raw), (1022, 1022))
                        tok = self.dino_model.forward_features(x.unsqueeze(0))["x_norm_patchtokens"]
                        g["dtok"][b][:N_DINO].copy_(tok.reshape(N_DINO, -1))

        with trace.range("encoder_chunk %d/%d x%d" % (c + 1, n, B)):
            self._chunk_run_or_replay(g, c, n, run, two_streams)

    def _chunk_run_or_replay(self, g, c, n, run, two_streams):
        enc, dino, B = g["enc"], g["dino"], g["B"]
        if dino is None or not hip.GRAPHS_ENABLED or hip.timer_active():
            return run()
        key = (B, tuple(g["input_sizes"]), c, n, g["bufset"], enc.cap, dino.cap, self.group_two_streams and two_streams)
        # A key holds every frame's input size: a dataset of mixed aspect ratios would make nearly every group a new key, each
        # costing an eager run, a device-wide synchronize and a capture that is never replayed.  So a key of MIXED
        # sizes runs eagerly the first time it is seen and is captured the second time; the cache is a bounded LRU (graphs hold
        # private memory pools).
        ent = self._group_graphs.get(key)
        if ent is None and len(set(g["input_sizes"])) > 1:
            self._group_graphs[key] = 1             # mixed frame sizes, seen once: eager, no capture yet
            self._group_graphs_trim()
            return run()
        if ent is None:
            ent = 1                                 # a group of equal-sized frames (every uniform stream): captured at first sight
        if ent == 1:
            run()                                   # this call's execution (also sets kernel attributes)
            torch.cuda.synchronize()
            ent = torch.cuda.CUDAGraph()
            with torch.cuda.graph(ent):
                run()                               # recorded, not executed
            self._group_graphs.pop(key, None)
            self._group_graphs[key] = ent           # most recently used last
            self._group_graphs_trim()
            return
        self._group_graphs[key] = self._group_graphs.pop(key)
        ent.replay()

    _GROUP_GRAPH_CAP = 64        # chunk graphs kept (a uniform stream needs n_chunks x ramp sizes x 2 buffer sets ~ 30)

    def _group_graphs_trim(self):
        while len(self._group_graphs) > self._GROUP_GRAPH_CAP:
            self._group_graphs.pop(next(iter(self._group_graphs)))

    @torch.no_grad()
    def group_bundle(self, g, b):
        """The decoder's per-image constants of frame b of a finished group, into the decoder plan's INACTIVE slot -> a bundle
        for adopt_prefetched()."""
        plan = self.model.decoder_plan()
        slot = 1 - plan.slot
        plan.set_image(g["feat"][b], g["dtok"][b], slot=slot, activate=False)
        return dict(original_size=tuple(g["sizes"][b]), input_size=tuple(g["input_sizes"][b]), feat=plan.states[slot]["feat"],
                    plan=plan, slot=slot)

    def adopt_prefetched(self, bundle):
        self.reset_image()
        self._adopt(bundle)

    # lazily materialised API views of the resident state
    @property
    def features(self):
        if self._feat_tok is None:
            return None
        return self._feat_tok.view(64, 64, 256).permute(2, 0, 1).unsqueeze(0)

    @features.setter
    def features(self, v):
        self._feat_tok = None if v is None else v[0].permute(1, 2, 0).reshape(4096, 256).contiguous()

    @property
    def dino_feats(self):
        if not self.is_image_set:
            return None
        return self._plan.state["dtok"][:N_DINO].float().view(1, 73, 73, -1)

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def predict_fg_map(self, img_size=None):
        """[1,n_class,256,256] foreground logits (predictor.py:113-121)."""
        self._require_image()
        logits = self._plan.fg_logits()                               # [5329, C]
        planes = logits.t().contiguous().view(-1, 73, 73)
        return hip.bilinear_f32(planes, (256, 256)).unsqueeze(0)

    def predict(self, point_coords=None, point_labels=None, box=None, mask_input=None, multimask_output=True,
                return_logits=False, attn_sim=None, target_embedding=None):
        """Numpy front-end of predict_torch (predictor.py:133-212)."""
        self._require_image()
        coords_t = labels_t = box_t = None
        if point_coords is not None:
            assert point_labels is not None, "point_labels must be supplied if point_coords is supplied."
            pc = self.transform.apply_coords(point_coords, self.original_size)
            coords_t = torch.as_tensor(pc, dtype=torch.float, device=self.device)[None, :, :]
            labels_t = torch.as_tensor(point_labels, dtype=torch.int, device=self.device)[None, :]
        if box is not None:                                     # predictor.py:176-179
            bx = self.transform.apply_boxes(np.asarray(box, dtype=np.float64), self.original_size)
            box_t = torch.as_tensor(bx, dtype=torch.float, device=self.device).reshape(-1, 4)[:1]
        if mask_input is not None:
            raise NotImplementedError("mask prompts are not on Crowd-SAM's inference path")
        masks, iou, cls, low = self.predict_torch(coords_t, labels_t, box_t, None, multimask_output,
                                                  return_logits=return_logits)
        return masks[0].cpu().numpy(), iou[0].cpu().numpy(), low[0].cpu().numpy(), masks[0]

    @torch.no_grad()
    def predict_torch(self, point_coords, point_labels, boxes=None, mask_input=None, multimask_output=True,
                      return_logits=False, attn_sim=None, target_embedding=None):
        """(masks [B,C,H,W], iou [B,C], class_scores [B,C,n_class], low_res [B,C,256,256])."""
        self._require_image()
        low, iou, cls = self.decode_points(point_coords, point_labels, boxes, mask_input, attn_sim, target_embedding)
        if not multimask_output:
            low, iou, cls = low[:, :1], iou[:, :1], cls[:, :1]
        masks = self.model.postprocess_masks(low, self.input_size, self.original_size)
        if not return_logits:
            masks = masks > self.model.mask_threshold
        return masks, iou, cls, low

    @torch.no_grad()
    def decode_points(self, point_coords, point_labels, boxes=None, mask_input=None, attn_sim=None,
                      target_embedding=None):
        """Prompt encoder + mask decoder for one-point prompts (coords [B,1,2] in the input frame) or box prompts (boxes [B,4]).
        Returns views into the plan's workspace (valid until the next decode)."""
        self._require_image()
        if mask_input is not None or attn_sim is not None or target_embedding is not None:
            raise NotImplementedError("mask prompts / attention priors are not on Crowd-SAM's inference path")
        if boxes is not None:
            # box prompts (predictor.py:214-292 `boxes`, prompt_encoder.py:95-102): two corner tokens and no padding point -- the
            # same seven tokens per prompt as one point, so the fused decoder serves them; a point AND a box would be eight
            if point_coords is not None:
                raise NotImplementedError("a box prompt combined with points makes 8 tokens per prompt: not supported")
            bx = torch.as_tensor(boxes).reshape(-1, 4).to(device=self.device, dtype=torch.float32).contiguous()
            return self._plan.run_batch(None, boxes_f32=bx)
        if point_coords is None or point_coords.dim() != 3 or point_coords.shape[1] != 1:
            raise NotImplementedError("the HIP decoder takes exactly one positive point (or one box) per prompt")
        # trap 6: the frame scaling was done by the caller in float64; (x+0.5)/1024 is exact in fp32
        c = torch.as_tensor(point_coords)[:, 0, :].to(device=self.device, dtype=torch.float32).contiguous()
        lab = None
        if point_labels is not None:
            pl = torch.as_tensor(point_labels).reshape(-1).to(torch.int32)
            if not bool((pl == 1).all()):           # background (0) / not-a-point (-1) prompts: prompt_encoder.py:88-92
                if not bool(((pl >= -1) & (pl <= 1)).all()):
                    raise NotImplementedError("point labels must be 1, 0 or -1")
                lab = pl.to(self.device)
        return self._plan.run_batch(c, labels_i32=lab)

    @torch.no_grad()
    def decode_coords_device(self, coords_f32):
        """decode_points for coordinates that are already on the device: f32 [B,2] (x, y) in the input frame, e.g. from
        csam_eps_select (which applies ResizeLongestSide.apply_coords in float64 like the host path)."""
        self._require_image()
        assert coords_f32.is_cuda and coords_f32.dtype == torch.float32 and coords_f32.dim() == 2
        return self._plan.run_batch(coords_f32)

    def get_image_embedding(self):
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) to generate an embedding.")
        return self.features

    @property
    def device(self):
        return self.model.device

    def reset_image(self):
        self.is_image_set = False
        self._feat_tok = None
        self._plan = None
        self.original_size = None
        self.input_size = None
        self.orig_h = self.orig_w = self.input_h = self.input_w = None

    def _require_image(self):
        if not self.is_image_set:
            raise RuntimeError("An image must be set with .set_image(...) before mask prediction.")
