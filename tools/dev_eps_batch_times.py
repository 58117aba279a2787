"""Developer timing: host + device cost of the pieces of ONE EPS batch (32 prompts, grid 192) -- where does the
per-batch wall time of the shipped configuration go?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crowdsam.model import CrowdSAM
from crowdsam_amd import synth, hip
from crowdsam.utils import DEFAULT_TEST_CONFIG
t = dict(DEFAULT_TEST_CONFIG); t.update(grid_size=192, stability_score_thresh=0.25)
cfg = {"environ": {"device": "cuda:0"}, "model": {"sam_model": "vit_l", "sam_arch": "crowdsam", "n_class": 1, "trainfree": False}, "test": t}
m = CrowdSAM(cfg, sam_state_dict=synth.make_sam_state_dict("vit_l"), dino_state_dict=synth.make_dino_state_dict())
np.random.seed(0)
img = synth.synthetic_crowd_frame(2)
m.generate(img); m.generate(img)
# replay the crop set-up by hand, then time batches
m.crop_image(img, [0, 0, 1024, 1024]); m.predictor.set_image(m._frame_u8)
pts = m.sample_prompts().astype("int"); np.random.shuffle(pts)
store = m._result_store(*m.predictor.original_size); store["counter"].zero_()
sync = torch.cuda.synchronize
acc = {}
def tick(k, t0):
    sync(); acc[k] = acc.get(k, 0.0) + (time.perf_counter() - t0) * 1e3; return time.perf_counter()
N = 12
p = m.predictor
for it in range(N + 2):
    if it == 2: acc.clear()
    sel, pts = pts[:32], pts[32:]
    sync(); t0 = time.perf_counter()
    tp = p.transform.apply_coords(sel, p.original_size)
    in_points = torch.as_tensor(tp)[:, None, :]
    t0 = tick("host_prep", t0)
    c = torch.as_tensor(in_points)[:, 0, :].to(device=m.device, dtype=torch.float32).contiguous()
    t0 = tick("h2d_coords", t0)
    th = time.perf_counter()
    low, iou, cls = p._plan.run_batch(c)
    acc["decode_host_only"] = acc.get("decode_host_only", 0.0) + (time.perf_counter() - th) * 1e3
    t0 = tick("decode_graph", t0)
    bd = None
    th = time.perf_counter()
    # the remaining part of _process_batch (post kernels), by calling it on fresh points would decode again: time the whole call instead
    bd = m._process_batch(sel, p.original_size, [0, 0, 1024, 1024], store)
    acc["process_batch_host_only"] = acc.get("process_batch_host_only", 0.0) + (time.perf_counter() - th) * 1e3
    t0 = tick("process_batch_total(decode+post)", t0)
    rem = torch.as_tensor(np.ascontiguousarray(pts), dtype=torch.int32).to(m.device)
    t0 = tick("h2d_remaining_points", t0)
    bits = torch.empty(len(pts), dtype=torch.uint8, device=m.device)
    hip.occupancy_lookup(rem, store["masks"], bd["occ"], 32, 1024, 1024, bits, slot=bd["slot"])
    t0 = tick("lookup_kernel", t0)
    keep = ~bits.cpu().numpy().astype(bool)
    t0 = tick("d2h_bits", t0)
    pts2 = pts[keep]
    t0 = tick("numpy_filter", t0)
print({k: round(v / N, 3) for k, v in acc.items()}, "remaining points", len(pts))
os.environ["X"] = "1"
