"""Developer: when do the results of a stream's first frames come back -- generate_stream at batch 1, groups of 4, and groups that ramp
1, 2, 4 (CrowdSAM.group_ramp); host wall clock per result, after a rehearsal stream of the same shape.
   python tools/debug/stream_start.py [frames]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from crowdsam.model import CrowdSAM
from crowdsam_amd import synth
from crowdsam.utils import DEFAULT_TEST_CONFIG
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
t = dict(DEFAULT_TEST_CONFIG)
t.update(grid_size=64, points_per_batch=4096, stability_score_thresh=0.25, pos_sim_thresh=-float("inf"),
         filter_thresh=float("inf"), max_prompts=4096)
cfg = {"environ": {"device": "cuda:0"}, "model": {"sam_model": "vit_l", "sam_arch": "crowdsam", "n_class": 1, "trainfree": False}, "test": t}
m = CrowdSAM(cfg, sam_state_dict=synth.make_sam_state_dict("vit_l"), dino_state_dict=synth.make_dino_state_dict())
np.random.seed(42)
m.box_nms_thresh = m.crop_nms_thresh = 1.0
m.pred_iou_thresh = 0.8890                      # bench.py CROWD_FROZEN
frames = [synth.synthetic_crowd_frame(i, 1024, 150) for i in range(n)]
for f in frames[:2]:
    m.generate(f)


def stream(batch, ramp, serial_before=0):
    m.group_ramp = ramp
    for f in frames[:serial_before]:
        m.generate(f)                  # bench.py's warm-up frames: serial calls between the rehearsal and the timed stream
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ends = []
    for out in m.generate_stream(frames, batch=batch):
        ends.append(time.perf_counter() - t0)
    steps = np.diff([0.0] + ends) * 1e3
    return steps


fresh = os.environ.get("FRESH")          # bench.py's order: the rehearsal runs on ONE frame repeated, the timed frames are new arrays
if fresh:
    keep = frames
for name, batch, ramp in (("batch 1", 1, False), ("groups of 4", 4, False), ("ramp 1 2 4", 4, True)):
    if fresh:
        frames = [keep[0]] * n
    stream(batch, ramp)
    if fresh:
        frames = [f.copy() for f in keep]
    for rep in range(3):
        s = stream(batch, ramp, serial_before=2 if rep == 2 else 0)
        print("%-12s total %.1f ms, first result %.1f | " % (name, s.sum(), s[0]) + " ".join("%.1f" % v for v in s), flush=True)
torch.cuda.synchronize()
t0 = time.perf_counter()
for f in frames[:4]:
    m.generate(f)
print("serial generate(): %.1f ms per frame" % ((time.perf_counter() - t0) * 250))
