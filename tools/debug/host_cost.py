"""Developer: host (CPU) time per frame of the pipelined loop, and where it goes (cProfile, top cumulative entries) -- the GPU-bound loop
leaves the host idle most of a frame; with N ranks on one host that slack is what they share."""
import sys, os, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from crowdsam.model import CrowdSAM, settle_host
from crowdsam_amd import synth
from crowdsam.utils import DEFAULT_TEST_CONFIG
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
t = dict(DEFAULT_TEST_CONFIG)
t.update(grid_size=64, points_per_batch=4096, stability_score_thresh=0.25, pos_sim_thresh=-float("inf"),
         filter_thresh=float("inf"), max_prompts=4096)
cfg = {"environ": {"device": "cuda:0"}, "model": {"sam_model": "vit_l", "sam_arch": "crowdsam", "n_class": 1, "trainfree": False}, "test": t}
m = CrowdSAM(cfg, sam_state_dict=synth.make_sam_state_dict("vit_l"), dino_state_dict=synth.make_dino_state_dict())
np.random.seed(42)
m.box_nms_thresh = m.crop_nms_thresh = 1.0
m.pred_iou_thresh = 0.8890                      # bench.py CROWD_FROZEN
frames = [synth.synthetic_crowd_frame(i, 1024, 150) for i in range(n)]
for _ in m.generate_stream(frames):
    pass
settle_host()
torch.cuda.synchronize()
w0, c0 = time.perf_counter(), time.process_time()
th0 = time.thread_time()
for _ in m.generate_stream(frames):
    pass
torch.cuda.synchronize()
w, c, th = time.perf_counter() - w0, time.process_time() - c0, time.thread_time() - th0
print("%d frames: wall %.1f ms per frame, process CPU %.1f ms per frame (all threads), main thread %.1f ms per frame" % (n, 1e3 * w / n, 1e3 * c / n, 1e3 * th / n))
pr = cProfile.Profile()
pr.enable()
for _ in m.generate_stream(frames):
    pass
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18)
print("\n".join(l[:150] for l in s.getvalue().splitlines()[:34]))
