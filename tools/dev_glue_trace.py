"""Developer: which Python lines of the hot path launch torch's own kernels (copies, elementwise, index ops)?  Runs the
bench's crowded frame under torch.profiler with stacks and prints, per source line of this repo, the aten ops and the
number of device kernels / copies they launch per image."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import collections
import numpy as np
import torch
from torch.profiler import ProfilerActivity, profile
from crowdsam.model import CrowdSAM
from crowdsam.utils import DEFAULT_TEST_CONFIG
from crowdsam_amd import synth

t = dict(DEFAULT_TEST_CONFIG)
t.update(grid_size=64, points_per_batch=4096, pos_sim_thresh=-float("inf"), filter_thresh=float("inf"), max_prompts=4096)   # bench default
cfg = {"environ": {"device": "cuda:0"}, "model": {"sam_model": "vit_l", "sam_arch": "crowdsam", "n_class": 1, "trainfree": False}, "test": t}
m = CrowdSAM(cfg, sam_state_dict=synth.blob_heads(synth.make_sam_state_dict("vit_l")), dino_state_dict=synth.make_dino_state_dict())
np.random.seed(42)
frames = [synth.synthetic_crowd_frame(i, 1024, 150) for i in range(5)]
for f in frames[:2]:
    m.generate(f)
torch.cuda.synchronize()
N = 3
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
             experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
    for f in frames[2:2 + N]:
        m.generate(f)
    torch.cuda.synchronize()
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
acc = collections.Counter()
dev_time = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::") or ev.device_time_total <= 0 and not ev.kernels:
        continue
    if ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue                     # count the outermost aten op only
    where = "?"
    for fr in ev.stack or []:
        if any(k in fr for k in ("crowdsam/", "crowdsam_amd/", "segment_anything_cs/")) and "tools/dev_glue" not in fr:
            where = fr[max(fr.find("crowdsam"), fr.find("segment_anything_cs")) if "crowdsam" in fr or "segment_anything_cs" in fr else 0:]
            i = min([x for x in (fr.find("crowdsam/"), fr.find("crowdsam_amd/"), fr.find("segment_anything_cs/")) if x >= 0])
            where = fr[i:]
            break
    nk = len(ev.kernels)
    if nk == 0:
        continue
    acc[(where, ev.name)] += nk
    dev_time[(where, ev.name)] += sum(k.duration for k in ev.kernels)
print("device kernels / copies launched by torch ops, per image (%d images):" % N)
tot = 0
for (where, name), n in sorted(acc.items(), key=lambda kv: -dev_time[kv[0]]):
    print("%6.1f launches %8.1f us  %-28s %s" % (n / N, dev_time[(where, name)] / N, name, where))
    tot += n
print("total %.1f torch-launched kernels per image, %.1f us" % (tot / N, sum(dev_time.values()) / N))
