"""developer: in the shipped EPS configuration (grid 192, 32 prompts per batch), how long do the sweep and the next frame's
prefetch take on the GPU when they run beside each other (pipelined) and alone (serial)?  Events on the streams involved."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from crowdsam.model import CrowdSAM
from crowdsam.utils import DEFAULT_TEST_CONFIG
from crowdsam_amd import hip, synth

t = dict(DEFAULT_TEST_CONFIG)
t.update(grid_size=192, points_per_batch=32, stability_score_thresh=0.25)
cfg = {"environ": {"device": "cuda:0"}, "model": {"sam_model": "vit_l", "sam_arch": "crowdsam", "n_class": 1, "trainfree": False}, "test": t}
m = CrowdSAM(cfg, sam_state_dict=synth.make_sam_state_dict("vit_l"), dino_state_dict=synth.make_dino_state_dict())
np.random.seed(42)
frames = [synth.synthetic_crowd_frame(i, 1024, 150) for i in range(12)]
ev = {"sweep": [], "prefetch": []}
o_sel, o_reset, o_pf = hip.eps_select, m.predictor.reset_image, m._prefetch
state = {"open": None}

def sel(*a, **k):
    if state["open"] is None:
        e = torch.cuda.Event(enable_timing=True); e.record(); state["open"] = e
    return o_sel(*a, **k)

def reset():
    if state["open"] is not None:
        e = torch.cuda.Event(enable_timing=True); e.record(); ev["sweep"].append((state["open"], e)); state["open"] = None
    return o_reset()

def pf(image, early):
    s = m._pf_stream
    e0 = torch.cuda.Event(enable_timing=True)
    r = None
    if s is None:
        r = o_pf(image, early); return r
    with torch.cuda.stream(s):
        e0.record()
    r = o_pf(image, early)
    e1 = torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(s):
        e1.record()
    ev["prefetch"].append((e0, e1))
    return r

hip.eps_select, m.predictor.reset_image, m._prefetch = sel, reset, pf


def masked_stream(keep_every):
    """HIP stream restricted to the CUs whose index i has i % 4 < keep_every (hipExtStreamCreateWithCUMask)."""
    import ctypes
    rt = ctypes.CDLL("libamdhip64.so")
    words = (ctypes.c_uint32 * 8)()
    for i in range(256):
        if i % 4 < keep_every:
            words[i // 32] |= 1 << (i % 32)
    st = ctypes.c_void_p()
    rc = rt.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(st.value)


modes = ["serial", "pipelined", "serial", "pipelined"]
if len(sys.argv) > 1:
    keep = int(sys.argv[1])                     # 3 -> prefetch on 192 of the 256 CUs, 2 -> on 128
    modes = ["pipelined", "pipelined+mask", "pipelined", "pipelined+mask"]
for mode in modes:
    if mode.endswith("+mask"):
        m._pf_stream, m.predictor._side_stream = masked_stream(keep), masked_stream(keep)
        if os.environ.get("EAGER_PREFETCH") == "1":
            pass
        mode = "pipelined"
        print("   (prefetch + DINOv2 side stream restricted to %d of 256 CUs)" % (64 * keep))
    elif mode == "pipelined" and len(sys.argv) > 1:
        m._pf_stream, m.predictor._side_stream = None, None
    ev["sweep"].clear(); ev["prefetch"].clear()
    m.generate(frames[0], next_image=frames[0]); m.generate(frames[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    work = torch.cuda.Stream(priority=-1) if os.environ.get("WORK_STREAM") == "1" else torch.cuda.current_stream()
    with torch.cuda.stream(work):
        if mode == "serial":
            for f in frames[2:]:
                m.generate(f)
        else:
            list(m.generate_stream(frames[2:]))
    work.synchronize()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / len(frames[2:]) * 1e3
    sw = [a.elapsed_time(b) for a, b in ev["sweep"][2:]]
    pfm = [a.elapsed_time(b) for a, b in ev["prefetch"][1:]]
    print("%-9s %.2f ms/image | EPS sweep on the GPU: mean %.2f ms (min %.2f max %.2f) | prefetch (encoders + constants): %s"
          % (mode, dt, np.mean(sw), min(sw), max(sw), "mean %.2f ms" % np.mean(pfm) if pfm else "-"))
