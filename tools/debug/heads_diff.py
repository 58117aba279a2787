import sys; sys.path.insert(0, "/root/repo")
import numpy as np, torch
from crowdsam_amd import synth
from crowdsam_amd.decoder import DecoderPlan
from tests.test_decoder_gpu import _set_image
cuda = torch.device("cuda:0")
sd = synth.make_sam_state_dict("vit_test128")
plan = DecoderPlan(sd, cuda, n_class=1, max_batch=64)
_set_image(plan, cuda)
plan.splitk = False
B = int(sys.argv[1]) if len(sys.argv) > 1 else 7
pts = np.random.RandomState(21).randint(0, 1024, size=(B, 2)).astype(np.float32)
coords = torch.from_numpy(pts).to(cuda).contiguous()
res = {}
for on in (True, False):
    plan.token_block = on
    plan.batch_graphs.clear()
    plan.run_batch(coords); torch.cuda.synchronize()
    res[on] = {k: plan.ws[k][: (B if k != "res_iou" else B * 4)].clone().float() for k in ("hyper", "iou", "res_iou")}
    res[on]["hh2"] = plan.ws["hh2"][:, :B].clone()
for k in ("hyper", "iou", "res_iou"):
    d = (res[True][k] - res[False][k]).abs()
    print(k, "max diff", d.max().item(), "rows differing:", torch.nonzero(d.view(B, -1).amax(1)).flatten().tolist())
d = (res[True]["hyper"] - res[False]["hyper"]).abs().view(B, 4, 32)
print("hyper diff per (b,l):", d.amax(2)[:8]); hh=(res[True]["hh2"]-res[False]["hh2"]).abs(); print("hh2 (old buffer, not written by the fused path) skip")
