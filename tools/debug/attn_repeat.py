"""debug: launch-to-launch repeatability of csam_win_attn and the non-biased csam_flash_attn on realistic operands
(ViT-L block 0 qkv for the window kernel; random qkv at the DINOv2 shape for flash)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crowdsam_amd import hip, synth
from crowdsam_amd.encoder import EncoderPlan
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
D, depth, heads, gidx = synth.SAM_CONFIGS["vit_l"]
sd = synth.make_sam_state_dict("vit_l")
plan = EncoderPlan(sd, "image_encoder.", D, depth, heads, gidx, dev)
img = torch.from_numpy(synth.synthetic_crowd_frame(7, 1024, 150)).permute(2, 0, 1).float().contiguous().to(dev)
ws = plan.ws
hip.sam_im2col(img, ws["col"])
x = hip.gemm_f16(ws["col"], plan.patch_w, out=ws["x"], bias=plan.patch_b, residual=plan.pos)
b = plan.blocks[0]
hip.layernorm(x, b["ln1_g"], b["ln1_b"], 1e-6, out=ws["h"])
hip.gemm_f16(ws["h"], b["qkv_w"], out=ws["qkv"], bias=b["qkv_b"])
torch.cuda.synchronize()
def rep(fn, name):
    outs = []
    for _ in range(N):
        outs.append(fn().clone())
    torch.cuda.synchronize()
    ref = torch.stack([o.float() for o in outs[:9]]).median(0).values
    bad = [i for i, o in enumerate(outs) if not torch.equal(o.float(), ref)]
    print(name, "launches with differences: %d of %d" % (len(bad), N))
    for i in bad[:3]:
        d = (outs[i].float() != ref).nonzero()
        print("   launch", i, "elements", len(d), "rows", d[:, 0].unique()[:8].tolist(), "max diff", float((outs[i].float() - ref).abs().max()))
out = torch.empty(4096, D, dtype=torch.float16, device=dev)
rep(lambda: (hip.win_attn(ws["qkv"], b["qkv_b"], b["relcat"], out, D, heads, 0.125), out)[1], "win_attn (ViT-L block 0)")
torch.manual_seed(2)
T = 5330
qkv = torch.randn(T, 3 * D, device=dev).half()
o2 = torch.empty(T, D, device=dev, dtype=torch.float16)
rep(lambda: (hip.flash_attn(qkv, o2, T, heads, 0.125, D), o2)[1], "flash_attn (T=5330, no bias)")
