"""debug: which kernel of the full-depth SAM encoder is not bitwise repeatable?  Runs the op sequence of
EncoderPlan.forward twice on the same input and compares an integer checksum of every op's output."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from crowdsam_amd import hip, synth
from crowdsam_amd.encoder import EncoderPlan
arch = sys.argv[1] if len(sys.argv) > 1 else "vit_l"
D, depth, heads, gidx = synth.SAM_CONFIGS[arch]
dev = torch.device("cuda:0")
sd = synth.make_sam_state_dict(arch)
plan = EncoderPlan(sd, "image_encoder.", D, depth, heads, gidx, dev)
img = torch.from_numpy(synth.synthetic_crowd_frame(7, 1024, 150)).permute(2, 0, 1).float().contiguous().to(dev)

def cs(t):
    it = torch.int16 if t.element_size() == 2 else torch.int32
    return int(t.contiguous().view(-1).view(it).to(torch.int64).sum().item())

def run():
    log = []
    ws, nH = plan.ws, plan.heads
    scale = plan.hd ** -0.5
    hip.sam_im2col(img, ws["col"]); log.append(("im2col", cs(ws["col"])))
    x = hip.gemm_f16(ws["col"], plan.patch_w, out=ws["x"], bias=plan.patch_b, residual=plan.pos); log.append(("patch", cs(x)))
    for i, b in enumerate(plan.blocks):
        hip.layernorm(x, b["ln1_g"], b["ln1_b"], 1e-6, out=ws["h"]); log.append((f"{i}.ln1", cs(ws["h"])))
        hip.gemm_f16(ws["h"], b["qkv_w"], out=ws["qkv"], bias=b["qkv_b"]); log.append((f"{i}.qkv", cs(ws["qkv"])))
        if not plan.fused_attn:
            plan._attn_generic(b)
        elif b["is_global"]:
            hip.relpos_raw(ws["qkv"], b["relcat"], ws["traw"], nH); log.append((f"{i}.relpos", cs(ws["traw"])))
            hip.flash_attn(ws["qkv"], ws["attn"], 4096, nH, scale, D, relpos=ws["traw"], q_prescaled=True)
        else:
            hip.win_attn(ws["qkv"], b["qkv_b"], b["relcat"], ws["attn"], D, nH, scale)
        log.append((f"{i}.attn{'G' if b['is_global'] else 'W'}", cs(ws["attn"])))
        hip.gemm_f16(ws["attn"], b["proj_w"], out=x, bias=b["proj_b"], residual=x); log.append((f"{i}.proj", cs(x)))
        hip.layernorm(x, b["ln2_g"], b["ln2_b"], 1e-6, out=ws["h"]); log.append((f"{i}.ln2", cs(ws["h"])))
        hip.gemm_f16(ws["h"], b["lin1_w"], out=ws["mlp"], bias=b["lin1_b"], act=hip.ACT_GELU); log.append((f"{i}.fc1", cs(ws["mlp"])))
        hip.gemm_f16(ws["mlp"], b["lin2_w"], out=x, bias=b["lin2_b"], residual=x); log.append((f"{i}.fc2", cs(x)))
    return log

ref = run()
for rep in range(4):
    cur = run()
    bad = [(a[0]) for a, b in zip(ref, cur) if a != b]
    print("repeat", rep, "first differing ops:", bad[:6], "of", len(bad))
# isolate: the same op repeated on frozen inputs
