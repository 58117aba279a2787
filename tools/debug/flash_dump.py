"""debug: tile-0 s_init of rt=1 as the kernel had it in registers vs recomputed from memory, on launches whose output differs"""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from crowdsam_amd import hip, synth
from crowdsam_amd.encoder import EncoderPlan
D, depth, heads, gidx = synth.SAM_CONFIGS["vit_l"]
dev = torch.device("cuda:0")
sd = synth.make_sam_state_dict("vit_l")
plan = EncoderPlan(sd, "image_encoder.", D, depth, heads, gidx, dev)
img = torch.from_numpy(synth.synthetic_crowd_frame(7, 1024, 150)).permute(2, 0, 1).float().contiguous().to(dev)
ws, nH = plan.ws, plan.heads
scale = plan.hd ** -0.5
target = 11
hip.sam_im2col(img, ws["col"])
x = hip.gemm_f16(ws["col"], plan.patch_w, out=ws["x"], bias=plan.patch_b, residual=plan.pos)
for i, b in enumerate(plan.blocks):
    hip.layernorm(x, b["ln1_g"], b["ln1_b"], 1e-6, out=ws["h"])
    hip.gemm_f16(ws["h"], b["qkv_w"], out=ws["qkv"], bias=b["qkv_b"])
    if i == target:
        break
    if b["is_global"]:
        hip.relpos_raw(ws["qkv"], b["relcat"], ws["traw"], nH)
        hip.flash_attn(ws["qkv"], ws["attn"], 4096, nH, scale, D, relpos=ws["traw"], q_prescaled=True)
    else:
        hip.win_attn(ws["qkv"], b["qkv_b"], b["relcat"], ws["attn"], D, nH, scale)
    hip.gemm_f16(ws["attn"], b["proj_w"], out=x, bias=b["proj_b"], residual=x)
    hip.layernorm(x, b["ln2_g"], b["ln2_b"], 1e-6, out=ws["h"])
    hip.gemm_f16(ws["h"], b["lin1_w"], out=ws["mlp"], bias=b["lin1_b"], act=hip.ACT_GELU)
    hip.gemm_f16(ws["mlp"], b["lin2_w"], out=x, bias=b["lin2_b"], residual=x)
b = plan.blocks[target]
hip.relpos_raw(ws["qkv"], b["relcat"], ws["traw"], nH)
torch.cuda.synchronize()
qkv, traw = ws["qkv"].clone(), ws["traw"].clone()
nwg = 32 * nH
dbg = torch.zeros(nwg, 256, 48, device=dev)
lib = hip.lib()
lib.csam_dbg_set_flash.argtypes = [ctypes.c_void_p]
assert lib.csam_dbg_set_flash(ctypes.c_void_p(dbg.data_ptr())) == 0
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
nbad = 0
for r in range(N):
    dbg.zero_()
    o = torch.empty(4096, D, dtype=torch.float16, device=dev)
    hip.flash_attn(qkv, o, 4096, nH, scale, D, relpos=traw, q_prescaled=True)
    torch.cuda.synchronize()
    s_used, tw_scaled, ref = dbg[:, :, :16], dbg[:, :, 16:32], dbg[:, :, 32:]
    bad = (s_used != ref)
    if bad.any():
        nbad += 1
        idx = bad.nonzero()
        wgs = idx[:, 0].unique().tolist()
        print("launch %d: %d register values of the tile-0 rt=1 s_init differ from memory; workgroups %s" % (r, len(idx), wgs[:6]))
        for wg in wgs[:2]:
            sub = idx[idx[:, 0] == wg]
            tids = sub[:, 1].unique()
            print("   wg %d: waves %s lanes %s regs(kt*4+j) %s" % (wg, (tids // 64).unique().tolist(), (tids % 64).unique().tolist()[:20],
                  sub[:, 2].unique().tolist()))
            t0 = int(tids[0]); k0 = int(sub[sub[:, 1] == t0][0, 2])
            print("      e.g. tid %d reg %d: used %.5f  memory %.5f  scaled-twr reg %.5f  (memory - c0 would be %.5f)" %
                  (t0, k0, float(s_used[wg, t0, k0]), float(ref[wg, t0, k0]), float(tw_scaled[wg, t0, k0]),
                   float(ref[wg, t0, k0] - (s_used[wg, t0, (k0 + 1) % 16] - tw_scaled[wg, t0, (k0 + 1) % 16]))))
print("launches with a wrong tile-0 s_init: %d of %d" % (nbad, N))
