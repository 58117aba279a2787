"""Developer benchmark: one decoder batch (B prompts) on synthetic weights; per-kernel-family timing via events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from crowdsam_amd import synth, hip
from crowdsam_amd.decoder import DecoderPlan

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
hip.GRAPHS_ENABLED = False
specs = [s for s in synth.sam_param_specs(128, 4, 2, (1, 3)) ]
sd = synth.make_state_dict(specs, 0)
plan = DecoderPlan(sd, torch.device("cuda"), 1, B)
feat = torch.randn(4096, 256, device="cuda")
dtok = torch.zeros(5376, 1024, dtype=torch.float16, device="cuda"); dtok[:5329] = torch.randn(5329, 1024, device="cuda").half()
plan.set_image(feat, dtok)
coords = torch.rand(B, 2, device="cuda") * 1023
for _ in range(2): plan.run_batch(coords)
torch.cuda.synchronize()
names = ["csam_i2t_fused", "csam_i2t_stream", "csam_i2t_rank", "csam_i2t_rank_proj", "csam_i2t_t2i", "csam_i2t_t2i_fold", "csam_t2i_stream", "csam_t2i_rank", "csam_t2i_fused", "csam_t2i_shared", "csam_upscale_fused", "csam_upscale_stream", "csam_gemm_f16", "csam_gemm_f16_batched", "csam_linear_f32",
         "csam_pool_adjoint", "csam_pool_adjoint_mfma", "csam_softmax_stats", "csam_layernorm", "csam_add_cast", "csam_token_self_attn", "csam_point_tokens", "csam_rowscale_bias"]
t = hip.KernelTimer(names); hip.set_timer(t)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 5
for _ in range(n): plan.run_batch(coords)
e1.record(); torch.cuda.synchronize(); hip.set_timer(None)
print(f"B={B}: {e0.elapsed_time(e1)/n:.3f} ms per batch, {e0.elapsed_time(e1)/n/B*1e3:.2f} us/prompt")
for k, v in sorted(t.summary().items(), key=lambda kv: -kv[1]["ms"]):
    print(f"  {k:28s} calls/batch {v['calls']//n:4d}  {v['ms']/n*1e3:9.1f} us/batch")
