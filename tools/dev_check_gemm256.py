import sys, torch
sys.path.insert(0, "/root/repo")
from crowdsam_amd import hip
torch.manual_seed(0)
for (M, N, K, act) in [(256, 2048, 64, 0), (300, 2048, 128, 0), (4096, 3072, 1024, 0), (5330, 3072, 1024, 0), (4096, 4096, 1024, hip.ACT_GELU), (4096, 2048, 4096, hip.ACT_RELU)]:
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") * 0.05).half()
    b = torch.randn(N, device="cuda")
    out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.float16)
    hip.gemm_f16(a, w, out=out, bias=b, act=act)
    torch.cuda.synchronize()
    ref = a.float() @ w.float().t() + b
    if act == hip.ACT_GELU: ref = torch.nn.functional.gelu(ref)
    if act == hip.ACT_RELU: ref = torch.relu(ref)
    err = (out.float() - ref).abs().max().item()
    print(M, N, K, act, "max err", err, "ref max", ref.abs().max().item(), flush=True)
