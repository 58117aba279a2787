"""CPU checks of the C-ABI boundary: the library loads without a GPU, exports every symbol that
include/csam.h declares, the ctypes table covers them, and the product path refuses to run on CPU."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "csam.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(csam_\w+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    from crowdsam_amd import build, hip
    build.build(verbose=False)
    lib = hip.lib()
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/csam.h but not exported"
    assert lib.csam_abi_version() >= 1
    assert lib.csam_last_error() is not None


def test_ctypes_table_matches_header():
    from crowdsam_amd import hip
    bound = set(hip.SIGNATURES) | set(hip.LONG_RETURNS) | {"csam_abi_version", "csam_last_error", "csam_adj_taps_bytes", "csam_adj_mfma_bytes"}
    assert set(_declared()) == bound


def test_argument_validation_needs_no_gpu():
    """Entry points validate arguments before touching the device: error codes + message on CPU."""
    from crowdsam_amd import hip
    lib = hip.lib()
    rc = lib.csam_gemm_f16(None, None, 0, None, 0, None, 0, 0, None, None, None, 0, 0, 0, 1, 1, 1)
    assert rc == -1 and b"null operand" in lib.csam_last_error()
    assert lib.csam_box_nms_workspace_bytes(4096) == 4096 * 4 + 4096 * 64 * 8 + 64


def test_product_path_fails_loudly_without_gpu():
    import segment_anything_cs as sa
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    sam = sa.sam_model_registry["vit_test128"](n_class=1)
    with pytest.raises(RuntimeError, match="MI355X only"):
        sam.image_encoder.plan()
    with pytest.raises(RuntimeError, match="MI355X only"):
        sam.decoder_plan()


def test_product_never_imports_oracle():
    """No module of the product path may import oracle/ (only tests, smoke() and bench's cpu leg)."""
    bad = []
    for pkg in ("crowdsam_amd", "segment_anything_cs", "crowdsam", "tools"):
        for dp, _, files in os.walk(os.path.join(ROOT, pkg)):
            for f in files:
                if f.endswith(".py"):
                    src = open(os.path.join(dp, f)).read()
                    if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_fragment_order_is_the_layout_the_header_documents():
    """hip.frag_order (plan-time permutation of the weights csam_token_block_* / csam_token_heads stream) against the index formula
    in include/csam.h: element (n, k) of a row-major [N][K] matrix at ((n / 16 * (K / 32) + k / 32) * 64 + (k % 32 / 8) * 16 + n % 16) * 8
    + k % 8 -- for one matrix and for a stack of matrices."""
    from crowdsam_amd import hip
    for lead, N, K in (((), 48, 96), ((3,), 32, 64)):
        w = torch.arange(int(torch.tensor(lead + (N, K)).prod()), dtype=torch.float32).reshape(*lead, N, K).to(torch.float16)
        f = hip.frag_order(w).reshape(*lead, -1)
        n = torch.arange(N)[:, None].expand(N, K)
        k = torch.arange(K)[None, :].expand(N, K)
        pos = ((n // 16 * (K // 32) + k // 32) * 64 + (k % 32 // 8) * 16 + n % 16) * 8 + k % 8
        assert sorted(pos.reshape(-1).tolist()) == list(range(N * K))
        if lead:
            for i in range(lead[0]):
                assert torch.equal(f[i][pos.reshape(-1)], w[i].reshape(-1))
        else:
            assert torch.equal(f[pos.reshape(-1)], w.reshape(-1))
