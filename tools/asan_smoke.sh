#!/bin/bash
# Sanitizer leg: an AddressSanitizer build of libcsam_hip (gfx950:xnack+, -fsanitize=address
# -shared-libsan) and the tiny-shape subset of the GPU tests under it.  Build here or on the GPU box:
#     bash tools/asan_smoke.sh build        # CSAM_BUILD_TAG=asan -> crowdsam_amd/libcsam_hip_asan.so
#     bash tools/asan_smoke.sh run          # on the GPU box; log -> gpurun_out/asan.txt (copy to profiles/<round>_asan.txt)
# This image ships the DEVICE half of the sanitizer (lib/llvm/lib/clang/*/lib/amdgcn/bitcode/asanrtl.bc) but neither clang's host
# runtime (libclang_rt.asan-x86_64.so) nor the instrumented HIP runtime (/opt/rocm/lib/asan): `run` substitutes gcc's libasan (same
# __asan ABI version) by LD_PRELOAD and reports honestly what happens -- a failure to start is a finding about the image, not a pass.
cd "$(dirname "$0")/.."
R=$(pwd)
case "$1" in
build)
  CSAM_BUILD_TAG=asan CSAM_ARCH=gfx950:xnack+ CSAM_EXTRA_FLAGS="-fsanitize=address -shared-libsan -g" \
    CSAM_EXTRA_FLAGS_SKIP=decoder_fused,gemm_f16 python -m crowdsam_amd.build 2>&1 | tail -3
  ;;
run)
  mkdir -p gpurun_out
  {
    echo "== sanitizer leg: $(date -u +%FT%TZ)"
    echo "host ASAN runtime: $(find /opt/rocm -name 'libclang_rt.asan*' | head -1 || true) (empty = absent); gcc libasan: $(gcc -print-file-name=libasan.so)"
    echo "instrumented HIP runtime (/opt/rocm/lib/asan): $(ls /opt/rocm/lib/asan 2>/dev/null | head -1) (empty = absent)"
    ldd crowdsam_amd/libcsam_hip_asan.so | grep -i asan
    export HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0 CSAM_LIB=$R/crowdsam_amd/libcsam_hip_asan.so
    export LD_PRELOAD=$(gcc -print-file-name=libasan.so)
    # instrumented: attention, elementwise, mask post, NMS, connected components, RLE, evaluator, fp32 heads, token blocks;
    # NOT instrumented (inline-asm kernels, see crowdsam_amd/build.py CSAM_EXTRA_FLAGS_SKIP): gemm_f16.hip, decoder_fused.hip
    timeout 500 python -m pytest -x -q -m gpu tests/test_post_gpu.py tests/test_regions_gpu.py tests/test_mask_nms.py \
      tests/test_encoder_gpu.py::test_layernorm tests/test_encoder_gpu.py::test_win_attn tests/test_gemm_gpu.py::test_linear_f32_small_heads 2>&1 | tail -25
    echo "exit code: $?"
  } > gpurun_out/asan.txt 2>&1
  tail -30 gpurun_out/asan.txt
  ;;
*) echo "usage: $0 build|run"; exit 2;;
esac
