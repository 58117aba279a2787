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
    export HSA_XNACK=1 ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:verify_asan_link_order=0:alloc_dealloc_mismatch=0:new_delete_type_mismatch=0:detect_odr_violation=0:halt_on_error=0 CSAM_LIB=$R/crowdsam_amd/libcsam_hip_asan.so
    export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libstdc++.so.6)"   # libstdc++ too: libasan must find __cxa_throw when it starts
    # the library's DT_NEEDED names clang's runtime: point that name at gcc's libasan (same __asan_* ABI, v8)
    mkdir -p /tmp/asanlib && ln -sf $(gcc -print-file-name=libasan.so) /tmp/asanlib/libclang_rt.asan-x86_64.so
    export LD_LIBRARY_PATH=/tmp/asanlib:$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), \"lib\"))"):/opt/rocm/lib:$LD_LIBRARY_PATH   # the dlopen interceptor drops RUNPATH lookups
    python -c "import ctypes; l = ctypes.CDLL('$R/crowdsam_amd/libcsam_hip_asan.so'); print('library loads, abi', l.csam_abi_version())" 2>&1 | tail -3
    for x in 1 0; do
      HSA_XNACK=$x python -c "import torch; torch.cuda.init(); print('HSA_XNACK=$x: device', torch.cuda.get_device_name(0)); a = torch.ones(8, device='cuda'); print((a + a).sum().item())" > /tmp/asan_dev.log 2>&1
      echo "HSA_XNACK=$x device init exit code $?: $(tail -3 /tmp/asan_dev.log | cut -c1-300 | tr '\n' ' ')"
    done
    # instrumented: attention, elementwise, mask post, NMS, connected components, RLE, evaluator, fp32 heads, token blocks;
    # deselected: the RLE tests -- the instrumented rle_scan_kernel (1024-thread workgroups + 360 B of scratch per lane) is rejected at
    # dispatch (HSA_STATUS_ERROR_INVALID_ISA) and the aborted queue takes the process with it
    # NOT instrumented (inline-asm kernels, see crowdsam_amd/build.py CSAM_EXTRA_FLAGS_SKIP): gemm_f16.hip, decoder_fused.hip
    timeout 500 python -m pytest -q -m gpu -k "not rle" tests/test_post_gpu.py tests/test_regions_gpu.py tests/test_mask_nms.py \
      tests/test_encoder_gpu.py::test_layernorm tests/test_encoder_gpu.py::test_win_attn tests/test_gemm_gpu.py::test_linear_f32_small_heads > /tmp/asan_pytest.log 2>&1; echo "pytest exit code: $?"; head -c 3000 /tmp/asan_pytest.log; echo ...; tail -25 /tmp/asan_pytest.log
  } > gpurun_out/asan.txt 2>&1
  tail -30 gpurun_out/asan.txt
  ;;
*) echo "usage: $0 build|run"; exit 2;;
esac
