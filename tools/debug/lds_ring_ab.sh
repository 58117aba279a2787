#!/bin/bash
# Developer: same-box A/B of the end-of-round-6 scheduling changes -- the shipped library against a build of the SAME
# source with every pin off (rounds 3-6's schedule): decoder sweep per kernel family at 4096 prompts, then the bench headline.
# Runs on the GPU box (hipcc is there too): bash tools/debug/lds_ring_ab.sh [rounds]   -> gpurun_out/lds_ring_ab.txt
cd "$(dirname "$0")/../.."
rounds="${1:-2}"
out=gpurun_out/lds_ring_ab.txt
mkdir -p gpurun_out
OFF="-DCSAM_UP_ONE_STORE_BLOCK=0 -DFUSE_PM_PIPE=0 -DFUSE_RD_PIPE=0 -DFUSE_GB_DEPTH=0 -DCSAM_UP_PIN=0 -DCSAM_SWAP_REDUCE=0"
CSAM_BUILD_TAG=old CSAM_EXTRA_FLAGS="-DCSAM_SWAP_REDUCE=0" CSAM_DEFS_decoder_fused="$OFF" CSAM_DEFS_decoder="-DPOOL_RING=0" CSAM_DEFS_gemm_f16="-DG128_READS_FIRST=0 -DG128_FULL_PATH=0 -DG128_HOIST=0 -DG4_RES_AHEAD=1 -DG4_FULL_PATH=0" \
  python -m crowdsam_amd.build > /dev/null 2>&1 || { echo "old-schedule build failed" | tee $out; exit 1; }
cp crowdsam_amd/libcsam_hip.so /tmp/lib_new.so
cp crowdsam_amd/libcsam_hip_old.so /tmp/lib_old.so
{
  echo "same-box A/B: new = shipped source, old = the same source with FUSE_PM_PIPE / FUSE_RD_PIPE / FUSE_GB_DEPTH / CSAM_UP_PIN / CSAM_SWAP_REDUCE / POOL_RING / G128_READS_FIRST / G4_FULL_PATH off, G4_RES_AHEAD 1"
  for r in $(seq 1 "$rounds"); do
    for v in new old; do
      cp /tmp/lib_$v.so crowdsam_amd/libcsam_hip.so
      echo "== $v (round $r): decoder batch of 4096 prompts, per kernel family"
      python tools/dev_bench_decoder.py 4096 2>&1 | grep -v amdgpu.ids | head -14
      echo "== $v (round $r): decoder batch of 32 prompts"
      python tools/dev_bench_decoder.py 32 2>&1 | grep -v amdgpu.ids | head -12
    done
  done
  for v in new old; do
    cp /tmp/lib_$v.so crowdsam_amd/libcsam_hip.so
    echo "== $v: bench.py --steps 20 --warmup 5 --no-extra-legs"
    python bench.py --steps 20 --warmup 5 --no-extra-legs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print({k: d[k] for k in ('value','ms_per_step')}, d.get('roofline',{}).get('frac'))"
  done
} 2>&1 | tee $out
cp /tmp/lib_new.so crowdsam_amd/libcsam_hip.so
