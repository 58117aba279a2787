#!/bin/bash
# Developer: same-box A/B of two builds of the library (crowdsam_amd/libcsam_hip.so vs crowdsam_amd/libcsam_old.so, the latter built by
# hand from an older source) -- boxes differ by a few per cent, so before/after numbers from different gpurun calls do not compare.
# usage: bash tools/debug/lib_ab.sh '<command>' [rounds]
cd "$(dirname "$0")/../.."
cmd="$1"; rounds="${2:-2}"
cp crowdsam_amd/libcsam_hip.so /tmp/lib_new.so
cp crowdsam_amd/libcsam_old.so /tmp/lib_old.so
for r in $(seq 1 "$rounds"); do
  for v in new old; do
    cp /tmp/lib_$v.so crowdsam_amd/libcsam_hip.so
    echo "== $v (round $r)"
    bash -c "$cmd" 2>&1 | grep -v amdgpu.ids
  done
done
cp /tmp/lib_new.so crowdsam_amd/libcsam_hip.so
