#!/bin/bash
# Developer: build libcsam_hip_<name>.so for each main-loop variant of gemm4w_kernel (G4_ABLATE knobs of tools/gen/gen_gemm4w_asm.py;
# ablations give wrong results -- timing only).  Only gemm_f16.hip is recompiled; the other objects come from crowdsam_amd/build/.
# usage: bash tools/debug/gemm4w_variants.sh name1=knob,knob name2=knob ...   (a bare name = that knob; "base" = no knob)
set -e
cd "$(dirname "$0")/../.."
python -m crowdsam_amd.build > /dev/null
for spec in "$@"; do
  name="${spec%%=*}"; knobs="${spec#*=}"; [ "$knobs" = "base" ] && knobs=""
  inc=/tmp/gemm4w_asm_$name.inc
  G4_ABLATE="$knobs" G4_PLACE="$G4_PLACE" python tools/gen/gen_gemm4w_asm.py > $inc
  /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value \
    "-DG4_ASM_INC=\"$inc\"" $G4_DEFS -c crowdsam_amd/csrc/gemm_f16.hip -o /tmp/gemm_f16_$name.o 2>&1 | grep -E "error" || true
  objs=$(ls crowdsam_amd/build/*.o | grep -v gemm_f16)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o crowdsam_amd/libcsam_hip_$name.so $objs /tmp/gemm_f16_$name.o
  echo "built crowdsam_amd/libcsam_hip_$name.so ($knobs)"
done
