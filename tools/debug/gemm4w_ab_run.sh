#!/bin/bash
# Developer: time the gemm4w main-loop variants built by gemm4w_variants.sh on one box (tools/dev_gemm4w.py --ab), new / old lib around them.
cd "$(dirname "$0")/../.."
python -c "import torch; print(torch.cuda.get_device_name(0))"
for v in "$@"; do
  [ "$v" = "base" ] && v=""
  echo "== variant '$v'"
  CSAM_LIB=$PWD/crowdsam_amd/libcsam_hip$v.so timeout 300 python tools/dev_gemm4w.py --ab 2>&1 | grep -v amdgpu.ids
done
