#!/bin/bash
# developer A/B: time one decoder batch with each alternative build of the library (crowdsam_amd/build.py CSAM_BUILD_TAG)
B=${1:-2048}
for lib in "" $(ls crowdsam_amd/libcsam_hip_*.so 2>/dev/null | sed "s/.*libcsam_hip\(_[a-z0-9]*\).so/\1/"); do
  f=crowdsam_amd/libcsam_hip${lib}.so
  [ -f $f ] || continue
  echo "== $f"
  CSAM_LIB=$PWD/$f python tools/dev_bench_decoder.py $B 2>&1 | grep -E "per batch|upscale|t2i_stream|i2t_stream|pool"
done
