#!/bin/bash
# Round profile set (run on the GPU box from the repo root): bash tools/final_profiles.sh r04
# -> gpurun_out/<tag>_*: GPU test log, bench lines (default / 20 steps / serial / encoder-only / EPS / stress), rocprofv3 kernel
#    stats, PMC HBM traffic, counter-derived MFMA utilisation, GEMM in-sequence probe.  Copy what is to be kept into profiles/.
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/${TAG}_gpu_tests.log
python bench.py > gpurun_out/${TAG}_bench_line_default_steps.json 2> gpurun_out/${TAG}_bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_line.json 2>> gpurun_out/${TAG}_bench.err
python bench.py --steps 20 --warmup 5 --serial --no-cpu-baseline --no-kernel-timer > gpurun_out/${TAG}_bench_line_serial.json 2>> gpurun_out/${TAG}_bench.err
python bench.py --encoder-only --steps 50 --warmup 5 > gpurun_out/${TAG}_bench_encoder_only.json 2>> gpurun_out/${TAG}_bench.err
for m in "" "--serial"; do
  python bench.py --mode eps --grid 192 --points-per-batch 32 --stability-thresh 0.25 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer $m > gpurun_out/${TAG}_bench_eps_mode${m/--/_}.json 2>> gpurun_out/${TAG}_bench.err
done
python bench.py --mode eps --grid 64 --points-per-batch 32 --stability-thresh 0.25 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer > gpurun_out/${TAG}_bench_eps_mode_grid64.json 2>> gpurun_out/${TAG}_bench.err
python bench.py --arch vit_h --grid 128 --frame 1500 --stability-thresh 0.0 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer --crowd-keep 0 > gpurun_out/${TAG}_bench_stress_vith.json 2>> gpurun_out/${TAG}_bench.err
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-kernel-timer 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/step, %.2f images/s, %.1f kept' % (d['ms_per_step'], d['value'], d['config']['kept_masks_per_image']))"; done > gpurun_out/${TAG}_bench_repeat.txt
python tools/dev_bench_gemm.py > gpurun_out/${TAG}_gemm_shapes.txt 2>&1
python tools/dev_gemm_ingraph.py > gpurun_out/${TAG}_gemm_ingraph.txt 2>&1
python tools/dev_gemm_breakdown.py > gpurun_out/${TAG}_gemm_breakdown.txt 2>&1
bash tools/prof_bench.sh ${TAG}_bench --steps 10 --warmup 3 --no-cpu-e2e --serial     # per-kernel averages: the serial trace
bash tools/prof_bench.sh ${TAG}_bench_pipelined --steps 10 --warmup 3 --no-cpu-e2e
bash tools/prof_bench.sh ${TAG}_encoder_only --encoder-only --steps 20 --warmup 3
bash tools/prof_bench.sh ${TAG}_eps_mode --mode eps --grid 192 --points-per-batch 32 --stability-thresh 0.25 --steps 6 --warmup 3 --crowd-keep 0
bash tools/collect_pmc.sh --crowd-keep 0 --serial     # one leg only: the collapsed leg would double the image count
cp gpurun_out/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
bash tools/pmc_mfma_calib.sh > /dev/null 2>&1
bash tools/pmc_mfma_util.sh encoder > /dev/null 2>&1
bash tools/pmc_mfma_util.sh frame > /dev/null 2>&1
CSAM_TIMING=1 python tools/dev_crowd_times.py 2>&1 | grep -v amdgpu.ids | head -3 > gpurun_out/${TAG}_crowd_stage_times.txt
python tools/debug/eps_overlap.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_eps_overlap.txt
bash tools/dev_ablate.sh 2048 > gpurun_out/${TAG}_upscale_ablation.txt 2>&1
