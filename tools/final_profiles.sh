#!/bin/bash
# Round profile set (run on the GPU box from the repo root): bash tools/final_profiles.sh r02
# -> gpurun_out/<tag>_*: GPU test log, bench lines (default / encoder-only / EPS / stress), rocprofv3 kernel stats,
#    PMC HBM traffic, SQ issue counters of the decoder kernels.  Copy what is to be kept into profiles/.
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/${TAG}_gpu_tests.log
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
python bench.py --encoder-only --steps 50 --warmup 5 > gpurun_out/${TAG}_bench_encoder_only.json 2>> gpurun_out/${TAG}_bench.err
python bench.py --mode eps --grid 192 --points-per-batch 32 --stability-thresh 0.25 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer > gpurun_out/${TAG}_bench_eps_mode.json 2>> gpurun_out/${TAG}_bench.err
python bench.py --mode eps --grid 64 --points-per-batch 32 --stability-thresh 0.25 --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timer > gpurun_out/${TAG}_bench_eps_mode_grid64.json 2>> gpurun_out/${TAG}_bench.err
bash tools/dev_ablate.sh 2048 > gpurun_out/${TAG}_upscale_ablation.txt 2>&1
python tools/dev_bench_gemm.py > gpurun_out/${TAG}_gemm_shapes.txt 2>&1
python tools/dev_bench_attn.py > gpurun_out/${TAG}_attn.txt 2>&1
python bench.py --arch vit_h --grid 128 --frame 1500 --stability-thresh 0.0 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer --crowd-keep 0 > gpurun_out/${TAG}_bench_stress_vith.json 2>> gpurun_out/${TAG}_bench.err
bash tools/prof_bench.sh ${TAG}_bench --steps 10 --warmup 3 --crowd-keep 0
bash tools/prof_bench.sh ${TAG}_encoder_only --encoder-only --steps 20 --warmup 3
bash tools/prof_bench.sh ${TAG}_eps_mode --mode eps --grid 192 --points-per-batch 32 --stability-thresh 0.25 --steps 6 --warmup 3 --crowd-keep 0
bash tools/collect_pmc.sh --crowd-keep 0
bash tools/pmc_sq.sh 2048
