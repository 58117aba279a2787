#!/bin/bash
# Round profile set (run on the GPU box from the repo root): bash tools/final_profiles.sh r03
# -> gpurun_out/<tag>_*: GPU test log, bench lines (default / encoder-only / EPS / stress), rocprofv3 kernel stats,
#    PMC HBM traffic, SQ issue counters of the decoder kernels.  Copy what is to be kept into profiles/.
TAG=${1:-r03}
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
bash tools/prof_bench.sh ${TAG}_bench --steps 10 --warmup 3 --no-cpu-e2e
bash tools/prof_bench.sh ${TAG}_encoder_only --encoder-only --steps 20 --warmup 3
bash tools/prof_bench.sh ${TAG}_eps_mode --mode eps --grid 192 --points-per-batch 32 --stability-thresh 0.25 --steps 6 --warmup 3 --crowd-keep 0
bash tools/collect_pmc.sh --crowd-keep 0
bash tools/pmc_sq.sh 2048
cp gpurun_out/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
cp gpurun_out/pmc_sq.txt gpurun_out/${TAG}_pmc_sq_decoder.txt
bash tools/pmc_sq_encoder.sh && cp gpurun_out/pmc_sq_encoder.txt gpurun_out/${TAG}_pmc_sq_encoder.txt
python tools/dev_crowd_times.py 2>&1 | grep -v amdgpu.ids | head -3 > gpurun_out/${TAG}_crowd_stage_times.txt
bash tools/dev_crowd_prof.sh > /dev/null 2>&1; cp gpurun_out/crowd_kernel_stats.txt gpurun_out/${TAG}_crowd_tail_kernels.txt
python tools/debug/fused_time.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_i2t_t2i_vs_separate.txt
for b in 530 700; do python tools/debug/fused_diff.py $b proj 2>&1 | grep mismatch >> gpurun_out/${TAG}_i2t_t2i_vs_separate.txt; done
./tools/probe/valu_mfma_overlap > gpurun_out/${TAG}_valu_mfma_overlap_probe.txt 2>&1
./tools/probe/mfma_srcc_lds_war > gpurun_out/${TAG}_mfma_srcc_lds_war_probe.txt 2>&1
python tools/debug/flash_repeat.py 11 1000 q > gpurun_out/${TAG}_flash_repeat.txt 2>&1
python tools/debug/tile_classes.py 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/${TAG}_crowd_tile_classes.txt
