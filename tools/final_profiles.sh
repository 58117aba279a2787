#!/bin/bash
# Round profile set (run on the GPU box from the repo root): bash tools/final_profiles.sh r05
# -> gpurun_out/<tag>_*: GPU test log, bench lines (default / 20 steps at --batch 4 and 1 / serial / encoder-only at 1 and 4 images
#    per pass / EPS / stress), rocprofv3 kernel stats, PMC HBM traffic, counter-derived MFMA utilisation, step traces.
#    Copy what is to be kept into profiles/.
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/${TAG}_gpu_tests.log
python bench.py > gpurun_out/${TAG}_bench_line_default_steps.json 2> gpurun_out/${TAG}_bench.err
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_line.json 2>> gpurun_out/${TAG}_bench.err
python bench.py --steps 20 --warmup 5 --weights random --no-cpu-baseline --no-extra-legs > gpurun_out/${TAG}_bench_line_random_weights.json 2>> gpurun_out/${TAG}_bench.err   # rounds 1-5's workload
python bench.py --steps 20 --warmup 5 --batch 1 --no-cpu-baseline --no-extra-legs > gpurun_out/${TAG}_bench_line_batch1.json 2>> gpurun_out/${TAG}_bench.err
python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-extra-legs > gpurun_out/${TAG}_bench_line_100_steps.json 2>> gpurun_out/${TAG}_bench.err
python bench.py --steps 20 --warmup 5 --serial --no-cpu-baseline --no-kernel-timer --no-extra-legs > gpurun_out/${TAG}_bench_line_serial.json 2>> gpurun_out/${TAG}_bench.err
python bench.py --encoder-only --steps 50 --warmup 5 > gpurun_out/${TAG}_bench_encoder_only.json 2>> gpurun_out/${TAG}_bench.err
python bench.py --encoder-only --batch 4 --steps 20 --warmup 3 > gpurun_out/${TAG}_bench_encoder_only_batch4.json 2>> gpurun_out/${TAG}_bench.err
for m in "" "--serial"; do
  python bench.py --mode eps --grid 192 --points-per-batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer $m > gpurun_out/${TAG}_bench_eps_mode${m/--/_}.json 2>> gpurun_out/${TAG}_bench.err
done
python bench.py --mode eps --grid 64 --points-per-batch 32 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer > gpurun_out/${TAG}_bench_eps_mode_grid64.json 2>> gpurun_out/${TAG}_bench.err
python bench.py --arch vit_h --grid 128 --frame 1500 --stability-thresh 0.0 --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timer --crowd-keep 0 > gpurun_out/${TAG}_bench_stress_vith.json 2>> gpurun_out/${TAG}_bench.err
# no step > 1.5 x median in 100 timed frames (5 runs of 20 steps, when each generate() returned)
for i in 1 2 3 4 5; do CSAM_BENCH_TRACE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer --no-extra-legs 2>&1 | grep -E "step returns|\"metric\"" | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('step returns'):
        print(ln.strip())
    elif ln.startswith('{'):
        d = json.loads(ln); print('   -> %.2f ms/step, %.1f kept, loop: %s' % (d['ms_per_step'], d['config']['kept_masks_per_image'], d['config']['loop'][:40]))"; done > gpurun_out/${TAG}_step_trace.txt
for i in 1 2 3; do python bench.py --no-cpu-baseline --no-kernel-timer --no-extra-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.2f ms/step, %.2f images/s, %.1f kept (no flags: %s)' % (d['ms_per_step'], d['value'], d['config']['kept_masks_per_image'], d['config']['loop'][:30]))"; done > gpurun_out/${TAG}_bench_repeat.txt
python tools/dev_gemm4w.py > gpurun_out/${TAG}_gemm_vendor_batched.txt 2>&1            # own vs vendor, fp16-out and fp32-epilogue shapes
python tools/probe/attn_lib_probe.py > gpurun_out/${TAG}_attn_vendor.txt 2>&1
bash tools/debug/torch_launches.sh > gpurun_out/${TAG}_torch_launches.txt 2>&1
python tools/dev_bench_gemm_batch.py > gpurun_out/${TAG}_gemm_shapes_batched.txt 2>&1
python tools/dev_bench_gemm_k.py > gpurun_out/${TAG}_gemm_fixed_cost.txt 2>&1
python tools/dev_bench_encoder_batch.py > gpurun_out/${TAG}_encoder_batch.txt 2>&1
bash tools/prof_bench.sh ${TAG}_bench --steps 10 --warmup 3 --no-cpu-e2e --serial               # <tag>_bench_kernel_stats.txt: per-kernel averages, the SERIAL trace
bash tools/prof_bench.sh ${TAG}_bench_pipelined --steps 20 --warmup 3 --no-cpu-e2e            # <tag>_bench_pipelined_kernel_stats.txt: the headline loop (--batch 4)
bash tools/prof_bench.sh ${TAG}_encoder_only --encoder-only --steps 20 --warmup 3
bash tools/prof_bench.sh ${TAG}_encoder_only_batch4 --encoder-only --batch 4 --steps 10 --warmup 3
bash tools/prof_bench.sh ${TAG}_eps_mode --mode eps --grid 192 --points-per-batch 32 --steps 6 --warmup 3 --crowd-keep 0
bash tools/collect_pmc.sh --crowd-keep 0 --serial     # one leg only: the collapsed leg would double the image count
cp gpurun_out/pmc_traffic.json gpurun_out/${TAG}_pmc_traffic.json
bash tools/pmc_mfma_util.sh encoder > /dev/null 2>&1
bash tools/pmc_mfma_util.sh frame > /dev/null 2>&1
bash tools/dev_stream_timeline.sh --batch 4 > /dev/null 2>&1
bash tools/pmc_mfma_util.sh encoder4 > /dev/null 2>&1
