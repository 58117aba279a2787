cd $GRAFT_REPO_ROOT
TAG=r05
python -m pytest tests/test_gemm_gpu.py -q 2>&1 | tail -2
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_line.json 2> gpurun_out/${TAG}_bench.err
python bench.py > gpurun_out/${TAG}_bench_line_default_steps.json 2>> gpurun_out/${TAG}_bench.err
for m in "" "--serial"; do
  python bench.py --mode eps --grid 192 --points-per-batch 32 --stability-thresh 0.25 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer $m > gpurun_out/${TAG}_bench_eps_mode${m/--/_}.json 2>> gpurun_out/${TAG}_bench.err
done
python bench.py --mode eps --grid 64 --points-per-batch 32 --stability-thresh 0.25 --steps 20 --warmup 5 --no-cpu-baseline --no-kernel-timer > gpurun_out/${TAG}_bench_eps_mode_grid64.json 2>> gpurun_out/${TAG}_bench.err
tail -c 600 gpurun_out/${TAG}_bench_line.json
