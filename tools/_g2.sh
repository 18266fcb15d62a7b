mkdir -p gpurun_out/g2
python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "fused_rotary or 256x256" 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-configs > $GRAFT_REPO_ROOT/gpurun_out/g2/bench_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/g2/bench_prof.err
find /tmp/prof_bench -name "*kernel_stats*" | head; cp $(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/g2/kernel_stats.csv
head -30 $GRAFT_REPO_ROOT/gpurun_out/g2/kernel_stats.csv | cut -c1-200
