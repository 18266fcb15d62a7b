cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -- python bench.py --steps 2 --warmup 1 > gpurun_out/bench_prof.log 2>&1
python tools/prof_summary.py /tmp/prof_bench > gpurun_out/r02_bench_v2_kernel_stats.csv 2>&1
python bench.py 2>/dev/null | tail -1 > gpurun_out/r02_bench_v2.json
