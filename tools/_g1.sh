mkdir -p gpurun_out/g1
(time python -m pytest tests -m gpu -q -x) > gpurun_out/g1/pytest.log 2>&1; tail -4 gpurun_out/g1/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --steps 5 --warmup 2 > gpurun_out/g1/bench.json 2> gpurun_out/g1/bench.err; tail -c 3000 gpurun_out/g1/bench.json; tail -5 gpurun_out/g1/bench.err
PCY_BENCH_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 python bench.py --steps 2 --warmup 1 --no-configs --no-cpu-baseline > gpurun_out/g1/bench_dist.json 2> gpurun_out/g1/bench_dist.err; tail -c 600 gpurun_out/g1/bench_dist.json; tail -3 gpurun_out/g1/bench_dist.err
