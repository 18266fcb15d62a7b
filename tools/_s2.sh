mkdir -p gpurun_out/s2
(time python -m pytest tests -m gpu -q -x) > gpurun_out/s2/pytest.log 2>&1
tail -5 gpurun_out/s2/pytest.log
python tools/probe_overlap.py > gpurun_out/s2/overlap.log 2>&1
cat gpurun_out/s2/overlap.log
python tools/bench_decode.py > gpurun_out/s2/dec.log 2>&1; tail -3 gpurun_out/s2/dec.log
PCY_AO_DELAY=800 python tools/bench_decode.py > gpurun_out/s2/dec_d800.log 2>&1; tail -3 gpurun_out/s2/dec_d800.log
