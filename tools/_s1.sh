mkdir -p gpurun_out/s1
(time python -m pytest tests/test_gpu_fulldepth.py -x -q -s) > gpurun_out/s1/fulldepth.log 2>&1
python tools/bench_gemv_l3.py > gpurun_out/s1/l3.log 2>&1
for m in 0 1; do for d in 0 200 500; do PCY_AO_MAP=$m PCY_AO_DELAY=$d python tools/bench_decode.py > gpurun_out/s1/dec_m${m}_d${d}.log 2>&1; done; done
tail -n 3 gpurun_out/s1/dec_*.log
