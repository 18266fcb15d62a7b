mkdir -p gpurun_out/s5
for p in 1 0; do for g in 1 0; do echo "PIPE=$p GRAPH=$g"; PCY_DECODE_PIPE=$p GRAPH=$g timeout 300 python tools/bench_decode.py 2>&1 | grep decode | tee gpurun_out/s5/dec_p${p}_g${g}.log; done; done
