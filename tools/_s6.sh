mkdir -p gpurun_out/s6
PCY_PIPE_TRACE=1 PCY_DECODE_PIPE=1 timeout 300 python tools/bench_decode.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/s6/trace.log
