mkdir -p gpurun_out/s3
(timeout 600 python -m pytest tests/test_gpu_models.py -q -x -k "pipelined or attn_o_fused or greedy_matches or small_prefill" ) > gpurun_out/s3/pytest_pipe.log 2>&1
tail -15 gpurun_out/s3/pytest_pipe.log
for p in 0 1; do PCY_DECODE_PIPE=$p timeout 300 python tools/bench_decode.py > gpurun_out/s3/dec_pipe$p.log 2>&1; tail -3 gpurun_out/s3/dec_pipe$p.log; done
