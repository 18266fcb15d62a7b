mkdir -p gpurun_out/s4
(timeout 900 python -m pytest tests/test_gpu_models.py -q -x -k "pipelined or attn_o_fused" ) > gpurun_out/s4/pytest_pipe.log 2>&1
tail -8 gpurun_out/s4/pytest_pipe.log
for p in 1 0; do PCY_DECODE_PIPE=$p timeout 300 python tools/bench_decode.py > gpurun_out/s4/dec_pipe$p.log 2>&1; tail -2 gpurun_out/s4/dec_pipe$p.log; done
cd /tmp && export TMPDIR=/tmp
PCY_DECODE_PIPE=1 timeout 600 rocprofv3 --kernel-trace -d /tmp/tr1 -- python $GRAFT_REPO_ROOT/tools/bench_decode.py > $GRAFT_REPO_ROOT/gpurun_out/s4/trace_run.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/trace_pipe.py /tmp/tr1 -3 40 > gpurun_out/s4/trace_pipe1.txt 2>&1
head -60 gpurun_out/s4/trace_pipe1.txt
