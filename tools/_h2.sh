cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -q -x -k "attn or attention or prefill or llama" 2>&1 | tail -3
python tools/bench_prefill1.py 2>&1 | grep prefill
rocprofv3 --kernel-trace --stats -d /tmp/prof_p -- python tools/bench_prefill1.py > /dev/null 2>&1
python tools/prof_summary.py /tmp/prof_p | head -16
python tools/bench_pairs.py 2>&1 | tail -3
