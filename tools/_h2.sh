cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_blk -- python tools/bench_decode.py > gpurun_out/blk.log 2>&1
python tools/prof_summary.py gpurun_out/prof_blk 2>&1 | head -8
