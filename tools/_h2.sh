cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
sed -i 's/for B in (1, 4, 5, 8, 16, 20, 32):/for B in (32,):/' tools/bench_decode_b.py
rocprofv3 --kernel-trace --stats -d /tmp/prof_b -- python tools/bench_decode_b.py > /dev/null 2>&1
python tools/prof_summary.py /tmp/prof_b | head -9
