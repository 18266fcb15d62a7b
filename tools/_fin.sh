cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -- python bench.py --steps 2 --warmup 1 > gpurun_out/bench_prof.log 2>&1
python tools/prof_summary.py /tmp/prof_bench > gpurun_out/r02_bench_final_kernel_stats.csv 2>&1
python bench.py 2>/dev/null | tail -1 > gpurun_out/r02_bench_final.json
python -c "
import json; d=json.load(open('gpurun_out/r02_bench_final.json')); print(d['value'], d['phases'], d['roofline']['achieved'], d['roofline']['frac'], d['retrieval']['proteins_per_s'], {k:(v.get('tokens_per_s') or v.get('pairs_per_s')) for k,v in d['configs'].items() if isinstance(v,dict)})"
