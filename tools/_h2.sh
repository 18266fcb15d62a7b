cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc_f -o f --output-format csv -- python tools/pmc_decode.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pmc_w -o w --output-format csv -- python tools/pmc_decode.py > /dev/null 2>&1
python tools/pmc_hbm_summary.py /tmp/pmc_f /tmp/pmc_w attn_block_kernel 438409216 "decode layer launch at t = 512..536: Wqkv + Wo + Wgu + Wdown + norms (436.2 MB) + 2.1 MB of cached K/V" > gpurun_out/r02_pmc_decode_layer.json; cat gpurun_out/r02_pmc_decode_layer.json | tail -12
python bench.py 2>/dev/null | tail -1 > gpurun_out/r02_bench_v3.json; python -c "
import json; d=json.load(open('gpurun_out/r02_bench_v3.json')); print(d['value'], d['phases']['decode_ms_per_token'], d['phases']['decode_step_GBps'], d['roofline'])"
