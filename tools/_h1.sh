timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_models.py -m gpu -q -x -k "decode_mlp or fused_layer" 2>&1 | tail -5
for c in 1 0; do PCY_MLP_CHAIN=$c python tools/bench_decode_mlp.py 2>&1 | grep "decode mlp"; done
timeout 300 python tools/bench_decode.py 2>&1 | grep decode
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc_f -o f --output-format csv -- python tools/bench_decode_mlp.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc_w -o w --output-format csv -- python tools/bench_decode_mlp.py > /dev/null 2>&1
python tools/pmc_hbm_summary.py gpurun_out/pmc_f gpurun_out/pmc_w mlp_chain_kernel 352321536 "decode MLP chain launch: Wgu [2*14336,4096] + Wdown [4096,14336]" > gpurun_out/r02_pmc_mlp_chain.json; cat gpurun_out/r02_pmc_mlp_chain.json
