python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "mlp_chain" 2>&1 | tail -12
for c in 1 0; do echo CHAIN=$c; PCY_MLP_CHAIN=$c timeout 300 python tools/bench_decode.py 2>&1 | grep decode; done
PCY_MC_TRACE=1 GRAPH=0 python tools/bench_decode.py 2>&1 | grep -A9 "layer 10"
