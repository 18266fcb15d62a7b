timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "fused_layer" 2>&1 | tail -12
for c in "1 1" "1 0" "0 0"; do set -- $c; echo CHAIN=$1 BLOCK=$2; PCY_MLP_CHAIN=$1 PCY_ATTN_BLOCK=$2 timeout 300 python tools/bench_decode.py 2>&1 | grep decode; done
