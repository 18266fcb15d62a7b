timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "fused_layer" 2>&1 | tail -3
for c in 1 0; do echo LAYER=$c; PCY_LAYER_FUSED=$c timeout 300 python tools/bench_decode.py 2>&1 | grep decode; done
PCY_MC_TRACE=1 GRAPH=0 python tools/bench_decode.py 2>&1 | grep -A13 "layer 16 attention block, 64"
