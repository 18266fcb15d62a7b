timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "layer_launch or layers_only" 2>&1 | tail -3
for c in 1 0; do echo STEP=$c; PCY_DECODE_STEP=$c python tools/bench_decode.py 2>&1 | grep decode; done
