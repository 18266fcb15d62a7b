timeout 900 python -m pytest tests/test_gpu_models.py tests/test_gpu_kernels.py -m gpu -q -x -k "decode" 2>&1 | tail -3
for c in 1 0; do echo LAYER=$c; PCY_DECODE_LAYER=$c timeout 300 python tools/bench_decode.py 2>&1 | grep decode; done
