timeout 600 python -m pytest tests/test_gpu_models.py -m gpu -q -x -k "layer_launch" 2>&1 | tail -2
python tools/bench_decode.py 2>&1 | grep decode
