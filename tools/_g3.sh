mkdir -p gpurun_out/g3
python -m pytest tests/test_gpu_kernels.py tests/test_gpu_unified.py -m gpu -q -x -k "sample or sampling" 2>&1 | tail -15
