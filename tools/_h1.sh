timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gemv" 2>&1 | tail -5
for l in 1 0; do echo LDS=$l; PCY_GEMV_LDS=$l python tools/bench_gemv_b.py 2>&1 | grep "B="; done
for l in 1 0; do echo LDS=$l; PCY_GEMV_LDS=$l python tools/bench_decode_b.py 2>&1 | grep "B="; done
