#!/bin/bash
# kernel names (Tensile configuration strings) and times of the hipBLASLt kernels torch.matmul picks for the ESM / Llama shapes
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ref -o ref -- python /root/repo/tools/bench_gemm_ref.py > /root/repo/gpurun_out/gemm_ref.log 2>&1
f=$(find /tmp/prof_ref -name "*kernel_stats.csv" | head -1)
cp "$f" /root/repo/gpurun_out/gemm_ref_kernel_stats.csv
head -c 8000 /root/repo/gpurun_out/gemm_ref_kernel_stats.csv
