#!/bin/bash
# Round-6 evidence run on the GPU box: kernel tables (rocprofv3 --kernel-trace --stats) and PMC passes (MFMA counters; HBM traffic of the
# dominant decode kernel), written under gpurun_out/r6e/ -- the summaries are then copied into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6e; mkdir -p $O
cd $R
# 1. the bench line + its kernel table
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/prof_bench -o b -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python tools/prof_summary.py $O/prof_bench > $O/bench_kernel_stats.csv 2>/dev/null; rm -rf $O/prof_bench
# 2. the single-protein encoder and the single-prompt prefill: kernel tables + MFMA counters
for t in esm1 prefill1; do
  rocprofv3 --kernel-trace --stats -d $O/prof_$t -o p -- python tools/bench_$t.py > /dev/null 2>&1
  python tools/prof_summary.py $O/prof_$t > $O/${t}_kernel_stats.csv 2>/dev/null; rm -rf $O/prof_$t
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_$t -o p --output-format csv -- python tools/bench_$t.py > /dev/null 2>&1
  python tools/pmc_mfma_summary.py $O/pmc_$t > $O/pmc_mfma_$t.json 2>/dev/null; rm -rf $O/pmc_$t
done
# 3. ESM batch 25 kernel table + MFMA counters
N=25 rocprofv3 --kernel-trace --stats -d $O/prof_esm25 -o p -- python tools/bench_esm1.py > /dev/null 2>&1
python tools/prof_summary.py $O/prof_esm25 > $O/esm_b25_kernel_stats.csv 2>/dev/null; rm -rf $O/prof_esm25
N=25 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/pmc_esm25 -o p --output-format csv -- python tools/bench_esm1.py > /dev/null 2>&1
python tools/pmc_mfma_summary.py $O/pmc_esm25 > $O/pmc_mfma_esm_b25.json 2>/dev/null; rm -rf $O/pmc_esm25
# 4. HBM traffic of the dominant kernel (decode step): FETCH_SIZE and WRITE_SIZE in separate passes
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d $O/pmc_dec_$c -o p --output-format csv -- python tools/pmc_decode.py > /dev/null 2>&1
done
python tools/pmc_hbm_summary.py $O/pmc_dec_FETCH_SIZE $O/pmc_dec_WRITE_SIZE decode_step_kernel 14029094912 "all 32 decoder layers of a decode step at t = 512..536: 32 x (436.2 MB of weights + 2.1 MB of cached K/V)" > $O/pmc_decode_step.json 2>$O/pmc_decode_step.err; rm -rf $O/pmc_dec_FETCH_SIZE $O/pmc_dec_WRITE_SIZE
ls -la $O
# 5. the small-batch decode step at 4 rows: HBM traffic of decode_step_nb_kernel (weights once + 4 rows of K/V + the hand-over vectors)
for c in FETCH_SIZE WRITE_SIZE; do
  ROWS=4 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_nb_$c -o p --output-format csv -- python tools/pmc_decode.py > /dev/null 2>&1
done
python tools/pmc_hbm_summary.py $O/pmc_nb_FETCH_SIZE $O/pmc_nb_WRITE_SIZE decode_step_nb_kernel 14235663360 "all 32 decoder layers of a 4-row decode step at t = 512..536: 32 x (436.2 MB of weights + 4 x 2.1 MB of cached K/V)" > $O/pmc_decode_step_nb4.json 2>$O/pmc_decode_step_nb4.err; rm -rf $O/pmc_nb_FETCH_SIZE $O/pmc_nb_WRITE_SIZE
# 6. kernel table of the per-batch decode timing (decode_step_nb_kernel per batch size)
CHECK=0 BATCHES=2,4,5,8 rocprofv3 --kernel-trace --stats -d $O/prof_nb -o p -- python tools/bench_decode_nb.py > $O/decode_nb.log 2>&1
python tools/prof_summary.py $O/prof_nb > $O/decode_nb_kernel_stats.csv 2>/dev/null; rm -rf $O/prof_nb
ls -la $O
# 7. round 6: the batched decode step (launch path with the rotated K order; the opt-in mid-batch step beside it): per-batch timing + kernel table
CHECK=0 BATCHES=10,16,20,32 python tools/bench_decode_mb.py > $O/decode_mb.log 2>&1
CHECK=0 BATCHES=10,20,32 PCY_MB_MAX=0 rocprofv3 --kernel-trace --stats -d $O/prof_b -o p -- python tools/bench_decode_mb.py > /dev/null 2>&1
python tools/prof_summary.py $O/prof_b > $O/decode_batched_kernel_stats.csv 2>/dev/null; rm -rf $O/prof_b
CHECK=0 BATCHES=10 PCY_MC_TRACE=1 python tools/bench_decode_mb.py > $O/decode_mb_trace.log 2>&1
# 8. the probes behind the rotated K order (built here when the binaries did not travel)
for pr in stream_rows stream_cus; do [ -x tools/probes/$pr ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/probes/$pr tools/probes/$pr.hip; done
./tools/probes/stream_rows > $O/probe_stream_rows.log 2>&1
./tools/probes/stream_cus > $O/probe_stream_cus.log 2>&1
ls -la $O
# 9. round 6: ProCyon-Split (Llama-2-7B geometry) -- the bench line, its kernel table, HBM traffic of decode_step_mha_kernel, bit-identity + in-kernel stamps
python bench.py --geometry split --steps 5 --warmup 2 > $O/bench_split.json 2> $O/bench_split.err
rocprofv3 --kernel-trace --stats -d $O/prof_split -o b -- python bench.py --geometry split --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python tools/prof_summary.py $O/prof_split > $O/split_kernel_stats.csv 2>/dev/null; rm -rf $O/prof_split
for c in FETCH_SIZE WRITE_SIZE; do
  GEO=split rocprofv3 --pmc $c --kernel-trace -d $O/pmc_mha_$c -o p --output-format csv -- python tools/pmc_decode.py > /dev/null 2>&1
done
python tools/pmc_hbm_summary.py $O/pmc_mha_FETCH_SIZE $O/pmc_mha_WRITE_SIZE decode_step_mha_kernel 13227261952 "all 32 decoder layers of a ProCyon-Split (Llama-2-7B) decode step at t = 512..536: 32 x (404.8 MB of weights + 8.6 MB of cached K/V)" > $O/pmc_decode_step_mha.json 2>$O/pmc_decode_step_mha.err; rm -rf $O/pmc_mha_FETCH_SIZE $O/pmc_mha_WRITE_SIZE
python tools/bench_decode_mha.py > $O/decode_mha.log 2>&1
CHECK=0 MODES=step GRAPH=0 PCY_MC_TRACE=1 python tools/bench_decode_mha.py > $O/decode_mha_trace.log 2>&1
python tools/bench_gemm_ref.py > $O/gemm_ref.txt 2>&1
ls -la $O
