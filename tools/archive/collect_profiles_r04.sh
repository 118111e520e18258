#!/bin/bash
# Round-4 profile collection on the GPU box (run through gpurun from the repo root).  Outputs under gpurun_out/prof4/; the summaries
# judged live under profiles/r04_* (copied by hand after a look).  PMC passes are their own runs with --kernel-trace only.
#   1. bench.py (default): JSON line with roofline (+ power / sclk), kernel classes, c2, fp32_exact, cpu_baseline
#   2. rocprofv3 --kernel-trace --stats of bench.py (2 timed steps + 1 warm-up = 3 passes) and of tools/bench_c2.py
#   3. --pmc SQ pass (GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES) of both
#   4. --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py (separate)
#   5. rocprofv3 --stats of the config-5 pipeline, the text2semantic bench and the vocoder bench
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof4
rm -rf $OUT; mkdir -p $OUT
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --no-fp32-exact --no-c2"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --steps 2 --warmup 1 > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_stats -- python $REPO/tools/bench_c2.py > $OUT/c2_stats.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_sq -- $B --steps 1 --warmup 1 > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/c2_pmc_sq -- python $REPO/tools/bench_c2.py > $OUT/c2_pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --steps 1 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B --steps 1 --warmup 1 > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_stats -- python $REPO/tools/bench_pipeline.py > $OUT/c5_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t2s_stats -- python $REPO/tools/bench_t2s.py > $OUT/t2s_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/voc_stats -- python $REPO/tools/vocoder_bench.py > $OUT/voc_stats.log 2>&1
cd $REPO
# reduce here (the raw traces are large; only summaries travel back)
for d in stats c2_stats c5_stats t2s_stats voc_stats; do
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${d}_kernel_stats.csv && python tools/stats_summary.py $f > $OUT/${d}_summary.txt 2>&1
done
f=$(find $OUT/pmc_sq -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/sq_summary.py $f $OUT/sq_counters.json > $OUT/sq_counters.txt 2>&1
f=$(find $OUT/c2_pmc_sq -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/sq_summary.py $f $OUT/c2_sq_counters.json > $OUT/c2_sq_counters.txt 2>&1
ff=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); fw=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_summary.py $ff $fw $OUT/pmc_summary.json > $OUT/pmc_summary.txt 2>&1
f=$(find $OUT/stats -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/gemm_by_grid.py $f > $OUT/gemm_by_grid.txt 2>&1
# drop the raw traces (keep gpurun_out small)
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
tail -3 $OUT/*.log | cut -c1-300
cut -c1-600 $OUT/bench_n1.json
