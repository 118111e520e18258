#!/bin/bash
# Round-5 profile collection on the GPU box (run through gpurun from the repo root).  Outputs under gpurun_out/prof5/; the summaries
# judged live under profiles/r05_* (copied after a look).  PMC passes are their own runs with --kernel-trace only.
#   1. bench.py (default): JSON line with roofline (+ power / sclk), kernel classes, c2, c5, fp32_exact, cpu_baseline
#   2. rocprofv3 --kernel-trace --stats of bench.py (2 timed steps + 1 warm-up), of tools/bench_c2.py and of the config-5 pipeline
#      (tools/bench_pipeline.py: serial / alternate / pipelined schedules in one trace; the overlap of the two stages is in its own
#      HIP-event device timeline, c5_pipeline.txt - a profiler's trace slows the decode chain and merges the three schedules)
#   3. --pmc SQ pass (GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES) of bench.py and config 2
#   5. config 5 (56 dialogues) and the ragged test directory without a profiler, the ragged directory's device idle gaps (gap_summary.py)
#   4. --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py (separate; pmc_summary.py: one row per GEMM epilogue instance)
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof5
rm -rf $OUT; mkdir -p $OUT
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
DIALOGUES=56 python tools/bench_pipeline.py > $OUT/c5_pipeline.txt 2>&1
PASSES=3 python tools/ragged_dir.py > $OUT/ragged_dir.txt 2>&1
cd /tmp && export TMPDIR=/tmp
PASSES=1 rocprofv3 --kernel-trace --output-format csv -d $OUT/ragged -- python $REPO/tools/ragged_dir.py > $OUT/ragged_trace.log 2>&1
f=$(find $OUT/ragged -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python $REPO/tools/gap_summary.py $f 12 -$(grep -o "[0-9.]* ms = files" $OUT/ragged_trace.log | head -1 | cut -d" " -f1) > $OUT/ragged_gaps.txt 2>&1
B="python $REPO/bench.py --no-cpu-baseline --no-fp32-exact --no-c2 --no-c5"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --steps 2 --warmup 1 > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_stats -- python $REPO/tools/bench_c2.py > $OUT/c2_stats.log 2>&1
DIALOGUES=28 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5_stats -- python $REPO/tools/bench_pipeline.py > $OUT/c5_stats.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_sq -- $B --steps 1 --warmup 1 > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/c2_pmc_sq -- python $REPO/tools/bench_c2.py > $OUT/c2_pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --steps 1 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B --steps 1 --warmup 1 > $OUT/pmc_write.log 2>&1
cd $REPO
for d in stats c2_stats c5_stats; do
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${d}_kernel_stats.csv && python tools/stats_summary.py $f > $OUT/${d}_summary.txt 2>&1
done
f=$(find $OUT/pmc_sq -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/sq_summary.py $f $OUT/sq_counters.json > $OUT/sq_counters.txt 2>&1
f=$(find $OUT/c2_pmc_sq -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/sq_summary.py $f $OUT/c2_sq_counters.json > $OUT/c2_sq_counters.txt 2>&1
ff=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); fw=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_summary.py $ff $fw $OUT/pmc_summary.json > $OUT/pmc_summary.txt 2>&1
f=$(find $OUT/stats -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/gemm_by_grid.py $f > $OUT/gemm_by_grid.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete
tail -3 $OUT/*.log | cut -c1-300
cut -c1-600 $OUT/bench_n1.json
