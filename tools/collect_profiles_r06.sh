#!/bin/bash
# Round-6 profile collection on the GPU box (run through gpurun from the repo root).  Outputs under gpurun_out/prof6/; the summaries
# judged live under profiles/r06_* (copied after a look).  PMC passes are their own runs with --kernel-trace only.
#   1. bench.py (default): JSON line with roofline (+ power / sclk), kernel classes, c1, c2, c5 (best schedule, ragged_eos), fp32_exact,
#      host, cpu_baseline
#   2. config 5 under every schedule (tools/bench_pipeline.py, 56 dialogues), the text2semantic decode at 1 ... 64 slots and its
#      continuous batching (tools/bench_t2s.py), the ragged test directory
#   3. rocprofv3 --kernel-trace --stats of bench.py (2 timed steps + 1 warm-up), of config 2, of the decode at 8 and 64 slots and of the
#      vocoder alone
#   4. --pmc SQ pass (GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES) and FETCH_SIZE / WRITE_SIZE passes of bench.py
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof6
rm -rf $OUT; mkdir -p $OUT
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
DIALOGUES=56 python tools/bench_pipeline.py > $OUT/c5_schedules.txt 2>&1
BATCHES=1,8,16,32,64 python tools/bench_t2s.py comix > $OUT/t2s_decode.txt 2>&1
PASSES=3 python tools/ragged_dir.py > $OUT/ragged_dir.txt 2>&1
cd /tmp && export TMPDIR=/tmp
B="python $REPO/bench.py --no-cpu-baseline --no-fp32-exact --no-c1 --no-c2 --no-c5"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --steps 2 --warmup 1 > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2_stats -- python $REPO/tools/bench_c2.py > $OUT/c2_stats.log 2>&1
for nb in 8 64; do
  MANY=0 BATCHES=$nb TOKENS=256 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t2s_b${nb}_stats -- python $REPO/tools/bench_t2s.py comix > $OUT/t2s_b${nb}_stats.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/voc_stats -- python $REPO/tools/vocoder_bench.py > $OUT/voc_stats.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_sq -- $B --steps 1 --warmup 1 > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --steps 1 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B --steps 1 --warmup 1 > $OUT/pmc_write.log 2>&1
cd $REPO
for d in stats c2_stats t2s_b8_stats t2s_b64_stats voc_stats; do
  f=$(find $OUT/$d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/${d}_kernel_stats.csv && python tools/stats_summary.py $f > $OUT/${d}_summary.txt 2>&1
done
f=$(find $OUT/pmc_sq -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python tools/sq_summary.py $f $OUT/sq_counters.json > $OUT/sq_counters.txt 2>&1
ff=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); fw=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python tools/pmc_summary.py $ff $fw $OUT/pmc_summary.json > $OUT/pmc_summary.txt 2>&1
f=$(find $OUT/stats -name "*kernel_trace.csv" | head -1); [ -n "$f" ] && python tools/gemm_by_grid.py $f > $OUT/gemm_by_grid.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -delete; find $OUT -name "*agent_info.csv" -delete
for f in $OUT/*.log; do tail -n 2 $f | cut -c1-200; done
cut -c1-600 $OUT/bench_n1.json
