#!/bin/bash
# Round profile collection on the GPU box (run through gpurun from the repo root):
#   bench JSON lines (default precision with the CPU baseline; opt-in f16), rocprofv3 kernel stats of the same command,
#   the two HBM PMC passes (FETCH_SIZE / WRITE_SIZE, separately, with --kernel-trace only) and one SQ pass
#   (cycles, MFMA-busy cycles -> effective clock and matrix-pipe utilisation per kernel: tools/sq_summary.py).
# Outputs land in gpurun_out/prof/; copy the summaries into profiles/ afterwards (tools/pmc_summary.py, stats_summary.py).
set -u
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
python bench.py > $OUT/bench_n1.json 2> $OUT/bench_n1.err
python bench.py --no-cpu-baseline --no-fp32-exact --precision f16 > $OUT/bench_n1_f16.json 2> $OUT/bench_n1_f16.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python $REPO/bench.py --no-cpu-baseline --no-fp32-exact --steps 2 --warmup 1 > $OUT/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $REPO/bench.py --no-cpu-baseline --no-fp32-exact --steps 1 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $REPO/bench.py --no-cpu-baseline --no-fp32-exact --steps 1 --warmup 1 > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/pmc_sq -- python $REPO/bench.py --no-cpu-baseline --no-fp32-exact --steps 1 --warmup 1 > $OUT/pmc_sq.log 2>&1
cd $REPO
find $OUT -name "*.csv" | head -20
cat $OUT/bench_n1.json | cut -c1-400
