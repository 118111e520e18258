#!/bin/bash
# Board power and clocks while bench.py runs (rocm-smi sampled every ~0.25 s): direct evidence for the power-bound claims of
# DESIGN.md 4.1 / 4.3.  Run through gpurun from the repo root; the log lands in gpurun_out/power_sample.log.
OUT=gpurun_out/power_sample.log
mkdir -p gpurun_out
: > $OUT
python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-fp32-exact > gpurun_out/power_bench.json 2> gpurun_out/power_bench.err &
BP=$!
i=0
while kill -0 $BP 2>/dev/null; do
    echo "--- sample $i $(date +%s.%N)" >> $OUT
    /opt/rocm/bin/rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|busy|use" >> $OUT
    i=$((i + 1))
    sleep 0.25
done
wait $BP
cut -c1-300 gpurun_out/power_bench.json
grep -c "sample" $OUT
