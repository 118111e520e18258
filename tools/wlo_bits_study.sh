#!/bin/bash
# Round-4 study: truncate the weights' lo halves to b significand bits - error of the 2-NFE C3 evaluation
# against the oracle and ms per bench step.  Run through gpurun from the repo root; log: gpurun_out/wlo_bits.log
OUT=gpurun_out/wlo_bits.log; mkdir -p gpurun_out; : > $OUT
for b in 11 8 6 4 0; do
  echo "== CVX_WLO_BITS=$b" >> $OUT
  CVX_WLO_BITS=$b python -m pytest tests/test_parity_at_size_gpu.py -q -s -k "c3_vomix_b8_t1000_two_nfe" 2>&1 | grep -E "rel-L2|passed|failed" >> $OUT
  CVX_WLO_BITS=$b python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-fp32-exact --no-c2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('ms_per_step', d['ms_per_step'], 'gemm avg_launch_ms', r['avg_launch_ms'], 'power_w', r.get('power_w'), 'sclk_mhz', r.get('sclk_mhz'), 'attention ms', d['kernel_classes_ms_per_step']['attention']['ms_per_step'])" >> $OUT
done
cat $OUT
