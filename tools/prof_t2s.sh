set -u
REPO=$(pwd); OUT=$REPO/gpurun_out/r6; mkdir -p $OUT
timeout 400 python -m pytest tests/test_t2s_gpu.py -x -q -m gpu 2>&1 | tail -5
for ppw in ${PPWS:-1 2}; do echo "== pairs per wave $ppw"; CVX_T2S_PPW=$ppw MANY=0 BATCHES=${BATCHES:-1,8,16,32,64} timeout 200 python tools/bench_t2s.py comix 2>&1 | grep batch; done
if [ "${MANY:-1}" = "1" ]; then echo "== many"; BATCHES=8 timeout 200 python tools/bench_t2s.py comix 2>&1 | grep utterances; fi
if [ "${PROF:-1}" = "1" ]; then
cd /tmp && export TMPDIR=/tmp
for nb in ${PROFB:-8 64}; do
  CVX_T2S_PPW=${PROFPPW:-0} MANY=0 BATCHES=$nb TOKENS=256 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t2s_b$nb -- python $REPO/tools/bench_t2s.py comix > $OUT/t2s_b$nb.log 2>&1
  f=$(find $OUT/t2s_b$nb -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/t2s_b${nb}_kernel_stats.csv
  rm -rf $OUT/t2s_b$nb
done
cd $REPO
for nb in ${PROFB:-8 64}; do echo "== batch $nb"; python tools/stats_summary.py $OUT/t2s_b${nb}_kernel_stats.csv 2>&1 | head -12; done
fi
