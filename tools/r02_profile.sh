#!/bin/bash
# One gpurun call: the full bench line, rocprofv3 kernel statistics of the same command, and
# the PMC passes of the Gram kernel (matrix-pipe utilisation; fabric read / write bytes; L2
# hits), each in its own rocprofv3 run with --kernel-trace only, as the pool requires.
# Outputs under gpurun_out/r02/; tools/r02_summarise.py turns them into profiles/r02_*.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r02
TAG=${1:-a}
mkdir -p $OUT
cd $R
timeout -k 10 600 python3 bench.py > $OUT/bench_line_$TAG.json 2> $OUT/bench_$TAG.err || { tail -5 $OUT/bench_$TAG.err; exit 1; }
echo "bench ok"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$TAG -o p -- \
  python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-backend --no-config3 \
  > $OUT/bench_line_profiled_$TAG.json 2> $OUT/stats_$TAG.err || { tail -5 $OUT/stats_$TAG.err; exit 1; }
rm -f $OUT/stats_$TAG/*kernel_trace.csv
echo "stats ok"
for set in "MfmaUtil VALUBusy" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | tr ' ' '_')
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_${TAG}_$tag -o p -- \
    python3 $R/tools/gram_only.py 1000000 0 > $OUT/pmc_${TAG}_$tag.log 2>&1 || { tail -5 $OUT/pmc_${TAG}_$tag.log; exit 1; }
  f=$(find $OUT/pmc_${TAG}_$tag -name "*counter_collection.csv" | head -1)
  grep -E "k_atb_dma2|Counter_Name" $f > $OUT/pmc_${TAG}_$tag.csv
  rm -rf $OUT/pmc_${TAG}_$tag
  echo "pmc $set ok"
done
