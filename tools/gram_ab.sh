#!/bin/bash
# Gram kernel A/B aid: parity tests that touch the Gram, its time at the headline size, and
# the fabric-traffic / L2 counters (each PMC set in its own rocprofv3 run).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/gram_ab_${1:-x}
mkdir -p $OUT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_generic.py -x -q -m gpu -k "gram or newton or random_terms or chunked" > $OUT/tests.log 2>&1 || { tail -20 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
OBHIP_GRAM_DBG=1 timeout -k 10 300 python3 tools/gram_only.py 1000000 0 2>&1 | grep -v amdgpu | tee $OUT/time_dbg.log
timeout -k 10 300 python3 tools/gram_only.py 1000000 0 2>&1 | grep -v amdgpu | tee $OUT/time.log
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $set | tr ' ' '_')
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmc_$tag -o p -- \
    python3 $R/tools/gram_only.py 1000000 0 > $OUT/pmc_$tag.log 2>&1 || { tail -5 $OUT/pmc_$tag.log; exit 1; }
  f=$(find $OUT/pmc_$tag -name "*counter_collection.csv" | head -1)
  python3 $R/tools/pmc_summary.py $f k_atb_dma2 | tee -a $OUT/pmc.txt
  rm -rf $OUT/pmc_$tag
done
