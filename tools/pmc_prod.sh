#!/bin/bash
# PMC passes for the product kernels (run on the GPU box through gpurun): LDS / VALU utilisation
# and LDS bank conflicts of k_mm_tl / k_tmm_tl / k_materialize_tl, one rocprofv3 pass per set.
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/pmc_prod
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "LdsUtil VALUBusy" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  tag=$(echo $set | tr ' ' '_')
  timeout -k 10 500 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/$tag -o p -- \
    python3 $R/bench.py --backend cg --steps 1 --warmup 0 --no-cpu-baseline --no-alt-backend --no-config3 > $OUT/$tag.log 2>&1
  f=$(find $OUT/$tag -name "*counter_collection.csv" | head -1)
  echo "== $set"
  python3 $R/tools/pmc_summary.py $f _tl | tee -a $OUT/summary.txt
done
