#!/bin/bash
# LDS / VALU counters of the fused Hessian-product kernels (PCG back end at the headline sizes),
# one rocprofv3 pass per counter set, round-3 kernel (OBHIP_HM_V1=1) and k_hm2.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04
mkdir -p $OUT
LEAN="--no-cpu-baseline --no-alt-backend --no-config3 --no-configs --no-fit-parity --no-obfit-eval"
: > $OUT/pmc_products.txt
for v1 in 1 0; do
  for set in "LdsUtil VALUBusy" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
    name=$(echo $set | tr ' ' '_')
    ( cd /tmp && export TMPDIR=/tmp && export OBHIP_HM_V1=$v1 && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv \
        -d $OUT/pmcp_$name -o p -- python3 $R/bench.py --backend cg --steps 1 --warmup 0 $LEAN \
        > $OUT/pmcp_$name.log 2>&1 ) || { tail -5 $OUT/pmcp_$name.log; continue; }
    f=$(find $OUT/pmcp_$name -name "*counter_collection.csv" | head -1)
    echo "== OBHIP_HM_V1=$v1: rocprofv3 --kernel-trace --pmc $set -- python3 bench.py --backend cg --steps 1 --warmup 0 $LEAN" >> $OUT/pmc_products.txt
    python3 $R/tools/pmc_summary.py $f k_hm >> $OUT/pmc_products.txt
    rm -rf $OUT/pmcp_$name
  done
done
cat $OUT/pmc_products.txt
