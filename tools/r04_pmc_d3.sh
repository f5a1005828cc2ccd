#!/bin/bash
# LDS / VALU counters of the fused hyper-gradient pass k_tmm_d3 (tools/grad_eval_bench.py: obfit's
# second-stage shape, n = 1e6, p = 4096, d = 8 mat25pow), one rocprofv3 pass per counter set.
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04
mkdir -p $OUT
: > $OUT/pmc_d3.txt
for set in "LdsUtil VALUBusy" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  name=$(echo $set | tr ' ' '_')
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv \
      -d $OUT/pmcd_$name -o p -- python3 $R/tools/grad_eval_bench.py > $OUT/pmcd_$name.log 2>&1 ) || { tail -5 $OUT/pmcd_$name.log; continue; }
  f=$(find $OUT/pmcd_$name -name "*counter_collection.csv" | head -1)
  echo "== rocprofv3 --kernel-trace --pmc $set -- python3 tools/grad_eval_bench.py" >> $OUT/pmc_d3.txt
  python3 $R/tools/pmc_summary.py $f "k_tmm_d3|k_tmm_ge0|k_tmm_tl" >> $OUT/pmc_d3.txt
  rm -rf $OUT/pmcd_$name
done
cat $OUT/pmc_d3.txt
