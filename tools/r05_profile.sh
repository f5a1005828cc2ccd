#!/bin/bash
# One gpurun call: the full bench line, rocprofv3 kernel statistics of the same command, the
# PMC passes of the Gram kernel (matrix-pipe utilisation; fabric read / write bytes; L2 hits),
# each in its own rocprofv3 run with --kernel-trace only, as the pool requires, and the bench
# lines of the per-rank shard sizes.  Outputs under gpurun_out/r05/; tools/r05_summarise.py turns
# them into profiles/r05_*.
#
#   tools/r05_profile.sh [TAG]          collect everything
#   tools/r05_profile.sh --check        regression guard of the Gram kernel's L2 behaviour: one
#       TCC_HIT / TCC_MISS pass and one timed, instrumented run on this box against the committed
#       profiles/r05_gram_traffic.json; fails (exit 1) when the L2 hit rate is below 0.78, when a
#       block needs more than 1.02 x the committed shader-clock ticks per 16-row chunk (the
#       clock-independent form of "the Gram got slower": the GPUs of the pool differ by +-3 % in
#       the clock they hold, which a bound on wall time alone would report as a regression), or
#       when the launch takes more than 1.06 x the committed wall time
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
pmc() {  # $1 = tag, $2... = counters
  local tag=$1; shift
  local name=$(echo "$*" | tr ' ' '_')
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc $* --output-format csv \
      -d $OUT/pmc_${tag}_$name -o p -- python3 $R/tools/gram_only.py 1000000 0 > $OUT/pmc_${tag}_$name.log 2>&1 ) \
    || { tail -5 $OUT/pmc_${tag}_$name.log; return 1; }
  local f=$(find $OUT/pmc_${tag}_$name -name "*counter_collection.csv" | head -1)
  grep -E "k_atb_dma2|Counter_Name" $f > $OUT/pmc_${tag}_$name.csv
  rm -rf $OUT/pmc_${tag}_$name
  echo "pmc $* ok"
}
if [ "$1" = "--check" ]; then
  pmc check TCC_HIT_sum TCC_MISS_sum || exit 1
  OBHIP_GRAM_DBG=1 timeout -k 10 300 python3 tools/gram_only.py 1000000 0 > $OUT/check_gram_only.log 2>&1 || { tail -5 $OUT/check_gram_only.log; exit 1; }
  python3 tools/r05_summarise.py --check
  exit $?
fi
TAG=${1:-a}
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$TAG -o p -- \
  python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt-backend --no-config3 --no-configs --no-fit-parity --no-obfit-eval \
  > $OUT/bench_line_profiled_$TAG.json 2> $OUT/stats_$TAG.err ) || { tail -5 $OUT/stats_$TAG.err; exit 1; }
rm -f $OUT/stats_$TAG/*kernel_trace.csv
echo "stats ok"
OBHIP_GRAM_DBG=1 timeout -k 10 300 python3 tools/gram_only.py 1000000 0 > $OUT/gram_only_$TAG.log 2>&1 || { tail -5 $OUT/gram_only_$TAG.log; exit 1; }
for set in "MfmaUtil VALUBusy" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  pmc $TAG $set || exit 1
done
# the full bench line AFTER the PMC passes, with profiles/r05_gram_traffic.json of this very
# library in place on the box, so that its roofline.traffic / mfma_util / l2_hit_rate are filled
python3 tools/r05_summarise.py $TAG > $OUT/summarise_on_box_$TAG.log 2>&1 || { tail -5 $OUT/summarise_on_box_$TAG.log; exit 1; }
timeout -k 10 900 python3 bench.py > $OUT/bench_line_$TAG.json 2> $OUT/bench_$TAG.err || { tail -5 $OUT/bench_$TAG.err; exit 1; }
echo "bench ok"
# the per-rank step of an N-GPU job on this one GPU: N virtual ranks (obhip_comm_init_sim: the real
# exchange-buffer layout, pack, unpack, replicated solve; the sum itself is one device pass)
LEAN="--no-cpu-baseline --no-alt-backend --no-config3 --no-configs --no-fit-parity --no-obfit-eval"
rm -f $OUT/bench_lines_shard_sizes_$TAG.jsonl
for spec in "125000 8" "250000 4" "500000 2"; do
  set -- $spec
  timeout -k 10 300 python3 bench.py --rows $1 --sim-ranks $2 --steps 10 --warmup 2 $LEAN \
    >> $OUT/bench_lines_shard_sizes_$TAG.jsonl 2>> $OUT/bench_$TAG.err || exit 1
done
timeout -k 10 300 python3 bench.py --rows 125000 --steps 10 --warmup 2 $LEAN \
  >> $OUT/bench_lines_shard_sizes_$TAG.jsonl 2>> $OUT/bench_$TAG.err || exit 1
echo "shard sizes ok"
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats125_$TAG -o p -- \
  python3 $R/bench.py --rows 125000 --sim-ranks 8 --steps 5 --warmup 1 $LEAN \
  > /dev/null 2> $OUT/stats125_$TAG.err ) || { tail -5 $OUT/stats125_$TAG.err; exit 1; }
rm -f $OUT/stats125_$TAG/*kernel_trace.csv
# BASELINE.json configs[1] (d = 10, n = 1e5, p = 1024): kernel statistics of its own
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/statsc1_$TAG -o p -- \
  python3 $R/bench.py --dims 10 --p 1024 --rows 100000 --steps 20 --warmup 2 $LEAN \
  > $OUT/bench_line_configs1_$TAG.json 2> $OUT/statsc1_$TAG.err ) || { tail -5 $OUT/statsc1_$TAG.err; exit 1; }
rm -f $OUT/statsc1_$TAG/*kernel_trace.csv
# ... and the matrix-pipe / L2 counters of its Gram launch (what bounds 0.74 of the peak there)
: > $OUT/pmc_configs1_$TAG.txt
for set in "MfmaUtil VALUBusy" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA"; do
  name=$(echo $set | tr ' ' '_')
  ( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv \
      -d $OUT/pmcc1_${TAG}_$name -o p -- python3 $R/bench.py --dims 10 --p 1024 --rows 100000 --steps 3 --warmup 1 $LEAN \
      > $OUT/pmcc1_${TAG}_$name.log 2>&1 ) || { tail -5 $OUT/pmcc1_${TAG}_$name.log; continue; }
  f=$(find $OUT/pmcc1_${TAG}_$name -name "*counter_collection.csv" | head -1)
  echo "== rocprofv3 --kernel-trace --pmc $set -- python3 bench.py --dims 10 --p 1024 --rows 100000 --steps 3 --warmup 1 $LEAN" >> $OUT/pmc_configs1_$TAG.txt
  python3 $R/tools/pmc_summary.py $f "k_atb_dma2|k_gram_reduce|k_chol" >> $OUT/pmc_configs1_$TAG.txt
  rm -rf $OUT/pmcc1_${TAG}_$name
done
# the PCG back end: kernel statistics and the LDS / VALU counters of the fused Hessian product,
# k_hm2 (OBHIP_SHARE=0: every term multiplied out on its own, rounds 1-4) and k_star (shared
# sub-products, round 5) on the same box
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/statscg_$TAG -o p -- \
  python3 $R/bench.py --backend cg --steps 3 --warmup 1 $LEAN \
  > $OUT/bench_line_cg_profiled_$TAG.json 2> $OUT/statscg_$TAG.err ) || { tail -5 $OUT/statscg_$TAG.err; exit 1; }
rm -f $OUT/statscg_$TAG/*kernel_trace.csv
: > $OUT/pmc_products_$TAG.txt
for sh in 0 1; do
  for set in "LdsUtil VALUBusy" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
    name=$(echo $set | tr ' ' '_')
    ( cd /tmp && export TMPDIR=/tmp && export OBHIP_SHARE=$sh && timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv \
        -d $OUT/pmcp_${TAG}_$name -o p -- python3 $R/bench.py --backend cg --steps 1 --warmup 0 $LEAN \
        > $OUT/pmcp_${TAG}_$name.log 2>&1 ) || { tail -5 $OUT/pmcp_${TAG}_$name.log; exit 1; }
    f=$(find $OUT/pmcp_${TAG}_$name -name "*counter_collection.csv" | head -1)
    echo "== OBHIP_SHARE=$sh: rocprofv3 --kernel-trace --pmc $set -- python3 bench.py --backend cg --steps 1 --warmup 0 $LEAN" >> $OUT/pmc_products_$TAG.txt
    python3 $R/tools/pmc_summary.py $f "k_hm|k_star|k_tmm_tl|k_predict" >> $OUT/pmc_products_$TAG.txt
    rm -rf $OUT/pmcp_${TAG}_$name
  done
done
echo "products ok"
echo "all ok"
