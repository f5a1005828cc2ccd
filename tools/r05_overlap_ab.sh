#!/bin/bash
# A/B of the staging / Gram overlap at the headline (round-4 verdict, item 3): whole design matrix
# (default), two and four row chunks staged one after the other, and the same with chunk k + 1
# staged on a second stream while chunk k is multiplied (OBHIP_GRAM_OVERLAP=1).
# Output: gpurun_out/r05/overlap_ab.txt
mkdir -p gpurun_out/r05
out=gpurun_out/r05/overlap_ab.txt
: > $out
LEAN="--no-cpu-baseline --no-alt-backend --no-config3 --no-configs --no-fit-parity --no-obfit-eval"
run() {
  echo "== $1" >> $out
  env $1 python bench.py --steps 5 --warmup 1 $LEAN > gpurun_out/r05/overlap_line.json 2>> gpurun_out/r05/overlap.err || return 1
  python - >> $out <<'PY'
import json
d = json.load(open("gpurun_out/r05/overlap_line.json"))
k = d["kernels_ms"]
print("ms_per_step %.2f  median %.2f  gram %.2f ms x %d  materialize %.2f ms x %d  reduce %.2f  predict err %.3g  newton resid %.3g" % (
    d["ms_per_step"], d["median_step_ms"], k["gram"]["avg_ms"], k["gram"]["launches"], k["materialize_B"]["avg_ms"],
    k["materialize_B"]["launches"], k["gram_reduce"]["avg_ms"], d["parity_check"]["predict_max_rel_err"],
    d["parity_check"]["newton_residual_rel"]))
PY
}
run "OBHIP_X=0" && run "OBHIP_GRAM_CHUNK_ROWS=500032" && run "OBHIP_GRAM_CHUNK_ROWS=500032 OBHIP_GRAM_OVERLAP=1" && \
run "OBHIP_GRAM_CHUNK_ROWS=250048" && run "OBHIP_GRAM_CHUNK_ROWS=250048 OBHIP_GRAM_OVERLAP=1" && run "OBHIP_X=0"
cat $out
