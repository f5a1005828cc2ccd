#!/bin/bash
# A/B of the shared sub-products (stars, csrc/share.cpp) on the box at hand: PCG back end at the
# headline sizes and the d = 8 mat25pow Hessian product, OBHIP_SHARE=0 against the default.
# Output: gpurun_out/r05/share_ab.txt
mkdir -p gpurun_out/r05
out=gpurun_out/r05/share_ab.txt
: > $out
run() {
  echo "== $1" >> $out
  env $1 python bench.py --backend cg --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-config3 \
      --no-fit-parity > gpurun_out/r05/share_ab_line.json 2>> gpurun_out/r05/share_ab.err || return 1
  python - >> $out <<'PY'
import json
d = json.load(open("gpurun_out/r05/share_ab_line.json"))
k = d["kernels_ms"]
print("ms_per_step %.2f  hessmult avg %.4f ms x %d  tmm_dual %s  predict %s  predict err %.3g  newton resid %s" % (
    d["ms_per_step"], k["hessmult"]["avg_ms"], k["hessmult"]["launches"],
    k.get("tmm_dual", {}).get("avg_ms"), k.get("predict", {}).get("avg_ms"), d["parity_check"]["predict_max_rel_err"],
    d["parity_check"].get("newton_residual_rel")))
PY
}
hm() {
  echo "== hm_bench $1 ($2)" >> $out
  env $1 python tools/hm_bench.py $2 >> $out 2>> gpurun_out/r05/share_ab.err || return 1
}
run "OBHIP_SHARE=0" && run "OBHIP_HM3=1" && run "OBHIP_HM3=1" && \
hm "OBHIP_SHARE=0" "1000000 4096 8 mat25pow 12" && hm "OBHIP_HM3=1" "1000000 4096 8 mat25pow 12"
cat $out
