#!/bin/bash
# A/B of the fused Hessian-product kernels on the box at hand (PCG back end at the headline sizes):
# round-3 kernel (OBHIP_HM_V1=1) against k_hm2's block shapes.  Output: gpurun_out/r04/hm_ab.txt
mkdir -p gpurun_out/r04
out=gpurun_out/r04/hm_ab.txt
: > $out
run() {
  echo "== $1" >> $out
  env $1 python bench.py --backend cg --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-config3 \
      --no-fit-parity > gpurun_out/r04/hm_ab_line.json 2>> gpurun_out/r04/hm_ab.err || return 1
  python - >> $out <<'PY'
import json
d = json.load(open("gpurun_out/r04/hm_ab_line.json"))
k = d["kernels_ms"]
print("ms_per_step %.2f  cg_iters %s  hessmult avg %.4f ms x %d  tmm %s  mm %s  predict err %.3g" % (
    d["ms_per_step"], d["config"].get("backend"), k["hessmult"]["avg_ms"], k["hessmult"]["launches"],
    k.get("tmm", {}).get("avg_ms"), k.get("mm", {}).get("avg_ms"), d["parity_check"]["predict_max_rel_err"]))
PY
}
run "OBHIP_HM_V1=1" && run "OBHIP_HM2_VARIANT=0" && run "OBHIP_HM2_VARIANT=5" && run "OBHIP_HM2_VARIANT=6" && run "OBHIP_HM2_VARIANT=2" && run "OBHIP_HM2_VARIANT=3" && \
run "OBHIP_HM_V1=1" && run "OBHIP_HM2_VARIANT=0"
cat $out
