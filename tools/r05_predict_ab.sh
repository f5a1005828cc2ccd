#!/bin/bash
# The fused predictor with the settings given as arguments ("VAR=value" words, one bench run each;
# default: with and without the early request of the next tile's inputs).  One gpurun call.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_parity.py tests/test_gpu_star.py -x -q -m gpu -k "newton_fit_and_predict or predict or star_kernels_on_selected" > gpurun_out/r05_t_pred.log 2>&1; tail -3 gpurun_out/r05_t_pred.log
for e in "${@:-OBHIP_PREDICT_PFX=0 OBHIP_PREDICT_PFX=1}"; do
  for s in $e; do
  env $s python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-config3 --no-fit-parity --no-obfit-eval --no-alt-backend > gpurun_out/r05/pred_line.json 2>> gpurun_out/r05/pred.err
  python - "$s" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r05/pred_line.json"))
print(sys.argv[1], "predict", d["kernels_ms"]["predict"]["avg_ms"], "split", d["fit_predict_split"]["predict_ms"], d["fit_predict_split"].get("predict_with_var_ms"), "err", d["parity_check"]["predict_max_rel_err"], "ms_per_step", d["ms_per_step"])
PY
  done
done
