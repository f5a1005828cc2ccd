#!/bin/bash
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "term_per_lane_variants or newton_fit_and_predict or predict" > gpurun_out/r05_t7.log 2>&1; tail -3 gpurun_out/r05_t7.log
for e in "OBHIP_HM3=0" "OBHIP_HM3=1"; do
  env $e python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-configs --no-config3 --no-fit-parity --no-obfit-eval --no-alt-backend > gpurun_out/r05/pred_line.json 2>> gpurun_out/r05/pred.err
  python - "$e" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r05/pred_line.json"))
print(sys.argv[1], "predict", d["kernels_ms"]["predict"]["avg_ms"], "split", d["fit_predict_split"]["predict_ms"], d["fit_predict_split"].get("predict_with_var_ms"), "err", d["parity_check"]["predict_max_rel_err"], "ms_per_step", d["ms_per_step"])
PY
done
