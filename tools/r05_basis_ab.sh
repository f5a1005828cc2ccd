#!/bin/bash
# basis build / fused predictor with the one-round interval search (equidistant knots: the guess is
# checked on the host, ModelDev::build) -- timings of the headline step's kernels
mkdir -p gpurun_out/r05
LEAN="--no-cpu-baseline --no-alt-backend --no-config3 --no-configs --no-fit-parity --no-obfit-eval"
python bench.py --steps 5 --warmup 1 $LEAN > gpurun_out/r05/basis_line.json 2>> gpurun_out/r05/basis.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05/basis_line.json"))
k = d["kernels_ms"]
print("ms_per_step %.2f  build_basis %.3f  predict %.3f  gram %.2f  predict err %.3g  newton resid %.3g" % (
    d["ms_per_step"], k["build_basis"]["avg_ms"], k["predict"]["avg_ms"], k["gram"]["avg_ms"],
    d["parity_check"]["predict_max_rel_err"], d["parity_check"]["newton_residual_rel"]))
PY
