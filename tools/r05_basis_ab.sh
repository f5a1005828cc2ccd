#!/bin/bash
# basis build (k_build_basis) with 4 / 8 / 16 tiles per block (OBHIP_BB_WAVES) -- timings of the
# headline step's kernels, alternating on one box
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r05
LEAN="--no-cpu-baseline --no-alt-backend --no-config3 --no-configs --no-fit-parity --no-obfit-eval"
for w in 4 8 16 4 8 16; do
OBHIP_BB_WAVES=$w python bench.py --steps 5 --warmup 1 $LEAN > gpurun_out/r05/basis_line.json 2>> gpurun_out/r05/basis.err || exit 1
python - $w <<'PY'
import json, sys
d = json.load(open("gpurun_out/r05/basis_line.json"))
k = d["kernels_ms"]
print("waves %s  ms_per_step %.2f  build_basis %.3f  predict %.3f  gram %.2f  predict err %.3g  newton resid %.3g" % (sys.argv[1],
    d["ms_per_step"], k["build_basis"]["avg_ms"], k["predict"]["avg_ms"], k["gram"]["avg_ms"],
    d["parity_check"]["predict_max_rel_err"], d["parity_check"]["newton_residual_rel"]))
PY
done
