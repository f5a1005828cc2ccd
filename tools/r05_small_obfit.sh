#!/bin/bash
mkdir -p gpurun_out/r05
out=gpurun_out/r05/small_obfit.txt
: > $out
OBHIP_CG_BATCH=1 python tools/r05_small_obfit.py >> $out 2>> gpurun_out/r05/small_obfit.err && \
python tools/r05_small_obfit.py >> $out 2>> gpurun_out/r05/small_obfit.err && \
OBHIP_CG_BATCH=16 python tools/r05_small_obfit.py >> $out 2>> gpurun_out/r05/small_obfit.err
cat $out
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cg_fit or obfit or term_per_lane_variants or random_lpdf" > gpurun_out/r05_t8.log 2>&1; tail -3 gpurun_out/r05_t8.log
