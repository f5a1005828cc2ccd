#!/bin/bash
# MfmaUtil / VALUBusy of the p = 16384 factorisation's kernels (one 60 000-row step at d = 40, the
# configs[4] shape), one rocprofv3 --pmc pass with --kernel-trace only.  One gpurun call.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=$R/gpurun_out/r05_chol16k
mkdir -p $OUT
( cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --pmc MfmaUtil VALUBusy --output-format csv \
    -d $OUT/pmc -o p -- python3 $R/bench.py --rows 60000 --d 40 --p 16384 --kinds mat25,mat25pow,mat25ang --steps 1 --warmup 0 \
    --no-cpu-baseline --no-alt-backend --no-config3 --no-configs --no-fit-parity --no-obfit-eval > $OUT/pmc.log 2>&1 ) || { tail -5 $OUT/pmc.log; exit 1; }
f=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
python3 - $f <<'PY' | tee $OUT/pmc_mfma_util.txt
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if "k_chol" in r["Kernel_Name"]:
        m = re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"])
        agg[m.group(0)][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("rocprofv3 --kernel-trace --pmc MfmaUtil VALUBusy -- python3 bench.py --rows 60000 --d 40 --p 16384 --kinds mat25,mat25pow,mat25ang --steps 1 --warmup 0 (...): per kernel, mean over its dispatches (and over the 10 % with the highest value: the large trailing passes)")
for k, v in agg.items():
    for name, vals in sorted(v.items()):
        vals.sort()
        top = vals[-max(1, len(vals) // 10):]
        print("%-24s %-10s dispatches %5d  mean %6.2f  top-10%% mean %6.2f  max %6.2f" % (k, name, len(vals), sum(vals) / len(vals), sum(top) / len(top), vals[-1]))
PY
rm -rf $OUT/pmc
