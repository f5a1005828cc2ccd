"""gpurun_out/r02/ (tools/r02_profile.sh) -> profiles/r02_*: the PMC passes of k_atb_dma2 as
two small JSON files bench.py reads for roofline.traffic / roofline.mfma_util, the per-kernel
rocprofv3 statistics of the bench command, and the bench lines themselves."""
import collections
import csv
import json
import os
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(R, "gpurun_out", "r02")
DST = os.path.join(R, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "a"


def counters(name):
    rows = list(csv.DictReader(open(os.path.join(SRC, "pmc_%s_%s.csv" % (tag, name)))))
    per = collections.defaultdict(list)
    for r in rows:
        if "k_atb_dma2" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in per.items()}


mf = counters("MfmaUtil_VALUBusy")
fe = counters("FETCH_SIZE")
wr = counters("WRITE_SIZE")
tc = counters("TCC_HIT_sum_TCC_MISS_sum")
cfg = {"d": 20, "rows": 1000000, "p": 4096, "knots": 40}
cmd = "rocprofv3 --kernel-trace --pmc %s --output-format csv -- python3 tools/gram_only.py 1000000 0 " \
      "(one pass per counter set, tools/r02_profile.sh)"
json.dump({"kernel": "k_atb_dma2", "config": cfg, "command": cmd % "MfmaUtil VALUBusy",
           "launches": mf["MfmaUtil"][1], "mfma_util": mf["MfmaUtil"][0] / 100.0,
           "valu_busy": mf["VALUBusy"][0] / 100.0,
           "note": "MfmaUtil / VALUBusy as rocprofv3 derives them (gfx94x formulas, "
                   "MI355X_MICROARCH.md 'rocprofv3 PMC slots'); per-dispatch average"},
          open(os.path.join(DST, "r02_pmc_gram_mfma.json"), "w"), indent=1)
fetch_raw_kb = fe["FETCH_SIZE"][0]
write_kb = wr["WRITE_SIZE"][0]
fetch_b = 2.0 * fetch_raw_kb * 1024      # gfx950: FETCH_SIZE counts 128-B requests at 64 B
write_b = write_kb * 1024
hit, miss = tc["TCC_HIT_sum"][0], tc["TCC_MISS_sum"][0]
json.dump({"kernel": "k_atb_dma2", "config": cfg, "command": cmd % "FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum",
           "launches": fe["FETCH_SIZE"][1], "FETCH_SIZE_KB_per_launch_raw": fetch_raw_kb,
           "WRITE_SIZE_KB_per_launch": write_kb, "TCC_HIT_per_launch": hit, "TCC_MISS_per_launch": miss,
           "l2_hit_rate": hit / (hit + miss),
           "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
           "traffic_bytes_per_launch": fetch_b + write_b,
           "algorithmic_bytes_per_launch": 8.0 * 1e6 * 4096 + write_b,
           "mfma_util": mf["MfmaUtil"][0] / 100.0,
           "note": "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane loads on "
                   "gfx950 (the panel copies are global_load_lds_dwordx4); WRITE_SIZE = the row-split "
                   "partial tiles.  Algorithmic bytes: the staged design matrix read once (8 n p) + "
                   "the partials."},
          open(os.path.join(DST, "r02_gram_traffic.json"), "w"), indent=1)
for name, dst in (("bench_line_%s.json" % tag, "r02_bench_line.json"),
                  ("bench_line_profiled_%s.json" % tag, "r02_bench_line_profiled.json")):
    if os.path.exists(os.path.join(SRC, name)):
        shutil.copy(os.path.join(SRC, name), os.path.join(DST, dst))
st = os.path.join(SRC, "stats_%s" % tag)
for f in os.listdir(st) if os.path.isdir(st) else []:
    if f.endswith("kernel_stats.csv"):
        shutil.copy(os.path.join(st, f), os.path.join(DST, "r02_kernel_stats_bench.csv"))
        rows = list(csv.DictReader(open(os.path.join(st, f))))
        with open(os.path.join(DST, "r02_kernel_stats_bench.txt"), "w") as o:
            o.write("rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 "
                    "--no-cpu-baseline --no-alt-backend --no-config3\n")
            for r in rows[:22]:
                nm = r["Name"].replace("(anonymous namespace)::", "").replace("obhip::", "")
                nm = nm[5:] if nm.startswith("void ") else nm
                nm = nm.split("(")[0]
                o.write("%-44s calls %5s  avg %10.3f ms  total %9.3f ms  %5.1f %%\n" % (
                    nm[:44], r["Calls"], float(r["AverageNs"]) / 1e6,
                    float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
print("mfma_util %.4f  traffic %.1f GB/launch (fetch %.1f, write %.1f)  L2 hit %.3f" % (
    mf["MfmaUtil"][0] / 100.0, (fetch_b + write_b) / 1e9, fetch_b / 1e9, write_b / 1e9, hit / (hit + miss)))
