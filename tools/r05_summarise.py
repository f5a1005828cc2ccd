"""gpurun_out/r05/ (tools/r05_profile.sh) -> profiles/r05_*: the PMC passes of k_atb_dma2 as a
small JSON file bench.py reads for roofline.traffic / mfma_util / l2_hit_rate -- stamped with the
content hash of the Gram kernel's sources as compiled into the library that was profiled, so
that bench.py can tell a stale profile --, the per-kernel rocprofv3 statistics of the bench
command, and the bench lines themselves.

  python3 tools/r05_summarise.py [TAG]     after tools/r05_profile.sh [TAG]
  python3 tools/r05_summarise.py --check   on the GPU box, by tools/r05_profile.sh --check
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(R, "gpurun_out", "r05")
DST = os.path.join(R, "profiles")
CFG = {"d": 20, "rows": 1000000, "p": 4096, "knots": 40}


def counters(tag, name):
    rows = list(csv.DictReader(open(os.path.join(SRC, "pmc_%s_%s.csv" % (tag, name)))))
    per = collections.defaultdict(list)
    for r in rows:
        if "k_atb_dma2" in r["Kernel_Name"]:
            per[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in per.items()}


def gram_only_log(path):
    """-> (gram kernel ms, source hash of the Gram sources in the profiled library, shader-clock
    ticks per 16-row chunk and MHz of the instrumented launch -- None without OBHIP_GRAM_DBG)"""
    txt = open(path).read()
    ms = float(re.search(r"^\s*gram\s+([0-9.]+) ms", txt, re.M).group(1))
    sha = re.search(r"source_hash_gram=(\w+)", txt).group(1)
    m = re.findall(r"-> (\d+) MHz, (\d+) ticks per 16-row chunk", txt)
    mhz, ticks = (float(m[-1][0]), float(m[-1][1])) if m else (None, None)
    return ms, sha, ticks, mhz


def check():
    ref = json.load(open(os.path.join(DST, "r05_gram_traffic.json")))
    tc = counters("check", "TCC_HIT_sum_TCC_MISS_sum")
    hit, miss = tc["TCC_HIT_sum"][0], tc["TCC_MISS_sum"][0]
    rate = hit / (hit + miss)
    ms, sha, ticks, mhz = gram_only_log(os.path.join(SRC, "check_gram_only.log"))
    ok = True
    print("L2 hit rate %.3f (committed %.3f, floor 0.78)" % (rate, ref["l2_hit_rate"]))
    if rate < 0.78:
        print("FAIL: the Gram kernel's L2 hit rate fell below 0.78")
        ok = False
    rt = ref.get("gram_ticks_per_chunk")
    if ticks is not None and rt:
        print("shader-clock ticks per 16-row chunk %.0f at %.0f MHz (committed %.0f, ceiling x 1.02 = %.0f; "
              "8192 = matrix pipe saturated)" % (ticks, mhz, rt, 1.02 * rt))
        if ticks > 1.02 * rt:
            print("FAIL: a block of the Gram kernel needs more than 2 % more cycles per chunk than committed")
            ok = False
    print("Gram launch %.2f ms (committed %.2f ms on another GPU of the pool; ceiling x 1.06 = %.2f)" % (
        ms, ref["gram_avg_ms"], 1.06 * ref["gram_avg_ms"]))
    if ms > 1.06 * ref["gram_avg_ms"]:
        print("FAIL: the Gram launch is more than 6 % slower than the committed profile")
        ok = False
    if sha != ref["source_hash_gram"]:
        print("note: Gram sources changed since the committed profile (%s -> %s): re-collect it "
              "with tools/r05_profile.sh once the numbers hold" % (ref["source_hash_gram"], sha))
    return 0 if ok else 1


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "a"
    mf = counters(tag, "MfmaUtil_VALUBusy")
    fe = counters(tag, "FETCH_SIZE")
    wr = counters(tag, "WRITE_SIZE")
    tc = counters(tag, "TCC_HIT_sum_TCC_MISS_sum")
    ms, sha, ticks, mhz = gram_only_log(os.path.join(SRC, "gram_only_%s.log" % tag))
    cmd = "rocprofv3 --kernel-trace --pmc %s --output-format csv -- python3 tools/gram_only.py 1000000 0 " \
          "(one pass per counter set, tools/r05_profile.sh)"
    fetch_raw_kb = fe["FETCH_SIZE"][0]
    write_kb = wr["WRITE_SIZE"][0]
    fetch_b = 2.0 * fetch_raw_kb * 1024      # gfx950: FETCH_SIZE counts 128-B requests at 64 B
    write_b = write_kb * 1024
    hit, miss = tc["TCC_HIT_sum"][0], tc["TCC_MISS_sum"][0]
    json.dump({"kernel": "k_atb_dma2", "config": CFG, "source_hash_gram": sha,
               "command": cmd % "MfmaUtil VALUBusy | FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum",
               "launches": fe["FETCH_SIZE"][1], "gram_avg_ms": ms,
               "gram_ticks_per_chunk": ticks, "gram_clock_mhz": mhz,
               "FETCH_SIZE_KB_per_launch_raw": fetch_raw_kb,
               "WRITE_SIZE_KB_per_launch": write_kb, "TCC_HIT_per_launch": hit, "TCC_MISS_per_launch": miss,
               "l2_hit_rate": hit / (hit + miss),
               "fetch_bytes_per_launch": fetch_b, "write_bytes_per_launch": write_b,
               "traffic_bytes_per_launch": fetch_b + write_b,
               "algorithmic_bytes_per_launch": 8.0 * 1e6 * 4096 + write_b,
               "mfma_util": mf["MfmaUtil"][0] / 100.0, "valu_busy": mf["VALUBusy"][0] / 100.0,
               "note": "FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for 16-byte-per-lane loads on "
                       "gfx950 (the panel copies are global_load_lds_dwordx4); WRITE_SIZE = the row-split "
                       "partial tiles.  Algorithmic bytes: the staged design matrix read once (8 n p) + "
                       "the partials.  MfmaUtil / VALUBusy as rocprofv3 derives them (gfx94x formulas), "
                       "per-dispatch average.  source_hash_gram: obhip_source_hash(1) of the profiled "
                       "library; bench.py prints these counters only for a library with the same hash."},
              open(os.path.join(DST, "r05_gram_traffic.json"), "w"), indent=1)
    for name, dst in (("bench_line_%s.json" % tag, "r05_bench_line.json"),
                      ("bench_line_profiled_%s.json" % tag, "r05_bench_line_profiled.json"),
                      ("bench_lines_shard_sizes_%s.jsonl" % tag, "r05_bench_lines_shard_sizes_1gpu.jsonl"),
                      ("bench_line_configs1_%s.json" % tag, "r05_bench_line_configs1_profiled.json"),
                      ("bench_line_cg_profiled_%s.json" % tag, "r05_bench_line_cg_profiled.json"),
                      ("pmc_products_%s.txt" % tag, "r05_pmc_products.txt"),
                      ("pmc_configs1_%s.txt" % tag, "r05_pmc_configs1.txt")):
        if os.path.exists(os.path.join(SRC, name)):
            shutil.copy(os.path.join(SRC, name), os.path.join(DST, dst))
    for sub, out, cmdline in (
            ("stats_%s" % tag, "r05_kernel_stats_bench",
             "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 3 --warmup 1 "
             "--no-cpu-baseline --no-alt-backend --no-config3 --no-configs --no-fit-parity --no-obfit-eval"),
            ("stats125_%s" % tag, "r05_kernel_stats_shard_125000_rows_8_virtual_ranks",
             "rocprofv3 --kernel-trace --stats -- python3 bench.py --rows 125000 --sim-ranks 8 --steps 5 "
             "--warmup 1 --no-cpu-baseline --no-alt-backend --no-config3 --no-configs --no-fit-parity --no-obfit-eval"),
            ("statsc1_%s" % tag, "r05_kernel_stats_configs1",
             "rocprofv3 --kernel-trace --stats -- python3 bench.py --dims 10 --p 1024 --rows 100000 --steps 20 "
             "--warmup 2 --no-cpu-baseline --no-alt-backend --no-config3 --no-configs --no-fit-parity --no-obfit-eval"),
            ("statscg_%s" % tag, "r05_kernel_stats_bench_cg",
             "rocprofv3 --kernel-trace --stats -- python3 bench.py --backend cg --steps 3 --warmup 1 "
             "--no-cpu-baseline --no-alt-backend --no-config3 --no-configs --no-fit-parity --no-obfit-eval")):
        st = os.path.join(SRC, sub)
        for f in os.listdir(st) if os.path.isdir(st) else []:
            if f.endswith("kernel_stats.csv"):
                shutil.copy(os.path.join(st, f), os.path.join(DST, out + ".csv"))
                rows = list(csv.DictReader(open(os.path.join(st, f))))
                with open(os.path.join(DST, out + ".txt"), "w") as o:
                    o.write(cmdline + "\n")
                    for r in rows[:24]:
                        nm = r["Name"].replace("(anonymous namespace)::", "").replace("obhip::", "")
                        nm = nm[5:] if nm.startswith("void ") else nm
                        nm = nm.split("(")[0]
                        o.write("%-44s calls %5s  avg %10.3f ms  total %9.3f ms  %5.1f %%\n" % (
                            nm[:44], r["Calls"], float(r["AverageNs"]) / 1e6,
                            float(r["TotalDurationNs"]) / 1e6, float(r["Percentage"])))
    print("mfma_util %.4f  traffic %.1f GB/launch (fetch %.1f, write %.1f)  L2 hit %.3f  gram %.2f ms  sha %s" % (
        mf["MfmaUtil"][0] / 100.0, (fetch_b + write_b) / 1e9, fetch_b / 1e9, write_b / 1e9,
        hit / (hit + miss), ms, sha))


if __name__ == "__main__":
    sys.exit(check() if "--check" in sys.argv else main())
