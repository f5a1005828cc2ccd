"""Sums rocprofv3 --pmc counter_collection.csv per kernel (tuning aid)."""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else "gram"
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in rows:
    if re.search(pat, r["Kernel_Name"]):
        m = re.search(r"k_\w+(<[^>]*>)?", r["Kernel_Name"])   # "(anonymous namespace)" holds the first "("
        k = m.group(0) if m else r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in agg.items():
    print(k)
    for a, b in sorted(v.items()):
        print("   %-28s %.4g" % (a, b))
