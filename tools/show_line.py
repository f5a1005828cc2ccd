"""Prints the figures of a bench.py line that one looks at first (aid for gpurun output)."""
import json
import sys
# (a launcher's own chatter may share the file -- gloo prints its connections to stdout: the bench
# line is the one that starts with a brace)
d = json.loads([ln for ln in open(sys.argv[1]) if ln.startswith("{")][-1])
print("value %.4g %s  ms_per_step %.2f  n_gpus %d" % (d["value"], d["unit"], d["ms_per_step"], d["n_gpus"]))
r = d.get("roofline") or {}
print("roofline: %s frac %.4f achieved %.2f %s traffic %s mfma_util %s l2 %s" % (
    r.get("kernel"), r.get("frac", 0), r.get("achieved", 0), r.get("unit"), r.get("traffic"), r.get("mfma_util"),
    r.get("l2_hit_rate")))
print("kernels_ms:", {k: round(v["ms_per_step"], 3) for k, v in d["kernels_ms"].items()})
if d.get("parity_check"):
    pc = dict(d["parity_check"])
    fv = pc.pop("fit_vs_oracle", None)
    print("parity_check:", pc)
    if fv:
        print("fit_vs_oracle:", fv)
print("exchange:", d.get("exchange"))
if d.get("alt_backend"):
    print("alt_backend:", d["alt_backend"])
for c in d.get("configs") or []:
    print("config: %s\n   ms_per_step %.3f value %.4g gram %s chol %s resid %.2g" % (
        c["workload"], c["ms_per_step"], c["value"], c.get("gram"), c.get("cholesky_ms"), c["newton_residual_rel"]))
    print("   kernels_ms:", c["kernels_ms"])
if d.get("cpu_baseline"):
    print("cpu_baseline: %.4g %s on %s cores" % (d["cpu_baseline"]["value"], d["cpu_baseline"]["unit"], d["cpu_baseline"]["cores"]))
