"""prints the table of a `bench.py --mode potentials` record: python scripts/show_potentials.py <record.json>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
for sysn, rec in d["systems"].items():
    print(sysn, rec["atoms"], "atoms; us per execution (host to host), frames", d["frames"], "x", d["param_sets"], "parameter sets")
    for lab, v in rec.items():
        if lab == "atoms":
            continue
        for prec, r in v.items():
            print("  ", lab[:44].ljust(44), prec, "  ".join("%s=%.1f(dev %.1f)" % (k, x["us_per_execution"], x.get("device_us_per_execution", float("nan"))) for k, x in r.items()))
print("config 2, one execute() per term:", json.dumps(d["config2_single_execute"]))
