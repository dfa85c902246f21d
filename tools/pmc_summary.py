#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel."""
import collections, csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: dict({c: sum(x) / len(x) for c, x in v.items()}, dispatches=len(next(iter(v.values())))) for k, v in agg.items() if "dz" in k}
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
for k, v in out.items():
    print(k, {c: round(x) for c, x in v.items()})
