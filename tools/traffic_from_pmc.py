#!/usr/bin/env python3
"""HBM traffic of the dominant kernel from the rocprofv3 --pmc passes (tools/collect_profiles.sh).

usage: traffic_from_pmc.py <tag>_pmc_summary.json <generations in the profiled run> <out.json>

FETCH_SIZE and WRITE_SIZE are in KiB per dispatch.  Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM
section): on gfx950 FETCH_SIZE counts 128-byte read requests as 64 bytes for wide coalesced reads (16 B per lane --
the Z-row gathers here), so it is doubled; WRITE_SIZE is taken as is (the guide calls it uncalibrated).
"""
import json, sys
summ = json.load(open(sys.argv[1])); gens = int(sys.argv[2])
cand = {k: v for k, v in summ.items() if "FETCH_SIZE" in v and ("k_generations" in k or "k_propose" in k)}
name = max(cand, key=lambda k: cand[k]["FETCH_SIZE"] * cand[k]["dispatches"])
v = cand[name]
tot_r = 2.0 * v["FETCH_SIZE"] * 1024 * v["dispatches"]; tot_w = v["WRITE_SIZE"] * 1024 * v["dispatches"]
out = {"kernel": name, "dispatches": v["dispatches"], "generations_in_run": gens,
       "fetch_size_kib_per_dispatch": v["FETCH_SIZE"], "write_size_kib_per_dispatch": v["WRITE_SIZE"],
       "read_bytes_per_generation": tot_r / gens, "write_bytes_per_generation": tot_w / gens,
       "bytes_per_generation": (tot_r + tot_w) / gens,
       "note": "FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md); separate --pmc passes for FETCH_SIZE and WRITE_SIZE"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(out)
