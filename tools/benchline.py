import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
kt=d.get("kernel_times",{})
cv=d.get("convergence",{})
print("value %.1f M/s  ms/step %.4f (%d blocks, best %.1f worst %.1f) | "%(d["value"]/1e6,d["ms_per_step"],d["timing"]["timed_blocks"],d["timing"]["value_best_block"]/1e6,d["timing"]["value_worst_block"]/1e6)
      + " ".join("%s %.1fus(x%d)"%(k,v["avg_us"],v["launches"]) for k,v in kt.items() if isinstance(v,dict) and v["launches"]),
      "| acc %.3f rhat(window) %s gens to rhat<1.2: %s | dense %s | roofline %s %.3f"%(d["acceptance_rate"],d.get("rhat_max"),cv.get("generations_to_rhat_below_1p2"),
      ("%.1f M/s"%(d["dense_value"]/1e6)) if "dense_value" in d else "-", d.get("roofline",{}).get("bound"), d.get("roofline",{}).get("frac",0)))
