import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
kt=d.get("kernel_times",{})
print("value %.1f M/s  ms/step %.4f | "%(d["value"]/1e6,d["ms_per_step"]) + " ".join("%s %.1fus(x%d)"%(k,v["avg_us"],v["launches"]) for k,v in kt.items() if isinstance(v,dict) and v["launches"]), "| acc %.3f rhat %.2f"%(d["acceptance_rate"],d["rhat_max"]))
