import json,sys
d=json.load(open(sys.argv[1]))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["step_frac_of_hbm_peak"])
for k,v in d["variants"].items(): print(k, round(v["frames_per_s"]), round(v["ms_per_step"],4), v.get("step_frac_of_hbm_peak"), v["roofline"]["bound"], round(v["roofline"]["frac"],3), v.get("mfma_roofline",{}).get("frac"))
print(d["cpu_baseline"]["value"], d["cpu_baseline"]["kind"])
