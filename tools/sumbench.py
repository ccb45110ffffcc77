import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d.get("value"))
for k,v in d.get("configs",{}).items():
    print(k, {x:(round(v[x],2) if isinstance(v[x],float) else v[x]) for x in ("images_per_s","images_per_s_incl_host_io","us_per_iteration","us_per_lockstep_iteration") if x in v}, round((v.get("roofline") or {}).get("avg_launch_us",0),1))
