import json, sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print({k:v for k,v in d.items() if k not in ("roofline","secondary","config","cpu_baseline")})
print("roofline", {k:v for k,v in d["roofline"].items() if not isinstance(v,(list,dict))})
print("cpu", d.get("cpu_baseline"))
for k,v in d.get("secondary",{}).items():
    print(k, {a:b for a,b in v.items() if a in ("value","unit","ms_per_step","error","dtype")}, "roofline" in v and {a:b for a,b in v["roofline"].items() if a in ("frac","achieved","peak")})
