import json,sys
d=json.load(open(sys.argv[1]))
for f in d["frames"]:
    print(f["frame"], f["ctu_kernel_ms"]); print(json.dumps(f.get("phase_share_of_total")))
    tot=0
    for k,v in f.get("primitives",{}).items():
        print("  %-28s %6.3f %8d calls %8.0f ticks/call" % (k, v["share_of_total"], v["calls"], v["ticks_per_call"])); tot+=v["share_of_total"]
    print("  sum", round(tot,3))
