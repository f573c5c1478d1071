import json,sys
d=json.load(open(sys.argv[1]))
rows=sorted(d.items(), key=lambda kv:-kv[1]["seconds"])
tot=sum(v["seconds"] for k,v in rows)
n=int(sys.argv[2]) if len(sys.argv)>2 else 16
print(" ".join("%s=%.0f"%(k,v["seconds"]*1e6/2) for k,v in rows[:n]), "TOTAL=%.0f"%(tot*1e6/2))
