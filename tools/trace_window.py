import csv,sys
rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id","?"), r.get("Stream_Id","?")))
rows.sort()
rows=rows[len(rows)//2:]
idx=[i for i,r in enumerate(rows) if r[2]=="k_diag_col"][2]
t0=rows[idx-8][0]
for s,e,n,q,st in rows[idx-8:idx+30]:
    print("%9.1f %9.1f  %-18s q=%s st=%s"%((s-t0)/1e3,(e-t0)/1e3,n,q,st))
