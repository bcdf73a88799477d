#!/bin/bash
# kernel-trace of the TV-L1 workload; time of the iteration kernels by pyramid level (development aid)
scripts/prof_any.sh tvl1b bench.py --workload tvl1 --steps 5 --warmup 1 --no-cpu-baseline "$@"
grep -o '"value": [0-9.]*, "unit": "pairs/s"' gpurun_out/prof_tvl1b.log
python3 - <<PY
import csv,collections,glob,statistics
f=glob.glob("gpurun_out/prof_tvl1b/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
agg=collections.defaultdict(list)
for r in rows:
    if "k_tvl1_" in r["Kernel_Name"]:
        g=(r["Kernel_Name"][33:60],int(r["Grid_Size_X"]),int(r["Grid_Size_Y"]))
        agg[g].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for g,ds in sorted(agg.items()):
    ds=sorted(ds)
    print(g,"calls",len(ds),"total ms",round(sum(ds)/1e3,2),"median",round(statistics.median(ds),1),"p10",round(ds[len(ds)//10],1),"max",round(ds[-1],1))
import collections as _c
other=_c.defaultdict(lambda:[0,0.0])
for r in rows:
    if "k_tvl1_" not in r["Kernel_Name"]:
        n=r["Kernel_Name"].split("(")[0][-40:]
        other[n][0]+=1; other[n][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3
for n,v in sorted(other.items(), key=lambda kv:-kv[1][1])[:8]:
    print("   other:", n, "calls", v[0], "total ms", round(v[1]/1e3,2))
print("VGPR/LDS/scratch", set((r["VGPR_Count"],r["LDS_Block_Size"],r["Scratch_Size"]) for r in rows if "k_tvl1_multi" in r["Kernel_Name"]))
PY
