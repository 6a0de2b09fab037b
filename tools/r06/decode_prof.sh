#!/bin/bash
# Round 6: per-kernel durations AND inter-kernel gaps of the cold decode layer (the last replays of tools/cold_bench.py layer B are the
# cold graph: 8 distinct layers per replay), from rocprofv3's kernel trace.   usage: tools/r06/decode_prof.sh TAG BATCH [ENV=VAL ...]
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; B=$2; shift 2
for kv in "$@"; do export "$kv"; done
O=$R/gpurun_out/r06; mkdir -p $O
export ATOM_LIB=$R/build/tools/libatom_hip.so      # the tools build: reads the ATOM_* tuning variables
cd /tmp
rm -rf /tmp/dl_$TAG
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/dl_$TAG -o dl -- python $R/tools/cold_bench.py layer $B > /tmp/dl_$TAG.log 2>&1
grep batch /tmp/dl_$TAG.log
python3 - <<PY > $O/decode_prof_$TAG.txt
import csv,glob,collections
f=glob.glob("/tmp/dl_$TAG/**/*kernel_trace.csv",recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the cold measurement = the last 5 x 4 replays x 8 layers; take the last 4 replays (32 layers)
names=[r["Kernel_Name"] for r in rows]
# launches per layer: count kernels between consecutive occurrences of the first kernel of a layer in the tail
tail=rows[-2000:]
first=tail[-1]["Kernel_Name"]
idx=[i for i,r in enumerate(tail) if r["Kernel_Name"]==first]
per=idx[-1]-idx[-2]
nl=32
seg=rows[-per*nl:]
dur=collections.defaultdict(list); gaps=[]
for i,r in enumerate(seg):
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    dur[r["Kernel_Name"][:100]].append((e-s)/1e3)
    if i: gaps.append((s-int(seg[i-1]["End_Timestamp"]))/1e3)
tot=(int(seg[-1]["End_Timestamp"])-int(seg[0]["Start_Timestamp"]))/1e3/nl
print("# $TAG batch $B  $*   (cold graph, last %d layers; %d launches per layer; %.1f us per layer by the trace's clock)"%(nl,per,tot))
order=[]
for r in seg[:per]:
    k=r["Kernel_Name"][:100]
    order.append(k)
seen=set()
for k in order:
    if k in seen: continue
    seen.add(k)
    v=dur[k]
    print("%7.2f us avg  %6.2f min  x%d per layer  %s"%(sum(v)/len(v),min(v),len(v)//nl,k))
print("sum of kernel time per layer %.1f us; gaps: avg %.2f us, per layer %.1f us"%(sum(sum(v) for v in dur.values())/nl,sum(gaps)/len(gaps),sum(gaps)/nl))
# the gap IN FRONT of each launch of a layer (position by position over the layers)
pos=collections.defaultdict(list)
for i,r in enumerate(seg):
    if i: pos[i%per].append((int(r["Start_Timestamp"])-int(seg[i-1]["End_Timestamp"]))/1e3)
print("gap in front of launch: "+"  ".join("%d: %.2f"%(k,sum(v)/len(v)) for k,v in sorted(pos.items())))
PY
cat $O/decode_prof_$TAG.txt
