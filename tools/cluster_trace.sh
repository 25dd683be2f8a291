#!/bin/bash
# Per-launch durations of the clustering kernels inside one program (run through gpurun):
#   tools/cluster_trace.sh <python script and arguments>
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_cl
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_cl -o cl -- python3 "$@" > /tmp/cl.log 2>&1 || { tail -5 /tmp/cl.log; exit 1; }
f=$(find /tmp/prof_cl -name "*kernel_trace.csv" | sed -n 1p)
python3 - $f <<"P"
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"]
    if "cl_" not in n and "rocprim" not in n:
        continue
    n = n.replace("void ", "")
    key = (n[:44], r.get("Grid_Size_X"), r.get("Grid_Size_Y"))
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    v = sorted(v)
    print(f"{k[0]:46s} grid {k[1]:>8s} x {k[2]:>2s}  calls {len(v):3d}  median {v[len(v)//2]:10.1f} us")
P
