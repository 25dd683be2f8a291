#!/bin/bash
# Per-launch durations of every library kernel inside the scene bench at 166,499 points (run through gpurun).
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_sc
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_sc -o sc -- python3 /root/repo/tools/bench_inference.py --no-cpu > /tmp/sc.log 2>&1 || { tail -5 /tmp/sc.log; exit 1; }
f=$(find /tmp/prof_sc -name "*kernel_trace.csv" | sed -n 1p)
python3 - $f <<"P"
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    n = r["Kernel_Name"].replace("void ", "")
    if not n.startswith("p2n_"):
        continue
    key = (n[:52], r.get("Grid_Size_X"))
    agg.setdefault(key, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in agg.items():
    v = sorted(v)
    print(f"{k[0]:54s} grid {k[1]:>9s}  calls {len(v):3d}  median {v[len(v)//2]:9.1f} us")
P
