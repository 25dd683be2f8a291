#!/bin/bash
# Per-kernel times of tools/knn_self_probe.py (rocprofv3 kernel trace), grouped by kernel and grid.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/knnprof
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/knnprof -o knn -- python3 /root/repo/tools/knn_self_probe.py || exit 1
python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/knnprof/**/*kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "knn" in r["Kernel_Name"]:
        d[(r["Kernel_Name"][:24], r["Grid_Size_X"], r["Grid_Size_Y"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in d.items():
    print(k, len(v), "avg %.1f us min %.1f" % (sum(v) / len(v), min(v)))
PY
