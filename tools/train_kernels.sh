#!/bin/bash
# Kernel time table of the training step (rocprofv3 kernel trace of tools/bench_train.py), top N rows.
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tr
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_tr -o tr -- python3 /root/repo/tools/bench_train.py --steps 5 > /dev/null 2>&1 || exit 1
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/prof_tr/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step: %.3f" % (tot / 7 / 1e6))
for r in rows[:${1:-24}]:
    print("%-64s %5s %9.1f us %5.1f%%" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / tot * 100))
PY
