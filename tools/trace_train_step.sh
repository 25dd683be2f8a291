#!/bin/bash
# per-dispatch timeline of ONE training step (tools/bench_train.py) -> gpurun_out/train_timeline.txt
set -e
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tt
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tt -o tt -- python3 /root/repo/tools/bench_train.py --steps 3 --warmup 2 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, os
rows = list(csv.DictReader(open(glob.glob("/tmp/tt/**/*kernel_trace.csv", recursive=True)[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last step: from the last p2n_pack_w preceded by a non-p2n kernel... simply take the tail after the last "zero_grad"-ish gap: find last occurrence of train_layer_kernel<0
idx = [i for i, r in enumerate(rows) if "p2n_train_layer_kernel<0" in r["Kernel_Name"]]
start = idx[-1] - 2
t0 = int(rows[start]["Start_Timestamp"])
out = open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "train_timeline.txt"), "w")
prev_end = t0
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    out.write("%9.1f %8.1f gap %6.1f  %s  grid %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r["Kernel_Name"][:80], r["Grid_Size_X"] + "x" + r["Grid_Size_Y"]))
    prev_end = e
out.close()
PY
