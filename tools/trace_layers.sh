cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_l
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_l -o l -- python3 /root/repo/tools/exp_layer.py > /dev/null 2>&1 || exit 1
python3 - <<'PY'
import csv, glob
f = glob.glob("/tmp/prof_l/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [r for r in rows if "train_layer" in r["Kernel_Name"] or "wgrad" in r["Kernel_Name"]]
# last full fwd+bwd iteration: take the last 17+9 launches
for r in ks[-26:]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("%-60s grid %-8s %8.1f us" % (r["Kernel_Name"][:60], r.get("Grid_Size",""), d))
PY
