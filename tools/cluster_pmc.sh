#!/bin/bash
# Counter passes over the clustering bench (run through gpurun): HBM bytes and pipe activity of the pair passes.
#   tools/cluster_pmc.sh <tag>
tag=${1:-rXX}
out=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    d=/tmp/prof_clp_$(echo $pass | tr -c 'A-Za-z0-9' '_')
    rm -rf $d
    timeout -s KILL 120 rocprofv3 --pmc $pass --output-format csv -d $d -o p -- python3 /root/repo/tools/bench_cluster.py --order random --cpu-n 0 --steps 2 > /tmp/clp.log 2>&1 || { tail -3 /tmp/clp.log; exit 1; }
    f=$(find $d -name "*counter_collection.csv" | sed -n 1p)
    python3 - "$f" "$pass" <<'P'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    n = r["Kernel_Name"].replace("void ", "")
    if "cl_pair" in n or "cl_reach" in n or "cl_resolve" in n:
        agg[n[:28]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(k, {a: sum(v[-2:]) / len(v[-2:]) for a, v in c.items()})
P
done > $out/${tag}_cluster_pmc.txt 2>&1
cat $out/${tag}_cluster_pmc.txt
