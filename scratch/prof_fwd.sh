#!/bin/bash
# MFMA-pipe and wait-state counters of the inference forward kernel (scratch/fwd_only.py)
set -e
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_fwd; rm -rf $O; mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- python3 scratch/fwd_only.py > $O/p1.out 2> $O/p1.err
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 scratch/fwd_only.py > $O/tr.out 2> $O/tr.err
python3 - <<PY
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob("$O/p1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mlp_fwd" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m={k:sum(v)/len(v) for k,v in acc.items()}
for k,v in m.items(): print(k, "%.4g"%v)
dur=None
for f in glob.glob("$O/tr/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mlp_fwd" in r["Name"]: dur=float(r["AverageNs"])
print("avg ns", dur)
gui=m["GRBM_GUI_ACTIVE"]/8
print("mfma busy %.3f  clock %.2f GHz  wait_any/wave %.3f wait_inst/wave %.3f" % (m["SQ_VALU_MFMA_BUSY_CYCLES"]/(gui*1024), gui/dur, m["SQ_WAIT_ANY"]/m["SQ_WAVE_CYCLES"], m["SQ_WAIT_INST_ANY"]/m["SQ_WAVE_CYCLES"]))
PY
