#!/bin/bash
# wait-state counters of the inference forward kernel
set -e
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_fwd; rm -rf $O; mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $O/p1 -- python3 scratch/fwd_only.py > $O/p1.out 2> $O/p1.err
echo p1
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM --output-format csv -d $O/p2 -- python3 scratch/fwd_only.py > $O/p2.out 2> $O/p2.err
echo p2
python3 - <<PY
import csv, glob, collections
for d in ("p1","p2"):
    for f in glob.glob("$O/%s/**/*counter_collection.csv"%d, recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "mlp_fwd" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items(): print(d, k, "n=%d mean=%.4g"%(len(v), sum(v)/len(v)))
PY
