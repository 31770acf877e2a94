#!/bin/bash
# PMC passes over the stand-alone MLP kernel timings: scratch/prof_mlp.sh <tag>
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/pmc_$1; rm -rf $O; mkdir -p $O
B="python3 scratch/time_mlp.py 3"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU --output-format csv -d $O/p1 -- $B > $O/p1.out 2> $O/p1.err || exit 1
echo "pass 1 done"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $O/p2 -- $B > $O/p2.out 2> $O/p2.err || exit 1
echo "pass 2 done"
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum --output-format csv -d $O/p3 -- $B > $O/p3.out 2> $O/p3.err || exit 1
echo "pass 3 done"
python3 - $O <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for p in ("p1", "p2", "p3"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{O}/{p}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:60]
            if "mlp" not in k: continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in sorted(acc.items()):
        # dispatches come in groups (3 sizes x (3+3) launches); report the mean of the largest third = fine-size launches
        print(p, k)
        for c, v in sorted(cs.items()):
            v = sorted(v)
            print(f"    {c:32s} n={len(v):3d} min {v[0]:.4g} median {v[len(v)//2]:.4g} max {v[-1]:.4g}")
PY
