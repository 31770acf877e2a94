#!/bin/bash
# rocprofv3 passes over a short bench run: scratch/prof.sh <tag>  -> gpurun_out/prof_<tag>/
# (kernel trace + stats, then three counter passes, each in a run of its own)
set -e
TAG=${1:-r3}
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O
B="python3 bench.py --steps 10 --warmup 3 --repeats 1 --big-batch 0 --no-default-flags --no-cpu-baseline --no-one-eighth --kernel-steps 4"
echo "$B" > $O/command.txt
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- $B > $O/trace_bench.json 2> $O/trace.err
echo "trace done"
timeout -s KILL 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -- $B > $O/pmc_mfma_bench.json 2> $O/pmc_mfma.err
echo "pmc mfma done"
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- $B > $O/pmc_fetch_bench.json 2> $O/pmc_fetch.err
echo "pmc fetch done"
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- $B > $O/pmc_write_bench.json 2> $O/pmc_write.err
echo "pmc write done"
du -sh $O
