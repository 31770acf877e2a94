"""Summarise rocprofv3 CSVs under gpurun_out/prof_r1 into small text files for profiles/."""
import csv, sys, collections, os, re
src = sys.argv[1]; dst = sys.argv[2]
os.makedirs(dst, exist_ok=True)
def short(n):
    n = re.sub(r"\(.*", "", n)
    n = n.replace("void ", "")
    return n[:90]
def find(sub, name):
    for root, _, files in os.walk(os.path.join(src, sub)):
        for f in files:
            if f.endswith(name): return os.path.join(root, f)
# kernel stats
p = find("trace", "kernel_stats.csv")
rows = list(csv.DictReader(open(p)))
with open(os.path.join(dst, "kernel_stats.txt"), "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --kernel-steps 2\n")
    f.write("# (10 timed graph-replayed steps + 3 warm-up + 5 eager steps + render section: 4 frames of 800x800)\n")
    f.write(f"{'kernel':92s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>7s}\n")
    for r in rows[:28]:
        f.write(f"{short(r['Name']):92s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.1f} {float(r['MinNs'])/1e3:10.1f} {float(r['MaxNs'])/1e3:10.1f} {float(r['Percentage']):7.2f}\n")
# pmc
def pmc(sub, out, note):
    p = find(sub, "counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(p)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(os.path.join(dst, out), "w") as f:
        f.write(note + "\n")
        for k, cs in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
            if not any(t in k for t in ("nerf::",)): continue
            line = f"{k:80s} n={len(next(iter(cs.values()))):4d} " + " ".join(f"{c}: mean {sum(v)/len(v):.4g} max {max(v):.4g}" for c, v in sorted(cs.items()))
            f.write(line + "\n")
pmc("pmc_mfma", "pmc_mfma.txt", "# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE (per dispatch; mean and max over dispatches)")
pmc("pmc_fetch", "pmc_fetch.txt", "# rocprofv3 --pmc FETCH_SIZE (KB per dispatch as reported; on gfx950 a wide coalesced stream is under-reported 2x)")
pmc("pmc_write", "pmc_write.txt", "# rocprofv3 --pmc WRITE_SIZE (KB per dispatch)")
