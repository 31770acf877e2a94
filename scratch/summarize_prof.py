"""Summarise the rocprofv3 CSVs of scratch/prof.sh into small text files for profiles/<tag>/ and the
per-launch HBM traffic table profiles/traffic.json that bench.py reports as roofline.traffic.

    python scratch/summarize_prof.py gpurun_out/prof_<tag> profiles/<tag>
"""
import collections
import csv
import json
import os
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
cmd = open(os.path.join(src, "command.txt")).read().strip()

# bench.py's kernel names -> device kernel names
BENCH_NAMES = {"mlp_bwd_weights": "mlp_wgrad_kernel", "mlp_bwd_data": "mlp_dgrad_kernel",
               "mlp_fwd_train": "mlp_fwd_kernel<0, true, false, true>", "mlp_fwd": "mlp_fwd16_kernel<0, true>"}


def short(n):
    n = re.sub(r"\(.*", "", n)
    return n.replace("void ", "")[:90]


def find(sub, name):
    """newest match: results of earlier runs merged into the same directory stay behind under other process ids"""
    hits = [os.path.join(root, f) for root, _, files in os.walk(os.path.join(src, sub)) for f in files if f.endswith(name)]
    if not hits:
        raise FileNotFoundError(f"{sub}/**/{name}")
    return max(hits, key=os.path.getmtime)


rows = list(csv.DictReader(open(find("trace", "kernel_stats.csv"))))
with open(os.path.join(dst, "kernel_stats.txt"), "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats -- {cmd}\n")
    f.write("# (graph capture, 3 warm-up + 10 timed graph-replayed training steps at 1024 rays, 3 eager steps, then the frame renders, four camera-path frames and the whole-frame kernel launches)\n")
    f.write(f"{'kernel':92s} {'calls':>6s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>7s}\n")
    for r in rows[:30]:
        f.write(f"{short(r['Name']):92s} {r['Calls']:>6s} {float(r['AverageNs'])/1e3:10.1f} "
                f"{float(r['MinNs'])/1e3:10.1f} {float(r['MaxNs'])/1e3:10.1f} {float(r['Percentage']):7.2f}\n")


def pmc(sub):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(find(sub, "counter_collection.csv"))):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def write(agg, out, note):
    with open(os.path.join(dst, out), "w") as f:
        f.write(note + "\n")
        for k, cs in sorted(agg.items(), key=lambda kv: -sum(sum(v) for v in kv[1].values())):
            if "nerf::" not in k:
                continue
            f.write(f"{k:80s} n={len(next(iter(cs.values()))):4d} " + " ".join(
                f"{c}: mean {sum(v)/len(v):.4g} max {max(v):.4g}" for c, v in sorted(cs.items())) + "\n")


mf, fe, wr = pmc("pmc_mfma"), pmc("pmc_fetch"), pmc("pmc_write")
write(mf, "pmc_mfma.txt", f"# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -- {cmd}\n"
      "# per dispatch; MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)")
write(fe, "pmc_fetch.txt", f"# rocprofv3 --pmc FETCH_SIZE -- {cmd}\n# KiB per dispatch as reported; gfx950 tallies a wide "
      "coalesced stream at half its bytes")
write(wr, "pmc_write.txt", f"# rocprofv3 --pmc WRITE_SIZE -- {cmd}\n# KiB per dispatch")

with open(os.path.join(dst, "mfma_util.txt"), "w") as f:
    f.write("# MFMA pipe busy fraction and effective clock per kernel (means over dispatches)\n")
    stats = {short(r["Name"]): float(r["AverageNs"]) for r in rows}
    for k, cs in mf.items():
        if "mlp_" not in k or "SQ_VALU_MFMA_BUSY_CYCLES" not in cs:
            continue
        busy = sum(cs["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(cs["SQ_VALU_MFMA_BUSY_CYCLES"])
        gui = sum(cs["GRBM_GUI_ACTIVE"]) / len(cs["GRBM_GUI_ACTIVE"])
        line = f"{k:70s} mfma_busy {busy / (gui / 8 * 1024):.3f}"
        if k in stats:
            line += f"  clock_GHz {gui / 8 / stats[k]:.2f}"
        f.write(line + "\n")

bench = json.loads(open(os.path.join(src, "trace_bench.json")).read().strip().splitlines()[-1])

# ---- the training step's kernels inside the replayed graphs, from the per-dispatch trace -----------------------
# Dispatches of one kernel in time order: bench.py's first TrainStep issues 2 eager warm-up steps on a side
# stream, then replays (warmup + steps) graphs -- the timed region; its second, time-stamped TrainStep issues 2
# eager warm-ups and replays 1 + kernel_steps graphs.  Two networks per step: 2 dispatches per step and kernel.
try:
    import shlex
    a = shlex.split(cmd)
    getopt = lambda name, default: int(a[a.index(name) + 1]) if name in a else default   # noqa: E731
    n_timed = getopt("--warmup", 10) + getopt("--steps", 50) * getopt("--repeats", 3)
    n_stamped = 1 + getopt("--kernel-steps", 10)
    trace = collections.defaultdict(list)
    for r in csv.DictReader(open(find("trace", "kernel_trace.csv"))):
        trace[short(r["Kernel_Name"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    step_kernels = ["mlp_fwd_kernel<0, true, false, true>", "mlp_dgrad_kernel<0, true>", "mlp_wgrad_kernel<0>",
                    "mlp_reduce_kernel", "mlp_post_kernel"]
    with open(os.path.join(dst, "step_kernels.txt"), "w") as f:
        f.write(f"# from the per-dispatch kernel trace of: {cmd}\n")
        f.write("# average duration (us) of the training step's MLP kernels over both networks, split by how the dispatch was issued:\n")
        f.write(f"#   timed   = the {n_timed} graph replays of the timed region and its warm-up\n")
        f.write(f"#   stamped = the {n_stamped} replays of the copy of the graph that carries bench.py's time-stamp kernels\n")
        f.write("#   eager   = the 2 + 2 warm-up steps issued one by one before the two captures\n")
        f.write(f"{'kernel':60s} {'timed':>9s} {'stamped':>9s} {'eager':>9s}\n")
        sums = {"timed": 0.0, "stamped": 0.0}
        for kn in step_kernels:
            k = [x for x in trace if kn in x]
            if not k:
                continue
            d = [(e - s0) / 1e3 for s0, e in sorted(trace[k[0]])]
            e1, t, e2, st = d[:4], d[4:4 + 2 * n_timed], d[4 + 2 * n_timed:8 + 2 * n_timed], d[8 + 2 * n_timed:8 + 2 * n_timed + 2 * n_stamped]
            avg = lambda v: sum(v) / len(v) if v else float("nan")   # noqa: E731
            f.write(f"{k[0][:60]:60s} {avg(t):9.1f} {avg(st):9.1f} {avg(e1 + e2):9.1f}\n")
            if any(x in kn for x in ("wgrad", "reduce", "post")):
                sums["timed"] += avg(t); sums["stamped"] += avg(st)
        f.write(f"{'weight-gradient call (wgrad + reduce + post)':60s} {sums['timed']:9.1f} {sums['stamped']:9.1f}\n")
        rl = bench.get("roofline", {})
        f.write(f"# bench.py reported for the same run: roofline.kernel {rl.get('kernel')} avg_launch_us {rl.get('avg_launch_us')} "
                f"(time stamps around the call in the stamped replays)\n")
    print(open(os.path.join(dst, "step_kernels.txt")).read())
except Exception as e:      # noqa: BLE001
    print("step_kernels.txt not written:", type(e).__name__, e)

traffic = {"command": cmd, "rays_per_gpu": bench["config"]["rays_per_gpu"], "precision": bench["dtype"],
           "note": "means over the dispatches of one run; hbm_bytes_per_launch = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes)",
           "kernels": {}}
for bname, kname in BENCH_NAMES.items():
    fk = [k for k in fe if kname in k]
    wk = [k for k in wr if kname in k]
    if not fk or not wk:
        continue
    fv, wv = fe[fk[0]]["FETCH_SIZE"], wr[wk[0]]["WRITE_SIZE"]
    fetch, wrt = sum(fv) / len(fv) * 1024, sum(wv) / len(wv) * 1024
    traffic["kernels"][bname] = {"fetch_size_bytes_reported": round(fetch), "write_size_bytes": round(wrt),
                                 "hbm_bytes_per_launch": round(2 * fetch + wrt), "dispatches": len(fv)}
with open(os.path.join(os.path.dirname(os.path.abspath(dst)), "traffic.json"), "w") as f:
    json.dump(traffic, f, indent=1)
print(json.dumps(traffic["kernels"], indent=1))
