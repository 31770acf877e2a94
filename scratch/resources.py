"""Per-kernel register / LDS usage of the MLP kernels (hipcc -Rpass-analysis=kernel-resource-usage)."""
import os, re, subprocess, sys
files = sys.argv[1:] or ["mlp_fwd", "mlp_bwd"]
for f in files:
    out = subprocess.run(["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-pass-failed",
                          "-mllvm", "-pragma-unroll-threshold=1000000", "-I../../include", *os.environ.get("NERF_EXTRA", "").split(), "-c", f + ".hip",
                          "-o", "/tmp/" + f + ".o", "-Rpass-analysis=kernel-resource-usage"],
                         cwd="/root/repo/nerf_amd/csrc", capture_output=True, text=True).stderr
    cur = {}
    for line in out.splitlines():
        m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?)\s*\[-Rpass", line) or re.search(r":\d+:\d+: remark:\s+(.*?)\s*\[-Rpass", line)
        if not m:
            m = re.search(r"\d+:\d+:\s+(?:remark:\s+)?(.*?)\s*\[-Rpass", line)
        if not m:
            continue
        t = m.group(1)
        if t.startswith("Function Name:") or t.startswith("Name:"):
            if cur: print(cur)
            name = subprocess.run(["c++filt", t.split(":", 1)[1].strip()], capture_output=True, text=True).stdout.strip()
            cur = {"kernel": name[:70]}
        elif ":" in t:
            k, v = t.split(":", 1)
            if k.strip() in ("VGPRs", "AGPRs", "SGPRs", "VGPRs Spill", "SGPRs Spill", "ScratchSize [bytes/lane]", "LDS Size [bytes/block]", "Occupancy [waves/SIMD]"):
                cur[k.strip().split(" [")[0]] = v.strip()
    if cur: print(cur)
