"""Audit of the hand-issued register loads in the chain kernels' disassembly.

An inline-asm `global_load_dword` is invisible to the compiler's memory counters: its destination register
counts as written where the statement ends, so the compiler may copy, spill or reuse it before the data has
landed.  For every such load inside a kernel (dword loads with a VGPR-pair address and `off`, the form the
source uses) this script follows the destination register up to the next `s_waitcnt vmcnt(N)` with an N the
source was built with (mlp_waits.hpp) or vmcnt(0) and reports any instruction that touches it in between.

    llvm-objdump -d <code object> > k.s ;  python audit_asm_loads.py k.s   (or: make audit)
"""
import os
import re
import sys

# the waits that end a load's window: the counts the source was built with (mlp_waits.hpp), and vmcnt(0)
_hdr = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "mlp_waits.hpp")).read()
_counts = re.findall(r"#define\s+NERF_INPUTS_WAIT_\w+\s+(\d+)", _hdr)
if len(_counts) != 2:
    sys.exit("audit_asm_loads.py: expected two NERF_INPUTS_WAIT_* counts in mlp_waits.hpp")
WAITS = {"vmcnt(%s)" % c for c in _counts} | {"vmcnt(0)"}
text = open(sys.argv[1]).read().splitlines()
kernel, bad, n_loads = None, 0, 0
pending = {}          # register -> line number of its load
for i, line in enumerate(text):
    m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
    if m:
        kernel, pending = m.group(1), {}
        continue
    ins = line.split("//")[0].strip()
    if not ins:
        continue
    m = re.match(r"global_load_dword (v\d+), v\[\d+:\d+\], off", ins)
    if m and "mlp_fwd_kernel" in (kernel or ""):
        pending[m.group(1)] = i
        n_loads += 1
        continue
    if ins.startswith("s_waitcnt") and any(w in ins for w in WAITS):
        pending = {}
        continue
    for reg, at in list(pending.items()):
        n = int(reg[1:])
        hit = re.search(r"\b%s\b" % reg, ins) is not None
        for a, b in re.findall(r"v\[(\d+):(\d+)\]", ins):
            hit = hit or int(a) <= n <= int(b)
        if hit:
            print(f"{kernel}: line {i + 1}: '{ins}' touches {reg} loaded at line {at + 1} before its wait")
            bad += 1
print(f"{n_loads} hand-issued loads followed (waits {sorted(WAITS)}), {bad} early uses")
if n_loads == 0:
    sys.exit("audit_asm_loads.py: no hand-issued load found -- the disassembly or the load's form changed")
sys.exit(1 if bad else 0)
