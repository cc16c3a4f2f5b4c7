#!/usr/bin/env python3
"""Static look at exposed memory waits in a kernel's ISA (hipcc -S output).

For every s_waitcnt it reports how many VALU instructions were issued since the youngest
LDS / scalar / vector memory instruction it has to wait for: a small number means the round
trip (LDS ~64+ cycles, 16 VALU slots) is exposed when one wave owns the SIMD.
usage: lds_waits.py kernel.s [max_cover]
"""
import re
import sys

lines = open(sys.argv[1]).read().splitlines()
maxcov = int(sys.argv[2]) if len(sys.argv) > 2 else 12
mem_re = re.compile(r"^\s+(ds_read|ds_write|s_load|s_buffer_load|global_load|global_store|buffer_load|flat_load)")
valu_re = re.compile(r"^\s+v_")
since = {"lgkm": None, "vm": None}
valu = 0
label = ""
out = []
for i, ln in enumerate(lines):
    if ln.startswith(".LBB"):
        label = ln.split(":")[0]
    m = mem_re.match(ln)
    if m:
        kind = "vm" if m.group(1).startswith(("global", "buffer", "flat")) else "lgkm"
        since[kind] = (valu, i, ln.strip().split()[0])
    elif valu_re.match(ln):
        valu += 1
    elif "s_waitcnt" in ln:
        for kind, key in (("lgkm", "lgkmcnt"), ("vm", "vmcnt")):
            if key in ln and since[kind] is not None:
                cover = valu - since[kind][0]
                if cover <= maxcov and not since[kind][2].startswith(("ds_write", "global_store")):
                    out.append((i + 1, label, cover, since[kind][2], ln.strip()))
for o in out:
    print("%6d %-10s cover=%2d after %-22s %s" % o)
print(len(out), "waits with cover <=", maxcov)
