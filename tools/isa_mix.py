#!/usr/bin/env python3
"""usage: tools/isa_mix.py <asm file> <kernel name substring>: static instruction mix of one kernel
(device-only assembly from tools/kernel_regs.sh)."""
import collections
import re
import sys
asm, pat = sys.argv[1], sys.argv[2]
mix, inside, total = collections.Counter(), False, 0
for line in open(asm):
    t = line.strip()
    m = re.match(r"^([A-Za-z_][\w$.]*):", t)
    if m and not t.startswith(".L"):
        inside = pat in m.group(1)
        continue
    if t.startswith(".Lfunc_end") or t.startswith(".amdhsa_kernel"):
        inside = False
    if not inside or not t or t[0] in ".;":
        continue
    op = t.split()[0]
    if not re.match(r"^[a-z]", op):
        continue
    total += 1
    kind = ("valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_")
            else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
    mix[kind] += 1
    if kind in ("lds", "vmem"):
        mix[op] += 1
print(total, dict(mix))
