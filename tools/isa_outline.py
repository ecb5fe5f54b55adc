#!/usr/bin/env python3
"""Outline of a kernel's ISA (barriers, waits, MFMA / LDS / DMA runs) from a hipcc -S dump: isa_outline.py file.s symbol-substring"""
import re
import sys
s = open(sys.argv[1]).read()
names = [m.group(1) for m in re.finditer(r"^(\S+):\s*; @", s, re.M) if sys.argv[2] in m.group(1)]
name = names[int(sys.argv[3]) if len(sys.argv) > 3 else 0]
i = s.index(name + ":")
j = s.index(".Lfunc_end", i)
out = []
for l in s[i:j].splitlines():
    l = l.strip()
    m = re.match(r"(s_barrier|s_waitcnt\S*.*|v_mfma\S+|ds_read\S+|ds_write\S+|global_load_lds\S+|s_setprio.*|s_cbranch\S+.*|\.LBB\S+|global_store\S+|global_load\S+|buffer_\S+|s_endpgm|scratch_\S+)", l)
    if m:
        k = l.split(";")[0].strip() if l.startswith(("s_waitcnt", ".LBB", "s_cbranch", "s_setprio")) else m.group(1)
        if out and out[-1][0] == k:
            out[-1][1] += 1
        else:
            out.append([k, 1])
print(name)
print(" | ".join(f"{k} x{n}" if n > 1 else k for k, n in out))
