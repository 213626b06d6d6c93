#!/usr/bin/env python3
"""Instruction statistics of the kernels in a gfx950 assembly listing (hipcc --cuda-device-only -S).

    python scripts/dev_isa_stats.py listing.s [name-filter]

Per kernel: instructions in total, vector-ALU instructions, wait states (s_nop), selected opcodes, registers.
Static counts over the whole body (all loops unrolled or not): compare two builds of one kernel, not two kernels."""
import collections
import re
import sys


def main():
    s = open(sys.argv[1]).read()
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    regs = {m.group(1): m.group(2) for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s*(\d+)", s)}
    for m in re.finditer(r"^(\S+):\s*; @\1\n(.*?)^\.Lfunc_end\d+:", s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if flt not in name:
            continue
        ins = [ln.split()[0] for ln in body.split("\n") if ln.startswith("\t") and not ln.lstrip().startswith((".", ";"))]
        c = collections.Counter(ins)
        valu = sum(n for k, n in c.items() if k.startswith("v_"))
        top = ", ".join(f"{k} {n}" for k, n in c.most_common(14))
        print(f"{name[:70]}\n  total {len(ins)} valu {valu} s_nop {c['s_nop']} vgpr {regs.get(name, '?')}\n  {top}")


if __name__ == "__main__":
    main()
