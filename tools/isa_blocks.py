#!/usr/bin/env python3
"""tools/isa_blocks.py -- per-basic-block instruction mix of one kernel in a gfx950 listing.

    python tools/isa_blocks.py k.s <mangled-kernel-name-substring>
Prints, per block: loop depth, VALU (of which v_readlane/v_writelane spills), SALU, SMEM, VMEM and
branch targets -- enough to count the instructions on a hot path by hand.
"""
import re
import sys


def main():
    text = open(sys.argv[1]).read()
    pat = sys.argv[2]
    m = re.search(r"^([^\n]*%s[^\n:]*):[^\n]*\n(.*?)s_endpgm" % re.escape(pat), text, re.S | re.M)
    lines = m.group(2).split("\n")
    blocks, cur = [], None
    for i, l in enumerate(lines):
        lab = re.match(r"^(\.LBB\d+_\d+):", l) or re.match(r"^; (%bb\.\d+):", l)
        if lab or cur is None:
            cur = {"name": lab.group(1) if lab else "entry", "line": i + 1, "valu": 0, "lane": 0, "salu": 0, "smem": 0, "vmem": 0, "mad": 0, "br": [], "depth": ""}
            d = re.search(r"Depth=(\d)", l)
            cur["depth"] = d.group(1) if d else ""
            blocks.append(cur)
            if lab:
                continue
        t = l.strip()
        if t.startswith(("v_readlane", "v_writelane")):
            cur["lane"] += 1
            cur["valu"] += 1
        elif t.startswith("v_"):
            cur["valu"] += 1
            cur["mad"] += t.startswith("v_mad_u64")
        elif t.startswith("s_load"):
            cur["smem"] += 1
        elif t.startswith(("global_", "flat_", "scratch_")):
            cur["vmem"] += 1
        elif t.startswith(("s_cbranch", "s_branch")):
            cur["br"].append(t.split()[-1])
            cur["salu"] += 1
        elif t.startswith("s_") and not t.startswith(("s_waitcnt", "s_nop")):
            cur["salu"] += 1
    for b in blocks:
        if b["valu"] + b["salu"] + b["vmem"] + b["smem"]:
            print(f"{b['name']:10s} L{b['line']:5d} d={b['depth']:1s} valu={b['valu']:4d} (mad64 {b['mad']:3d}, lane {b['lane']:3d}) salu={b['salu']:4d} smem={b['smem']:2d} vmem={b['vmem']:2d} -> {' '.join(b['br'])}")


if __name__ == "__main__":
    main()
