#!/usr/bin/env python3
"""Summarise rocprofv3 (ROCm 7.2, rocpd sqlite output) results as text for profiles/.

    python tools/rocpd_summary.py gpurun_out/prof_<tag> > profiles/<name>.txt

Kernel trace -> per-kernel calls / total / average duration; PMC passes -> per kernel, the mean
over dispatches of every counter summed over its instances (dimensions).
"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def short(name):
    """Kernel name without namespaces and argument list, template arguments kept."""
    name = name.replace("(anonymous namespace)::", "").replace("mhx::", "").replace("void ", "")
    name = name.replace("unsigned long", "u64").replace("unsigned int", "u32")
    cut = name.find(">(")
    return (name[: cut + 1] if cut > 0 else name.split("(")[0])[:110]


def trace_summary(db):
    con = sqlite3.connect(db)
    rows = con.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# kernel trace: {db}")
    print("# steady = mean without each kernel's first dispatch (first touch of freshly allocated output pages) and")
    print("#          without dispatches that ran with the event counters on; this is what bench.py's timed steps see")
    print(f"{'calls':>6} {'total_us':>12} {'avg_us':>10} {'steady_us':>10} {'median_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}  kernel")
    for name, calls, tot, avg, mn, mx in rows:
        durs = [d for (d,) in con.execute("select duration from kernels where name = ? order by start", (name,))]
        steady = durs[1:-1] if len(durs) > 3 else durs
        med = sorted(durs)[len(durs) // 2]
        print(f"{calls:6d} {tot/1e3:12.1f} {avg/1e3:10.1f} {sum(steady)/len(steady)/1e3:10.1f} {med/1e3:10.1f} {mn/1e3:10.1f} {mx/1e3:10.1f} {100*tot/total:6.2f}  {short(name)}")
    for r in con.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, grid_x, grid_y, workgroup_x, lds_size, scratch_size from kernels group by name"):
        print(f"#   {short(r[0])}: vgpr={r[1]} agpr={r[2]} sgpr={r[3]} grid=({r[4]},{r[5]}) wg={r[6]} lds={r[7]} scratch={r[8]}")


def pmc_summary(db):
    con = sqlite3.connect(db)
    per = defaultdict(lambda: defaultdict(float))  # (kernel, counter) -> dispatch -> sum over instances
    for kernel, disp, counter, value in con.execute("select kernel_name, dispatch_id, counter_name, value from counters_collection"):
        per[(kernel, counter)][disp] += value
    print(f"# pmc: {db}")
    for (kernel, counter), d in sorted(per.items()):
        vals = list(d.values())
        print(f"{counter:28s} mean/dispatch={sum(vals)/len(vals):18.1f}  dispatches={len(vals):3d}  {short(kernel)}")


def main():
    root = sys.argv[1]
    for db in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        if "trace" in os.path.basename(os.path.dirname(db)):
            trace_summary(db)
        else:
            pmc_summary(db)
        print()


if __name__ == "__main__":
    main()
