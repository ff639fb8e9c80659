#!/usr/bin/env python3
"""tools/timeline.py <rocprofv3 output dir> [n] -- the last n kernel dispatches of a --kernel-trace run in start order:
start offset, duration and the gap to the previous dispatch's end (microseconds)."""
import glob
import os
import sqlite3
import sys

from rocpd_summary import short


def main():
    root, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 24
    for db in sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True)):
        con = sqlite3.connect(db)
        rows = con.execute("select name, start, end from kernels order by start").fetchall()[-n:]
        t0, prev_end = rows[0][1], None
        print(f"# {db}")
        for name, start, end in rows:
            gap = "" if prev_end is None else f"{(start - prev_end) / 1e3:8.1f}"
            print(f"{(start - t0) / 1e3:10.1f} us  dur {(end - start) / 1e3:8.1f}  gap {gap:>8}  {short(name)}")
            prev_end = end


if __name__ == "__main__":
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    main()
