#!/usr/bin/env python
"""Turn a rocprofv3 results .db (rocpd sqlite, `rocprofv3 --kernel-trace --stats`)
into the per-kernel summary table committed under profiles/.

  python tools/rocprof_summary.py gpurun_out/prof_x/x_results.db "command line" > profiles/<name>.txt
"""
import sqlite3
import sys


def main():
    db, cmd = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    print("# rocprofv3 --kernel-trace --stats summary (durations in microseconds)")
    if cmd:
        print("# command: " + cmd)
    print("%-72s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, tot, avg, pct in rows:
        print("%-72s %8d %14.3f %12.3f %7.2f" % (name[:72], calls, tot, avg, pct))


if __name__ == "__main__":
    main()
