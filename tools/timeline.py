"""Dump the kernel timeline of a rocprofv3 rocpd database as CSV: start_us, dur_us, queue, stream, name.

    python tools/timeline.py results.db > gpurun_out/timeline.csv
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = "kernels" if "kernels" in names else None
    if view is None:
        print("tables:", names, file=sys.stderr)
        sys.exit(1)
    cols = [r[1] for r in c.execute(f"pragma table_info({view})")]
    print("# columns:", cols, file=sys.stderr)
    q = "queue_id" if "queue_id" in cols else "0"
    s = "stream_id" if "stream_id" in cols else "0"
    rows = list(c.execute(f"select start, end, {q}, {s}, name from {view} order by start"))
    t0 = rows[0][0]
    print("start_us,dur_us,queue,stream,name")
    for st, en, qq, ss, name in rows:
        print(f"{(st - t0) / 1e3:.2f},{(en - st) / 1e3:.2f},{qq},{ss},\"{name[:80]}\"")


if __name__ == "__main__":
    main(sys.argv[1])
