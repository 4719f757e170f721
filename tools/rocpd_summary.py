"""Summarise a rocprofv3 (ROCm 7 rocpd sqlite) kernel trace into the per-kernel table the judge reads.

    python tools/rocpd_summary.py gpurun_out/prof_c5/c5_results.db > profiles/r01_c5_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    tot = sum(r[2] for r in rows)
    print(f"# rocprofv3 --kernel-trace --stats summary of {path}")
    print(f"# total kernel time {tot / 1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches (durations in us)")
    print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'pct':>6}  name")
    for name, calls, total, avg, pct in rows:
        # the whole name: bench.py joins its roofline to these rows by instantiation (rounds 1-5 cut names at 107 characters)
        print(f"{calls:7d} {total:12.1f} {avg:10.2f} {pct:6.2f}  {name}")


if __name__ == "__main__":
    main(sys.argv[1])
