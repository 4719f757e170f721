"""The style encoder's forward chain in one traced c3 step (stream 1 from the step start to its first long idle gap)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select start, end, stream_id, name from kernels order by start"))
# step start ~ the first spectral-norm power-iteration kernel of a step in the middle of the trace
sn = [r for r in rows if "sn_wt_u_kernel" in r[3]]
starts = [sn[0]]
for r in sn[1:]:
    if r[0] - starts[-1][0] > 20e6:
        starts.append(r)
i = len(starts) // 2
t0, nxt = starts[i][0], starts[i + 1][0]
for r in rows:
    if t0 <= r[0] < nxt and r[2] == 1 and (r[0] - t0) < 12e6:
        print(f"s1 {(r[0] - t0) / 1e6:8.3f} +{(r[1] - r[0]) / 1e3:7.1f} us {r[3][:100]}")
main = [r for r in rows if t0 <= r[0] < nxt and r[2] == 0 and (r[0] - t0) < 12e6]
prev = t0
for r in main:
    if r[0] - prev > 100e3:
        print(f"main gap {(prev - t0) / 1e6:8.3f} .. {(r[0] - t0) / 1e6:8.3f} before {r[3][:60]}")
    prev = max(prev, r[1])
