"""When does the style encoder's backward start?  From a rocprofv3 kernel trace (rocpd database) of the c3 step: the end of
style_fc_bwd_kernel (d_style complete) on the main stream, the first style-encoder-stream kernel after it, and what the
main stream runs in between.     python tools/se_bwd_start.py results.db"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = list(c.execute("select start, end, stream_id, name from kernels order by start"))
fcs = [r for r in rows if "style_fc_bwd" in r[3]]
print("style_fc_bwd launches:", len(fcs))
for fc in fcs[4:7]:
    t0 = fc[1]
    nxt_adam = next((r for r in rows if r[0] > t0 and "adamw" in r[3]), None)
    if not nxt_adam:
        continue
    win = [r for r in rows if t0 <= r[0] <= nxt_adam[0]]
    print(f"\nd_style done at 0; adamw at {(nxt_adam[0] - t0) / 1e6:.3f} ms; stream of fc_bwd {fc[2]}")
    per = {}
    for r in win:
        per.setdefault(r[2], []).append(r)
    for s, ks in per.items():
        print(f"  stream {s}: {len(ks)} kernels, first at {(ks[0][0] - t0) / 1e6:.3f} ms ({ks[0][3][:50]}), "
              f"last ends {(ks[-1][1] - t0) / 1e6:.3f} ms, busy {sum(k[1] - k[0] for k in ks) / 1e6:.3f} ms")
    # first 12 kernels of each non-main stream in the window
    for s, ks in per.items():
        if s == fc[2]:
            continue
        for k in ks[:6]:
            print(f"    s{s} {(k[0] - t0) / 1e6:8.3f} +{(k[1] - k[0]) / 1e3:7.1f} us {k[3][:70]}")

# full list of the style-encoder streams' kernels in the last printed window
if len(sys.argv) > 2:
    fc = fcs[6]
    t0 = fc[1]
    nxt_adam = next((r for r in rows if r[0] > t0 and "adamw" in r[3]), None)
    for r in rows:
        if t0 <= r[0] <= nxt_adam[0] and r[2] in (1, 3):
            print(f"s{r[2]} {(r[0] - t0) / 1e6:8.3f} +{(r[1] - r[0]) / 1e3:7.1f} us {r[3][:110]}")
