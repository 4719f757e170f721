"""Per-stream picture of ONE training step from a rocprofv3 kernel trace (rocpd sqlite database).

    python tools/stream_busy.py results.db [skip_steps_from_end]

The step window runs from the end of one step's last adamw_kernel to the end of the next step's.  Prints, per HIP
stream: kernels, busy time (sum of durations), span; then the union busy time of all streams, the main stream's idle
time inside the window split into "another stream was busy" / "chip idle", the tail after the main stream's last
non-optimizer kernel, and the top kernels of every stream by time.
"""
import collections
import sqlite3
import sys


def main(path, skip=1):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    q = "stream_id" if "stream_id" in cols else "queue_id"
    rows = list(c.execute(f"select start, end, {q}, name from kernels order by start"))
    ends = [r[1] for r in rows if "adamw_kernel" in r[3]]
    # group adamw launches that belong to one step (gaps < 2 ms)
    groups = []
    for e in ends:
        if groups and e - groups[-1][-1] < 15e6:  # (one step: the predictor's AdamW runs ~8 ms before the style encoder's)
            groups[-1].append(e)
        else:
            groups.append([e])
    if len(groups) < skip + 2:
        print("not enough steps in the trace", file=sys.stderr)
        sys.exit(1)
    t0, t1 = groups[-2 - skip][-1], groups[-1 - skip][-1]
    win = [(s, e, st, n) for s, e, st, n in rows if s >= t0 and e <= t1]
    print(f"step window {(t1 - t0) / 1e6:.3f} ms, {len(win)} kernels")
    per = collections.defaultdict(list)
    for s, e, st, n in win:
        per[st].append((s, e, n))
    main_stream = max(per, key=lambda k: len(per[k]))

    def union(iv):
        iv = sorted(iv)
        out = []
        for s, e in iv:
            if out and s <= out[-1][1]:
                out[-1][1] = max(out[-1][1], e)
            else:
                out.append([s, e])
        return out

    def length(iv):
        return sum(e - s for s, e in iv)

    all_u = union([(s, e) for s, e, _, _ in win])
    print(f"union busy of all streams {length(all_u) / 1e6:.3f} ms  (chip idle {(t1 - t0 - length(all_u)) / 1e6:.3f} ms)")
    for st, ks in sorted(per.items(), key=lambda kv: -len(kv[1])):
        u = union([(s, e) for s, e, _ in ks])
        tag = " (main)" if st == main_stream else ""
        print(f"stream {st}{tag}: {len(ks)} kernels, sum of durations {sum(e - s for s, e, _ in ks) / 1e6:.3f} ms, "
              f"busy {length(u) / 1e6:.3f} ms, from {(min(s for s, _, _ in ks) - t0) / 1e6:.3f} to {(max(e for _, e, _ in ks) - t0) / 1e6:.3f} ms")
    mu = union([(s, e) for s, e, _ in per[main_stream]])
    gaps = [(a[1], b[0]) for a, b in zip(mu, mu[1:])]
    other = union([(s, e) for st, ks in per.items() if st != main_stream for s, e, _ in ks])
    covered = 0
    for gs, ge in gaps:
        for s, e in other:
            lo, hi = max(gs, s), min(ge, e)
            if hi > lo:
                covered += hi - lo
    gl = length(gaps)
    print(f"main stream: {len(gaps)} gaps, {gl / 1e6:.3f} ms in total ({covered / 1e6:.3f} ms of it with another stream busy); "
          f"gaps > 20 us: {sum(1 for s, e in gaps if e - s > 2e4)} totalling {sum(e - s for s, e in gaps if e - s > 2e4) / 1e6:.3f} ms")
    for st, ks in sorted(per.items(), key=lambda kv: -len(kv[1])):
        agg = collections.Counter()
        cnt = collections.Counter()
        for s, e, n in ks:
            key = n.replace("void sty::", "").replace("sty::", "")[:70]
            agg[key] += e - s
            cnt[key] += 1
        print(f"--- stream {st}: top kernels")
        for k, v in agg.most_common(45 if st == main_stream else 14):
            print(f"   {v / 1e6:8.3f} ms {cnt[k]:5d}  {k}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
