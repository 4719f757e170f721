"""Where the main stream of ONE training step waits (rocprofv3 kernel trace, rocpd sqlite database).

    python tools/step_gaps.py results.db [skip_steps_from_end] [min_gap_us]

Same step window as tools/stream_busy.py.  For every gap of the main stream longer than min_gap_us: when it starts, how
long it is, the kernels before and after it, and what the other streams ran inside it (name, launches, time).
"""
import collections
import re
import sqlite3
import sys


def short(n):
    return re.sub(r"\(.*", "", n.replace("sty::", "").replace("void ", "").replace("(anonymous namespace)::", ""))


def main(path, skip=1, min_gap=100.0):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    q = "stream_id" if "stream_id" in cols else "queue_id"
    rows = list(c.execute(f"select start, end, {q}, name from kernels order by start"))
    ends = [r[1] for r in rows if "adamw_kernel" in r[3]]
    groups = []
    for e in ends:
        if groups and e - groups[-1][-1] < 15e6:  # (one step: the predictor's AdamW runs ~8 ms before the style encoder's)
            groups[-1].append(e)
        else:
            groups.append([e])
    t0, t1 = groups[-2 - skip][-1], groups[-1 - skip][-1]
    win = [(s, e, st, n) for s, e, st, n in rows if s >= t0 and e <= t1]
    per = collections.defaultdict(list)
    for s, e, st, n in win:
        per[st].append((s, e, n))
    ms = max(per, key=lambda k: len(per[k]))
    print(f"step window {(t1 - t0) / 1e6:.3f} ms, main stream {ms}: {len(per[ms])} kernels")
    mk = sorted(per[ms])
    prev_end, prev_name = t0, "(step start)"
    for s, e, n in mk + [(t1, t1, "(step end)")]:
        gap = (s - prev_end) / 1e3
        if gap >= min_gap:
            print(f"\n{(prev_end - t0) / 1e6:8.3f} ms  gap {gap:8.1f} us   after {short(prev_name)}   before {short(n)}")
            for st, ks in per.items():
                if st == ms:
                    continue
                agg = collections.OrderedDict()
                for ks_, ke_, kn in sorted(ks):
                    ov = min(ke_, s) - max(ks_, prev_end)
                    if ov > 0:
                        a = agg.setdefault(short(kn), [0, 0.0])
                        a[0] += 1
                        a[1] += ov / 1e3
                if agg:
                    tot = sum(v[1] for v in agg.values())
                    print(f"    stream {st}: {tot:8.1f} us busy")
                    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:6]:
                        print(f"        {v[1]:8.1f} us {v[0]:4d}  {k}")
        if e > prev_end:
            prev_end, prev_name = e, n


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1, float(sys.argv[3]) if len(sys.argv) > 3 else 100.0)
