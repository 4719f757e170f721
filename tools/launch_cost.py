"""Host cost of hipLaunchKernel from a rocprofv3 --hip-runtime-trace (+ --kernel-trace) csv output directory:
percentiles over the second half of the run, and the launches that blocked (> 100 us) with the kernel they launched.

    python tools/launch_cost.py <output dir>
"""
import collections
import csv
import glob
import sys

d = sys.argv[1]
rows = list(csv.DictReader(open(glob.glob(d + "/*/*hip_api_trace.csv")[0])))
names = {}
kt = glob.glob(d + "/*/*kernel_trace.csv")
if kt:
    for r in csv.DictReader(open(kt[0])):
        names[r["Correlation_Id"]] = r["Kernel_Name"]
L = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Correlation_Id"])
     for r in rows if r["Function"] == "hipLaunchKernel"]
L.sort()
L = L[len(L) // 2:]  # steady state: second half
dur = sorted(x[1] for x in L)
n = len(dur)
print("launches", n, "sum ms", sum(dur) / 1e6)
for p in (10, 50, 90, 99, 99.9):
    print(f"  p{p}: {dur[min(n - 1, int(n * p / 100))] / 1e3:.2f} us")
big = collections.defaultdict(lambda: [0, 0.0])
for _, t, c in L:
    if t > 100e3:
        k = names.get(c, "?")[:110]
        big[k][0] += 1
        big[k][1] += t / 1e6
print("launches that blocked the host for > 100 us:")
for k, (c, t) in sorted(big.items(), key=lambda kv: -kv[1][1]):
    print(f"  {c:4d} x, {t:9.2f} ms in total   {k}")
