"""Host issue time against GPU start time, kernel by kernel, around the start of one training step
(rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv).  A kernel that starts a few microseconds after the
host issued it was waiting for the HOST; one that starts much later was waiting for the GPU (dependency or dispatcher).

    python tools/issue_delay.py <output dir> [steps_from_end] [rows]

Caveat (measured): with the HIP API trace on, a launch costs the host ~60 us instead of ~3 us, so the traced run is
host-bound where the untraced one is not; read the table for ordering and dependencies, not for absolute slack.
"""
import csv
import glob
import sys

d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 6
nrows = int(sys.argv[3]) if len(sys.argv) > 3 else 60
api = {}
for r in csv.DictReader(open(glob.glob(d + "/*/*hip_api_trace.csv")[0])):
    if r["Function"] in ("hipLaunchKernel", "hipExtLaunchKernel", "hipModuleLaunchKernel"):
        api[r["Correlation_Id"]] = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
K = []
for r in csv.DictReader(open(glob.glob(d + "/*/*kernel_trace.csv")[0])):
    K.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"], r["Correlation_Id"]))
K.sort()
ends = [i for i, k in enumerate(K) if "adamw_kernel" in k[3]]
groups = []
for i in ends:
    if groups and K[i][1] - K[groups[-1][-1]][1] < 2e6:
        groups[-1].append(i)
    else:
        groups.append([i])
g = groups[-1 - skip]
i0 = g[0] - 15
t0 = K[g[-1]][1]
print("t = 0: end of the previous step's last adamw_kernel; times in us")
print(f"{'issued':>10} {'gpu start':>10} {'gpu end':>10} {'wait':>9}  queue  kernel")
for s, e, q, n, c in K[i0:i0 + nrows]:
    a = api.get(c)
    iss = (a[1] - t0) / 1e3 if a else float("nan")
    name = n.replace("sty::", "").replace("void ", "")[:70]
    print(f"{iss:10.1f} {(s - t0) / 1e3:10.1f} {(e - t0) / 1e3:10.1f} {(s - t0) / 1e3 - iss:9.1f}  {q:>5}  {name}")
