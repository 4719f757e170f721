"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output).

MI355X_MICROARCH.md, HBM section: FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B as rocprofv3 reports them;
on gfx950 FETCH_SIZE counts 128-B read requests as 64 B, so reads are doubled.  The guide calibrates that factor for
16-byte-per-lane streaming reads only; round 6 calibrated every access shape this library uses on known byte counts
(tools/probes/fetch_calib.hip, 1 GiB streamed per launch = 4 x the Infinity Cache; profiles/r06_fetch_calibration.txt):
FETCH_SIZE x 1024 x 2.000 for 2 / 4 / 8 / 16 bytes per lane, plain global loads and buffer loads, whole lines and 64-byte
runs alike; WRITE_SIZE x 1024 x 1.000 for 2 / 4 / 8 / 16-byte stores.  So ONE read factor and no write factor, for every
kernel.  Output: JSON {kernel: {launches, fetch_bytes_per_launch, write_bytes_per_launch}}."""
import collections
import csv
import glob
import json
import sys


def collect(d, name):
    tot, n = collections.Counter(), collections.Counter()
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != name:
                continue
            k = r["Kernel_Name"]
            tot[k] += float(r["Counter_Value"])
            n[k] += 1
    return tot, n


def main(fetch_dir, write_dir, forwards=0):
    ft, fn = collect(fetch_dir, "FETCH_SIZE")
    wt, wn = collect(write_dir, "WRITE_SIZE")
    out = {}
    for k in ft:
        if "sty::" not in k:
            continue
        out[k] = {"launches": fn[k], "fetch_bytes_per_launch": 2.0 * 1024.0 * ft[k] / fn[k],
                  "write_bytes_per_launch": 1024.0 * wt.get(k, 0.0) / max(1, wn.get(k, 0)),
                  "note": "FETCH_SIZE x1024 x2.000, WRITE_SIZE x1024 x1.000 (profiles/r06_fetch_calibration.txt: every access width)"}
    if forwards:
        # the whole command: every kernel of every pass (steps + warm-up + the profiled step; preparation kernels once)
        fb = sum(v["fetch_bytes_per_launch"] * v["launches"] for v in out.values())
        wb = sum(v["write_bytes_per_launch"] * v["launches"] for v in out.values())
        out["_total"] = {"passes": forwards, "fetch_bytes_per_pass": fb / forwards, "write_bytes_per_pass": wb / forwards,
                         "note": "sum over all kernels of the profiled command / number of passes of the workload in it"}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0)
