"""profiles/r06_fetch_calibration.txt from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv) over tools/probes/fetch_calib.hip:
known bytes of every streaming kernel / (counter value x 1024) = the factor a counter reading has to be multiplied with for that
access width.    python tools/fetch_calib.py <fetch_dir> <write_dir>"""
import collections
import csv
import glob
import json
import re
import sys

KNOWN = float(1 << 30)


def collect(d, name):
    tot, n = collections.Counter(), collections.Counter()
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                tot[r["Kernel_Name"]] += float(r["Counter_Value"])
                n[r["Kernel_Name"]] += 1
    return {k: tot[k] / n[k] for k in tot}


def main(fetch_dir, write_dir):
    f, w = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    out = {}
    print("# rocprofv3 FETCH_SIZE / WRITE_SIZE against known byte counts (1 GiB streamed per launch, 4 x the Infinity Cache), MI355X")
    print("# factor = known bytes / (counter x 1024): what a reading of that access shape has to be multiplied with")
    print(f"{'kernel':58s} {'known MB':>9s} {'FETCH x1024 MB':>15s} {'factor':>7s} {'WRITE x1024 MB':>15s} {'factor':>7s}")
    for k in sorted(set(f) | set(w)):
        if "read_" not in k and "write_" not in k:
            continue
        known = KNOWN / 2 if "_half" in k else KNOWN
        fb, wb = 1024.0 * f.get(k, 0.0), 1024.0 * w.get(k, 0.0)
        is_read = "read_" in k
        ff = known / fb if is_read and fb else float("nan")
        wf = known / wb if not is_read and wb else float("nan")
        name = re.sub(r"\(.*", "", k.replace("void ", ""))
        out[name] = {"known_bytes": known, "fetch_counter_bytes": fb, "write_counter_bytes": wb,
                     "fetch_factor": None if ff != ff else ff, "write_factor": None if wf != wf else wf}
        print(f"{name:58s} {known / 1e6:9.1f} {fb / 1e6:15.1f} {ff:7.3f} {wb / 1e6:15.1f} {wf:7.3f}")
    rf = [v["fetch_factor"] for k, v in out.items() if v["fetch_factor"] and "_half" not in k]
    wf = [v["write_factor"] for v in out.values() if v["write_factor"]]
    summary = {"fetch_factor_min": min(rf) if rf else None, "fetch_factor_max": max(rf) if rf else None,
               "write_factor_min": min(wf) if wf else None, "write_factor_max": max(wf) if wf else None}
    print("# " + json.dumps(summary))
    json.dump({"kernels": out, "summary": summary}, open("gpurun_out/fetch_calibration.json", "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
