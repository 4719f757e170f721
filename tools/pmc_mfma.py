"""Per-kernel matrix-core utilisation from one rocprofv3 --pmc pass (csv output, with --kernel-trace) of the serialised c3 step.

    SQ_VALU_MFMA_BUSY_CYCLES  cycles a SIMD's matrix pipe was busy, summed over all SIMDs of the chip (MI355X_MICROARCH.md: 32 per
                              v_mfma_f32_32x32x16_bf16, 64 per v_mfma_f32_32x32x2_f32)
    duration                  End - Start of the same dispatch in the kernel trace of the SAME run (kernels run one at a time
                              under counter collection)
    utilisation = MFMA_BUSY / (duration x 2.4 GHz x 1024 SIMDs): the fraction of the chip's matrix-pipe cycles at the peak clock
    the launch used -- the counter-side twin of (achieved TFLOP/s / peak) in bench.py's table, independent of the flop model
    (it counts every MFMA issued, padding rows / columns and halo columns included, so it sits above the algorithmic fraction).
    GRBM_GUI_ACTIVE is reported as collected (it sums the eight XCDs' counters: / 8 / duration = the effective clock).
Output: one line per kernel (launch averages), sorted by total matrix-pipe cycles."""
import collections
import csv
import glob
import sys

SIMDS = 256 * 4
PEAK_HZ = 2.4e9


def main(d):
    dur = {}
    for f in glob.glob(d + "/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    tot = collections.defaultdict(lambda: collections.Counter())
    n = collections.Counter()
    seen = set()
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "sty::" not in k:
                continue
            tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
            key = (k, r["Dispatch_Id"])
            if key not in seen:
                seen.add(key)
                n[k] += 1
                tot[k]["_dur"] += dur.get(r["Dispatch_Id"], 0.0)
    rows = []
    for k, c in tot.items():
        busy, t = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), c.get("_dur", 0.0)
        if busy <= 0 or t <= 0:
            continue
        rows.append((busy, k, n[k], 1e6 * t / n[k], busy / n[k], busy / (t * PEAK_HZ * SIMDS),
                     c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / t / 1e9, c.get("SQ_WAVES", 0.0) / n[k]))
    rows.sort(reverse=True)
    print("# matrix-pipe utilisation per kernel, every kernel alone on the chip (counter collection serialises the dispatches):")
    print("# util = SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs); clk = GRBM_GUI_ACTIVE / 8 XCDs / duration (GHz)")
    print(f"{'launches':>8} {'avg_us':>9} {'mfma_busy':>14} {'util':>7} {'clk':>5} {'waves':>8}  kernel")
    for busy, k, ln, us, b1, util, clk, waves in rows:
        print(f"{ln:8d} {us:9.1f} {b1:14.0f} {util:7.3f} {clk:5.2f} {waves:8.0f}  {k[:118]}")


if __name__ == "__main__":
    main(sys.argv[1])
