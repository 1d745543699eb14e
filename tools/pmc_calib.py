"""FETCH_SIZE / WRITE_SIZE calibration: counter value / true byte count for each access pattern of tools/probes/fetch_calib.hip.
usage: python tools/pmc_calib.py fetch.csv write.csv        (the csv files of two `rocprofv3 --pmc X --kernel-trace --output-format csv` passes
over tools/probes/fetch_calib_bin).  The counters are in KiB.  A ratio of 0.5 = the counter tallies 128-byte requests at 64 B
(MI355X_MICROARCH.md, HBM section: the case of wide coalesced reads) and has to be doubled; 1.0 = to be used as it is."""
import csv
import re
import sys
from collections import defaultdict

BYTES = float(1 << 30)
NAMES = [("k_read_contig<float __vector(4)>", "read16   (16 B/lane, contiguous)"), ("k_read_rows<float __vector(2)>", "read8r   (8 B/lane, 64-byte row segments)"),
         ("k_read_rows<f32x1>", "read4r   (4 B/lane, 32-byte row segments)"), ("k_read_rows<float __vector(4)>", "read16r  (16 B/lane, 128-byte row segments)"),
         ("k_write_contig<float __vector(4)>", "write16  (16 B/lane, contiguous)"), ("k_write_rows<float __vector(2)>", "write8r  (8 B/lane, 64-byte row segments)"),
         ("k_write_rows<float __vector(4)>", "write16r (16 B/lane, 128-byte row segments)")]


def load(path, counter):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a = agg[re.sub(r"^void ", "", r["Kernel_Name"])]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    print(f"{'pattern':46s} {'FETCH_SIZE/bytes':>17s} {'WRITE_SIZE/bytes':>17s}   (per launch, 1 GiB moved)")
    for key, label in NAMES:
        def ratio(agg):
            hit = [v for k, v in agg.items() if k.startswith(key.split("<")[0]) and key.split("<")[1].rstrip(">") in k]
            return (hit[0][1] * 1024 / hit[0][0] / BYTES) if hit and hit[0][0] else float("nan")
        print(f"{label:46s} {ratio(f):17.3f} {ratio(w):17.3f}")
    print("# raw kernel names:", sorted(f)[:12])


if __name__ == "__main__":
    main()
