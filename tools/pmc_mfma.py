"""MFMA utilisation per kernel from a rocprofv3 --pmc pass (csv) with SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE,
SQ_INSTS_VALU_MFMA_MOPS_BF16 and SQ_BUSY_CU_CYCLES.  usage: python tools/pmc_mfma.py counters.csv
MfmaUtil follows rocprofv3's own derived-metric expression (rocprofv3 -L):
    reduce(SQ_VALU_MFMA_BUSY_CYCLES, sum) / (reduce(GRBM_GUI_ACTIVE, max) * SIMD_NUM) * 100,   SIMD_NUM = 256 CUs x 4.
Cross-check: SQ_INSTS_VALU_MFMA_MOPS_BF16 counts 512-flop units, so MOPS * 512 / duration is the delivered bf16 MFMA rate."""
import csv
import re
import sys
from collections import defaultdict

SIMD_NUM = 1024


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    return re.sub(r"^void ", "", name)[:60]


def main():
    per = defaultdict(lambda: defaultdict(float))
    dur = defaultdict(float)
    n = defaultdict(int)
    seen = set()
    for r in csv.DictReader(open(sys.argv[1])):
        k = short(r["Kernel_Name"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
            n[k] += 1
    print("# SQ counters of this rocprofv3 build are collected on ONE of the 8 XCDs (SQ_BUSY_CU_CYCLES / (GRBM_GUI_ACTIVE * 256) tops out at")
    print("# 12.5 %), so rocprofv3's MfmaUtil expression under-reports by 8x; 'MFMA busy' below = SQ_VALU_MFMA_BUSY_CYCLES /")
    print("# (SQ_BUSY_CU_CYCLES * 4 SIMDs): the share of busy-CU SIMD time with the MFMA pipe occupied, independent of the sampling.")
    print(f"{'calls':>6s} {'avg_us':>9s} {'MfmaUtil%':>10s} {'MFMA busy%':>11s} {'bf16 TFLOP/s':>13s} {'of 2500':>8s}  kernel")
    for k in sorted(per, key=lambda k: -dur[k])[:14]:
        c = per[k]
        gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        util = 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * SIMD_NUM) if gui else float("nan")
        tf = c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) * 512 / (dur[k] * 1e-6) / 1e12 if dur[k] else 0.0
        busy = 100.0 * c.get("SQ_BUSY_CU_CYCLES", 0.0) / (gui * 256) if gui else float("nan")
        cu = c.get("SQ_BUSY_CU_CYCLES", 0.0)
        mb = 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cu * 4) if cu else float("nan")
        print(f"{n[k]:6d} {dur[k] / n[k]:9.1f} {util:10.1f} {mb:11.1f} {tf:13.1f} {100 * tf / 2500:7.1f}%  {k}")


if __name__ == "__main__":
    main()
