"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output).
usage: python tools/pmc_stats.py fetch.csv write.csv [out.json [git_sha]]
FETCH_SIZE / WRITE_SIZE are reported in KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts
128-byte read requests at 64 B, i.e. reports half of the bytes of wide coalesced reads -> doubled here.  WRITE_SIZE is
used as reported (uncalibrated per the guide; the elementwise kernels below serve as the calibration points)."""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:70]


def load(path, counter):
    agg = defaultdict(lambda: [0, 0.0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        a = agg[short(r["Kernel_Name"])]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
        a[2] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    return agg


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    print(f"{'calls':>6s} {'read_MB/launch':>15s} {'write_MB/launch':>16s} {'total_MB/launch':>16s} {'avg_us(pmc run)':>16s}  kernel")
    rows = []
    for k in f:
        n = f[k][0]
        rd = 2.0 * f[k][1] * 1024 / n / 1e6
        wr = (w[k][1] * 1024 / w[k][0] / 1e6) if k in w and w[k][0] else float("nan")
        rows.append((rd * n + (wr * n if wr == wr else 0), n, rd, wr, f[k][2] / n, k))
    for tot, n, rd, wr, us, k in sorted(rows, reverse=True)[:40]:
        print(f"{n:6d} {rd:15.1f} {wr:16.1f} {rd + wr:16.1f} {us:16.1f}  {k}")
    print(f"# total HBM traffic of the run: {sum(r[0] for r in rows) / 1e3:.1f} GB")
    if len(sys.argv) > 3:  # machine-readable record of the dominant kernel family for bench.py's roofline.traffic
        import json
        fam = [r for r in rows if r[5].startswith("gemm_nt5_kernel")]
        n = sum(r[1] for r in fam)
        rec = {"kernel": "gemm_nt5_kernel (all epilogues)", "launches": n,
               "read_bytes_per_launch": sum(r[2] * r[1] for r in fam) * 1e6 / n,
               "write_bytes_per_launch": sum(r[3] * r[1] for r in fam) * 1e6 / n,
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 1 --warmup 1`; "
                         "FETCH_SIZE doubled (gfx950 counts 128-byte requests at 64 B), KiB units"}
        rec["bytes_per_launch"] = rec["read_bytes_per_launch"] + rec["write_bytes_per_launch"]
        # which code these passes saw: bench.py compares csrc_sha16 with the kernel sources it runs and marks a stale quote on its line
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from bench import code_identity
        rec["git_sha"], rec["csrc_sha16"] = code_identity()
        if len(sys.argv) > 4:
            rec["git_sha"] = sys.argv[4]
        # per kernel instantiation, keyed as bench.py names them ("gemm_nt5_kernel<2,false,40>", "gemm_tn5_kernel<true>")
        rec["by_kernel"] = {}
        for tot, n, rd, wr, us, k in rows:
            if k.startswith("gemm_"):
                key = k.split("(")[0].replace(", ", ",")
                # the static instantiations (last template argument RESCUE = false, round 6) keep the names bench.py has used since round 1
                key = re.sub(r"^(gemm_nt5_kernel<\d+,false,\d+),false>$", r"\1>", key)
                key = re.sub(r"^(gemm_tn5_kernel<(?:true|false)),false>$", r"\1>", key)
                rec["by_kernel"][key] = {"launches": n, "read_bytes_per_launch": rd * 1e6, "write_bytes_per_launch": (wr if wr == wr else 0.0) * 1e6,
                                         "bytes_per_launch": (rd + (wr if wr == wr else 0.0)) * 1e6}
        json.dump(rec, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
