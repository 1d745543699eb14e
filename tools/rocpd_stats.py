"""Summarises a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share.
usage: python tools/rocpd_stats.py results.db [skip_first_fraction]   (developer tool; output is committed under profiles/)"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {namecol}, start, end from kernels order by start").fetchall()
    agg = {}
    for n, s, e in rows:
        a = agg.setdefault(short(n), [0, 0])
        a[0] += 1
        a[1] += e - s
    tot = sum(a[1] for a in agg.values())
    span = rows[-1][2] - rows[0][1]
    print(f"# {len(rows)} dispatches, kernel time {tot / 1e6:.2f} ms, span first->last {span / 1e6:.2f} ms")
    print(f"{'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}  kernel")
    for n, (k, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:7d} {t / 1e6:10.3f} {t / k / 1e3:10.2f} {100.0 * t / tot:6.2f}  {n}")
    # how many kernels share the chip, over the busiest contiguous stretch (the timed steps: from the first dispatch of the second half
    # of the trace to the last one): time with 0 / 1 / 2 / 3+ kernels in flight
    half = rows[len(rows) // 2:]
    ev = sorted([(s, 1) for _, s, _ in half] + [(e, -1) for _, _, e in half])
    depth, last, hist = 0, ev[0][0], {}
    for t, d in ev:
        hist[min(depth, 3)] = hist.get(min(depth, 3), 0) + (t - last)
        depth, last = depth + d, t
    whole = sum(hist.values())
    print("# second half of the trace (%.1f ms): " % (whole / 1e6) + ", ".join(
        f"{'3+' if k == 3 else k} kernel{'s' if k != 1 else ''} in flight {100.0 * v / whole:.1f} %" for k, v in sorted(hist.items())))


if __name__ == "__main__":
    main()
