"""Per-kernel statistics and launch-to-launch gaps from a rocprofv3 (rocpd sqlite) kernel trace.
    python tools/rocpd_summary.py results.db [name filter] [--timeline N]
Prints calls, total / average / min / max duration per kernel, and for the filtered kernels the idle time between
consecutive dispatches on the GPU (end of one to start of the next)."""
import sqlite3
import sys


def main():
    path = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
    n_tl = int(sys.argv[sys.argv.index("--timeline") + 1]) if "--timeline" in sys.argv else 0
    db = sqlite3.connect(path)
    c = db.cursor()
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    stats = {}
    for name, s, e in rows:
        short = name.split("(")[0].replace("void gcmi::", "").replace("gcmi::", "")
        st = stats.setdefault(short, [0, 0, 1 << 62, 0])
        d = e - s
        st[0] += 1
        st[1] += d
        st[2] = min(st[2], d)
        st[3] = max(st[3], d)
    total = sum(v[1] for v in stats.values())
    print("%-64s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for k, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        print("%-64s %8d %12.1f %10.2f %10.2f %10.2f %6.1f" % (k[:64], v[0], v[1] / 1e3, v[1] / v[0] / 1e3, v[2] / 1e3,
                                                              v[3] / 1e3, 100.0 * v[1] / max(total, 1)))
    sel = [(n, s, e) for n, s, e in rows if flt in n]
    if len(sel) > 1:
        gaps = [sel[i + 1][1] - sel[i][2] for i in range(len(sel) - 1)]
        gaps_sorted = sorted(gaps)
        busy = sum(e - s for _, s, e in sel)
        span = sel[-1][2] - sel[0][1]
        print("\n%d dispatches matching %r: span %.1f us, busy %.1f us (%.1f %%), median gap %.2f us, p90 %.2f us, max %.1f us"
              % (len(sel), flt, span / 1e3, busy / 1e3, 100.0 * busy / span, gaps_sorted[len(gaps) // 2] / 1e3,
                 gaps_sorted[int(len(gaps) * 0.9)] / 1e3, gaps_sorted[-1] / 1e3))
    if "--gaps" in sys.argv and len(sel) > 1:
        k = int(sys.argv[sys.argv.index("--gaps") + 1])
        big = sorted(range(len(sel) - 1), key=lambda i: -(sel[i + 1][1] - sel[i][2]))[:k]
        total_gap = sum(max(0, sel[i + 1][1] - sel[i][2]) for i in range(len(sel) - 1))
        print("\nidle between matching dispatches: %.1f us in all; the %d largest gaps (us, at us since first, after -> before):"
              % (total_gap / 1e3, k))
        for i in sorted(big):
            print("  %9.1f  @%10.1f  %s -> %s" % ((sel[i + 1][1] - sel[i][2]) / 1e3, (sel[i][2] - sel[0][1]) / 1e3,
                                                 sel[i][0].split("(")[0][-40:], sel[i + 1][0].split("(")[0][-40:]))
    if n_tl:
        mid = len(sel) // 2
        t0 = sel[mid][1]
        print("\ntimeline of %d dispatches from the middle of the run (us since the first):" % n_tl)
        for n, s, e in sel[mid:mid + n_tl]:
            print("  %9.2f  +%7.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n.split("(")[0].replace("void gcmi::", "")[:70]))


if __name__ == "__main__":
    main()
