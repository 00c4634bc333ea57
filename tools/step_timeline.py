"""One training step as a kernel timeline, from a rocprofv3 --kernel-trace output directory.

    rocprofv3 --kernel-trace -d gpurun_out/prof -- python3 bench.py --profile-only --steps 3 --warmup 2
    python tools/step_timeline.py gpurun_out/prof > profiles/rNN_step_timeline.txt

Prints the kernels between the last two optimiser launches (start, gap to the previous kernel, duration)
and per-kernel totals.  Reads the rocpd sqlite database rocprofv3 writes (views `kernels`)."""
import collections
import glob
import os
import sqlite3
import sys


def main():
    root = sys.argv[1]
    dbs = sorted(glob.glob(os.path.join(root, "**", "*.db"), recursive=True))
    if not dbs:
        raise SystemExit("no .db under %s" % root)
    con = sqlite3.connect(dbs[-1])
    cols = [r[1] for r in con.execute("PRAGMA table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = con.execute("SELECT %s, start, end FROM kernels ORDER BY start" % name_col).fetchall()
    marks = [i for i, r in enumerate(rows) if r[0].startswith("adam_kernel") or "adam_kernel" in r[0][:40]]
    if len(marks) < 2:
        raise SystemExit("fewer than two optimiser launches in the trace")
    step = rows[marks[-2] + 1:marks[-1] + 1]
    t0 = step[0][1]
    print("one training step, rocprofv3 --kernel-trace; columns: start us, gap us, duration us, kernel")
    prev_end = t0
    tot = collections.OrderedDict()
    for name, s, e in step:
        short = name.split("(")[0][-60:]
        print("%9.1f gap %6.1f dur %8.1f  %s" % ((s - t0) / 1e3, max(0, s - prev_end) / 1e3, (e - s) / 1e3, short))
        prev_end = e
        k = short.split("<")[0].split("::")[-1].strip()
        c = tot.setdefault(k, [0, 0.0])
        c[0] += 1
        c[1] += (e - s) / 1e3
    print("\nstep span %.1f us; per kernel (launches, total us):" % ((step[-1][2] - t0) / 1e3))
    for k, (n, us) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
        print("  %-34s %3d %9.1f" % (k, n, us))


if __name__ == "__main__":
    main()
