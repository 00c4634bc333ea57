"""Per-kernel statistics (calls, total / average / min / max duration, share) of a rocprofv3 --kernel-trace run,
read from the rocpd database it writes -- the table `rocprofv3 --stats` prints.

    python tools/kernel_stats.py gpurun_out/prof > profiles/rNN_kernel_stats.csv"""
import collections
import glob
import os
import sqlite3
import sys


def main():
    dbs = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True))
    if not dbs:
        raise SystemExit("no .db under %s" % sys.argv[1])
    con = sqlite3.connect(dbs[-1])
    cols = [r[1] for r in con.execute("PRAGMA table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    by = collections.defaultdict(list)
    for name, s, e in con.execute("SELECT %s, start, end FROM kernels" % name_col):
        by[name.split("(")[0]].append(e - s)
    total = sum(sum(v) for v in by.values())
    print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
    for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print('"%s",%d,%d,%.1f,%.2f,%d,%d' % (name, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / total, min(v), max(v)))


if __name__ == "__main__":
    main()
