"""Cadence of the optimiser launches in a rocprofv3 --kernel-trace of a fit() run: the time between consecutive
adam_kernel launches (= one step + whatever the batch pipeline made it wait), and where the collation kernels ran.

    rocprofv3 --kernel-trace -d gpurun_out/prof_fit -- python3 tools/fit_throughput.py --batches 65536 --epochs 6
    python tools/fit_cadence.py gpurun_out/prof_fit"""
import glob
import os
import sqlite3
import sys

import numpy as np


def main():
    dbs = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*.db"), recursive=True))
    con = sqlite3.connect(dbs[-1])
    cols = [r[1] for r in con.execute("PRAGMA table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = con.execute("SELECT %s, start, end FROM kernels ORDER BY start" % name_col).fetchall()
    adam = np.array([r[1] for r in rows if "adam_kernel" in r[0][:60]], dtype=np.float64)
    d = np.diff(adam) / 1e3
    print("optimiser launches: %d" % adam.size)
    runs = np.split(d, np.where(d > 50000)[0] + 1)  # fits are separated by host work
    for i, r in enumerate(runs):
        r = r[r <= 50000]
        if r.size < 3:
            continue
        print("fit %d: %d steps, step-to-step us: median %.0f  mean %.0f  min %.0f  max %.0f  first three %s" %
              (i, r.size, np.median(r), r.mean(), r.min(), r.max(), np.round(r[:3]).tolist()))
        slow = [(int(k), int(v)) for k, v in enumerate(r) if v > 1.25 * np.median(r)]
        print("   steps more than 25 %% above the median (index, us): %s" % slow)
    busy = sum(e - s for _, s, e in rows) / 1e3
    print("kernel time total %.0f us over a span of %.0f us" % (busy, (rows[-1][2] - rows[0][1]) / 1e3))


if __name__ == "__main__":
    main()
