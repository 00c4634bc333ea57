"""Per-kernel averages of whatever counters a rocprofv3 --pmc pass collected.

    rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d gpurun_out/pmc_a -- python3 tools/kbench.py --only seg_gemm --iters 2
    python3 tools/pmc_kernels.py gpurun_out/pmc_a [name filter]

One line per (kernel, template args): launches, mean duration, mean of every counter (summed over the
counter's instances / dimensions per dispatch)."""
import collections
import glob
import os
import re
import sqlite3
import sys


def main():
    d = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    dbs = glob.glob(os.path.join(d, "**", "*.db"), recursive=True)
    if not dbs:
        raise SystemExit("no rocprofv3 database under " + d)
    c = sqlite3.connect(dbs[0]).cursor()
    cols = [x[0] for x in c.execute("select * from counters_collection limit 1").description]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    rows = c.execute("select dispatch_id, %s, counter_name, value, start, end from counters_collection" % name_col)
    per = {}
    for did, name, cname, val, s, e in rows:
        rec = per.setdefault(did, [re.sub(r"\(.*", "", name)[-70:], (e - s) / 1e3, collections.defaultdict(float)])
        rec[2][cname] += float(val)
    by = collections.defaultdict(list)
    for name, dur, cs in per.values():
        if flt in name:
            by[name].append((dur, cs))
    for name, lst in sorted(by.items()):
        mx = max(x[0] for x in lst)
        big = [x for x in lst if x[0] > 0.5 * mx]
        line = "%-62s n=%d dur=%.1fus" % (name, len(big), sum(x[0] for x in big) / len(big))
        for cn in sorted(big[0][1]):
            line += "  %s=%.3g" % (cn, sum(x[1][cn] for x in big) / len(big))
        print(line)


if __name__ == "__main__":
    main()
