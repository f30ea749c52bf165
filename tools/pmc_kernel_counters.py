"""Per-kernel sums of arbitrary PMC counters from rocprofv3 --pmc passes (one results .db per pass).  usage: python tools/pmc_kernel_counters.py <substring> <db> [<db> ...]"""
import sqlite3
import sys

sub = sys.argv[1]
tot = {}
for path in sys.argv[2:]:
    db = sqlite3.connect(path)
    for name, counter, n, s in db.execute("select kernel_name, counter_name, count(*), sum(value) from counters_collection group by kernel_name, counter_name"):
        if sub in name:
            tot.setdefault(name[:70], {})[counter] = (n, s)
for k, v in tot.items():
    print(k)
    for c, (n, s) in sorted(v.items()):
        print(f"   {c:32s} launches {n:5d}  total {s:.4g}  per launch {s / max(1, n):.4g}")
