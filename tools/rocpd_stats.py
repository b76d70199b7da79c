#!/usr/bin/env python3
"""Per-kernel summary (calls, total/avg duration, share) from a rocprofv3 rocpd sqlite database
(`rocprofv3 --kernel-trace --stats -d DIR -o NAME` writes NAME_results.db).  Prints CSV."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void bs::", "").replace("(anonymous namespace)::", "")
    return name[:150]


def main(path, skip_first=0):
    db = sqlite3.connect(path)
    rows = db.execute("select name, start, end from kernels order by start").fetchall()
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0])
        a[0] += 1
        a[1] += e - s
    tot = sum(a[1] for a in agg.values())
    print("kernel,calls,total_ms,avg_us,percent")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"\"{k}\",{n},{t / 1e6:.3f},{t / n / 1e3:.2f},{100.0 * t / tot:.2f}")
    print(f"\"TOTAL\",{sum(a[0] for a in agg.values())},{tot / 1e6:.3f},,100")


if __name__ == "__main__":
    main(sys.argv[1])
