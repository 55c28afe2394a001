"""Summarise a rocprofv3 --kernel-trace --stats CSV (kernel_stats.csv) into a short table:
kernel short-name, calls, total ms, avg us, % — template arguments of torch kernels are cut."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([A-Za-z0-9_:]+(?:<[A-Za-z0-9_, ]+>)?)", name)
    s = m.group(1) if m else name
    return s[:70]


def main(path, out=None):
    rows = list(csv.DictReader(open(path)))
    agg = {}
    for r in rows:
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
    tot = sum(v[1] for v in agg.values())
    lines = ["kernel,calls,total_ms,avg_us,percent"]
    for k, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append(f"{k},{c},{ns / 1e6:.3f},{ns / c / 1e3:.2f},{100 * ns / tot:.2f}")
    txt = "\n".join(lines)
    if out:
        open(out, "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
