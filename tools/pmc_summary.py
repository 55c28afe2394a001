"""Aggregate rocprofv3 --pmc counter_collection.csv per (kernel short name, grid) -> per-launch means."""
import collections, csv, re, sys
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    m = re.match(r"(?:void )?([A-Za-z0-9_:]+(?:<[A-Za-z0-9_, ]+>)?)", n)
    return (m.group(1) if m else n)[:50]
def main(path, filt=""):
    rows = list(csv.DictReader(open(path)))
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
    for r in rows:
        k = (short(r["Kernel_Name"]), r["Grid_Size"])
        if filt and filt not in k[0]: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        c = len(n[k])
        print(k, "launches", c)
        for name, x in sorted(v.items()):
            print(f"    {name:28s} {x / c:16.1f}")
if __name__ == "__main__":
    main(*sys.argv[1:])
