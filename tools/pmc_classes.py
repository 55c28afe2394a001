"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel CLASS -> JSON (per-launch means).

    python tools/pmc_classes.py out.json NAME=path/to/counter_collection.csv [NAME=...]

Kernel classes: tapgemm (all instantiations), splitk_reduce, flash, temporal, groupnorm (stats/finalize/apply/fused),
layernorm, other.  Counters are summed over a dispatch's rows (one row per XCD/SE instance) and averaged per launch."""
import collections, csv, json, re, sys

CLASSES = [("tapgemm", r"tapgemm_kernel|panel_kernel"), ("splitk_reduce", r"splitk_reduce"), ("flash_attention", r"flash_kernel"),
           ("temporal_attention", r"temporal_kernel"), ("groupnorm", r"gn_(stats|finalize|finalize_cs|apply|fused|regs)"),
           ("layernorm", r"layernorm"), ("linear_f32", r"linear_f32")]


def cls(name):
    for c, pat in CLASSES:
        if re.search(pat, name):
            return c
    return "other"


def main(out, *specs):
    res = {}
    for spec in specs:
        tag, path = spec.split("=", 1)
        per = collections.defaultdict(lambda: collections.defaultdict(float))
        disp = collections.defaultdict(set)
        for r in csv.DictReader(open(path)):
            c = cls(r["Kernel_Name"])
            per[c][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[c].add(r["Dispatch_Id"])
        for c, v in per.items():
            e = res.setdefault(c, {})
            e["launches"] = max(e.get("launches", 0), len(disp[c]))
            for k, x in v.items():
                e[k + "_per_launch"] = x / len(disp[c])
                e[k + "_total"] = x
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1)[:3000])


if __name__ == "__main__":
    main(sys.argv[1], *sys.argv[2:])
