#!/usr/bin/env python3
"""LDS bank-conflict attribution: per-launch SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (/ SQ_INSTS_LDS) of one kernel family over the
rocprofv3 --pmc output directories of several probe builds (tools/build_x3r_lds_variants.sh):  lds_attr.py <kernel substring> name=dir ..."""
import csv, glob, os, sys


def collect(d, sub):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if sub not in r["Kernel_Name"]:
                continue
            c = acc.setdefault(r["Counter_Name"], [0.0, 0])
            c[0] += float(r["Counter_Value"]); c[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}, max([v[1] for v in acc.values()] or [0])


def main():
    sub = sys.argv[1]
    base = None
    print(f"kernel family *{sub}*: per launch (all workgroups): LDS-array cycles, bank-conflict cycles, conflict / active, LDS instructions")
    for spec in sys.argv[2:]:
        name, d = spec.split("=", 1)
        m, n = collect(d, sub)
        act, con, ins = m.get("SQ_LDS_IDX_ACTIVE", 0.0), m.get("SQ_LDS_BANK_CONFLICT", 0.0), m.get("SQ_INSTS_LDS", float("nan"))
        if base is None:
            base = (act, con)
        print(f"  {name:10s} launches {n:4d}  active {act:12.0f}  conflict {con:11.0f}  frac {con / act if act else 0:6.3f}  insts {ins:11.0f}"
              f"   vs base: active {act - base[0]:+12.0f}  conflict {con - base[1]:+11.0f}")


if __name__ == "__main__":
    main()
