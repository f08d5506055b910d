#!/usr/bin/env python3
"""Per-kernel SQ / GRBM counter summary from one rocprofv3 --pmc pass (no trace domains beside --kernel-trace):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
              SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d <dir> -- python bench.py ...
    python tools/pmc_sq.py <dir> out.json

mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) * 256 CUs * 4 SIMDs)   (the gfx94x MfmaUtil formula; ROCm 7.2 has
no gfx950 derived-counter section — MI355X_MICROARCH.md "rocprofv3 PMC slots").  Units, checked on this pool (r02b): rocprofv3 reports
GRBM_GUI_ACTIVE summed over the 8 XCDs (894,498 per rdb_kernel launch = 8 x 111.8 k cycles = 8 x 46.6 us x 2.4 GHz), and
SQ_VALU_MFMA_BUSY_CYCLES is exactly 32 x the number of issued v_mfma_f32_32x32x16_bf16 (29,786,112 = 32 x 1818 MFMAs x 512 workgroups per
dense-block launch, the count derived from the source).  SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles (same guide)."""
import csv, glob, json, os, re, sys


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ksym import hot as short       # the rocprofv3 name (family), the key bench.py's roofline.kernel uses too


def main():
    d, out = sys.argv[1:3]
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            a = acc.setdefault(k, {})
            c = a.setdefault(r["Counter_Name"], [0.0, 0])
            c[0] += float(r["Counter_Value"])
            c[1] += 1
    res = {}
    for k, a in sorted(acc.items()):
        m = {n: v[0] / v[1] for n, v in a.items()}
        row = {"launches": max(v[1] for v in a.values()), "per_launch": m}
        gui = m.get("GRBM_GUI_ACTIVE")
        gui = gui / 8.0 if gui else gui          # summed over the 8 XCDs (see the module docstring)
        if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
            row["mfma_util"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 256 * 4)
        if gui and "SQ_BUSY_CU_CYCLES" in m:
            row["cu_busy_frac_raw"] = m["SQ_BUSY_CU_CYCLES"] / (gui * 256)
        if m.get("SQ_LDS_IDX_ACTIVE"):
            row["lds_bank_conflict_frac"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
        if m.get("SQ_WAVE_CYCLES"):
            row["wait_any_frac"] = m.get("SQ_WAIT_ANY", 0.0) / m["SQ_WAVE_CYCLES"]
            row["wait_inst_lds_frac"] = m.get("SQ_WAIT_INST_LDS", 0.0) / m["SQ_WAVE_CYCLES"]
        res[k] = row
    json.dump(res, open(out, "w"), indent=1)
    for k, r in res.items():
        print(k, {x: (round(v, 4) if isinstance(v, float) else v) for x, v in r.items() if x != "per_launch"})


if __name__ == "__main__":
    main()
