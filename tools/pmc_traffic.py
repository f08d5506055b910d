#!/usr/bin/env python3
"""HBM-side traffic per launch of the hot kernels from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE cannot share
a pass on gfx950: TCC has 4 slots, they cost 3 + 2), written to profiles/traffic.json for bench.py's `roofline.traffic`.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline
    python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write profiles/traffic.json

Units and corrections (MI355X_MICROARCH.md, HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports exactly
half of the bytes of wide coalesced (16 B/lane) reads — every kernel here reads that way — so it is doubled; WRITE_SIZE
is taken as is (uncalibrated)."""
import csv, glob, json, os, sys


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ksym import hot as symbol      # the rocprofv3 name (family), the key bench.py's roofline.kernel uses too


def collect(d, counter):
    tot, cnt = {}, {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter: continue
            s = symbol(r["Kernel_Name"])
            if s is None: continue
            tot[s] = tot.get(s, 0.0) + float(r["Counter_Value"])
            cnt[s] = cnt.get(s, 0) + 1
    return {s: tot[s] / cnt[s] for s in tot}, cnt


def main():
    fd, wd, out = sys.argv[1:4]
    fetch, nf = collect(fd, "FETCH_SIZE")
    write, nw = collect(wd, "WRITE_SIZE")
    res = {}
    for s in sorted(set(fetch) | set(write)):
        rd = 2.0 * fetch.get(s, 0.0) * 1024.0          # gfx950: FETCH_SIZE = 1/2 of wide coalesced reads
        wr = write.get(s, 0.0) * 1024.0
        res[s] = {"hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "bytes_per_launch": rd + wr,
                  "launches_sampled": [nf.get(s, 0), nw.get(s, 0)],
                  "note": "FETCH_SIZE x2 (gfx950 wide-read correction) + WRITE_SIZE, KiB -> bytes, separate --pmc passes"}
    # which build / configuration the counters belong to: bench.py reports roofline.traffic only for a matching run
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from satlas_super_resolution_amd import build as bld
    import datetime
    res["_meta"] = {"source_hash": bld.source_hash(), "batch": int(os.environ.get("SSR_PMC_BATCH", "32")),
                    "frames": int(os.environ.get("SSR_PMC_FRAMES", "8")), "dtype": os.environ.get("SSR_PMC_DTYPE", "bf16"),
                    "collected": datetime.date.today().isoformat()}
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
