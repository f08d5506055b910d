#!/usr/bin/env python3
"""Recompute bench.py's roofline fraction from the committed rocprofv3 summary, no decoder ring:

    python tools/roofline_check.py profiles/r06x_bench.json profiles/r06x_kernel_stats_serial.csv [steps_profiled] > profiles/r06x_roofline_check.txt

For roofline.kernel (and every other kernel family of kernel_time_breakdown_ms): calls per step and average duration from the serial
kernel trace (SSR_OVERLAP_D=0 SSR_G_SPLIT=0: every launch has the chip to itself, as in the bench's instrumented step), FLOPs per launch
from the bench line, and frac = flops_per_launch / avg_ns / peak - beside the bench's own event-timed figure."""
import csv, json, os, sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from ksym import family


def main():
    bench, stats = sys.argv[1], sys.argv[2]
    line = [ln for ln in open(bench) if ln.startswith("{")][-1]
    d = json.loads(line)
    roof = d["roofline"]
    rows = {}
    for r in csv.DictReader(open(stats)):
        f = family(r["Name"])
        a = rows.setdefault(f, [0, 0.0])
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
    # steps inside the profiled run: the trace's call count of the roofline kernel / the bench's launches per step
    k = roof["kernel"]
    if k not in rows:
        print(f"{k}: not in {stats} (families present: {sorted(rows)[:12]} ...)")
        sys.exit(1)
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else rows[k][0] / roof["launches_per_step"]
    peak = roof["peak"]
    print(f"bench line: {bench}  (dtype {d['dtype']}, {d['ms_per_step']:.3f} ms per step, {d['value']:.1f} img/s)")
    print(f"kernel trace: {stats}  ({steps:.2f} steps profiled, derived from the roofline kernel's call count)")
    print(f"roofline.kernel = {k}   rocprof symbols: {roof.get('rocprof_symbols')}")
    calls, tot = rows[k]
    avg_ns = tot / calls
    fl = roof["flops_per_launch"]
    print(f"  flops per launch (algorithmic, bench)     {fl / 1e9:10.4f} GFLOP")
    print(f"  calls per step (trace)                    {calls / steps:10.2f}   (bench: {roof['launches_per_step']})")
    print(f"  average duration (trace)                  {avg_ns / 1e3:10.3f} us   (bench, HIP events: {roof['avg_launch_us']:.3f} us)")
    print(f"  achieved (trace)                          {fl / avg_ns / 1e3:10.2f} TFLOP/s (bench: {roof['achieved']:.2f})")
    print(f"  frac of peak {peak:.1f} TFLOP/s (trace)        {fl / avg_ns / 1e3 / peak:10.4f}   (bench: {roof['frac']:.4f})")
    print(f"  per step (trace)                          {tot / steps / 1e6:10.3f} ms")
    print("all kernel families of the trace, per step:")
    for f, (c, t) in sorted(rows.items(), key=lambda kv: -kv[1][1])[:16]:
        print(f"  {f[:70]:70s} {c / steps:8.1f} calls  {t / c / 1e3:10.2f} us avg  {t / steps / 1e6:8.3f} ms")


if __name__ == "__main__":
    main()
