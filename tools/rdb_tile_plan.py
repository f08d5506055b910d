#!/usr/bin/env python3
"""Budget calculator for the fused dense-block kernel (csrc/rdb_fwd.hip): LDS bytes, issued MFMAs per pixel (halo recompute +
M-tile padding), per-wave critical path, for a core tile of TH x TW pixels — the numbers behind DESIGN.md 8.1.  Pure arithmetic,
no GPU.   python tools/rdb_tile_plan.py [TH TW ...]"""
import math
import sys

NF, GC = 64, 32
ALG_MFMA_PER_PX = 2 * 9 * (64 * 32 + 96 * 32 + 128 * 32 + 160 * 32 + 192 * 64) / 32768.0     # 32x32x16 bf16 MFMA = 32768 FLOP


def plan(th, tw, row_bytes=80, ring_stages=2, mfma_waves=4, ring_bytes=None):
    regs = [(th + 2 * (5 - k), tw + 2 * (5 - k)) for k in range(0, 5)]        # x (halo 5), x1..x4 (halo 4..1)
    px = [h * w for h, w in regs]
    # the kernel keeps ONE row pitch (the width of the input halo region) for every slice: the conflict-free lane -> pixel map
    # needs all pitches congruent mod 16, and a tap shift must be the same immediate for every lane
    pitch = tw + 10
    rows = [(h - 1) * pitch + w for h, w in regs]
    lds_act = rows[0] * 2 * row_bytes + sum(rows[1:]) * row_bytes           # x: two 32-channel planes
    ring = ring_stages * 288 * 64 if ring_bytes is None else ring_bytes
    lds = lds_act + ring + 768 + 64 + row_bytes
    # issued MFMAs: conv k (k = 1..4) over region k in 32-pixel M-tiles, k+1 slabs of 18 k-steps; conv5: core, 2 N-tiles, 12 slabs x 9
    tiles = [math.ceil(p / 32) for p in px[1:]] + [math.ceil(th * tw / 32)]
    issued = sum(tiles[k - 1] * (k + 1) * 18 for k in range(1, 5)) + tiles[4] * 2 * 12 * 9
    # slowest wave: every slab is consumed in lock step (ring), so a stage's pace is ceil(tiles / waves) tile-times
    crit = sum(math.ceil(tiles[k - 1] / mfma_waves) * (k + 1) * 18 for k in range(1, 5)) + math.ceil(tiles[4] * 2 / mfma_waves) * 12 * 9
    core = th * tw
    return dict(tile=f"{th}x{tw}", lds_kb=round(lds / 1024, 1), fits=lds <= 160 * 1024, issued_per_px=round(issued / core, 1),
                over_issue=round(issued / core / ALG_MFMA_PER_PX, 2), crit_mfma_per_wave=crit,
                balance=round(issued / mfma_waves / crit, 2),
                cap_at_100pct_busy=round(1 / (issued / core / ALG_MFMA_PER_PX) * (issued / mfma_waves / crit), 2))


if __name__ == "__main__":
    shapes = [(8, 8), (8, 16), (16, 16)]
    if len(sys.argv) > 2:
        shapes = [(int(sys.argv[i]), int(sys.argv[i + 1])) for i in range(1, len(sys.argv) - 1, 2)]
    print(f"algorithmic MFMAs per pixel: {ALG_MFMA_PER_PX:.2f}")
    for th, tw in shapes:
        for rb, rs, mw, ringb in ((80, 2, 4, None), (64, 2, 4, None), (64, 1, 4, None), (64, 1, 8, 12288)):
            print(dict(row_bytes=rb, ring=("%d x 18 KB" % rs) if ringb is None else "%d B of tap slices" % ringb, mfma_waves=mw,
                       **plan(th, tw, rb, rs, mw, ringb)))
