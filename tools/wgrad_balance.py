"""Load balance of the batched weight-gradient launches of the default bench configuration: for every WgradBatch the
item list (in launch order) is list-scheduled onto 256 CUs; prints ideal vs simulated makespan in tile-iterations."""
import gc, heapq, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from satlas_super_resolution_amd import engine
from satlas_super_resolution_amd.train_step import ESRGANTrainStep, StepConfig

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
g_kw = dict(num_in_ch=24, num_out_ch=3, scale=4, num_feat=64, num_block=23, num_grow_ch=32)
d_kw = dict(num_in_ch=3, num_feat=64, skip_connection=True)
ts = ESRGANTrainStep(g_kw, d_kw, B, 32, 32, "bf16", StepConfig())
for obj in gc.get_objects():
    if isinstance(obj, engine.WgradBatch) and obj.layer_tab is not None:
        cost = []
        for it in obj.items:
            L = obj.layers[it.layer]
            nci = min(64 if obj.k == 3 else 32, L.Cin_w - it.ci0)
            f = 1.0 if (obj.k != 3 or nci > 32) else 0.6
            cost.append((it.tile_end - it.tile_begin) * f)
        heap = [0.0] * 256
        for c in cost:
            heapq.heapreplace(heap, heap[0] + c)
        print(f"k={obj.k} layers={len(obj.layers)} items={len(cost)} total={sum(cost):.0f} ideal/CU={sum(cost)/256:.1f} "
              f"makespan={max(heap):.1f} eff={sum(cost)/256/max(heap):.2f} biggest={max(cost):.0f}")
