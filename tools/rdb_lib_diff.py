"""Where the 8x16 dense-block kernel of the LIBRARY build differs from the 8x8 kernel (debug aid): python tools/rdb_lib_diff.py N H W"""
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import test_gpu_rdb_tile as T
from satlas_super_resolution_amd import hip
N, H, W = (int(a) for a in sys.argv[1:4])
b = T._bufs(N, H, W, seed=N * 1000 + H)
s0, o0 = T._run(hip, b, N, H, W, False, False, 0, b["cur"])
s1, o1 = T._run(hip, b, N, H, W, False, False, 16, b["cur"])
for name, u, v in (("slices", s0, s1), ("out", o0, o1)):
    d = (u.view(torch.int16) != v.view(torch.int16))
    print(name, "differing:", int(d.sum()), "of", d.numel())
    if int(d.sum()):
        print("  by channel block of 32:", [int(d[..., c:c + 32].sum()) for c in range(0, 192, 32)])
        print("  by image:", [int(d[n].sum()) for n in range(N)])
        print("  by row:", [int(d[:, y].sum()) for y in range(H)])
        print("  by col:", [int(d[:, :, x].sum()) for x in range(W)])
        idx = d.nonzero()[:5].tolist()
        for i in idx: print("   ", i, float(u[tuple(i)]), float(v[tuple(i)]))
