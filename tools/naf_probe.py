"""Accumulation time per addition with either kind of window table (one level per window / every bit position + NAF scalars) for one query
of 2^16 .. 2^19 points, c = 16: python tools/naf_probe.py (GPU box)."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from zecale_amd import zkhip
zkhip.init(0)
dev = torch.device("cuda", 0)
g1 = bench.g1_generator_limbs()
for logn in (16, 17, 18, 19):
    n = 1 << logn
    ks = torch.from_numpy(bench.random_fr_canonical(7, n).view(np.int64)).to(dev)
    pts = torch.empty((n, 24), dtype=torch.int64, device=dev)
    zkhip.fixed_base_mul_dev(g1, ks.data_ptr(), n, pts.data_ptr(), montgomery=False)
    sc = torch.from_numpy(bench.random_fr_uniform(9, n).view(np.int64)).to(dev)
    torch.cuda.synchronize()
    for naf in (0, 1):
        zkhip.set_table_naf(naf)
        os.environ["ZKHIP_NAF_TABLE_GB"] = "100"
        b = zkhip.Bases.upload_dev(pts.data_ptr(), n)
        b.precompute(16)
        zkhip.set_table_naf(-1)
        ms = []
        for _ in range(4):
            b.msm_dev(sc.data_ptr(), n, montgomery=True)
            ms.append(zkhip.last_accumulate_ms())
        adds = n * (378 / 18.0 if naf else 24)
        print("n=2^%d naf=%d table %.1f GB accumulate %.3f ms  %.3f ns/add" % (logn, naf, n * (378 if naf else 24) * 193 / 1e9, min(ms), min(ms) * 1e6 / adds), flush=True)
        b.free()
