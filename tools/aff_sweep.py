"""Accumulation time of one 2^20-term table-backed G1 MSM for every number of batched-affine levels (tuning aid)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from zecale_amd import zkhip

zkhip.init(0)
dev = torch.device("cuda", 0)
logn = int(os.environ.get("LOGN", "20"))
n = 1 << logn
g1 = bench.g1_generator_limbs()
ks = torch.from_numpy(bench.random_fr_canonical(0x5EED, n).view(np.int64)).to(dev)
pts = torch.empty((n, 24), dtype=torch.int64, device=dev)
zkhip.fixed_base_mul_dev(g1, ks.data_ptr(), n, pts.data_ptr(), montgomery=False)
torch.cuda.synchronize()
bases = zkhip.Bases.upload_dev(pts.data_ptr(), n)
bases.precompute()
sc = torch.from_numpy(bench.random_fr_canonical(0xABC0, n).view(np.int64)).to(dev)
torch.cuda.synchronize()
ref = None
for levels in [int(x) for x in os.environ.get("LEVELS", "0,1,2,3").split(",")]:
    zkhip.set_affine_levels(levels)
    ts, accs = [], []
    for rep in range(4):
        t = time.time()
        out = bases.msm_dev(sc.data_ptr(), n, montgomery=False)
        ts.append((time.time() - t) * 1e3)
        accs.append(zkhip.last_accumulate_ms())
    aff = zkhip.jac_to_affine(out)
    if ref is None:
        ref = aff
    print("levels", levels, "m", os.environ.get("ZKHIP_AFF_M", "64"), "msm_ms %.2f" % min(ts[1:]), "accumulate_ms %.2f" % min(accs[1:]),
          "same_result", bool((aff == ref).all()), flush=True)
