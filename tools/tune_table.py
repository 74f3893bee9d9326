"""Sweep the table window c for a few MSM sizes (GPU box).  Usage: python tools/tune_table.py 20:18,19,20,21 16:14,15,16,17"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
from zecale_amd import zkhip
from bench import g1_generator_limbs, random_fr_canonical

zkhip.init(0)
dev = torch.device("cuda", 0)
g1 = g1_generator_limbs()
for spec in sys.argv[1:]:
    logn, cs = spec.split(":")
    n = int(float(logn)) if "." not in logn else int(2 ** float(logn))
    n = (1 << int(logn)) if logn.isdigit() else n
    ks = torch.from_numpy(random_fr_canonical(1, n).view(np.int64)).to(dev)
    pts = torch.empty((n, 24), dtype=torch.int64, device=dev)
    zkhip.fixed_base_mul_dev(g1, ks.data_ptr(), n, pts.data_ptr(), montgomery=False)
    sc = [torch.from_numpy(random_fr_canonical(10 + i, n).view(np.int64)).to(dev) for i in range(3)]
    torch.cuda.synchronize()
    for c in [int(x) for x in cs.split(",")]:
        b = zkhip.Bases.upload_dev(pts.data_ptr(), n)
        t0 = time.time()
        if c > 0:
            b.precompute(c)
        tb = time.time() - t0
        for i in range(2):
            b.msm_dev(sc[i].data_ptr(), n, montgomery=False)
        t0 = time.time()
        acc = []
        for i in range(6):
            b.msm_dev(sc[i % 3].data_ptr(), n, montgomery=False)
            acc.append(zkhip.last_accumulate_ms())
        dt = (time.time() - t0) / 6
        print(f"n={n} c={c} build={tb:.2f}s msm={dt*1e3:.2f} ms accumulate={np.mean(acc):.2f} ms  {n/dt/1e6:.1f} Mscalar/s", flush=True)
        b.free()
