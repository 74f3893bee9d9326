"""What ONE rank of `bench.py --gpus N`'s partitioned-key leg does, measured on one GPU for N = 1, 2, 4, 8:
the rank's slice of the 2^22 key (equal finite terms: dist.key_slices_by_finite_terms) is uploaded with its window tables and
zkhip_groth16_prove_partial (upload of z, QAP map - both replicated on every rank - and the five MSMs over the slice) is timed, for
the first and the last rank of the partition.  The exchange (5 x 288 bytes over RCCL) and the host tail are not part of it.
A PREDICTION of the N-GPU time per proof from one GPU's measurements, not a scaling measurement.

    python3 tools/partition_model.py [log_n] > profiles/r05_partition_model.txt
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from zecale_amd import dist as zdist  # noqa: E402
from zecale_amd import zkhip  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 22
zkhip.init(0)
t = time.time()
fs = bench.FullSizeProver(zkhip, log_n)
print("2^%d - 8 constraints; setup %.1f s; finite terms of the whole key: %s" % (log_n, time.time() - t, fs.crs.finite_terms()), flush=True)
pk, m, l, dom = fs.kp.pk_arrays()
fs.crs.free()
for world in [int(x) for x in os.environ.get("ZKHIP_MODEL_WORLDS", "1,2,4,8").split(",")]:
    for rank in sorted({0, world - 1}):
        ranges = zdist.key_slices_by_finite_terms(pk, m, l, dom, world, rank)
        by_index = zdist.key_slices(m, l, dom, world, rank)
        crs = zkhip.Crs.upload_slice(pk, m, l, dom, *ranges)
        zkhip.groth16_prove_partial(crs, fs.r1, fs.z)
        ts = []
        for _ in range(3):
            t = time.time()
            zkhip.groth16_prove_partial(crs, fs.r1, fs.z)
            ts.append((time.time() - t) * 1e3)
        ph = bench.phase_dict(zkhip.last_prove_timings())
        print("N = %d rank %d: slice %s (by index it would be %s), finite terms %s, window %d: prove_partial %.1f ms (min of 3: %.1f); phases %s; split %s; accumulation launches %.1f ms"
              % (world, rank, ranges, by_index, crs.finite_terms(), crs.table_window, sum(ts) / 3, min(ts), {k: v for k, v in ph.items() if k != "host_tail"}, zkhip.last_prove_split(),
                 zkhip.last_accumulate_ms()), flush=True)
        crs.free()
