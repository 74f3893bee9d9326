#!/usr/bin/env python3
"""bench.py - G1 MSM over BW6-761 at 2^20 (BASELINE.json config 2) on N GPUs, one process per GPU.

A "step" is one pass of the hot path over one batch: one 2^20-term G1 multi-scalar multiplication
(fresh scalars already resident in HBM; base points resident as the proving key is).  For N > 1 the
path shards by independent units: every rank owns its own 2^20-term slice of a
N * 2^20-term MSM (weak scaling); the only exchange is an all-gather of the N partial sums
(288 B each) followed by N-1 group additions on every rank.

Prints ONE JSON line (rank 0) with the driver's fields plus `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LOG_N = 20
ALG_BYTES_PER_TERM = 240          # 192 B affine base + 48 B scalar, each read once
HBM_PEAK_GBPS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md
# second (honest) bound: Fq multiplications. 25.2 M mixed additions x (8 M + 2 S) ... see DESIGN.md
FQ_MUL_PEAK_PER_S = 19.5e9        # measured chip-wide peak of fp_mul (tools/ubench/fqmul_bench.hip)


def random_fr_canonical(seed, n):
    x = (np.arange(1, n * 6 + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)).astype(np.uint64)
    x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    a = x.reshape(n, 6)
    a[:, 5] &= np.uint64((1 << 56) - 1)
    return a


def cpu_baseline(bases_sample, scal_sample):
    """CPU restatement (oracle/bw6_oracle.c: BDLO12 chunked over OpenMP threads), timed on this host."""
    from oracle import oracle as O
    O.load()
    threads = O.max_threads()
    t = time.time()
    out = O.msm(bases_sample, scal_sample, chunks=threads, with_mixed=True)
    dt = time.time() - t
    return out, dt, threads


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-log", type=int, default=17)
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl" if torch.cuda.is_available() else "gloo", rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP kernels are the only compute path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from zecale_amd import zkhip
    zkhip.init(local)
    n = 1 << args.log_n

    # synthetic inputs (seeded): bases k_i * G1 generated ON the GPU by the product's fixed-base
    # kernel (k_i from splitmix64), scalars uniform canonical < 2^376, one fresh set per step.
    g1 = np.array(list(zkhip_g1_generator()), dtype=np.uint64)
    ks = torch.from_numpy(random_fr_canonical(0x5EED + 1000 * rank, n).view(np.int64)).to(dev)
    bases_dev = torch.empty((n, 24), dtype=torch.int64, device=dev)
    zkhip.fixed_base_mul_dev(g1, ks.data_ptr(), n, bases_dev.data_ptr(), montgomery=False)
    torch.cuda.synchronize()
    bases = zkhip.Bases.upload_dev(bases_dev.data_ptr(), n)
    n_sets = args.steps + args.warmup
    scal_dev = [torch.from_numpy(random_fr_canonical(0xABC0 + 17 * i + 1000 * rank, n).view(np.int64)).to(dev)
                for i in range(min(n_sets, 4))]
    torch.cuda.synchronize()

    def step(i):
        s = scal_dev[i % len(scal_dev)]
        part = bases.msm_dev(s.data_ptr(), n, montgomery=False)
        if world > 1:
            import torch.distributed as dist
            mine = torch.from_numpy(part.view(np.int64)).to(dev)
            allp = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(allp, mine)
            acc = allp[0].cpu().numpy().view(np.uint64)
            for q in allp[1:]:
                acc = zkhip.jac_add(acc, q.cpu().numpy().view(np.uint64))
            return acc
        return part

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    barrier()
    acc_ms = []
    t0 = time.time()
    for i in range(args.steps):
        res = step(args.warmup + i)
        acc_ms.append(zkhip.last_accumulate_ms())
    barrier()
    dt = time.time() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        total_terms = n * world * args.steps
        value = total_terms / dt / 1e6
        kernel_ms = float(np.mean(acc_ms))
        achieved = ALG_BYTES_PER_TERM * n / (kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "G1-MSM Mscalar/s (BW6_761, 2^%d terms per GPU)" % args.log_n,
            "value": round(value, 3), "unit": "Mscalar/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32x27 (29-bit limbs, 761-bit Montgomery integers)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: single MI355X G1_BW6_761 Pippenger MSM, 2^%d random scalars/points"
                                   % args.log_n, "terms_per_gpu": n, "window_bits": 16, "bases": "resident (proving key)",
                       "parallelism": "point-partitioned x%d, all-gather of partial sums" % world},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": None,
                         "kernel": "k_accumulate", "kernel_ms": round(kernel_ms, 3),
                         "note": "integer-VALU bound, not HBM bound (SURVEY 0.5): see fq_mul_frac",
                         "fq_mul_frac": round((n * 24 * 9.52) / (kernel_ms * 1e-3) / FQ_MUL_PEAK_PER_S, 4)},
        }
        if not args.no_cpu_baseline:
            ns = 1 << min(args.cpu_sample_log, args.log_n)
            bs = bases_dev[:ns].cpu().numpy().view(np.uint64)
            ss = scal_dev[0][:ns].cpu().numpy().view(np.uint64)
            # canonical words used as Montgomery residues on both sides: same MSM instance for CPU and GPU
            cpu_out, cpu_dt, threads = cpu_baseline(bs, ss)
            b2 = zkhip.Bases.upload(bs)
            gpu_out = b2.msm(ss, montgomery=True)
            from oracle import oracle as O
            parity = bool((zkhip.jac_to_affine(gpu_out) == O.jac_to_affine(cpu_out)).all())
            out["cpu_baseline"] = {"value": round(ns / cpu_dt / 1e6, 5), "unit": "Mscalar/s", "cores": threads, "kind": "port",
                                   "sample": "one 2^%d-term G1 MSM (same generator), CPU restatement of libff multi_exp "
                                             "(BDLO12, OpenMP chunks), %.1f s; not libsnark itself" % (ns.bit_length() - 1, cpu_dt),
                                   "parity_with_gpu_on_sample": parity}
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def zkhip_g1_generator():
    """G1 generator in ABI form (reference client/test_commands/test_bw6_761_groth16_contract.py:28-31)."""
    q = 0x0122e824fb83ce0ad187c94004faff3eb926186a81d14688528275ef8087be41707ba638e584e91903cebaff25b423048689c8ed12f9fd9071dcd3dc73ebff2e98a116c25667a8f8160cf8aeeaf0a437e6913e6870000082f49d00000000008b
    gx = 0x01075b020ea190c8b277ce98a477beaee6a0cfb7551b27f0ee05c54b85f56fc779017ffac15520ac11dbfcd294c2e746a17a54ce47729b905bd71fa0c9ea097103758f9a280ca27f6750dd0356133e82055928aca6af603f4088f3af66e5b43d
    gy = 0x0058b84e0a6fc574e6fd637b45cc2a420f952589884c9ec61a7348d2a2e573a3265909f1af7e0dbac5b8fa1771b5b806cc685d31717a4c55be3fb90b6fc2cdd49f9df141b3053253b2b08119cad0fb93ad1cb2be0b20d2a1bafc8f2db4e95363
    for v in (gx, gy):
        m = (v << 768) % q
        for i in range(12):
            yield (m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF


if __name__ == "__main__":
    main()
