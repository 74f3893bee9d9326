#!/usr/bin/env python3
"""bench.py - the hot path of the Zecale wrapping prover on N MI355X GPUs, one process per GPU.

Default workload (BASELINE.json configs[1], the configuration `metric` is quoted on):
  one G1 multi-scalar multiplication over BW6-761 with 2^20 terms per GPU.
A "step" is one pass of the hot path over one batch: one MSM with a fresh scalar vector that is
already resident in HBM; the base points are resident too (they are the proving key, uploaded once:
reference aggregator_server/aggregator_server.cpp:483-514).  For N > 1 the path shards by
independent units: rank r owns its own 2^20-term slice of an N * 2^20-term MSM (weak
scaling); the only exchange is an all-gather of the N partial sums (288 B each) over RCCL and
N - 1 group additions on every rank (zecale_amd/dist.py).

The default line also carries, measured after the timed region on rank 0 (N = 1):
  `plain_path`       the same MSM on a base set WITHOUT window tables (one-shot keys)
  `prover_2_20`      BASELINE configs[2] at the size `metric` quotes: a satisfiable 2^20-constraint system, trusted setup on
                     the GPU, proofs through two prover instances, the last proof verified with the host pairing check
  `wrapping_prover`  the real batch-2 aggregator circuit (44,183 constraints) through the streaming prover, witness generation
                     included, with its own `roofline` (k_accumulate<5>) and `cpu_baseline` (the C restatement proving the same batch)

`--workload prover` times configs[2] as the main metric (`--gpus N`: ONE proof per step over a key partitioned N ways),
`--workload aggregator` the wrapping circuit (`--gpus N`: replicas).

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
ALG_BYTES_PER_TERM = 240      # 192 B affine base + 48 B scalar, each read once
HBM_PEAK_GBPS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md (spec; 6.29 TB/s measured copy)
FQ_MUL_PEAK_PER_S = 19.5e9    # chip-wide peak of the Fq Montgomery multiplier, measured (tools/ubench/fqmul_bench.hip)
MULS_PER_MIXED_ADD = 10       # madd-2008-s: 8 M + 2 S, every one through the same multiplier
R_MOD = 0x01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001


def _splitmix(seed, count):
    x = (np.arange(1, count + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(seed)).astype(np.uint64)
    x ^= x >> np.uint64(30); x *= np.uint64(0xBF58476D1CE4E5B9)
    x ^= x >> np.uint64(27); x *= np.uint64(0x94D049BB133111EB)
    x ^= x >> np.uint64(31)
    return x


def random_fr_canonical(seed, n):
    """n x 6 limbs of canonical scalars < 2^376 < r from a splitmix64 stream (key generation, test inputs)."""
    a = _splitmix(seed, n * 6).reshape(n, 6)
    a[:, 5] &= np.uint64((1 << 56) - 1)
    return a


_R_LIMBS = [(R_MOD >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)]


def random_fr_uniform(seed, n):
    """n x 6 limbs UNIFORM in [0, r): 377-bit draws from a splitmix64 stream, rejected when >= r (16 % are)."""
    a = _splitmix(seed, n * 6).reshape(n, 6)
    a[:, 5] &= np.uint64((1 << 57) - 1)
    rnd = 0
    while True:
        ge = np.zeros(n, dtype=bool)          # lexicographic a >= r from the top limb down
        undecided = np.ones(n, dtype=bool)
        for k in range(5, -1, -1):
            rk = np.uint64(_R_LIMBS[k])
            ge |= undecided & (a[:, k] > rk)
            undecided &= a[:, k] == rk
        ge |= undecided
        bad = np.nonzero(ge)[0]
        if bad.size == 0:
            return a
        rnd += 1
        fresh = _splitmix(seed ^ (0xD1B54A32D192ED03 * rnd & 0xFFFFFFFFFFFFFFFF), bad.size * 6).reshape(bad.size, 6)
        fresh[:, 5] &= np.uint64((1 << 57) - 1)
        a[bad] = fresh


def g1_generator_limbs():
    """G1 generator in ABI form (reference client/test_commands/test_bw6_761_groth16_contract.py:28-31)."""
    q = 0x0122e824fb83ce0ad187c94004faff3eb926186a81d14688528275ef8087be41707ba638e584e91903cebaff25b423048689c8ed12f9fd9071dcd3dc73ebff2e98a116c25667a8f8160cf8aeeaf0a437e6913e6870000082f49d00000000008b
    gx = 0x01075b020ea190c8b277ce98a477beaee6a0cfb7551b27f0ee05c54b85f56fc779017ffac15520ac11dbfcd294c2e746a17a54ce47729b905bd71fa0c9ea097103758f9a280ca27f6750dd0356133e82055928aca6af603f4088f3af66e5b43d
    gy = 0x0058b84e0a6fc574e6fd637b45cc2a420f952589884c9ec61a7348d2a2e573a3265909f1af7e0dbac5b8fa1771b5b806cc685d31717a4c55be3fb90b6fc2cdd49f9df141b3053253b2b08119cad0fb93ad1cb2be0b20d2a1bafc8f2db4e95363
    out = []
    for v in (gx, gy):
        m = (v << 768) % q
        out += [(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(12)]
    return np.array(out, dtype=np.uint64)


def fr_mont(x):
    m = (x << 384) % R_MOD
    return np.array([(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(6)], dtype=np.uint64)


def zkhip_fr_one():
    return [int(v) for v in fr_mont(1)]


def measured_traffic():
    """HBM bytes per k_accumulate launch from the committed rocprofv3 --pmc passes (profiles/), or None."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get("k_accumulate_hbm_bytes_per_launch")
        except Exception:
            return None
    return None


def host_threads():
    """Host cores this process may actually use: the GPU box caps a one-GPU job with a cgroup CPU quota (16 of its 256 hardware
    threads); OpenMP would otherwise start one thread per hardware thread and split an MSM into that many tiny chunks."""
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else max(1, int(int(q) / int(per)))
    except Exception:
        pass
    avail = len(os.sched_getaffinity(0))
    return min(avail, quota) if quota else avail


def chain_system(log_n, seed=11):
    """A satisfiable system with 2^log_n - 8 constraints, 4 primary inputs (the wrapping circuit's count): constraint i multiplies two
    earlier variables into a new one.  Returns (A, B, C) CSR triples, z (Montgomery limbs), m, l."""
    n, l = (1 << log_n) - 8, 4
    m = n + l + 1
    rng = np.random.default_rng(seed)
    vals = [1, 3, 5, 7, 11] + [0] * n
    lo = rng.integers(0, 1 << 62, size=n)
    grow = np.arange(l + 1, l + 1 + n, dtype=np.uint64)
    a_idx = (lo.astype(np.uint64) % grow).astype(np.uint32)
    b_idx = ((lo.astype(np.uint64) >> np.uint64(31)) % grow).astype(np.uint32)
    al, bl = a_idx.tolist(), b_idx.tolist()
    for i in range(n):
        vals[l + 1 + i] = vals[al[i]] * vals[bl[i]] % R_MOD
    shift = 1 << 384
    z = np.frombuffer(b"".join((v * shift % R_MOD).to_bytes(48, "little") for v in vals), dtype=np.uint64).reshape(m, 6).copy()
    rp = np.arange(n + 1, dtype=np.uint32)
    ones = np.tile(z[0], (n, 1))
    return (rp, a_idx, ones), (rp, b_idx, ones), (rp, np.arange(l + 1, m, dtype=np.uint32), ones), z, m, l


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-n", type=int, default=LOG_N)
    ap.add_argument("--workload", choices=["msm", "prover", "aggregator"], default="msm")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-table", action="store_true", help="plain base sets: no precomputed window tables")
    ap.add_argument("--window", type=int, default=0, help="override the MSM window c (tuning; 0 = automatic, at most 18)")
    ap.add_argument("--no-batch-msms", action="store_true", help="one launch sequence per MSM instead of one per proof")
    ap.add_argument("--serial", action="store_true", help="one MSM / one proof in flight (per-phase timings) instead of the streaming forms")
    ap.add_argument("--gpu-slots", type=int, default=14, help="aggregator pipeline: proofs in flight on the GPU")
    ap.add_argument("--witness-workers", type=int, default=8, help="aggregator pipeline: witnesses generated side by side (3 host threads each)")
    ap.add_argument("--nested-inputs", type=int, default=1, help="aggregator workload: inputs per nested proof (9 = the shape of a Zeth proof, BASELINE configs[4])")
    ap.add_argument("--gpu-witness", action="store_true", help="aggregator pipeline: the witness workers generate the assignment on the GPU")
    ap.add_argument("--cpu-sample-log", type=int, default=20)
    ap.add_argument("--no-secondary", action="store_true", help="msm workload: skip the measurements that follow the timed region")
    args = ap.parse_args()

    # the aggregator pipeline keeps several proofs in flight, each on its own streams: give the HIP runtime more than its
    # default of 4 hardware queues (read once, when the runtime initialises)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP kernels are the only compute path")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    force_dist = os.environ.get("ZKHIP_BENCH_FORCE_DIST") == "1"      # exercise the RCCL plumbing with a single rank
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from zecale_amd import dist as zdist
    from zecale_amd import zkhip
    zkhip.init(local)
    zkhip.set_crs_precompute(not args.no_table)
    zkhip.set_batch_msms(not args.no_batch_msms)
    zkhip.set_msm_window(args.window)
    n = 1 << args.log_n
    g1 = g1_generator_limbs()

    def gen_bases(seed, count):
        """count points k_i * G1 (k_i from splitmix64), generated ON the GPU by the product's fixed-base kernel."""
        ks = torch.from_numpy(random_fr_canonical(seed, count).view(np.int64)).to(dev)
        out = torch.empty((count, 24), dtype=torch.int64, device=dev)
        zkhip.fixed_base_mul_dev(g1, ks.data_ptr(), count, out.data_ptr(), montgomery=False)
        torch.cuda.synchronize()
        return out

    def barrier():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    extra = {}
    collected = []            # msm workload: (index of the scalar vector, result) of MSMs finished inside the timed region
    if args.workload == "msm":
        bases_dev = gen_bases(0x5EED + 1000 * rank, n)
        bases = zkhip.Bases.upload_dev(bases_dev.data_ptr(), n)
        if not args.no_table:
            bases.precompute()          # setup-time, like loading the proving key: not part of a step
        extra["table_window"] = bases.table_window
        # scalars: uniform in [0, r), as libff holds them (Montgomery residues: the ABI's form, include/zkhip.h)
        n_sets = min(args.steps + args.warmup, 4)
        scal_dev = [torch.from_numpy(random_fr_uniform(0xABC0 + 17 * i + 1000 * rank, n).view(np.int64)).to(dev) for i in range(n_sets)]
        torch.cuda.synchronize()

        combine = (lambda part: zdist.combine_partial_sums(part, device=dev)) if (world > 1 or force_dist) else (lambda part: part)
        if args.serial:
            def step(i):
                s = scal_dev[i % n_sets]
                part = bases.msm_dev(s.data_ptr(), n, montgomery=True)
                collected.append((i % n_sets, part))
                return combine(part)
        else:
            # a stream of MSMs on the resident bases, two in flight (zkhip_msm_submit / zkhip_msm_collect): step i enqueues
            # MSM i and collects MSM i-1; drain() inside the timed region collects the last one.  Every step is one full MSM.
            # N > 1: the 288-byte exchange of MSM i-1 is started when it is collected and read one step later
            # (dist.combine_partial_sums_async): the all-gather kernel needs a free compute unit, which the accumulation of MSM i
            # does not leave for milliseconds at a time.
            inflight, exchanging = [], []
            distributed = world > 1 or force_dist
            for slot in (0, 1):       # set-up: allocate both slots' work space once (like any buffer allocation, not part of a step)
                bases.msm_submit(scal_dev[0].data_ptr(), n, slot=slot, montgomery=True)
                zkhip.msm_collect(slot)

            def settle(keep):
                out = None
                while len(exchanging) > keep:
                    out = exchanging.pop(0).result()
                return out

            def collect_oldest():
                slot, which = inflight.pop(0)
                part = zkhip.msm_collect(slot)
                collected.append((which, part))
                return part

            def step(i):
                s = scal_dev[i % n_sets]
                bases.msm_submit(s.data_ptr(), n, slot=i % 2, montgomery=True)
                inflight.append((i % 2, i % n_sets))
                if len(inflight) > 1:
                    part = collect_oldest()
                    if not distributed:
                        return part
                    out = settle(0)
                    exchanging.append(zdist.combine_partial_sums_async(part, device=dev))
                    return out

            def drain():
                out = None
                while inflight:
                    part = collect_oldest()
                    if distributed:
                        exchanging.append(zdist.combine_partial_sums_async(part, device=dev))
                    else:
                        out = part
                return settle(0) if distributed else out
            extra["drain"] = drain
            extra["msm_in_flight"] = 2
        units_per_step = n * world
    elif args.workload == "aggregator":
        # the real wrapping circuit on the committed nested fixtures (reference testdata/dummy_app: vk.json, extproof1/2.json)
        nvk_l, npr, nin, trapdoor = aggregator_inputs(args.nested_inputs)
        agg = zkhip.AggregatorCircuit(2, args.nested_inputs)
        desc = zkhip.r1cs_desc_from_aggregator(agg)
        kp = zkhip.Keypair(desc, *trapdoor)
        # a STREAM of proofs gains from the larger kind of window table (every bit position + NAF scalars: DESIGN.md section 5);
        # one proof at a time keeps the default
        naf_key = (not args.serial) and not args.no_table and os.environ.get("ZKHIP_TABLE_NAF") is None
        if naf_key:
            zkhip.set_table_naf(1)
        crs, r1 = kp.upload_crs(), zkhip.r1cs_from_desc(desc)
        zkhip.set_table_naf(-1)
        extra["table_kind"] = "every bit position, scalars in non-adjacent form (zkhip_set_table_naf)" if naf_key else "one level per window (default)"
        rr, ss = random_fr_uniform(5, 1)[0], random_fr_uniform(6, 1)[0]
        wit_ms = []
        if args.serial:
            # two-stage overlap only: a host thread generates the witness of batch i+1 (ctypes releases the GIL) while
            # the GPU proves batch i; one proof in flight, so the per-phase timings are meaningful.
            from concurrent.futures import ThreadPoolExecutor
            pool = ThreadPoolExecutor(max_workers=1)

            def make_witness():
                t = time.time()
                z = agg.witness(nvk_l, npr, nin)
                wit_ms.append((time.time() - t) * 1e3)
                return z
            pending = [pool.submit(make_witness)]

            def step(i):
                z = pending.pop().result()
                pending.append(pool.submit(make_witness))
                return zkhip.groth16_prove(crs, r1, z, rr, ss)
        else:
            # the streaming prover (zkhip_aggregator_pipeline_*): every step submits one batch (witness generation +
            # proof, nothing cached) and collects the oldest outstanding one; drain() inside the timed region collects
            # the rest, so exactly `steps` wrapping proofs are produced between the two barriers.
            pipe = zkhip.AggregatorPipeline(agg, crs, gpu_slots=args.gpu_slots, witness_workers=args.witness_workers, gpu_witness=args.gpu_witness)
            # (GPU witness: launches of 16 witnesses take ~60 ms each, so the stream is kept several launches deep)
            tickets, depth = [], (96 if args.gpu_witness else args.gpu_slots + args.witness_workers + 2)
            for t_ in [pipe.submit(nvk_l, npr, nin, rr, ss) for _ in range(2 * args.gpu_slots)]:   # set-up: every slot allocates its work space
                pipe.wait(t_)

            def step(i):
                tickets.append(pipe.submit(nvk_l, npr, nin, rr, ss))
                if len(tickets) > depth:
                    return pipe.wait(tickets.pop(0))

            def drain():
                out = None
                while tickets:
                    out = pipe.wait(tickets.pop(0))
                return out
            extra["drain"] = drain
        units_per_step = world
        n = agg.num_constraints
    else:
        # BASELINE configs[2] / [3]: a satisfiable system of 2^log_n - 8 constraints (4 primary inputs, the wrapping circuit's count;
        # about a third of its B query is the point at infinity, as in a real key), trusted setup on the GPU from a fixed trapdoor,
        # so that the proofs VERIFY.  The circuit, the key and the witness are the same on every rank.
        fs = FullSizeProver(zkhip, args.log_n, world, rank)
        n, crs, r1 = fs.n, fs.crs, fs.r1
        a_rng, h_rng, l_rng = fs.ranges
        z, rr, ss = fs.z, fs.rr, fs.ss
        if world > 1 or force_dist or args.serial:
            def step(i):
                if world > 1 or force_dist:
                    fs.last = zdist.prove_distributed(crs, r1, fs.consts, z, rr, ss, device=dev)
                else:
                    fs.last = zkhip.groth16_finish(fs.consts, zkhip.groth16_prove_partial(crs, r1, z), rr, ss)
                return fs.last
        else:
            # one GPU, whole key: two prover instances (zkhip_prover: own streams and work space), one host thread each, keep two
            # proofs in flight - the upload, the QAP map, the latency-bound end of the bucket reduction and the host tail of one
            # proof run under the accumulation of the other.  Every step is one full proof; drain() collects inside the timed region.
            step, drain = fs.two_in_flight()
            extra["drain"] = drain
            extra["proofs_in_flight"] = 2
            extra["accumulate_ms"] = lambda: max(p.last_accumulate_ms() for p in fs.provers)
        units_per_step = 1
        extra["scaling_override"] = "strong"

    if args.workload in ("prover", "aggregator"):
        extra["table_window"] = crs.table_window
    drain = extra.pop("drain", lambda: None)
    for i in range(args.warmup):
        step(i)
    drain()
    barrier()
    del collected[:]
    kernel_ms, phase = [], []
    cpu0 = os.times()
    t0 = time.time()
    for i in range(args.steps):
        step(args.warmup + i)
        kernel_ms.append(extra["accumulate_ms"]() if "accumulate_ms" in extra else zkhip.last_accumulate_ms())
        if (args.workload == "prover" and "proofs_in_flight" not in extra) or (args.workload == "aggregator" and args.serial):
            phase.append(zkhip.last_prove_timings())
    drain()
    barrier()
    dt = time.time() - t0
    cpu1 = os.times()
    extra["host_cores_busy"] = round(((cpu1.user + cpu1.system) - (cpu0.user + cpu0.system)) / dt, 2)   # this rank's process, timed region
    if os.environ.get("ZKHIP_BENCH_THREADS"):        # where the host time goes: CPU seconds per thread name (whole process life)
        acc = {}
        for t in os.listdir("/proc/self/task"):
            try:
                name = open("/proc/self/task/%s/comm" % t).read().strip()
                f = open("/proc/self/task/%s/stat" % t).read().rsplit(")", 1)[1].split()
                acc[name] = acc.get(name, 0.0) + (int(f[11]) + int(f[12])) / os.sysconf("SC_CLK_TCK")
            except Exception:
                pass
        extra["thread_cpu_s"] = {k: round(v, 2) for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:8]}
    if world > 1 or force_dist:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        k_ms = float(np.mean(kernel_ms))
        if args.workload == "msm":
            value, unit = units_per_step * args.steps / dt / 1e6, "Mscalar/s"
            metric = "G1-MSM Mscalar/s (BW6_761, 2^%d terms per GPU)" % args.log_n
            workload = "BASELINE configs[1]: single MI355X G1_BW6_761 Pippenger MSM, 2^%d random scalars/points" % args.log_n
            terms_in_kernel = n
        elif args.workload == "aggregator":
            value, unit = units_per_step * args.steps / dt, "proofs/s"
            metric = "wrapping proofs/sec (batch-2 BLS12_377 -> BW6_761 aggregation, %d constraints)" % n
            workload = ("BASELINE configs[0]/[2] shape: aggregator circuit batch=2 dummy_app proofs (reference fixtures), %s witness "
                        "generation + Groth16 BW6_761 prover on the GPU, end to end" % ("GPU" if args.gpu_witness else "host"))
            if args.nested_inputs > 1:
                workload = ("BASELINE configs[4] shape on one GPU: aggregator circuit batch=2 with %d inputs per nested proof (a Zeth proof has 9; "
                            "fixture points, result bits 0 - same circuit, witness and prover work as valid proofs), %s witness generation + "
                            "Groth16 BW6_761 prover on the GPU, end to end" % (args.nested_inputs, "GPU" if args.gpu_witness else "host"))
            m_, l_, d_ = agg.num_variables, agg.num_primary_inputs(), 1 << r1.log_d
            terms_in_kernel = (3 * m_ + (d_ - 1) + (m_ - l_ - 1)) if (tw_batched(extra, args)) else m_ - l_ - 1
            if args.serial:
                extra["phase_ms"] = {k: round(float(np.mean([p[k] for p in phase])), 3) for k in phase[0]}
                extra["phase_ms"]["witness_host"] = round(float(np.mean(wit_ms)), 3)
            else:
                extra["pipeline"] = {"gpu_slots": args.gpu_slots, "witness_workers": args.witness_workers, "witness_on_gpu": bool(args.gpu_witness),
                                     "note": "proofs in flight overlap; --serial gives the per-phase timings of one proof"}
        else:
            value, unit = units_per_step * args.steps / dt, "proofs/s"
            metric = "wrapping proofs/sec (Groth16 over BW6_761, 2^%d constraints)" % args.log_n
            workload = ("BASELINE configs[2]: full Groth16 BW6_761 prover (SpMV + 7 NTT + 5 MSM), satisfiable R1CS of 2^%d - 8 constraints, "
                        "4 primary inputs, trusted setup on the GPU; the last proof is verified (host pairing check)" % args.log_n)
            # one accumulation launch serves all five MSMs of a proof (table-backed key), else the last MSM of a proof is L
            al, hl, ll = a_rng[1] - a_rng[0], h_rng[1] - h_rng[0], l_rng[1] - l_rng[0]      # this rank's slice of the key
            terms_in_kernel = (3 * al + hl + ll) if tw_batched(extra, args) else ll
            if phase:
                extra["phase_ms"] = {k: round(float(np.mean([p[k] for p in phase])), 3) for k in phase[0]}
            extra["last_proof_verifies"] = fs.verify_last()
        tw_batched_flag = tw_batched(extra, args)
        tw = extra.pop("table_window", None)
        digits = -(-378 // tw) if tw else 24 if terms_in_kernel > (1 << 18) else None
        naf_tables = str(extra.get("table_kind", "")).startswith("every bit") and bool(tw)
        if naf_tables:
            digits = round(378.0 / (tw + 2), 2)          # width-(tw+1) non-adjacent form: one digit per tw + 2 bits on average
        timed = k_ms > 0
        if not timed:
            k_ms = float("nan")
        achieved = ALG_BYTES_PER_TERM * terms_in_kernel / (k_ms * 1e-3) / 1e9 if timed else None
        out = {
            "metric": metric, "value": round(value, 4), "unit": unit, "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": workload, "terms_per_gpu": n,
                       "scalars": "uniform in [0, r), Montgomery residues as libff holds them" if args.workload == "msm" else "the circuit's witness",
                       "bases": ("resident in HBM (proving key) with window tables: 2^j P_i for EVERY bit position j (378 levels, 378 x the key's "
                                 "memory), scalars in width-%d non-adjacent form, built at key-load time" % (tw + 1)) if naf_tables else
                                ("resident in HBM (proving key) with window tables: 2^(%d w) P_i for the %d window positions, built at "
                                 "key-load time, %d x the key's memory" % (tw, digits, digits)) if tw else "resident in HBM (proving key)",
                       "arithmetic": "761-bit Montgomery integers as 27 x 29-bit limbs in u32, products via v_mad_u64_u32",
                       "parallelism": "point-partitioned x%d, RCCL all-gather of 288-byte partial sums" % world},
            "roofline": roofline_obj("zkhip::k_accumulate<1>" if (args.workload == "msm" or not tw_batched_flag) else "zkhip::k_accumulate<5>",
                                     terms_in_kernel, digits, k_ms if timed else None,
                                     measured_traffic() if (timed and args.workload == "msm") else None),
        }
        if args.workload == "msm" and not args.serial and timed and digits:
            # the timed region streams two MSMs at a time, so its k_accumulate shares the chip with the reduction tail of the
            # previous MSM; for the kernel on its own, two more MSMs one at a time (outside the timed region)
            alone = []
            for i in range(2):
                bases.msm_dev(scal_dev[i % n_sets].data_ptr(), n, montgomery=True)
                alone.append(zkhip.last_accumulate_ms())
            ka = float(np.mean(alone))
            out["roofline"]["kernel_ms_alone"] = round(ka, 3)
            out["roofline"]["fq_mul_frac_alone"] = round(terms_in_kernel * digits * MULS_PER_MIXED_ADD / (ka * 1e-3) / FQ_MUL_PEAK_PER_S, 4)
        if "scaling_override" in extra:
            out["scaling"] = extra.pop("scaling_override")
            out["config"]["parallelism"] = "proving key partitioned x%d, RCCL all-gather of 5 x 288-byte partial sums per proof" % world
        if args.workload == "aggregator":
            out["config"]["parallelism"] = "replicas x%d: one streaming prover per GPU, independent batches, no collective" % world
        extra.pop("accumulate_ms", None)
        out.update(extra)
        if not args.no_cpu_baseline:
            from oracle import oracle as O
            O.load()
            threads = host_threads()
            O.set_threads(threads)
            threads = O.max_threads()
            if args.workload == "msm":
                # the SAME bases and the SAME scalar vector 0 as the timed loop (seeds above), so the oracle's output checks a
                # result that the timed, table-backed submit / collect path produced
                ns = 1 << min(args.cpu_sample_log, args.log_n)
                bs = gen_bases(0x5EED, ns).cpu().numpy().view(np.uint64)
                ss_ = random_fr_uniform(0xABC0, ns)
                t, ct = time.time(), os.times()
                cpu_out = O.msm(bs, ss_, chunks=threads, with_mixed=True)
                cpu_dt = time.time() - t
                ct2 = os.times()
                busy = ((ct2.user + ct2.system) - (ct.user + ct.system)) / cpu_dt    # cores actually kept busy (a container may cap them)
                cpu_aff = O.jac_to_affine(cpu_out)
                timed_hits = [part for which, part in collected if which == 0] if ns == n else []
                checked = "a result of the timed loop (scalar vector 0)"
                if not timed_hits:                                   # too few steps (or a smaller CPU sample): same path, one more MSM
                    bsub = bases if ns == n else None
                    if bsub is not None:
                        bsub.msm_submit(scal_dev[0].data_ptr(), n, slot=0, montgomery=True)
                        timed_hits = [zkhip.msm_collect(0)]
                        checked = "one more submit / collect on the timed path (no timed step used scalar vector 0)"
                    else:
                        b3 = zkhip.Bases.upload(bs)
                        if not args.no_table:
                            b3.precompute()
                        sd = torch.from_numpy(ss_.view(np.int64)).to(dev)
                        torch.cuda.synchronize()
                        b3.msm_submit(sd.data_ptr(), ns, slot=0, montgomery=True)
                        timed_hits = [zkhip.msm_collect(0)]
                        b3.free()
                        checked = "the table-backed submit / collect path on the CPU sample's 2^%d terms" % (ns.bit_length() - 1)
                parity = all(bool((zkhip.jac_to_affine(p) == cpu_aff).all()) for p in timed_hits)
                # the plain path (no window table) on the same input: parity and its own rate
                b2 = zkhip.Bases.upload(bs)
                plain_ok = bool((zkhip.jac_to_affine(b2.msm(ss_, montgomery=True)) == cpu_aff).all())
                if ns == n and not args.no_secondary:
                    sd = scal_dev[0]
                    ts = []
                    for _ in range(3):
                        torch.cuda.synchronize()
                        t = time.time()
                        b2.msm_dev(sd.data_ptr(), n, montgomery=True)
                        ts.append(time.time() - t)
                    out["plain_path"] = {"value": round(n / min(ts[1:]) / 1e6, 3), "unit": "Mscalar/s", "ms_per_msm": round(min(ts[1:]) * 1e3, 3),
                                         "note": "same MSM on a base set without window tables (24 digit positions, one at a time), parity with the oracle checked"}
                b2.free()
                cpu_val = ns / cpu_dt / 1e6
                out["cpu_baseline"] = {
                    "value": round(cpu_val, 6), "unit": unit, "cores": threads, "kind": "port",
                    "sample": "one 2^%d-term G1 MSM, CPU restatement of libff multi_exp (BDLO12, %d OpenMP chunks, -O2), %.1f s wall, "
                              "%.1f cores busy on average (process CPU time / wall); not libsnark itself (its sources are absent from the "
                              "reference tree)" % (ns.bit_length() - 1, threads, cpu_dt, busy),
                    "parity_with_gpu_on_sample": bool(parity and plain_ok), "gpu_result_checked": checked,
                    "timed_results_checked": len(timed_hits)}
            else:
                # proofs/s of the C restatement extrapolated from one timed MSM (a whole 2^20 proof takes a minute on these cores)
                ns = 1 << min(args.cpu_sample_log, 18)
                bs = gen_bases(0x5EED, ns).cpu().numpy().view(np.uint64)
                ss_ = random_fr_uniform(0xABC0, ns)
                t = time.time()
                O.msm(bs, ss_, chunks=threads, with_mixed=True)
                cpu_dt = time.time() - t
                terms = 5.0 * n
                out["cpu_baseline"] = {"value": round(ns / cpu_dt / terms, 6), "unit": unit, "cores": threads, "kind": "port",
                                       "sample": "one 2^%d-term G1 MSM of the C restatement (%.1f s); proofs/s extrapolated as 5 MSMs of %d "
                                                 "terms per proof" % (ns.bit_length() - 1, cpu_dt, n)}
        if args.workload == "msm" and world == 1 and not force_dist and not args.no_secondary:
            bases.free()
            out["prover_2_20"] = prover_secondary(zkhip, args)
            out["wrapping_prover"] = wrapping_prover_secondary(zkhip, args, cpu=not args.no_cpu_baseline)
        print(json.dumps(out))
    if world > 1 or force_dist:
        dist.destroy_process_group()


def roofline_obj(kernel, terms, digits, k_ms, traffic):
    timed = k_ms is not None and k_ms > 0
    achieved = ALG_BYTES_PER_TERM * terms / (k_ms * 1e-3) / 1e9 if timed else None
    return {"bound": "hbm", "achieved": round(achieved, 3) if timed else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 6) if timed else None, "traffic": traffic,
            "kernel": kernel, "kernel_ms": round(k_ms, 3) if timed else None,
            "algorithmic_bytes_per_launch": ALG_BYTES_PER_TERM * terms,
            "note": "this path is integer-multiply bound, not HBM bound (SURVEY 0.5): fq_mul_frac = Fq "
                    "multiplications per second in the kernel / measured chip peak of the multiplier",
            "fq_mul_frac": round(terms * digits * MULS_PER_MIXED_ADD / (k_ms * 1e-3) / FQ_MUL_PEAK_PER_S, 4) if (timed and digits) else None,   # upper bound when some scalars are 0
            "mixed_additions_per_term": digits}


class FullSizeProver:
    """BASELINE configs[2]: satisfiable system, GPU trusted setup, this rank's slice of the key, the witness."""

    def __init__(self, zkhip, log_n, world=1, rank=0):
        from zecale_amd import dist as zdist
        self.zk = zkhip
        A, B, C, self.z, self.m, self.l = chain_system(log_n)
        self.n = len(A[0]) - 1
        self.csr = (A, B, C)
        self.desc, self._keep = zkhip.make_r1cs_desc(A, B, C, self.m, self.l)
        self.r1 = zkhip.R1cs(A, B, C, self.m, self.l)
        d = 1 << self.r1.log_d
        self.kp = zkhip.Keypair(self.desc, fr_mont(0x1234567), fr_mont(0x2345678), fr_mont(0x3456789), fr_mont(0x456789a))
        self.vk = self.kp.vk()
        self.ranges = zdist.key_slices(self.m, self.l, d, world, rank)
        self.consts = self.kp.consts()
        if world == 1:
            self.crs = self.kp.upload_crs()
            self.crs.ranges = self.ranges                       # the whole key is the slice (0, n)
        else:
            pk, m, l, dom = self.kp.pk_arrays()
            self.crs = zkhip.Crs.upload_slice(pk, m, l, dom, *self.ranges)
            del pk
        self.rr, self.ss = random_fr_uniform(5, 1)[0], random_fr_uniform(6, 1)[0]
        self.provers, self.last = [], None

    def two_in_flight(self):
        from concurrent.futures import ThreadPoolExecutor
        zk = self.zk
        self.provers = [zk.Prover(self.crs, self.desc) for _ in range(2)]
        for p_ in self.provers:          # set-up: every instance allocates its work space on its first proof
            p_.prove(self.z, self.rr, self.ss)
        pool2 = ThreadPoolExecutor(max_workers=2)
        futs = []

        def step(i):
            futs.append(pool2.submit(self.provers[i % 2].prove, self.z, self.rr, self.ss))
            if len(futs) > 1:
                self.last = futs.pop(0).result()
                return self.last

        def drain():
            while futs:
                self.last = futs.pop(0).result()
            return self.last
        return step, drain

    def verify_last(self):
        return bool(self.last is not None and self.zk.groth16_verify(self.vk, self.z[1:1 + self.l], self.last))

    def free(self):
        for p_ in self.provers:
            p_.free()
        self.crs.free(); self.kp.free(); self.r1.free()


def aggregator_inputs(inputs_per_proof=1):
    """Nested key and two nested proofs from the reference's data fixtures (tests/golden/dummy_app = testdata/dummy_app: vk.json,
    extproof1.json, extproof2.json), decoded by the package's own JSON codec; toxic waste and (r, s) of the synthetic wrapping key.
    inputs_per_proof = 9 gives the SHAPE of a Zeth nested proof (aggregator_test.cpp:222-254; no Zeth fixtures exist): the key's
    ABC vector is padded with other curve points of the fixtures and every proof gets nine inputs, so the nested proofs do not
    verify (result bits 0) - the wrapping circuit, its witness and its proof cost exactly what valid ones would."""
    from zecale_amd import encoding as E
    gold = os.path.join(ROOT, "tests", "golden", "dummy_app")
    load = lambda name: json.load(open(os.path.join(gold, name)))
    nvk_l = E.nested_verification_key_from_json(load("vk.json"))
    txs = [E.nested_transaction_from_json(load("extproof%d.json" % k)) for k in (1, 2)]
    npr = np.concatenate([t[1] for t in txs])
    nin = np.concatenate([t[2] for t in txs])
    if inputs_per_proof > 1:
        k = inputs_per_proof
        more = [E.nested_transaction_from_json(load("extproof%d.json" % j)) for j in range(1, 7)]
        g1_points = [t[1][:12] for t in more] + [t[1][36:48] for t in more]          # A and C of the six fixtures: points of G1
        abc = np.concatenate([nvk_l[60:]] + g1_points)[: 12 * (k + 1)]
        assert len(abc) == 12 * (k + 1), "not enough fixture points for that many inputs"
        nvk_l = np.concatenate([nvk_l[:60], abc])
        nin = np.concatenate([np.concatenate([(t[2].reshape(-1, 6)[0] + np.array([j, 0, 0, 0, 0, 0], dtype=np.uint64)).reshape(1, 6) for j in range(k)]) for t in txs])
    fr = lambda x: np.array(E.fr_from_json(hex(x)), dtype=np.uint64)
    trapdoor = [fr(0x1234567), fr(0x2345678), fr(0x3456789), fr(0x456789a)]
    return nvk_l, npr, nin, trapdoor


def prover_secondary(zkhip, args, steps=6, warmup=1):
    """BASELINE configs[2] at the size `metric` quotes, in the same run after the MSM's timed region (N = 1 only): the
    `--workload prover` loop - a satisfiable 2^20-constraint system, trusted setup on the GPU, two prover instances in flight -
    with the last proof verified by the host pairing check before the number is reported."""
    t_setup = time.time()
    fs = FullSizeProver(zkhip, LOG_N)
    setup_s = time.time() - t_setup
    step, drain = fs.two_in_flight()
    for i in range(warmup):
        step(i)
    drain()
    t0 = time.time()
    for i in range(steps):
        step(warmup + i)
    drain()
    dt = time.time() - t0
    acc_ms = max(p.last_accumulate_ms() for p in fs.provers)
    # one proof alone for the phase timings
    proof = zkhip.groth16_prove(fs.crs, fs.r1, fs.z, fs.rr, fs.ss)
    phases = zkhip.last_prove_timings()
    same = bool((proof == fs.last).all())
    ok = fs.verify_last()
    bad = fs.z[1:1 + fs.l].copy(); bad[2] = fr_mont(6)
    rejects = not zkhip.groth16_verify(fs.vk, bad, fs.last)
    al, hl, ll = (b - a for a, b in fs.ranges)
    tw = fs.crs.table_window
    digits = -(-378 // tw) if tw else 24
    out = {"metric": "wrapping proofs/sec (Groth16 over BW6_761, 2^%d constraints): BASELINE configs[2], same run, after the timed region" % LOG_N,
           "value": round(steps / dt, 3), "unit": "proofs/s", "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3),
           "constraints": fs.n, "proofs_in_flight": 2, "last_proof_verifies": ok, "wrong_input_rejected": rejects,
           "serial_proof_identical": same, "setup_s": round(setup_s, 2),
           "phase_ms_one_proof_alone": {k: round(v, 3) for k, v in phases.items()},
           "roofline": roofline_obj("zkhip::k_accumulate<5>", 3 * al + hl + ll, digits, acc_ms, None)}
    fs.free()
    return out


def wrapping_prover_secondary(zkhip, args, steps=480, warmup=48, cpu=True):
    """The other half of BASELINE.json's metric, measured in the same run after the MSM's timed region (N = 1 only): wrapping
    proofs/s of the real batch-2 aggregator circuit through the streaming prover, witness generation included, nothing cached.
    The same loop as `--workload aggregator`; the last proof is verified (host pairing check) before the number is reported.
    `roofline`: k_accumulate<5> of one proof on its own.  `cpu_baseline`: the C restatement proving the IDENTICAL batch (host
    witness + QAP map + five MSMs + tail) on the host cores, its proof compared limb for limb with the GPU's."""
    nvk_l, npr, nin, trapdoor = aggregator_inputs()
    agg = zkhip.AggregatorCircuit(2, 1)
    desc = zkhip.r1cs_desc_from_aggregator(agg)
    kp = zkhip.Keypair(desc, *trapdoor)
    naf_key = os.environ.get("ZKHIP_TABLE_NAF") is None        # the streaming prover's key: the larger kind of window table
    if naf_key:
        zkhip.set_table_naf(1)
    crs = kp.upload_crs()
    zkhip.set_table_naf(-1)
    rr, ss = random_fr_uniform(5, 1)[0], random_fr_uniform(6, 1)[0]
    pipe = zkhip.AggregatorPipeline(agg, crs, gpu_slots=args.gpu_slots, witness_workers=args.witness_workers)
    depth = args.gpu_slots + args.witness_workers + 2

    def run(k):                                        # (reads `pipe` and `depth` of the enclosing scope at call time)
        tickets, last = [], None
        for _ in range(k):
            tickets.append(pipe.submit(nvk_l, npr, nin, rr, ss))
            if len(tickets) > depth:
                last = pipe.wait(tickets.pop(0))
        while tickets:
            last = pipe.wait(tickets.pop(0))
        return last
    run(warmup)
    c0, t0 = os.times(), time.time()
    prim, proof = run(steps)
    dt = time.time() - t0
    c1 = os.times()
    ok = bool(zkhip.groth16_verify(kp.vk(), prim, proof))
    out = {"metric": "wrapping proofs/sec (batch-2 BLS12_377 -> BW6_761 aggregation, %d constraints), same run, after the timed region"
                     % agg.num_constraints,
           "value": round(steps / dt, 3), "unit": "proofs/s", "steps": steps, "warmup": warmup, "ms_per_step": round(dt / steps * 1e3, 3),
           "last_proof_verifies": ok, "gpu_slots": args.gpu_slots, "witness_workers": args.witness_workers,
           "host_cores_busy": round(((c1.user + c1.system) - (c0.user + c0.system)) / dt, 2),
           "includes": "host witness generation + QAP + 5 MSMs + host tail per proof, reference dummy_app fixtures, nothing cached",
           "table_kind": "every bit position, scalars in non-adjacent form (zkhip_set_table_naf)" if naf_key else "one level per window"}
    pipe.free()
    # the same stream with the assignment generated ON THE GPU (SURVEY 8 rows a2-a5 as a kernel: witness.hip), two batcher threads
    try:
        gpipe = zkhip.AggregatorPipeline(agg, crs, gpu_slots=args.gpu_slots, witness_workers=2, gpu_witness=True)
        pipe, depth_host, depth = gpipe, depth, 96              # (launches of 16 witnesses: the stream is kept several launches deep)
        run(2 * warmup)
        c0g, t0g = os.times(), time.time()
        prim_g, proof_g = run(steps)
        dtg = time.time() - t0g
        c1g = os.times()
        out["gpu_witness"] = {"value": round(steps / dtg, 3), "unit": "proofs/s", "steps": steps, "ms_per_step": round(dtg / steps * 1e3, 3),
                              "host_cores_busy": round(((c1g.user + c1g.system) - (c0g.user + c0g.system)) / dtg, 2),
                              "last_proof_verifies": bool(zkhip.groth16_verify(kp.vk(), prim_g, proof_g)),
                              "proof_equals_host_witness_proof": bool((proof_g == proof).all()),
                              "includes": "GPU witness generation (k_witness, 16 witnesses per launch) + QAP + 5 MSMs + host tail per proof"}
        gpipe.free()
        depth = depth_host
    except Exception as e:                                      # (reported, never silently dropped)
        out["gpu_witness"] = {"error": str(e)}
    # one proof on its own: the accumulation kernel's time and the terms it processed
    r1 = zkhip.r1cs_from_desc(desc)
    t = time.time()
    z = agg.witness(nvk_l, npr, nin)
    wit_ms = (time.time() - t) * 1e3
    zkhip.groth16_prove(crs, r1, z, rr, ss)
    one = zkhip.groth16_prove(crs, r1, z, rr, ss)
    acc_ms = zkhip.last_accumulate_ms()
    phases = zkhip.last_prove_timings()
    pk, m, l, dom = kp.pk_arrays()
    finite = lambda a: int(np.count_nonzero(a.reshape(-1, 24).any(axis=1)))
    terms = sum(finite(pk[k]) for k in ("A", "B2", "B1", "H", "L"))
    tw = crs.table_window
    digits = -(-378 // tw) if tw else 24
    if naf_key and tw:
        digits = round(378.0 / (tw + 2), 2)              # non-adjacent form: one digit per tw + 2 bits on average
    out["roofline"] = roofline_obj("zkhip::k_accumulate<5>", terms, digits, acc_ms, None)
    out["roofline"]["terms"] = "the finite bases of the five query vectors of this key (a base at infinity produces no entry)"
    out["one_proof_alone_ms"] = {"witness_host": round(wit_ms, 3), **{k: round(v, 3) for k, v in phases.items()}}
    same = bool((one == proof).all())
    if cpu:
        from oracle import oracle as O
        O.load()
        threads = host_threads()
        O.set_threads(threads)
        threads = O.max_threads()
        A, B, C = agg.get_constraint_system()
        t, ct = time.time(), os.times()
        z_cpu = agg.witness(nvk_l, npr, nin)                      # the host witness generator is the product's (rows a2-a5 are host code)
        h = O.qap_h(A, B, C, z_cpu, agg.num_constraints, l)
        cpu_proof = O.groth16_prove(pk, z_cpu, l, h, rr, ss, chunks=threads)
        cpu_dt = time.time() - t
        ct2 = os.times()
        out["cpu_baseline"] = {"value": round(1.0 / cpu_dt, 4), "unit": "proofs/s", "cores": threads, "kind": "port",
                               "sample": "ONE wrapping proof of the identical batch by the C restatement (r1cs_to_qap_witness_map + r1cs_gg_ppzksnark_prover: "
                                         "7 FFTs, 5 BDLO12 multi_exps in %d OpenMP chunks) after the product's host witness generator, %.2f s wall, "
                                         "%.1f cores busy" % (threads, cpu_dt, ((ct2.user + ct2.system) - (ct.user + ct.system)) / cpu_dt),
                               "parity_with_gpu_on_sample": bool((cpu_proof == one).all())}
    out["pipeline_proof_equals_serial_proof"] = same
    crs.free(); kp.free(); agg.free(); r1.free()
    return out


def tw_batched(extra, args):
    return bool(extra.get("table_window")) and not args.no_batch_msms


if __name__ == "__main__":
    main()
