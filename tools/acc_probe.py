#!/usr/bin/env python3
"""Where does k_accumulate<5> spend its time on the wrapping key?  

Runs the real batch-2 wrapping circuit (44,183 constraints) through the plain prover entry point, ONE proof at a time, for a
grid of key options (table kind x window), with ZKHIP_DEBUG_DUMP on: the library then records begin / end clocks of every wave
of the accumulation launch and dumps the bucket populations.  Prints, per configuration:
  entries M, buckets, slice length S and slices T, the kernel's event time, the mad-pipe fraction at that time,
  entries per bucket (mean / p50 / p99 / max), runs opened per slice, F pieces,
  per-wave duration (min / p50 / p99 / max) and the spread of the wave END times (the tail of a one-fill launch).
With --stream N it also measures the streaming prover's throughput for the configuration (no debug dump there).

  python3 tools/acc_probe.py --grid "naf:16,win:16,win:18" [--stream 300] [--nested-inputs 1]
Run on the GPU box (gpurun); `rocprofv3 --pmc ... -- python3 tools/acc_probe.py --grid naf:16 --proofs 4` gives the SQ counters of the
same launches (tools/collect_acc5_counters.sh).
"""
import argparse
import os
import struct
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MADS = 13149            # v_mad_u64_u32 per mixed addition (6 products, 2 squarings, 1 dual product)
PEAK = 476e9 * 64       # measured issue peak of that instruction, lane-mads per second (tools/ubench/fqmul_occ_bench.hip)


def read_dump(path):
    raw = open(path, "rb").read()
    nb, nw, S, T, tight, K, c, merged = struct.unpack("<8Q", raw[:64])
    cnt = np.frombuffer(raw, dtype=np.uint32, count=nb, offset=64)
    tm = np.frombuffer(raw, dtype=np.uint64, count=4 * nw, offset=64 + 4 * nb).reshape(-1, 4)
    return dict(nb=nb, S=S, T=T, tight=tight, K=K, c=c, merged=merged, cnt=cnt, tm=tm)


def slice_stats(cnt, S_host, T, tight):
    """Replays the kernel's slicing on the host (slices of equal WEIGHT: 1 for the first entry of a bucket, 8 for every other one):
    runs opened, F pieces, additions, iterations of a wave."""
    cnt = cnt.astype(np.int64)
    M = int(cnt.sum())
    ne = cnt > 0
    w = np.where(ne, 8 * cnt - 7, 0)
    G = int(w.sum())
    a_host = 8 * S_host
    A = a_host
    if tight:
        a = (G + T) // T
        A = min(a_host, 128) if a < 128 else min(a, a_host)
    goff = (np.cumsum(w) - w)[ne]
    c = cnt[ne]
    t_first = goff // A
    t_last = np.where(c >= 2, (goff + 1 + 8 * (c - 2)) // A, t_first)
    pieces = t_last - t_first + 1
    runs = int(pieces.sum())
    fpieces = pieces - 1
    lanes = (G - 1) // A + 1 if G else 0
    return dict(M=M, S=A / 8.0, lanes=lanes, runs=runs, runs_per_slice=runs / max(lanes, 1),
                cut_buckets=int((fpieces > 0).sum()), short=int(((fpieces >= 2) & (fpieces <= 4)).sum()), long=int((fpieces > 4).sum()),
                additions=M - runs, iterations=-(-A // 8))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--grid", default="naf:0,win:0", help="comma list of kind:window (kind naf|win, window 0 = automatic)")
    ap.add_argument("--proofs", type=int, default=3, help="serial proofs per configuration (the last one is reported)")
    ap.add_argument("--stream", type=int, default=0, help="also push this many proofs through the streaming prover")
    ap.add_argument("--prove-stream", type=int, default=0, help="GPU side only: this many proofs of ONE precomputed assignment through --gpu-slots "
                    "prover instances on a host thread each (no witness generation: the device's capacity, with less host noise)")
    ap.add_argument("--app", action="store_true", help="--prove-stream: the MASKED assignment of the registered application through zkhip_prover_prove_app "
                    "(the streaming pipeline's steady state with the per-application constants on) instead of the full one")
    ap.add_argument("--dev", action="store_true", help="--prove-stream --app: the masked assignment already in DEVICE memory (zkhip_prover_prove_app_dev): no upload per proof")
    ap.add_argument("--pinned", action="store_true", help="--prove-stream: the assignment lives in PINNED host memory (zkhip_host_alloc) - the per-proof upload "
                    "is then a DMA the stream waits for, not a staged copy the host thread drives")
    ap.add_argument("--repeat", type=int, default=1, help="repeat the --stream / --prove-stream measurement this many times")
    ap.add_argument("--gpu-slots", type=int, default=24)
    ap.add_argument("--depth", type=int, default=0, help="--stream: batches the caller keeps outstanding (0 = gpu slots + witness workers + 2)")
    ap.add_argument("--witness-workers", type=int, default=10)
    ap.add_argument("--nested-inputs", type=int, default=1)
    ap.add_argument("--no-dump", action="store_true", help="no ZKHIP_DEBUG_DUMP (for counter passes: nothing but the proofs)")
    ap.add_argument("--keep", default="", help="directory that receives a copy of every raw dump (offline analysis)")
    ap.add_argument("--cpus", default="", help="restrict the process to these CPUs before anything starts (e.g. 0-63,128-191)")
    args = ap.parse_args()
    if args.cpus:
        cp = set()
        for part in args.cpus.split(','):
            a, _, b = part.partition('-')
            cp.update(range(int(a), int(b or a) + 1))
        os.sched_setaffinity(0, cp)
    dump_dir = None
    if not args.no_dump:
        dump_dir = tempfile.mkdtemp(prefix="zkdump")
        os.environ["ZKHIP_DEBUG_DUMP"] = dump_dir
    import bench
    from zecale_amd import zkhip
    zkhip.init(0)
    nvk_l, npr, nin, trapdoor = bench.aggregator_inputs(args.nested_inputs)
    agg = zkhip.AggregatorCircuit(2, args.nested_inputs)
    desc = zkhip.r1cs_desc_from_aggregator(agg)
    kp = zkhip.Keypair(desc, *trapdoor)
    r1 = zkhip.r1cs_from_desc(desc)
    rr, ss = bench.random_fr_uniform(5, 1)[0], bench.random_fr_uniform(6, 1)[0]
    z = agg.witness(nvk_l, npr, nin)
    pk, m, l, dom = kp.pk_arrays()
    finite = sum(int(np.count_nonzero(pk[k].reshape(-1, 24).any(axis=1))) for k in ("A", "B2", "B1", "H", "L"))
    print("circuit: %d constraints, %d variables, domain %d, finite bases of the five queries %d" % (agg.num_constraints, m, dom, finite), flush=True)
    ref_proof = None
    for item in args.grid.split(","):
        kind, c = item.split(":")
        c = int(c)
        crs = kp.upload_crs(zkhip.key_opts(table_naf=(kind == "naf"), window=c))
        tw = crs.table_window
        for _ in range(args.proofs):
            proof = zkhip.groth16_prove(crs, r1, z, rr, ss)
        if ref_proof is None:
            ref_proof = proof
        same = bool((proof == ref_proof).all())
        acc_ms = zkhip.last_accumulate_ms()
        ph = zkhip.last_prove_timings()
        line = {"kind": kind, "window": tw, "acc_ms": round(acc_ms, 3), "msm_phase_ms": round(float(ph["msm_A"]), 3), "same_proof": same}
        if dump_dir:
            path = os.path.join(dump_dir, "acc_K5_c%d_m%d.bin" % (tw, 2 if kind == "naf" else 1))
            if args.keep:
                import shutil
                os.makedirs(args.keep, exist_ok=True)
                shutil.copy(path, os.path.join(args.keep, "%s_%d.bin" % (kind, tw)))
            d = read_dump(path)
            st = slice_stats(d["cnt"], d["S"], d["T"], d["tight"])
            cnt = d["cnt"][d["cnt"] > 0]
            tm = d["tm"][d["tm"][:, 1] > 0].astype(np.int64)
            dur = (tm[:, 1] - tm[:, 0]) / 100.0            # wall_clock64: 100 MHz -> microseconds
            t0 = tm[:, 0].min()
            end = (tm[:, 1] - t0) / 100.0
            beg = (tm[:, 0] - t0) / 100.0
            line.update(M=st["M"], digits_per_finite_base=round(st["M"] / finite, 2), S=st["S"], S_host=d["S"], lanes=st["lanes"], tight=d["tight"],
                        runs_per_slice=round(st["runs_per_slice"], 2), additions=st["additions"],
                        cut=st["cut_buckets"], short=st["short"], long=st["long"],
                        bucket_mean=round(float(cnt.mean()), 1), bucket_p50=int(np.percentile(cnt, 50)), bucket_p99=int(np.percentile(cnt, 99)), bucket_max=int(cnt.max()),
                        nonempty=int(len(cnt)), nb=d["nb"],
                        frac_entries=round(st["M"] * MADS / (acc_ms * 1e-3) / PEAK, 3),
                        frac_additions=round(st["additions"] * MADS / (acc_ms * 1e-3) / PEAK, 3),
                        waves=len(dur), wave_us_min=round(float(dur.min()), 1), wave_us_p50=round(float(np.percentile(dur, 50)), 1),
                        wave_us_p99=round(float(np.percentile(dur, 99)), 1), wave_us_max=round(float(dur.max()), 1),
                        begin_us_p50=round(float(np.percentile(beg, 50)), 1), begin_us_max=round(float(beg.max()), 1),
                        end_us_p01=round(float(np.percentile(end, 1)), 1), end_us_p50=round(float(np.percentile(end, 50)), 1), end_us_max=round(float(end.max()), 1),
                        us_per_iteration_p50=round(float(np.percentile(dur, 50)) / st["iterations"], 2), iterations=st["iterations"])
        if args.prove_stream:
            import threading
            saved = os.environ.pop("ZKHIP_DEBUG_DUMP", None)
            provers = [zkhip.Prover(crs, desc) for _ in range(args.gpu_slots)]
            for p_ in provers:
                p_.set_streaming(True)
                p_.prove(z, rr, ss)                      # work space
            app = zkhip.AggregatorApp(agg, crs, nvk_l) if args.app else None
            zm = app.witness(npr, nin) if args.app else None
            dbuf = zkhip.DeviceBuffer(zm) if (args.app and args.dev) else None
            pin = None
            if args.pinned:
                import ctypes
                src = zm if args.app else z
                pin = zkhip.PinnedBuffer(src)
                view = np.ctypeslib.as_array(ctypes.cast(pin.ptr, ctypes.POINTER(ctypes.c_uint64)), (src.size,)).reshape(src.shape)
                if args.app:
                    zm = view
                else:
                    z = view
            rates = []
            for _ in range(args.repeat):
                counter, lock, outs = [args.prove_stream], threading.Lock(), []

                def worker(p_):
                    while True:
                        with lock:
                            if counter[0] <= 0:
                                return
                            counter[0] -= 1
                        outs.append(p_.prove_app_dev(app, dbuf.ptr, rr, ss) if (args.app and args.dev) else (p_.prove_app(app, zm, rr, ss) if args.app else p_.prove(z, rr, ss)))
                ths = [threading.Thread(target=worker, args=(p_,)) for p_ in provers]
                t = time.time()
                [x.start() for x in ths]; [x.join() for x in ths]
                rates.append(round(args.prove_stream / (time.time() - t), 1))
                assert all((o == ref_proof).all() for o in outs[-4:])
            line.update(prove_stream_proofs_per_s=rates, masked_assignment=bool(args.app), pinned_assignment=bool(args.pinned), device_assignment=bool(args.app and args.dev))
            if dbuf is not None:
                dbuf.free()
            if pin is not None:
                pin.free()
            if app is not None:
                app.free()
            for p_ in provers:
                p_.free()
            if saved is not None:
                os.environ["ZKHIP_DEBUG_DUMP"] = saved
        if args.stream:
            saved = os.environ.pop("ZKHIP_DEBUG_DUMP", None)
            pipe = zkhip.AggregatorPipeline(agg, crs, gpu_slots=args.gpu_slots, witness_workers=args.witness_workers)
            depth = args.depth or args.gpu_slots + args.witness_workers + 2

            def run(k):
                tickets, last = [], None
                for _ in range(k):
                    tickets.append(pipe.submit(nvk_l, npr, nin, rr, ss))
                    if len(tickets) > depth:
                        last = pipe.wait(tickets.pop(0))
                while tickets:
                    last = pipe.wait(tickets.pop(0))
                return last
            run(2 * args.gpu_slots)
            rates = []
            for _ in range(args.repeat):
                t = time.time()
                prim, pr = run(args.stream)
                rates.append(round(args.stream / (time.time() - t), 1))
            line.update(stream_proofs_per_s=rates if args.repeat > 1 else rates[0], stream_same_proof=bool((pr == ref_proof).all()))
            pipe.free()
            if saved is not None:
                os.environ["ZKHIP_DEBUG_DUMP"] = saved
        print(line, flush=True)
        crs.free()


if __name__ == "__main__":
    main()
