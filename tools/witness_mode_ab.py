"""Host / GPU / hybrid witness generation on ONE box, interleaved and repeated: the streaming prover on the nine-input
circuit (valid nested proofs) or, with `1`, on the one-input circuit.  Prints proofs/s and host cores busy per run.
    python3 tools/witness_mode_ab.py [nested_inputs=9] [repeats=3] [steps=576]
Environment knobs it is meant to be run under: ZKHIP_HYBRID_HOST_WORKERS (default 2)."""
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from zecale_amd import zkhip  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 9
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 576
zkhip.init(0)
args = types.SimpleNamespace(gpu_slots=int(os.environ.get("GS", "32")), witness_workers=int(os.environ.get("WW", "10")))
MODES = os.environ.get("MODES", "host,gpu,hybrid").split(",")
nvk_l, npr, nin, trapdoor = bench.aggregator_inputs(k)
agg = zkhip.AggregatorCircuit(2, k)
desc = zkhip.r1cs_desc_from_aggregator(agg)
kp = zkhip.Keypair(desc, *trapdoor)
crs = kp.upload_crs(zkhip.key_opts(table_naf=True))
rr, ss = bench.random_fr_uniform(5, 1)[0], bench.random_fr_uniform(6, 1)[0]
submit = lambda p_: p_.submit(nvk_l, npr, nin, rr, ss)
for rep in range(reps):
    for name, kw in (("host", {}), ("gpu", dict(gpu_witness=True)), ("hybrid", dict(gpu_witness=True, hybrid=True))):
        if name not in MODES:
            continue
        o, _ = bench.stream_rates(zkhip, args, agg, crs, kp.vk(), submit, steps, 96, nested_vk=nvk_l, **kw)
        print("k=%d rep %d %-6s %8.1f proofs/s on %.2f host cores, verifies %s, bits %s" % (k, rep, name, o["value"], o["host_cores_busy"], o["last_proof_verifies"], o["result_bits"]), flush=True)

# CHECK=<n>: n proofs of ONE batch and ONE (r, s) through the GPU generator at full rate, 256 outstanding - every proof must be the same
# bytes as the serial proof of the host assignment (the chip is fully loaded by the provers while the witness kernels run: the
# condition under which the value array's store -> prefetch distance of k_witness's short ring is exercised hardest)
if os.environ.get("CHECK"):
    n = int(os.environ["CHECK"])
    r1 = zkhip.r1cs_from_desc(desc)
    want = zkhip.groth16_prove(crs, r1, agg.witness(nvk_l, npr, nin), rr, ss)
    for mode, kw in (("gpu", dict(gpu_witness=True)), ("hybrid", dict(gpu_witness=True, hybrid=True))):
        pipe = zkhip.AggregatorPipeline(agg, crs, gpu_slots=32, witness_workers=6, **kw)
        pipe.register_app(nvk_l)
        tickets, bad, done = [], 0, 0
        import time
        t0 = time.time()
        for i in range(n):
            tickets.append(submit(pipe))
            if len(tickets) > 256:
                prim, proof = pipe.wait(tickets.pop(0)); done += 1
                bad += 0 if (proof == want).all() else 1
        while tickets:
            prim, proof = pipe.wait(tickets.pop(0)); done += 1
            bad += 0 if (proof == want).all() else 1
        print("CHECK k=%d %s: %d proofs at %.1f proofs/s, different from the serial proof: %d" % (k, mode, done, done / (time.time() - t0), bad), flush=True)
        pipe.free()
    r1.free()
