"""Soak run of the streaming prover on the GPU box: N batches with varying nested proofs, inputs (some bumped -> invalid nested
proof -> result bit 0) and (r, s); every wrapping proof is verified with the host pairing verifier and its public inputs checked.
Usage: python tools/soak_pipeline.py [N] [gpu] [hybrid] [nocache] [twokeys] [nine]
  gpu: assignments generated on the GPU; hybrid: two host generators beside it; nocache: per-application constants off;
  twokeys: every third batch belongs to a SECOND application (the fixture's key with ABC_0 / ABC_1 exchanged: result bits 0);
  nine (round 6): the nine-input circuit with the VALID trapdoor-built statements of tests/golden/nested_k9.json - random pairs of its
  three proofs, a fifth of the nested proofs get ONE input bumped at a random position (result bit 0 for that proof), full-size inputs"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zecale_amd import encoding as E
from zecale_amd import zkhip
import bench

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
zkhip.init(0)
gold = os.path.join(bench.ROOT, "tests", "golden", "dummy_app")
load = lambda n: json.load(open(os.path.join(gold, n)))
NINE = "nine" in sys.argv[2:]
K = 9 if NINE else 1
if NINE:
    j9 = json.load(open(os.path.join(bench.ROOT, "tests", "golden", "nested_k9.json")))
    nvk = E.nested_verification_key_from_json(j9["vk"])
    txs = [(None,) + tuple(E.nested_extended_proof_from_json(e)) for e in j9["proofs"]]          # (name, proof limbs, input limbs) as below
else:
    nvk = E.nested_verification_key_from_json(load("vk.json"))
    txs = [E.nested_transaction_from_json(load("extproof%d.json" % k)) for k in range(1, 7)]
_, _, _, trapdoor = bench.aggregator_inputs()
agg = zkhip.AggregatorCircuit(2, K)
kp = zkhip.Keypair(zkhip.r1cs_desc_from_aggregator(agg), *trapdoor)
vk, crs = kp.vk(), kp.upload_crs()
GPU_WITNESS = "gpu" in sys.argv[2:]
HYBRID, NOCACHE, TWOKEYS = "hybrid" in sys.argv[2:], "nocache" in sys.argv[2:], "twokeys" in sys.argv[2:]
nvk2 = nvk.copy()
nvk2[60:72], nvk2[72:84] = nvk[72:84], nvk[60:72]
pipe = zkhip.AggregatorPipeline(agg, crs, gpu_slots=14, witness_workers=2 if GPU_WITNESS else 8, gpu_witness=GPU_WITNESS, app_cache=not NOCACHE, hybrid=HYBRID)
rng = np.random.default_rng(1)
rs = bench.random_fr_canonical(77, 2 * N)
jobs, bad, t0 = [], 0, time.time()
for i in range(N):
    a, b = rng.integers(0, len(txs), 2)
    bump = [int(rng.random() < 0.2), int(rng.random() < 0.2)]
    nin = np.concatenate([txs[a][2], txs[b][2]]).copy()
    one = np.array(E.fr_from_json("0x" + "0" * 95 + "1"), dtype=np.uint64)
    expect = 0
    for p in range(2):
        if bump[p]:
            at = p * K + int(rng.integers(0, K))      # ONE of the proof's inputs, at a random position
            x = int(E.fr_to_json(nin[at]), 16) + 1
            nin[at] = np.array(E.fr_from_json(hex(x)), dtype=np.uint64)
        else:
            expect |= 1 << p
    key = nvk2 if (TWOKEYS and i % 3 == 2) else nvk
    if key is nvk2:
        expect = 0                                   # the fixtures' proofs do not verify under the second key
    jobs.append((pipe.submit(key, np.concatenate([txs[a][1], txs[b][1]]), nin, rs[2 * i], rs[2 * i + 1]), expect, nin))
    if len(jobs) > (96 if GPU_WITNESS else 32):
        t, exp, nin_ = jobs.pop(0)
        prim, proof = pipe.wait(t)
        ok = zkhip.groth16_verify(vk, prim, proof) and int(E.fr_to_json(prim[1]), 16) == exp and (prim[2:] == nin_.reshape(-1, 6)).all()
        bad += 0 if ok else 1
while jobs:
    t, exp, nin_ = jobs.pop(0)
    prim, proof = pipe.wait(t)
    ok = zkhip.groth16_verify(vk, prim, proof) and int(E.fr_to_json(prim[1]), 16) == exp and (prim[2:] == nin_.reshape(-1, 6)).all()
    bad += 0 if ok else 1
dt = time.time() - t0
hits = pipe.app_hits()
print(f"soak ({'nine VALID inputs per nested proof, ' if NINE else ''}{'GPU' if GPU_WITNESS else 'host'} witness{', hybrid' if HYBRID else ''}{', no cache' if NOCACHE else ''}{', two applications' if TWOKEYS else ''}; {hits} batches from an application's constants): {N} wrapping proofs in {dt:.1f} s ({N/dt:.1f} proofs/s including host verification of each), failures: {bad}")
pipe.free(); crs.free(); kp.free(); agg.free()
sys.exit(1 if bad else 0)
