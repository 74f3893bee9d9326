"""Nine inputs per nested proof: the every-bit-position (NAF) key against the default key, proof element by proof element."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from zecale_amd import zkhip
zkhip.init(0)
nvk_l, npr, nin, trapdoor = bench.aggregator_inputs(9)
agg = zkhip.AggregatorCircuit(2, 9)
desc = zkhip.r1cs_desc_from_aggregator(agg)
kp = zkhip.Keypair(desc, *trapdoor)
r1 = zkhip.r1cs_from_desc(desc)
z = agg.witness(nvk_l, npr, nin)
print("satisfied:", r1.is_satisfied(z), "n_primary", agg.num_primary_inputs())
rr, ss = bench.random_fr_uniform(5, 1)[0], bench.random_fr_uniform(6, 1)[0]
prim = z[1:1 + agg.num_primary_inputs()]
res = {}
for name, opts in (("default", zkhip.key_opts(table_naf=False)), ("naf", zkhip.key_opts(table_naf=True)), ("plain", zkhip.key_opts(precompute=False))):
    crs = kp.upload_crs(opts)
    p = zkhip.groth16_prove(crs, r1, z, rr, ss)
    res[name] = p
    print(name, "kind", crs.table_kind, "window", crs.table_window, "verifies", zkhip.groth16_verify(kp.vk(), prim, p), "finite", crs.finite_terms())
    pipe = zkhip.AggregatorPipeline(agg, crs, gpu_slots=2, witness_workers=2)
    t = pipe.submit(nvk_l, npr, nin, rr, ss)
    prim2, p2 = pipe.wait(t)
    print("   pipeline: prim equal", bool((prim2 == prim).all()), "proof equal", bool((p2 == p).all()), "verifies", zkhip.groth16_verify(kp.vk(), prim2, p2))
    pipe.free(); crs.free()
for k in ("naf", "plain"):
    print(k, "A", bool((res[k][:24] == res["default"][:24]).all()), "B", bool((res[k][24:48] == res["default"][24:48]).all()), "C", bool((res[k][48:] == res["default"][48:]).all()))
