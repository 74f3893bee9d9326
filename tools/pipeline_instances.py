"""Is the wrapping stream's rate a property of the pipeline INSTANCE?  One process, one key: N pipelines created, timed in steady
state and freed one after the other (round 5: inside the full bench the same stream lands at 390-400 or at 420-440 proofs/s while
the chip's clock is the same and its power is 10 % lower in the slow case - tools/smi_during_bench.sh).
Usage: python tools/pipeline_instances.py [N] [steps] [keep]   (keep: the pipelines stay alive until the end)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from zecale_amd import zkhip

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1500
zkhip.init(0)
nvk_l, npr, nin, trapdoor = bench.aggregator_inputs(1)
agg = zkhip.AggregatorCircuit(2, 1)
kp = zkhip.Keypair(zkhip.r1cs_desc_from_aggregator(agg), *trapdoor)
crs = kp.upload_crs(zkhip.key_opts(table_naf=True))
rr, ss = bench.random_fr_uniform(5, 1)[0], bench.random_fr_uniform(6, 1)[0]
submit = lambda p_: p_.submit(nvk_l, npr, nin, rr, ss)
slots, workers = int(os.environ.get("SLOTS", "32")), 10
for i in range(N):
    pipe = zkhip.AggregatorPipeline(agg, crs, gpu_slots=slots, witness_workers=workers)
    pipe.register_app(nvk_l)
    dt, _, c0, c1 = bench.pipeline_steady_rate(pipe, submit, 2 * slots + workers, 96, steps)
    dt2, _, _, _ = bench.pipeline_steady_rate(pipe, submit, 2 * slots + workers, 0, steps)
    print("instance %d: %.1f then %.1f proofs/s, %.2f host cores" % (i, steps / dt, steps / dt2, ((c1.user + c1.system) - (c0.user + c0.system)) / dt), flush=True)
    pipe.free()
