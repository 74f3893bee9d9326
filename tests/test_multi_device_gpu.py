"""The GPUs of one node behind the library's own surfaces, rehearsed with two (three) contexts on device 0 (VERDICT r3 items 2 and 8;
the reference server is one process: aggregator_server.cpp:106-118, 390-416):
  zkhip_msm_stream_*    handle-owned MSM streams - two base sets streamed from two threads at once, against the CPU oracle;
  zkhip_multi_prover_*  the wrapping key partitioned into slices, prover instances side by side: the whole-key proof limb for limb;
  zkhip_dispatcher_*    replicas behind a dispatcher: every batch proved and verified, both entries used."""
import threading

import numpy as np
import pytest

from oracle import pyref as R
from tests.helpers import aff_limbs, random_fr_canonical, random_fr_uniform

pytestmark = pytest.mark.gpu


def test_two_threads_stream_two_base_sets(zk, oracle_lib):
    """Each thread owns a zkhip_msm_stream (depth 3) over its own base set - one plain, one table-backed with NAF scalars - and keeps
    three MSMs in flight while the other does the same: no process-wide slot numbers to collide on.  Device and host scalars."""
    O = oracle_lib
    g = aff_limbs(R.G1_GEN)
    sets = []
    for k, (n, table) in enumerate([(5000, None), (3000, True)]):
        bases = zk.fixed_base_mul(g, random_fr_canonical(900 + k, n), montgomery=False)
        b = zk.Bases.upload(bases)
        if table is not None:
            b.precompute(9, table_naf=table)
        scal = [random_fr_uniform(910 + 10 * k + i, n) for i in range(4)]
        sets.append(dict(n=n, bases=bases, b=b, scal=scal, exp=[O.jac_to_affine(O.msm(bases, s)) for s in scal]))
    errors = []

    def worker(st_):
        try:
            stream = zk.MsmStream(st_["b"], depth=3)
            dev = [zk.DeviceBuffer(s) for s in st_["scal"]]
            pinned = zk.PinnedBuffer(st_["scal"][3])
            for rnd in range(3):
                tickets = [stream.submit(dev[i].ptr, st_["n"]) for i in range(3)]
                with pytest.raises(zk.ZkhipError):
                    stream.submit(dev[0].ptr, st_["n"])                      # all three contexts are in flight
                assert (zk.jac_to_affine(stream.collect(tickets[1])) == st_["exp"][1]).all()       # out of order
                t3 = stream.submit_host(pinned.ptr, st_["n"])                # host scalars: the copy travels on the context's stream
                assert (zk.jac_to_affine(stream.collect()) == st_["exp"][0]).all()                 # ticket 0: the oldest
                assert (zk.jac_to_affine(stream.collect(tickets[2])) == st_["exp"][2]).all()
                assert (zk.jac_to_affine(stream.collect(t3)) == st_["exp"][3]).all()
                t0, t1 = stream.last_accumulate_interval()
                assert t1 > t0 and stream.last_accumulate_ms() > 0
            with pytest.raises(zk.ZkhipError):
                stream.collect()                                              # nothing in flight
            pageable = np.ascontiguousarray(st_["scal"][2])
            t = stream.submit_host(pageable.ctypes.data, st_["n"])            # pageable host memory works too (staged by the runtime)
            assert (zk.jac_to_affine(stream.collect(t)) == st_["exp"][2]).all()
            stream.free(); pinned.free()
            for d in dev:
                d.free()
        except Exception as e:      # noqa: BLE001
            errors.append(e)
    ths = [threading.Thread(target=worker, args=(s,)) for s in sets]
    [t.start() for t in ths]; [t.join() for t in ths]
    assert not errors, errors
    for s in sets:
        s["b"].free()


@pytest.fixture(scope="module")
def wrapping(zk):
    import bench
    nvk_l, npr, nin, trapdoor = bench.aggregator_inputs()
    agg = zk.AggregatorCircuit(2, 1)
    desc = zk.r1cs_desc_from_aggregator(agg)
    kp = zk.Keypair(desc, *trapdoor)
    yield dict(agg=agg, desc=desc, kp=kp, nvk=nvk_l, npr=npr, nin=nin)
    kp.free(); agg.free()


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0]], ids=["two_contexts", "three_contexts"])
def test_partitioned_key_in_one_process_equals_the_whole_key_proof(zk, wrapping, devices):
    """zkhip_multi_prover: the real wrapping key cut into len(devices) slices, a prover instance per slice (all on GPU 0 here), host
    threads side by side, partial sums added on the host: the proof of the whole key, limb for limb, for both kinds of table."""
    w = wrapping
    z = w["agg"].witness(w["nvk"], w["npr"], w["nin"])
    rr, ss = random_fr_uniform(5, 1)[0], random_fr_uniform(6, 1)[0]
    r1 = zk.r1cs_from_desc(w["desc"])
    crs = w["kp"].upload_crs()
    whole = zk.groth16_prove(crs, r1, z, rr, ss)
    crs.free(); r1.free()
    for naf in (False, True):
        mp = zk.MultiProver(w["kp"], w["desc"], devices, zk.key_opts(table_naf=naf))
        assert mp.size == len(devices)
        for _ in range(2):
            got = mp.prove(z, rr, ss)
            assert (got == whole).all()
        t = mp.timings()
        assert t["slowest_slice"] > 0 and t["host_tail"] > 0
        mp.free()
    assert zk.groth16_verify(w["kp"].vk(), z[1:5], whole)


def test_dispatcher_spreads_batches_over_two_contexts(zk, wrapping):
    """zkhip_dispatcher: two resident copies of the key and two pipelines on GPU 0; twelve batches (the two nested proofs in either
    order), all proved and verified, both entries of the list used, the primary inputs in the order of the batch."""
    w = wrapping
    disp = zk.AggregatorDispatcher(w["agg"], w["kp"], [0, 0], zk.key_opts(table_naf=False), gpu_slots=3, witness_workers=2)
    rr, ss = random_fr_uniform(7, 1)[0], random_fr_uniform(8, 1)[0]
    flip = lambda a, k: np.concatenate([a.reshape(2, -1)[1], a.reshape(2, -1)[0]]) if k else a.reshape(-1)
    tickets = [(k & 1, disp.submit(w["nvk"], flip(w["npr"], k & 1), flip(w["nin"], k & 1), rr, ss)) for k in range(12)]
    vk = w["kp"].vk()
    firsts = set()
    for k, t in tickets:
        prim, proof = disp.wait(t)
        assert zk.groth16_verify(vk, prim, proof)
        assert (prim[2:] == flip(w["nin"], k).reshape(2, 6)).all()
        firsts.add(proof[:24].tobytes())
    assert len(firsts) == 2                                       # same (r, s): one proof per order of the nested proofs
    st = disp.stats()
    assert sum(st) == 12 and min(st) >= 3, st
    assert disp.outstanding() == [0, 0]
    with pytest.raises(zk.ZkhipError) as e:
        disp.wait(12345)                                          # no such ticket
    assert e.value.code == -5                                     # ZKHIP_ERR_NO_TICKET, not the ERR_ARG of a failed batch
    # a batch that FAILS (an off-curve nested proof: witness generation refuses it with ZKHIP_ERR_ARG) is a known
    # ticket - collecting it must take it off its entry's count, or least-loaded routing steers away from that GPU for good
    bad = w["npr"].copy(); bad[48 + 6] ^= np.uint64(1)            # y of the second proof's A
    bad_tickets = [disp.submit(w["nvk"], bad, w["nin"], rr, ss) for _ in range(4)]
    assert sum(disp.outstanding()) == 4
    for t in bad_tickets:
        with pytest.raises(zk.ZkhipError) as e:
            disp.wait(t)
        assert e.value.code == -1                                 # ZKHIP_ERR_ARG: the batch itself
    assert disp.outstanding() == [0, 0]
    before = disp.stats()
    good = [disp.submit(w["nvk"], w["npr"], w["nin"], rr, ss) for _ in range(6)]
    for t in good:
        prim, proof = disp.wait(t)
        assert zk.groth16_verify(vk, prim, proof)
    after = disp.stats()
    assert [a - b for a, b in zip(after, before)] == [3, 3]       # both entries still take their share
    disp.free()
    with pytest.raises(zk.ZkhipError):
        zk.AggregatorDispatcher(w["agg"], w["kp"], [0, 99])       # no such GPU: refused, nothing leaked
