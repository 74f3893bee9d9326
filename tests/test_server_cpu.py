"""The gRPC surface (zecale_amd/server.py) replaying the reference's end-to-end client script against a live server on the
loopback interface, with a stubbed prover (no GPU here): scripts/test-client:41-96 - register the dummy application, a second
registration fails, four nested transactions go in (fees 12, 11, 10, 9), two batches come out in fee order with their parameters,
a third request fails.  Handler semantics: aggregator_server/aggregator_server.cpp:130-348; pool: application_pool.tcc:49-63."""
import json

import grpc
import numpy as np
import pytest

from oracle import pyref as R
from tests.helpers import golden
from zecale_amd import encoding as E
from zecale_amd import server as S


class StubProver:
    """Returns the reference's own wrapping proof fixture with the batch's nested inputs patched in: enough to check the
    service's plumbing (ordering, parameters, encodings) without a device."""
    snark_name = "GROTH16"

    def __init__(self):
        self.calls = []
        self.vk_json = golden("dummy_app/aggregator_vk.json")

    def verification_key_json(self):
        return self.vk_json

    def nested_vk_hash(self, limbs):
        from zecale_amd import zkhip
        return zkhip.aggregator_vk_hash(limbs, 1)          # host code: no device needed

    def prove(self, nested_vk_limbs, proofs, inputs):
        self.calls.append((nested_vk_limbs, proofs, inputs))
        ep = json.loads(json.dumps(golden("dummy_app/batch1.json")["ext_proof"]))
        ep["inputs"] = ep["inputs"][:2] + [E.fr_to_json(x.reshape(6)) for x in inputs]
        return ep


@pytest.fixture()
def live():
    prover = StubProver()
    server, port, service = S.serve(prover, "127.0.0.1:0", max_workers=4)
    yield S.AggregatorClient("127.0.0.1:%d" % port), prover, service
    server.stop(0)


def _details(excinfo):
    return excinfo.value.code(), excinfo.value.details()


def test_descriptors_match_the_reference_proto():
    """Service, method and message names and the zecale field numbers as in proto/zecale/api/aggregator.proto:9-79."""
    svc = S.POOL.FindServiceByName("zecale_proto.Aggregator")
    assert [m.name for m in svc.methods] == ["GetConfiguration", "GetVerificationKey", "GetNestedVerificationKeyHash", "RegisterApplication",
                                             "SubmitNestedTransaction", "GenerateAggregatedTransaction"]
    assert svc.methods_by_name["GenerateAggregatedTransaction"].input_type.full_name == "zecale_proto.AggregatedTransactionRequest"
    assert svc.methods_by_name["RegisterApplication"].output_type.full_name == "zecale_proto.VerificationKeyHash"
    nt = S.POOL.FindMessageTypeByName("zecale_proto.NestedTransaction")
    assert {f.name: f.number for f in nt.fields} == {"application_name": 1, "extended_proof": 2, "parameters": 3, "fee_in_wei": 4}
    at = S.POOL.FindMessageTypeByName("zecale_proto.AggregatedTransaction")
    assert {f.name: f.number for f in at.fields} == {"application_name": 1, "extended_proof": 2, "nested_parameters": 3}
    assert at.fields_by_name["nested_parameters"].is_repeated
    cfg = S.POOL.FindMessageTypeByName("zecale_proto.AggregatorConfiguration")
    assert [f.name for f in cfg.fields] == ["nested_snark_name", "wrapper_snark_name", "nested_pairing_parameters", "wrapper_pairing_parameters"]


def test_configuration_and_keys(live):
    client, prover, _ = live
    cfg = client.get_configuration()
    assert cfg["nested_snark_name"] == cfg["wrapper_snark_name"] == "GROTH16"
    assert cfg["wrapper_pairing_parameters"]["name"] == "bw6-761" and int(cfg["wrapper_pairing_parameters"]["r"], 16) == R.R_MOD
    assert int(cfg["wrapper_pairing_parameters"]["q"], 16) == R.Q_MOD and int(cfg["nested_pairing_parameters"]["q"], 16) == R.BLS_Q
    assert int(cfg["nested_pairing_parameters"]["r"], 16) == R.BLS_R
    g1 = tuple(int(c, 16) for c in cfg["wrapper_pairing_parameters"]["generator_g1"])
    g2 = tuple(int(c, 16) for c in cfg["wrapper_pairing_parameters"]["generator_g2"])
    assert g1 == R.G1_GEN and g2 == R.G2_GEN
    n1 = tuple(int(c, 16) for c in cfg["nested_pairing_parameters"]["generator_g1"])
    assert (n1[1] ** 2 - n1[0] ** 3 - 1) % R.BLS_Q == 0 and R.ec_mul(R.BLS_R, n1, R.BLS_Q) is None
    (x1, x0), (y1, y0) = cfg["nested_pairing_parameters"]["generator_g2"]            # c1 first, as in the fixtures
    assert ((int(x0, 16), int(x1, 16)), (int(y0, 16), int(y1, 16))) == R.BLS_G2_GEN
    assert client.get_verification_key() == prover.vk_json


def test_client_script_flow(live):
    client, prover, service = live
    app_vk = golden("dummy_app/vk.json")
    h = client.get_nested_verification_key_hash(app_vk)
    assert int(h, 16) == int(E.fr_to_json(prover.nested_vk_hash(E.nested_verification_key_from_json(app_vk))), 16)
    assert client.register_application(app_vk, "dummy_app") == h
    with pytest.raises(grpc.RpcError) as e:                        # scripts/test-client:53-55: re-registration fails
        client.register_application(app_vk, "dummy_app")
    assert _details(e) == (grpc.StatusCode.INVALID_ARGUMENT, "application already registered")
    with pytest.raises(grpc.RpcError) as e:
        client.get_aggregated_transaction("dummy_app")
    assert _details(e) == (grpc.StatusCode.INVALID_ARGUMENT, "insufficient entries in pool")
    txs = [golden("dummy_app/extproof%d.json" % k) for k in (3, 1, 4, 2)]      # submitted out of fee order
    for tx in txs:
        client.submit_nested_transaction(tx)
    assert service.pools["dummy_app"].tx_pool_size() == 4
    by_fee = sorted(txs, key=lambda t: -t["fee_in_wei"])
    for b in range(2):
        batch = client.get_aggregated_transaction("dummy_app")
        assert batch["app_name"] == "dummy_app"
        want = by_fee[2 * b:2 * b + 2]                                # fees 12, 11 then 10, 9 (nested_transaction.tcc:78-83)
        norm = lambda h_: h_[2:] if h_.startswith("0x") else h_
        assert batch["nested_parameters"] == [norm(t["parameters"]).rjust(len(norm(t["parameters"])) + len(norm(t["parameters"])) % 2, "0") for t in want]
        assert [int(x, 16) for x in batch["ext_proof"]["inputs"][2:]] == [int(t["extended_proof"]["inputs"][0], 16) for t in want]
        vk_limbs, proofs, inputs = prover.calls[b]
        assert (vk_limbs == E.nested_verification_key_from_json(app_vk)).all()
        for got, t in zip(proofs, want):
            assert (got == E.nested_extended_proof_from_json(t["extended_proof"])[0]).all()
        # the response decodes with the aggregated-transaction codec of the client side
        name, proof, inp, params = E.aggregated_transaction_from_json(batch)
        assert name == "dummy_app" and proof.shape == (72,) and len(params) == 2
    with pytest.raises(grpc.RpcError) as e:                        # scripts/test-client:94-96: no third batch
        client.get_aggregated_transaction("dummy_app")
    assert _details(e) == (grpc.StatusCode.INVALID_ARGUMENT, "insufficient entries in pool")


def test_error_paths(live):
    client, _, _ = live
    tx = golden("dummy_app/extproof1.json")
    with pytest.raises(grpc.RpcError) as e:                        # unknown application: std::map::at throws (aggregator_server.cpp:248)
        client.submit_nested_transaction(tx)
    assert e.value.code() == grpc.StatusCode.INVALID_ARGUMENT
    client.register_application(golden("dummy_app/vk.json"), "dummy_app")
    bad = json.loads(json.dumps(tx))
    bad["extended_proof"]["inputs"] = bad["extended_proof"]["inputs"] * 2
    with pytest.raises(grpc.RpcError) as e:
        client.submit_nested_transaction(bad)
    assert _details(e) == (grpc.StatusCode.INVALID_ARGUMENT, "invalid number of inputs")
    short_vk = json.loads(json.dumps(golden("dummy_app/vk.json")))
    short_vk["ABC"] = short_vk["ABC"][:1]
    with pytest.raises(grpc.RpcError) as e:
        client.register_application(short_vk, "other")
    assert e.value.code() == grpc.StatusCode.INVALID_ARGUMENT
    one = json.loads(json.dumps(tx))
    client.submit_nested_transaction(one)                          # a single queued transaction is not a batch
    with pytest.raises(grpc.RpcError) as e:
        client.get_aggregated_transaction("dummy_app")
    assert e.value.details() == "insufficient entries in pool"


def test_pool_orders_by_fee_and_only_returns_whole_batches():
    pool = S.ApplicationPool("app", golden("dummy_app/vk.json"))
    assert pool.get_next_batch() == []
    for fee in (3, 9, 1):
        pool.add_tx({"fee_in_wei": fee})
    assert [t["fee_in_wei"] for t in pool.get_next_batch()] == [9, 3]
    assert pool.tx_pool_size() == 1 and pool.get_next_batch() == []


def test_malformed_proof_is_refused_at_submission_and_a_failed_batch_goes_back(live):
    """(1) A nested proof with a point off its curve is refused by SubmitNestedTransaction - queued, it would be batched with another
    user's transaction and take it down with it.  (2) When the prover fails for a TRANSIENT reason, the batch that was popped under
    the lock returns to the pool in its old order instead of being lost."""
    from zecale_amd import zkhip
    client, prover, service = live
    agg = zkhip.AggregatorCircuit(S.BATCH_SIZE, S.NUM_INPUTS_PER_NESTED_PROOF)          # host code: the circuit's own curve checks
    prover.check_nested_proof = lambda vk_limbs, proof: bool(agg.check_inputs(vk_limbs, np.concatenate([proof] * S.BATCH_SIZE)))
    client.register_application(golden("dummy_app/vk.json"), "dummy_app")
    bad = json.loads(json.dumps(golden("dummy_app/extproof1.json")))
    y = int(bad["extended_proof"]["proof"]["a"][1], 16)
    bad["extended_proof"]["proof"]["a"][1] = hex(y ^ 2)                                   # A leaves its curve
    with pytest.raises(grpc.RpcError) as e:
        client.submit_nested_transaction(bad)
    assert _details(e) == (grpc.StatusCode.INVALID_ARGUMENT, "nested proof has a point that is not on its curve")
    assert service.pools["dummy_app"].tx_pool_size() == 0
    for k in (1, 2, 3):
        client.submit_nested_transaction(golden("dummy_app/extproof%d.json" % k))
    good_prove = prover.prove

    def failing(*a):
        raise S.TransientProverError("device lost")
    prover.prove = failing
    with pytest.raises(grpc.RpcError) as e:
        client.get_aggregated_transaction("dummy_app")
    assert e.value.details() == "device lost"
    assert service.pools["dummy_app"].tx_pool_size() == 3                                 # nothing was dropped
    prover.prove = good_prove
    batch = client.get_aggregated_transaction("dummy_app")                               # and the order is the old one: fees 11, 10
    fees = sorted((golden("dummy_app/extproof%d.json" % k) for k in (1, 2, 3)), key=lambda t: -t["fee_in_wei"])
    assert [int(x, 16) for x in batch["ext_proof"]["inputs"][2:]] == [int(t["extended_proof"]["inputs"][0], 16) for t in fees[:2]]
    agg.free()


def test_a_batch_that_fails_for_good_does_not_block_the_pool(live):
    """a batch whose proof fails DETERMINISTICALLY (a degenerate nested proof makes witness generation return
    ZKHIP_ERR_ARG; prove()'s own input check raises ValueError) must not go back to the head of the fee-ordered queue, where every
    later call would pop it and fail again: the reference drops it (aggregator_server.cpp:283-340).  A transient failure is retried a
    bounded number of times, then set aside too.  Either way the pool makes progress."""
    from zecale_amd import zkhip
    client, prover, service = live
    client.register_application(golden("dummy_app/vk.json"), "dummy_app")
    for k in (1, 2, 3, 4):
        client.submit_nested_transaction(golden("dummy_app/extproof%d.json" % k))
    pool = service.pools["dummy_app"]
    good_prove, calls = prover.prove, []

    def poisoned(vk, proofs, inputs):                     # the two highest fees form a batch that can never be proved
        calls.append(1)
        if len(calls) == 1:
            raise zkhip.ZkhipError("zkhip error -1 (bad argument): degenerate nested proof", -1)
        return good_prove(vk, proofs, inputs)
    prover.prove = poisoned
    with pytest.raises(grpc.RpcError) as e:
        client.get_aggregated_transaction("dummy_app")
    assert "degenerate nested proof" in e.value.details()
    assert pool.tx_pool_size() == 2 and len(pool.quarantined) == 2                        # dropped, not requeued
    batch = client.get_aggregated_transaction("dummy_app")                               # the next call proves the NEXT batch
    assert len(batch["nested_parameters"]) == 2 and pool.tx_pool_size() == 0
    # a ValueError from the prover's input check is a property of the batch as well
    for k in (1, 2):
        client.submit_nested_transaction(golden("dummy_app/extproof%d.json" % k))

    def refuses(*a):
        raise ValueError("nested proof or verification key has a point that is not on its curve")
    prover.prove = refuses
    with pytest.raises(grpc.RpcError):
        client.get_aggregated_transaction("dummy_app")
    assert pool.tx_pool_size() == 0 and len(pool.quarantined) == 4
    # a transient failure that never goes away: three attempts, then the batch is set aside and the pool moves on
    for k in (1, 2, 3, 4):
        client.submit_nested_transaction(golden("dummy_app/extproof%d.json" % k))
    attempts = []

    def flaky(vk, proofs, inputs):
        fee_batch = len(attempts) < 3                     # fails while the head batch is the one being proved
        attempts.append(1)
        if fee_batch:
            raise S.TransientProverError("device lost")
        return good_prove(vk, proofs, inputs)
    prover.prove = flaky
    for _ in range(3):
        with pytest.raises(grpc.RpcError):
            client.get_aggregated_transaction("dummy_app")
    assert pool.tx_pool_size() == 2 and len(pool.quarantined) == 6                        # the head batch is out after its third failure
    assert len(client.get_aggregated_transaction("dummy_app")["nested_parameters"]) == 2
    assert S.is_transient_failure(zkhip.ZkhipError("x", -3)) and S.is_transient_failure(zkhip.ZkhipError("x", -4))
    assert not S.is_transient_failure(zkhip.ZkhipError("x", -1)) and not S.is_transient_failure(RuntimeError("x"))
    prover.prove = good_prove


def test_handler_pool_is_as_deep_as_the_prover():
    class P(StubProver):
        gpu_slots = 14
    server, port, _ = S.serve(P(), "127.0.0.1:0")
    try:
        assert server._state.thread_pool._max_workers >= 14 + 4
    finally:
        server.stop(0)
