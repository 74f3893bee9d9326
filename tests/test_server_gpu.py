"""The gRPC service with the real prover behind it (zecale_amd.server.GpuProver): the flow of the reference's scripts/test-client:41-96
on a live loopback server, every aggregated transaction verified against the key GetVerificationKey returns - what the contract
does on-chain (contracts/ZecaleDispatcher.sol:79-169) - and the start-up path of aggregator_server.cpp:483-514 (the keypair file
written by the first start is loaded by the second)."""
import threading

import numpy as np
import pytest

from oracle import pyref as R
from tests.helpers import fr_int, golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("devices", [None, [0, 0]], ids=["one_gpu", "devices_0_0"])
def test_client_script_flow_on_the_gpu(zk, tmp_path, devices):
    """devices_0_0: the same flow with the server started as `--devices 0,0` - one process, a dispatcher over two contexts of GPU 0
    (how a one-GPU box rehearses the 8-GPU server)."""
    from zecale_amd import encoding as E
    from zecale_amd import server as S
    kpf = str(tmp_path / "zeth_setup" / "zecale_keypair.bin")
    prover = S.GpuProver(kpf, device=0, gpu_slots=2, witness_workers=2, devices=devices)
    assert prover.gpu_slots == (4 if devices else 2)
    server, port, service = S.serve(prover, "127.0.0.1:0", max_workers=4)
    try:
        client = S.AggregatorClient("127.0.0.1:%d" % port)
        vk_json = client.get_verification_key()
        vk = E.verification_key_from_json(vk_json)
        assert len(vk_json["ABC"]) == 5                               # 4 primary inputs (aggregator_server.cpp:490)
        app_vk = golden("dummy_app/vk.json")
        h = client.register_application(app_vk, "dummy_app")
        nvk = dict(alpha=tuple(int(c, 16) for c in app_vk["alpha"]),
                   beta=tuple((int(c[1], 16), int(c[0], 16)) for c in app_vk["beta"]), delta=tuple((int(c[1], 16), int(c[0], 16)) for c in app_vk["delta"]),
                   ABC=[tuple(int(c, 16) for c in p) for p in app_vk["ABC"]])
        assert int(h, 16) == R.nested_vk_hash(nvk) == int(client.get_nested_verification_key_hash(app_vk), 16)
        txs = {k: golden("dummy_app/extproof%d.json" % k) for k in (1, 2, 3, 4, 5, 6)}
        for k in (1, 2, 3, 4):
            client.submit_nested_transaction(txs[k])
        for first in (1, 3):                                          # fees 12, 11 | 10, 9: batches (1, 2) and (3, 4)
            batch = client.get_aggregated_transaction("dummy_app")
            name, proof, inputs, params = E.aggregated_transaction_from_json(batch)
            assert zk.groth16_verify(vk, inputs, proof)               # Groth16BW6_761.verify on the client side
            xs = [int(txs[first + d]["extended_proof"]["inputs"][0], 16) for d in (0, 1)]
            assert [fr_int(x) for x in inputs] == [int(h, 16), 3] + xs
            assert len(params) == 2
        # two requests at once: both batches are proved side by side by the streaming prover
        for k in (5, 6, 1, 2):
            client.submit_nested_transaction(txs[k])
        results, errors = [], []

        def worker():
            try:
                results.append(client.get_aggregated_transaction("dummy_app"))
            except Exception as e:      # noqa: BLE001
                errors.append(e)
        ths = [threading.Thread(target=worker) for _ in range(2)]
        [t.start() for t in ths]; [t.join() for t in ths]
        assert not errors and len(results) == 2
        seen = set()
        for batch in results:
            _, proof, inputs, _ = E.aggregated_transaction_from_json(batch)
            assert zk.groth16_verify(vk, inputs, proof) and fr_int(inputs[1]) == 3
            seen.add(tuple(fr_int(x) for x in inputs[2:]))
        assert seen == {(7, 8), (11, 12)}                             # fees 12, 11 (proofs 1, 2) and 8, 7 (proofs 5, 6)
        if devices:
            assert sum(prover.pipe.stats()) == 4 and min(prover.pipe.stats()) >= 1      # both entries of the list proved batches
    finally:
        server.stop(0)
        prover.close()
    # second start: the keypair file is loaded, the verification key is the same
    prover2 = S.GpuProver(kpf, device=0, gpu_slots=1, witness_workers=1)
    try:
        assert E.verification_key_to_json(prover2.vk) == vk_json
    finally:
        prover2.close()
