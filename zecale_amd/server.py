"""The `zecale_proto.Aggregator` gRPC surface in front of the GPU wrapping prover.

Mirrors the reference service class, handler by handler:
  service / messages      /root/reference/proto/zecale/api/aggregator.proto:9-79
  handler semantics       aggregator_server/aggregator_server.cpp:130-348 (exceptions -> INVALID_ARGUMENT with e.what(), :338-345)
  pool                    libzecale/core/application_pool.tcc:35-63 (fee-ordered priority queue; a batch only when NumProofs are queued)
  start-up                aggregator_server.cpp:418-527 (load the keypair file or run the trusted setup and write it), :390-416 (0.0.0.0:50052,
                          insecure, synchronous)
  client side             client/zecale/core/aggregator_client.py:18-90 (AggregatorClient below keeps its method names)

The message classes are built at import time from descriptors written out below (grpc_tools / protoc are not in this image).
The three zeth messages the service imports (zeth/api/snark_messages.proto, ec_group_messages.proto) live in the reference's
empty submodule: their shapes are restated from the JSON shapes the fixtures pin (every coordinate a JSON string,
point lists a JSON array) and from recollection of zeth's public API - field NUMBERS are therefore WIRE-UNVERIFIED against a
reference build.  Everything else (service name, method names, zecale messages and their numbers) is read from the tree.

Where the reference has no locking around its pools and its circuit (aggregator_server.cpp:112-118, a synchronous server with a
thread pool), this service takes one mutex around the pools; proofs themselves go through the streaming prover, which is
thread-safe, so several GenerateAggregatedTransaction calls prove side by side on the GPU.
"""
import heapq
import itertools
import json
import os
import threading
from concurrent import futures

import grpc
import numpy as np
from google.protobuf import descriptor_pb2, descriptor_pool, empty_pb2, message_factory

from . import encoding as E

SERVICE = "zecale_proto.Aggregator"
BATCH_SIZE = 2                      # aggregator_server.cpp:71
NUM_INPUTS_PER_NESTED_PROOF = 1     # aggregator_server.cpp:72
DEFAULT_ENDPOINT = "0.0.0.0:50052"  # aggregator_server.cpp:393


# ---------------------------------------------------------------------------------------------- descriptors
def _msg(fdp, name, fields, oneofs=()):
    m = fdp.message_type.add()
    m.name = name
    for o in oneofs:
        m.oneof_decl.add().name = o
    T = descriptor_pb2.FieldDescriptorProto
    for f in fields:
        fname, number, ftype = f[0], f[1], f[2]
        fd = m.field.add()
        fd.name, fd.number = fname, number
        fd.label = T.LABEL_REPEATED if (len(f) > 3 and f[3] == "repeated") else T.LABEL_OPTIONAL
        if ftype in ("string", "bytes", "int32"):
            fd.type = {"string": T.TYPE_STRING, "bytes": T.TYPE_BYTES, "int32": T.TYPE_INT32}[ftype]
        else:
            fd.type, fd.type_name = T.TYPE_MESSAGE, ftype
        if len(f) > 3 and isinstance(f[3], int):
            fd.oneof_index = f[3]
    return m


def _build_pool():
    pool = descriptor_pool.DescriptorPool()
    pool.AddSerializedFile(empty_pb2.DESCRIPTOR.serialized_pb)
    # zeth/api/ec_group_messages.proto - [UPSTREAM-RECALL], wire-unverified: coordinates are JSON-encoded strings
    ec = descriptor_pb2.FileDescriptorProto(name="zeth/api/ec_group_messages.proto", package="zeth_proto", syntax="proto3")
    _msg(ec, "Group1Point", [("x_coord", 1, "string"), ("y_coord", 2, "string")])
    _msg(ec, "Group2Point", [("x_coord", 1, "string"), ("y_coord", 2, "string")])
    _msg(ec, "PairingParameters", [("name", 1, "string"), ("r", 2, "string"), ("q", 3, "string"),
                                   ("generator_g1", 4, ".zeth_proto.Group1Point"), ("generator_g2", 5, ".zeth_proto.Group2Point")])
    pool.Add(ec)
    # zeth/api/snark_messages.proto - [UPSTREAM-RECALL], wire-unverified
    sn = descriptor_pb2.FileDescriptorProto(name="zeth/api/snark_messages.proto", package="zeth_proto", syntax="proto3",
                                            dependency=["zeth/api/ec_group_messages.proto"])
    _msg(sn, "VerificationKeyPGHR13", [("a", 1, ".zeth_proto.Group2Point"), ("b", 2, ".zeth_proto.Group1Point"), ("c", 3, ".zeth_proto.Group2Point"),
                                       ("gamma", 4, ".zeth_proto.Group2Point"), ("gamma_beta_g1", 5, ".zeth_proto.Group1Point"),
                                       ("gamma_beta_g2", 6, ".zeth_proto.Group2Point"), ("z", 7, ".zeth_proto.Group2Point"), ("ic", 8, "string")])
    _msg(sn, "VerificationKeyGROTH16", [("alpha_g1", 1, ".zeth_proto.Group1Point"), ("beta_g2", 2, ".zeth_proto.Group2Point"),
                                        ("delta_g2", 3, ".zeth_proto.Group2Point"), ("abc_g1", 4, "string")])
    _msg(sn, "VerificationKey", [("pghr13_verification_key", 1, ".zeth_proto.VerificationKeyPGHR13", 0),
                                 ("groth16_verification_key", 2, ".zeth_proto.VerificationKeyGROTH16", 0)], oneofs=("VK",))
    _msg(sn, "ExtendedProofPGHR13", [("a", 1, ".zeth_proto.Group1Point"), ("a_p", 2, ".zeth_proto.Group1Point"), ("b", 3, ".zeth_proto.Group2Point"),
                                     ("b_p", 4, ".zeth_proto.Group1Point"), ("c", 5, ".zeth_proto.Group1Point"), ("c_p", 6, ".zeth_proto.Group1Point"),
                                     ("h", 7, ".zeth_proto.Group1Point"), ("k", 8, ".zeth_proto.Group1Point"), ("inputs", 9, "string")])
    _msg(sn, "ExtendedProofGROTH16", [("a", 1, ".zeth_proto.Group1Point"), ("b", 2, ".zeth_proto.Group2Point"), ("c", 3, ".zeth_proto.Group1Point"),
                                      ("inputs", 4, "string")])
    _msg(sn, "ExtendedProof", [("pghr13_extended_proof", 1, ".zeth_proto.ExtendedProofPGHR13", 0),
                               ("groth16_extended_proof", 2, ".zeth_proto.ExtendedProofGROTH16", 0)], oneofs=("EP",))
    pool.Add(sn)
    # zecale/api/aggregator.proto - [REF] proto/zecale/api/aggregator.proto:43-79 (numbers as in the tree)
    ag = descriptor_pb2.FileDescriptorProto(name="zecale/api/aggregator.proto", package="zecale_proto", syntax="proto3",
                                            dependency=["zeth/api/snark_messages.proto", "zeth/api/ec_group_messages.proto", "google/protobuf/empty.proto"])
    _msg(ag, "AggregatorConfiguration", [("nested_snark_name", 1, "string"), ("wrapper_snark_name", 2, "string"),
                                         ("nested_pairing_parameters", 3, ".zeth_proto.PairingParameters"),
                                         ("wrapper_pairing_parameters", 4, ".zeth_proto.PairingParameters")])
    _msg(ag, "VerificationKeyHash", [("hash", 1, "string")])
    _msg(ag, "ApplicationDescription", [("application_name", 1, "string"), ("vk", 2, ".zeth_proto.VerificationKey")])
    _msg(ag, "NestedTransaction", [("application_name", 1, "string"), ("extended_proof", 2, ".zeth_proto.ExtendedProof"),
                                   ("parameters", 3, "bytes"), ("fee_in_wei", 4, "int32")])
    _msg(ag, "AggregatedTransactionRequest", [("application_name", 1, "string")])
    _msg(ag, "AggregatedTransaction", [("application_name", 1, "string"), ("extended_proof", 2, ".zeth_proto.ExtendedProof"),
                                       ("nested_parameters", 3, "bytes", "repeated")])
    svc = ag.service.add()
    svc.name = "Aggregator"
    for name, req, resp in RPCS:
        m = svc.method.add()
        m.name, m.input_type, m.output_type = name, "." + req, "." + resp
    pool.Add(ag)
    return pool


# aggregator.proto:9-41
RPCS = [("GetConfiguration", "google.protobuf.Empty", "zecale_proto.AggregatorConfiguration"),
        ("GetVerificationKey", "google.protobuf.Empty", "zeth_proto.VerificationKey"),
        ("GetNestedVerificationKeyHash", "zeth_proto.VerificationKey", "zecale_proto.VerificationKeyHash"),
        ("RegisterApplication", "zecale_proto.ApplicationDescription", "zecale_proto.VerificationKeyHash"),
        ("SubmitNestedTransaction", "zecale_proto.NestedTransaction", "google.protobuf.Empty"),
        ("GenerateAggregatedTransaction", "zecale_proto.AggregatedTransactionRequest", "zecale_proto.AggregatedTransaction")]
POOL = _build_pool()


def message_class(full_name):
    if full_name == "google.protobuf.Empty":
        return empty_pb2.Empty
    return message_factory.GetMessageClass(POOL.FindMessageTypeByName(full_name))


# ---------------------------------------------------------------------------------------------- JSON <-> proto
# libzeth encodes every coordinate with field_element_to_json: a JSON value (quoted hex string, or an array of them for Fq2);
# lists (ABC, inputs) are JSON arrays.  These helpers go between the fixtures' JSON shapes (encoding.py) and the messages.
def _point_to_proto(msg, p):
    msg.x_coord, msg.y_coord = json.dumps(p[0]), json.dumps(p[1])


def _point_from_proto(msg):
    return [json.loads(msg.x_coord), json.loads(msg.y_coord)]


def verification_key_to_proto(vk_json):
    m = message_class("zeth_proto.VerificationKey")()
    g = m.groth16_verification_key
    _point_to_proto(g.alpha_g1, vk_json["alpha"]); _point_to_proto(g.beta_g2, vk_json["beta"]); _point_to_proto(g.delta_g2, vk_json["delta"])
    g.abc_g1 = json.dumps(vk_json["ABC"])
    return m


def verification_key_from_proto(m):
    if m.WhichOneof("VK") != "groth16_verification_key":
        raise ValueError("expected a GROTH16 verification key")
    g = m.groth16_verification_key
    return {"alpha": _point_from_proto(g.alpha_g1), "beta": _point_from_proto(g.beta_g2), "delta": _point_from_proto(g.delta_g2),
            "ABC": json.loads(g.abc_g1)}


def extended_proof_to_proto(ep_json):
    m = message_class("zeth_proto.ExtendedProof")()
    g = m.groth16_extended_proof
    _point_to_proto(g.a, ep_json["proof"]["a"]); _point_to_proto(g.b, ep_json["proof"]["b"]); _point_to_proto(g.c, ep_json["proof"]["c"])
    g.inputs = json.dumps(ep_json["inputs"])
    return m


def extended_proof_from_proto(m):
    if m.WhichOneof("EP") != "groth16_extended_proof":
        raise ValueError("expected a GROTH16 extended proof")
    g = m.groth16_extended_proof
    return {"proof": {"a": _point_from_proto(g.a), "b": _point_from_proto(g.b), "c": _point_from_proto(g.c)}, "inputs": json.loads(g.inputs)}


def _hex(x, digits):
    return "0x" + format(x, "0%dx" % digits)


def pairing_parameters(which):
    """libzeth::pairing_parameters_to_proto<ppT> for the two curves of the server (aggregator_server.cpp:40-48); constants:
    SURVEY App. A.1 (client/test_commands/test_bw6_761_groth16_contract.py:26-35) and the BLS12-377 generators the fixtures confirm."""
    m = message_class("zeth_proto.PairingParameters")()
    if which == "wrapper":
        from .csrc_constants import BW6_G1, BW6_G2
        m.name, m.r, m.q = "bw6-761", _hex(E.R_MOD, 96), _hex(E.Q_MOD, 192)
        _point_to_proto(m.generator_g1, [_hex(BW6_G1[0], 192), _hex(BW6_G1[1], 192)])
        _point_to_proto(m.generator_g2, [_hex(BW6_G2[0], 192), _hex(BW6_G2[1], 192)])
    else:
        from .csrc_constants import BLS_G1, BLS_G2, BLS_R
        m.name, m.r, m.q = "bls12-377", _hex(BLS_R, 64), _hex(E.R_MOD, 96)
        _point_to_proto(m.generator_g1, [_hex(BLS_G1[0], 96), _hex(BLS_G1[1], 96)])
        _point_to_proto(m.generator_g2, [[_hex(BLS_G2[0][1], 96), _hex(BLS_G2[0][0], 96)], [_hex(BLS_G2[1][1], 96), _hex(BLS_G2[1][0], 96)]])
    return m


# ---------------------------------------------------------------------------------------------- pool
class ApplicationPool:
    """libzecale::application_pool<npp, nsnark, NumProofs> (application_pool.hpp:21-64): the transactions of one application,
    highest fee first (nested_transaction::operator< compares fee_wei, nested_transaction.tcc:78-83)."""

    def __init__(self, name, vk_json, num_proofs=BATCH_SIZE):
        self.name, self.vk_json, self.num_proofs = name, vk_json, num_proofs
        self.vk_limbs = E.nested_verification_key_from_json(vk_json)
        self._heap, self._order = [], itertools.count()
        self.quarantined = []          # transactions of batches that failed for good (see requeue / drop)

    def add_tx(self, tx):
        heapq.heappush(self._heap, (-tx["fee_in_wei"], next(self._order), tx))

    def tx_pool_size(self):
        return len(self._heap)

    def pop_batch_entries(self):
        """Whole batches only (application_pool.tcc:49-63): [] unless NumProofs transactions are queued.  Heap entries
        (-fee, arrival order, tx), so that a batch whose proof failed can go back in its old place (requeue)."""
        if len(self._heap) < self.num_proofs:
            return []
        return [heapq.heappop(self._heap) for _ in range(self.num_proofs)]

    def get_next_batch(self):
        return [e[2] for e in self.pop_batch_entries()]

    def requeue(self, entries, max_failures=3):
        """A batch that was popped but not proved because of a TRANSIENT prover / device error goes back into the queue with its fee
        order and arrival order - at most `max_failures` times per transaction: after that it is set aside (`quarantined`) like a
        batch that failed deterministically, so that one poisonous high-fee transaction cannot block the pool for ever.
        Returns the number of transactions that went back."""
        back = 0
        for e in entries:
            e[2]["failures"] = e[2].get("failures", 0) + 1
            if e[2]["failures"] >= max_failures:
                self.quarantined.append(e[2])
            else:
                heapq.heappush(self._heap, e)
                back += 1
        return back

    def drop(self, entries):
        """A batch whose proof failed for a reason that will not go away (malformed or degenerate input): the reference loses it
        (aggregator_server.cpp:283-340 pops the batch before proving and never puts it back); kept aside for the operator."""
        self.quarantined.extend(e[2] for e in entries)


# ---------------------------------------------------------------------------------------------- provers
class TransientProverError(RuntimeError):
    """A prover failure that says nothing about the batch (lost device, exhausted memory): the batch may be retried."""


def is_transient_failure(e):
    """Which prover failures keep a batch in its pool: the library's HIP / state / device codes and TransientProverError.
    ValueError (input checks), ZKHIP_ERR_ARG (witness generation met a degenerate point, sizes do not match) and everything
    unknown are properties of the batch."""
    if isinstance(e, TransientProverError):
        return True
    return getattr(e, "code", None) in (-2, -3, -4)                # ZKHIP_ERR_NO_DEVICE, ZKHIP_ERR_HIP, ZKHIP_ERR_STATE


class GpuProver:
    """The wrapping prover behind the service: circuit, keypair (file or fresh setup), HBM-resident key, streaming pipeline."""
    snark_name = "GROTH16"

    def __init__(self, keypair_file=None, device=0, gpu_slots=32, witness_workers=10, gpu_witness=False, devices=None, hybrid_witness=False):
        """devices: the GPUs of the node this ONE server process drives (the reference server is one process that owns its prover:
        aggregator_server.cpp:106-118, 390-416) - a resident copy of the key and a streaming pipeline on each, behind a dispatcher
        (zkhip_dispatcher_*); an index may repeat (two contexts on one GPU).  None / one entry: that GPU alone (`device`)."""
        from . import zkhip
        self.zk = zkhip
        devices = [int(d) for d in devices] if devices else [int(device)]
        device = devices[0]
        self.devices = devices
        self.gpu_slots = gpu_slots * len(devices)        # (the handler pool is as deep as the node has proofs in flight)
        zkhip.init(device)
        self.agg = zkhip.AggregatorCircuit(BATCH_SIZE, NUM_INPUTS_PER_NESTED_PROOF)
        desc = zkhip.r1cs_desc_from_aggregator(self.agg)
        if keypair_file and os.path.exists(keypair_file):                       # aggregator_server.cpp:483-495
            self.kp = zkhip.Keypair.read(keypair_file)
        else:
            self.kp = zkhip.Keypair(desc, *[zkhip.fr_random() for _ in range(4)])
            if keypair_file:                                                   # :497-513
                d = os.path.dirname(os.path.abspath(keypair_file))
                os.makedirs(d, exist_ok=True)
                self.kp.write(keypair_file)
        self.vk = self.kp.vk()
        if self.vk["ABC"].shape[0] != self.agg.num_primary_inputs() + 1:      # the server's "invalid VK" check (:490, :504)
            raise ValueError("invalid VK")
        # a server proves a stream of batches: the larger kind of window table pays.  The option travels with
        # THIS key (zkhip_key_opts), not with the process: another key loaded on another thread keeps its own.
        opts = zkhip.key_opts(table_naf=True if os.environ.get("ZKHIP_TABLE_NAF") is None else None)
        if len(devices) > 1:
            self.crs = None                               # (the dispatcher uploads a copy of the key to every entry of the list)
            self.pipe = zkhip.AggregatorDispatcher(self.agg, self.kp, devices, opts, gpu_slots=gpu_slots, witness_workers=witness_workers,
                                                   gpu_witness=gpu_witness or hybrid_witness, hybrid=hybrid_witness)
        else:
            self.crs = self.kp.upload_crs(opts)
            self.pipe = zkhip.AggregatorPipeline(self.agg, self.crs, gpu_slots=gpu_slots, witness_workers=witness_workers,
                                                 gpu_witness=gpu_witness or hybrid_witness, hybrid=hybrid_witness)

    def verification_key_json(self):
        return E.verification_key_to_json(self.vk)

    def nested_vk_hash(self, nested_vk_limbs):
        return self.zk.aggregator_vk_hash(nested_vk_limbs, NUM_INPUTS_PER_NESTED_PROOF)

    def register_application(self, nested_vk_limbs):
        """RegisterApplication's part in the prover (aggregator_server.cpp:170-235 stores the key): the streaming prover computes the
        key's constants now (zkhip_aggregator_app: the key's share of every assignment and of four of the five MSMs), so that the
        application's first batch does not pay for them.  A degenerate key gets no handle and is proved by the plain path."""
        try:
            self.pipe.register_app(nested_vk_limbs)
            return True
        except self.zk.ZkhipError:
            return False

    def check_nested_proof(self, nested_vk_limbs, proof_limbs):
        """Well-formedness of ONE nested proof at submission time (libsnark's proof.is_well_formed(): every point on its curve).
        Host code.  A malformed transaction is refused before it can sit in a batch with somebody else's honest one."""
        return bool(self.agg.check_inputs(nested_vk_limbs, np.concatenate([proof_limbs] * BATCH_SIZE)))

    def prove(self, nested_vk_limbs, proofs, inputs):
        """proofs: BATCH_SIZE x 48 limbs; inputs: BATCH_SIZE x k x 6 limbs -> extended proof JSON of the wrapping proof."""
        if not self.agg.check_inputs(nested_vk_limbs, np.concatenate(proofs)):
            raise ValueError("nested proof or verification key has a point that is not on its curve")
        t = self.pipe.submit(nested_vk_limbs, np.concatenate(proofs), np.concatenate(inputs), self.zk.fr_random(), self.zk.fr_random())
        prim, proof = self.pipe.wait(t)
        return E.extended_proof_to_json(proof, prim)

    def close(self):
        self.pipe.free()
        if self.crs is not None:
            self.crs.free()
        self.kp.free(); self.agg.free()


# ---------------------------------------------------------------------------------------------- service
class AggregatorService:
    """Handlers of zecale_proto.Aggregator over a prover object (GpuProver, or a stub in the CPU tests)."""

    def __init__(self, prover):
        self.prover = prover
        self.pools = {}
        self.mu = threading.Lock()

    # --- RPCs (aggregator_server.cpp:130-348) ---
    def GetConfiguration(self, request, context):
        cfg = message_class("zecale_proto.AggregatorConfiguration")()
        cfg.nested_snark_name = cfg.wrapper_snark_name = self.prover.snark_name          # proto_utils.tcc:18-28
        cfg.nested_pairing_parameters.CopyFrom(pairing_parameters("nested"))
        cfg.wrapper_pairing_parameters.CopyFrom(pairing_parameters("wrapper"))
        return cfg

    def GetVerificationKey(self, request, context):
        return self._guard(context, lambda: verification_key_to_proto(self.prover.verification_key_json()),
                           message_class("zeth_proto.VerificationKey"))

    def _hash_response(self, vk_json):
        limbs = E.nested_verification_key_from_json(vk_json)
        if (limbs.size - 60) // 12 != NUM_INPUTS_PER_NESTED_PROOF + 1:
            raise ValueError("nested verification key has the wrong number of ABC elements")
        r = message_class("zecale_proto.VerificationKeyHash")()
        r.hash = json.dumps(E.fr_to_json(self.prover.nested_vk_hash(limbs)))              # field_element_to_json: a JSON string
        return r, limbs

    def GetNestedVerificationKeyHash(self, request, context):
        return self._guard(context, lambda: self._hash_response(verification_key_from_proto(request))[0],
                           message_class("zecale_proto.VerificationKeyHash"))

    def RegisterApplication(self, request, context):
        def run():
            name = request.application_name
            with self.mu:
                if name in self.pools:
                    raise ValueError("application already registered")                   # aggregator_server.cpp:186-190
                vk_json = verification_key_from_proto(request.vk)
                resp, limbs = self._hash_response(vk_json)
                self.pools[name] = ApplicationPool(name, vk_json)
            register = getattr(self.prover, "register_application", None)              # (outside the lock: ~0.2 s on the GPU)
            if register is not None:
                register(limbs)
            return resp
        return self._guard(context, run, message_class("zecale_proto.VerificationKeyHash"))

    def SubmitNestedTransaction(self, request, context):
        def run():
            with self.mu:
                pool = self._pool(request.application_name)
                ep = extended_proof_from_proto(request.extended_proof)
                if len(ep["inputs"]) != NUM_INPUTS_PER_NESTED_PROOF:
                    raise ValueError("invalid number of inputs")                         # aggregator_server.cpp:254-257
                proof, inputs = E.nested_extended_proof_from_json(ep)
                # refuse a malformed proof HERE: once queued it would be batched with another user's transaction and take it down
                # with it when the batch fails (the reference checks well-formedness when it decodes the proof)
                check = getattr(self.prover, "check_nested_proof", None)
                if check is not None and not check(pool.vk_limbs, proof):
                    raise ValueError("nested proof has a point that is not on its curve")
                pool.add_tx({"proof": proof, "inputs": inputs, "parameters": bytes(request.parameters),
                             "fee_in_wei": int(request.fee_in_wei) & 0xFFFFFFFF})         # uint32_t(fee), proto_utils.tcc:44
            return empty_pb2.Empty()
        return self._guard(context, run, empty_pb2.Empty)

    def GenerateAggregatedTransaction(self, request, context):
        def run():
            name = request.application_name
            with self.mu:
                pool = self._pool(name)
                entries = pool.pop_batch_entries()
                if not entries:
                    raise RuntimeError("insufficient entries in pool")                   # aggregator_server.cpp:298-300
                batch = [e[2] for e in entries]
                vk_limbs = pool.vk_limbs
            try:
                ep = self.prover.prove(vk_limbs, [tx["proof"] for tx in batch], [tx["inputs"] for tx in batch])
            except Exception as e:                                                        # noqa: BLE001
                # The batch was taken out of the pool under the lock.  A TRANSIENT failure (the device, the HIP runtime, the
                # library's state) must not lose the transactions in it: they go back in their old order, a bounded number of
                # times.  Anything else is a property of the batch (a degenerate nested proof, a bad input count): putting it back
                # at the head of a fee-ordered queue would make every later call pop it and fail again - the reference drops it.
                with self.mu:
                    if is_transient_failure(e):
                        pool.requeue(entries)
                    else:
                        pool.drop(entries)
                raise
            resp = message_class("zecale_proto.AggregatedTransaction")()
            resp.application_name = name
            resp.extended_proof.CopyFrom(extended_proof_to_proto(ep))
            for tx in batch:
                resp.nested_parameters.append(tx["parameters"])
            return resp
        return self._guard(context, run, message_class("zecale_proto.AggregatedTransaction"))

    # --- helpers ---
    def _pool(self, name):
        if name not in self.pools:
            raise KeyError("map::at")                                                    # std::map::at throws; e.what() reaches the client
        return self.pools[name]

    @staticmethod
    def _guard(context, fn, response_class):
        try:
            return fn()
        except Exception as e:          # noqa: BLE001 - every std::exception becomes INVALID_ARGUMENT with its text (:338-345)
            msg = e.args[0] if (isinstance(e, KeyError) and e.args) else str(e)
            context.set_code(grpc.StatusCode.INVALID_ARGUMENT)
            context.set_details(str(msg))
            return response_class()

    def generic_handler(self):
        handlers = {}
        for name, req, resp in RPCS:
            handlers[name] = grpc.unary_unary_rpc_method_handler(getattr(self, name), request_deserializer=message_class(req).FromString,
                                                                 response_serializer=message_class(resp).SerializeToString)
        return grpc.method_handlers_generic_handler(SERVICE, handlers)


def serve(prover, endpoint=DEFAULT_ENDPOINT, max_workers=None):
    """RunServer (aggregator_server.cpp:390-416): insecure, listens on `endpoint`; returns (server, bound port).
    A GenerateAggregatedTransaction handler blocks for the length of a proof, so the handler pool is at least as deep as the prover
    has slots (else the slots could never all be busy) plus a few threads for the short RPCs."""
    if max_workers is None:
        max_workers = max(8, int(getattr(prover, "gpu_slots", 0)) + 4)
    server = grpc.server(futures.ThreadPoolExecutor(max_workers=max_workers))
    service = AggregatorService(prover)
    server.add_generic_rpc_handlers((service.generic_handler(),))
    port = server.add_insecure_port(endpoint)
    server.start()
    return server, port, service


# ---------------------------------------------------------------------------------------------- client
class AggregatorClient:
    """client/zecale/core/aggregator_client.py:18-90 over JSON dictionaries (the shapes of testdata/dummy_app/*.json)."""

    def __init__(self, endpoint):
        self.endpoint = endpoint

    def _call(self, name, request):
        req, resp = next((r, s) for n, r, s in RPCS if n == name)
        with grpc.insecure_channel(self.endpoint) as channel:
            fn = channel.unary_unary("/%s/%s" % (SERVICE, name), request_serializer=message_class(req).SerializeToString,
                                     response_deserializer=message_class(resp).FromString)
            return fn(request)

    def get_configuration(self):
        c = self._call("GetConfiguration", empty_pb2.Empty())
        pp = lambda p: {"name": p.name, "r": p.r, "q": p.q, "generator_g1": _point_from_proto(p.generator_g1), "generator_g2": _point_from_proto(p.generator_g2)}
        return {"nested_snark_name": c.nested_snark_name, "wrapper_snark_name": c.wrapper_snark_name,
                "nested_pairing_parameters": pp(c.nested_pairing_parameters), "wrapper_pairing_parameters": pp(c.wrapper_pairing_parameters)}

    def get_verification_key(self):
        return verification_key_from_proto(self._call("GetVerificationKey", empty_pb2.Empty()))

    def get_nested_verification_key_hash(self, vk_json):
        return json.loads(self._call("GetNestedVerificationKeyHash", verification_key_to_proto(vk_json)).hash)

    def register_application(self, vk_json, app_name):
        d = message_class("zecale_proto.ApplicationDescription")()
        d.application_name = app_name
        d.vk.CopyFrom(verification_key_to_proto(vk_json))
        return json.loads(self._call("RegisterApplication", d).hash)

    def submit_nested_transaction(self, tx_json):
        """tx_json: the shape of testdata/dummy_app/extproof1.json (app_name, extended_proof, parameters hex, fee_in_wei)."""
        t = message_class("zecale_proto.NestedTransaction")()
        t.application_name = tx_json["app_name"]
        t.extended_proof.CopyFrom(extended_proof_to_proto(tx_json["extended_proof"]))
        h = tx_json["parameters"]
        h = h[2:] if h.startswith("0x") else h
        t.parameters = bytes.fromhex(h if len(h) % 2 == 0 else "0" + h)
        t.fee_in_wei = int(tx_json["fee_in_wei"])
        self._call("SubmitNestedTransaction", t)

    def get_aggregated_transaction(self, name):
        r = message_class("zecale_proto.AggregatedTransactionRequest")()
        r.application_name = name
        a = self._call("GenerateAggregatedTransaction", r)
        return {"app_name": a.application_name, "ext_proof": extended_proof_from_proto(a.extended_proof),
                "nested_parameters": [bytes(p).hex() for p in a.nested_parameters]}


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="aggregator_server on an MI355X (reference aggregator_server.cpp:418-527)")
    ap.add_argument("--keypair", "-k", default=os.path.join(os.environ.get("ZETH_SETUP_DIR", os.path.expanduser("~/zeth_setup")), "zecale_keypair.bin"),
                    help="file to load the keypair from (generated and written there when missing)")
    ap.add_argument("--endpoint", default=DEFAULT_ENDPOINT)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--devices", default="", help="comma list of GPUs this one process drives (replicas behind a dispatcher), e.g. 0,1,2,3,4,5,6,7; "
                                                  "overrides --device")
    ap.add_argument("--gpu-witness", action="store_true", help="generate the assignments on the GPU: half the host cores per GPU, 4 % fewer proofs/s "
                                                               "(round 5: 421 proofs/s on 2.2 cores against 439 on 4.1)")
    ap.add_argument("--hybrid-witness", action="store_true", help="host generators beside the GPU generator, taking what its launches leave: between the two "
                                                                   "modes in host cores, up to the host mode's proofs/s")
    args = ap.parse_args(argv)
    print("[INFO] Init params of both curves")
    prover = GpuProver(args.keypair, device=args.device, gpu_witness=args.gpu_witness, hybrid_witness=args.hybrid_witness,
                       devices=[int(x) for x in args.devices.split(",")] if args.devices else None)
    print("[INFO] Circuit has %d constraints" % prover.agg.num_constraints)
    print("[INFO] Setup successful, starting the server...")
    server, port, _ = serve(prover, args.endpoint)
    print("[INFO] Server listening on %s" % args.endpoint)
    server.wait_for_termination()


if __name__ == "__main__":
    main()
