"""The reference's JSON encodings on either side of the prover, between JSON and the limb arrays the
C ABI takes (little-endian u64 limbs, Montgomery form):

  field element   "0x" + fixed-width big-endian hex of the canonical value: 192 digits for BW6-761 Fq, 96 for BW6-761 Fr =
                  BLS12-377 Fq (testdata/dummy_app/batch1.json:5,10; vk.json:2)
  G1 / BW6 G2     [x, y]
  BLS12-377 G2    [[x.c1, x.c0], [y.c1, y.c0]]            (c1 first)
  Groth16 vk      {"alpha": G1, "beta": G2, "delta": G2, "ABC": [G1 x (l+1)]}       (no gamma: libzeth's variant)
  proof           {"a": G1, "b": G2, "c": G1};  extended proof adds "inputs": [Fr ...]
  nested tx       {"app_name", "extended_proof", "parameters", "fee_in_wei"}        (client/zecale/core/nested_transaction.py:25-43)
  aggregated tx   {"app_name", "ext_proof", "nested_parameters"}                    (client/zecale/core/aggregated_transaction.py:23-38)

Pure Python (big integers); needs neither the library nor a device.  Writers of these files in the reference:
wsnark::verification_key_write_json (aggregator_server.cpp:185,223), extended_proof::write_json (:322)."""
import numpy as np

Q_MOD = 0x0122e824fb83ce0ad187c94004faff3eb926186a81d14688528275ef8087be41707ba638e584e91903cebaff25b423048689c8ed12f9fd9071dcd3dc73ebff2e98a116c25667a8f8160cf8aeeaf0a437e6913e6870000082f49d00000000008b
R_MOD = 0x01ae3a4617c510eac63b05c06ca1493b1a22d9f300f5138f1ef3622fba094800170b5d44300000008508c00000000001   # = BLS12-377 Fq
NESTED_FR_HEX_DIGITS = 64      # BLS12-377 Fr (253 bits) as written by the nested prover (extproof1.json:9)
_MASK = (1 << 64) - 1


def _to_limbs(x, mod, n):
    m = (x << (64 * n)) % mod
    return [(m >> (64 * i)) & _MASK for i in range(n)]


def _from_limbs(limbs, mod, n):
    v = 0
    for i, l in enumerate(np.asarray(limbs, dtype=np.uint64).reshape(-1).tolist()[:n]):
        v |= int(l) << (64 * i)
    return (v * pow(1 << (64 * n), -1, mod)) % mod


def _hex(x, digits):
    return "0x" + format(x, "0%dx" % digits)


def _int(s, mod):
    x = int(s, 16)
    if x >= mod:
        raise ValueError("field element out of range")
    return x


# ---- wrapping curve (BW6-761): Fq 12 limbs, Fr 6 limbs ---------------------------------------------------------------
def fq_to_json(limbs):
    return _hex(_from_limbs(limbs, Q_MOD, 12), 192)


def fq_from_json(s):
    return _to_limbs(_int(s, Q_MOD), Q_MOD, 12)


def fr_to_json(limbs):
    return _hex(_from_limbs(limbs, R_MOD, 6), 96)


def fr_from_json(s):
    return _to_limbs(_int(s, R_MOD), R_MOD, 6)


def point_to_json(aff):
    """BW6-761 G1 or G2 point (24 limbs, all-zero = infinity, written as [0, 0])."""
    a = np.asarray(aff, dtype=np.uint64).reshape(24)
    return [fq_to_json(a[:12]), fq_to_json(a[12:])]


def point_from_json(p):
    return np.array(fq_from_json(p[0]) + fq_from_json(p[1]), dtype=np.uint64)


def verification_key_to_json(vk):
    """vk: dict alpha, beta, delta (24 limbs each), ABC ((l+1) x 24 limbs) as zkhip.Keypair.vk() returns it."""
    return {"alpha": point_to_json(vk["alpha"]), "beta": point_to_json(vk["beta"]), "delta": point_to_json(vk["delta"]),
            "ABC": [point_to_json(p) for p in np.asarray(vk["ABC"], dtype=np.uint64).reshape(-1, 24)]}


def verification_key_from_json(j):
    return {"alpha": point_from_json(j["alpha"]), "beta": point_from_json(j["beta"]), "delta": point_from_json(j["delta"]),
            "ABC": np.array([point_from_json(p) for p in j["ABC"]], dtype=np.uint64).reshape(-1, 24)}


def proof_to_json(proof):
    """proof: 72 limbs a | b | c as zkhip_groth16_prove returns it."""
    p = np.asarray(proof, dtype=np.uint64).reshape(72)
    return {"a": point_to_json(p[:24]), "b": point_to_json(p[24:48]), "c": point_to_json(p[48:])}


def proof_from_json(j):
    return np.concatenate([point_from_json(j["a"]), point_from_json(j["b"]), point_from_json(j["c"])])


def extended_proof_to_json(proof, inputs):
    return {"proof": proof_to_json(proof), "inputs": [fr_to_json(x) for x in np.asarray(inputs, dtype=np.uint64).reshape(-1, 6)]}


def extended_proof_from_json(j):
    """-> (proof 72 limbs, inputs k x 6 limbs)"""
    return proof_from_json(j["proof"]), np.array([fr_from_json(x) for x in j["inputs"]], dtype=np.uint64).reshape(-1, 6)


def aggregated_transaction_to_json(app_name, proof, inputs, nested_parameters):
    """nested_parameters: list of hex strings, one per nested transaction of the batch, kept as text (the application's
    opaque payload; the reference fixture extproof6.json holds an odd number of hex digits, so it is not decoded here)."""
    return {"app_name": app_name, "ext_proof": extended_proof_to_json(proof, inputs), "nested_parameters": [str(p) for p in nested_parameters]}


def aggregated_transaction_from_json(j):
    proof, inputs = extended_proof_from_json(j["ext_proof"])
    return j["app_name"], proof, inputs, list(j["nested_parameters"])


# ---- nested curve (BLS12-377): Fq = BW6 Fr (6 limbs), G1 12 limbs, G2 24 limbs (x.c0, x.c1, y.c0, y.c1) -------------
def nested_g1_from_json(p):
    return fr_from_json(p[0]) + fr_from_json(p[1])


def nested_g1_to_json(limbs):
    a = np.asarray(limbs, dtype=np.uint64).reshape(12)
    return [fr_to_json(a[:6]), fr_to_json(a[6:])]


def nested_g2_from_json(p):
    (x1, x0), (y1, y0) = p                    # JSON order is [c1, c0]
    return fr_from_json(x0) + fr_from_json(x1) + fr_from_json(y0) + fr_from_json(y1)


def nested_g2_to_json(limbs):
    a = np.asarray(limbs, dtype=np.uint64).reshape(24)
    return [[fr_to_json(a[6:12]), fr_to_json(a[:6])], [fr_to_json(a[18:]), fr_to_json(a[12:18])]]


def nested_verification_key_from_json(j):
    """-> the flat limb array zkhip_aggregator_witness takes: alpha (12) | beta (24) | delta (24) | ABC (12 each)."""
    out = nested_g1_from_json(j["alpha"]) + nested_g2_from_json(j["beta"]) + nested_g2_from_json(j["delta"])
    for p in j["ABC"]:
        out += nested_g1_from_json(p)
    return np.array(out, dtype=np.uint64)


def nested_verification_key_to_json(limbs):
    a = np.asarray(limbs, dtype=np.uint64).reshape(-1)
    return {"alpha": nested_g1_to_json(a[:12]), "beta": nested_g2_to_json(a[12:36]), "delta": nested_g2_to_json(a[36:60]),
            "ABC": [nested_g1_to_json(a[60 + 12 * i:72 + 12 * i]) for i in range((a.size - 60) // 12)]}


def nested_extended_proof_from_json(j):
    """-> (proof limbs a (12) | b (24) | c (12), inputs k x 6 limbs: nested Fr values embedded in the wrapping Fr)."""
    pr = j["proof"]
    proof = np.array(nested_g1_from_json(pr["a"]) + nested_g2_from_json(pr["b"]) + nested_g1_from_json(pr["c"]), dtype=np.uint64)
    inputs = np.array([_to_limbs(int(x, 16), R_MOD, 6) for x in j["inputs"]], dtype=np.uint64).reshape(-1, 6)
    return proof, inputs


def nested_extended_proof_to_json(proof, inputs):
    a = np.asarray(proof, dtype=np.uint64).reshape(48)
    return {"proof": {"a": nested_g1_to_json(a[:12]), "b": nested_g2_to_json(a[12:36]), "c": nested_g1_to_json(a[36:])},
            "inputs": [_hex(_from_limbs(x, R_MOD, 6), NESTED_FR_HEX_DIGITS) for x in np.asarray(inputs, dtype=np.uint64).reshape(-1, 6)]}


def nested_transaction_from_json(j):
    """-> (app_name, proof limbs, input limbs, parameters hex string, fee_in_wei int)   (testdata/dummy_app/extproof1.json)"""
    proof, inputs = nested_extended_proof_from_json(j["extended_proof"])
    return j["app_name"], proof, inputs, str(j["parameters"]), int(j["fee_in_wei"])


def nested_transaction_to_json(app_name, proof, inputs, parameters, fee_in_wei):
    return {"app_name": app_name, "extended_proof": nested_extended_proof_to_json(proof, inputs), "parameters": str(parameters),
            "fee_in_wei": int(fee_in_wei)}
