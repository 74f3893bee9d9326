"""bench.py's JSON line ENDS with a short `summary`: the driver keeps the last 2,000 characters of the output, and
the metric - "wrapping proofs/sec + G1-MSM Mscalar/s ... vs CPU" - has more legs than the one `value`.  CPU only: the object is
built from committed bench lines; every figure must be a copy of the object it names, and the whole summary must fit the tail."""
import json
import os

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_summary_copies_the_figures_of_the_line_and_fits_the_drivers_tail():
    line = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_line_driver_command.json")))
    line.pop("summary", None)
    s = bench.summary_obj(line)
    assert s["msm_2_20_Mscalar_s"] == line["value"] and s["msm_cpu_Mscalar_s"] == line["cpu_baseline"]["value"]
    w, z = line["wrapping_prover"], line["zeth_shaped"]
    assert s["wrapping_proofs_s"]["host_witness"] == w["value"] and s["wrapping_proofs_s"]["gpu_witness"] == w["gpu_witness"]["value"]
    assert s["wrapping_proofs_s"]["cpu_port"] == w["cpu_baseline"]["value"] and s["wrapping_proofs_s"]["result_bits"] == 3
    assert s["zeth_shaped_proofs_s"]["host_witness"] == z["value"] and s["zeth_shaped_proofs_s"]["cpu_port"] == z["cpu_baseline"]["value"]
    assert s["zeth_shaped_proofs_s"]["result_bits"] == 3 and s["zeth_shaped_proofs_s"]["cpu_proof_identical"] is True
    assert s["prover_2_20_proofs_s"]["gpu"] == line["prover_2_20"]["value"] and s["prover_2_22_proofs_s"]["gpu"] == line["prover_2_22"]["value"]
    assert s["all_last_proofs_verify"] is True and s["roofline_frac"] == line["roofline"]["frac"]
    assert len(json.dumps(s)) < 1500                                  # well inside the 2,000 characters the driver keeps
    # a line without the secondaries (N > 1, --no-secondary): nothing invented
    bare = {k: line[k] for k in ("metric", "value", "unit", "roofline")}
    sb = bench.summary_obj(bare)
    assert sb["msm_2_20_Mscalar_s"] == line["value"] and "wrapping_proofs_s" not in sb and sb["all_last_proofs_verify"] is None


def test_replica_witness_mode_follows_the_cores_of_a_rank(monkeypatch):
    """bench.GpuReplicaStream picks the host generator where a rank has six or more host cores, the GPU generator otherwise (DESIGN
    section 8: the hybrid mode is dominated at nine inputs); the environment overrides.  The rule itself, without a GPU."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'os.environ.get("ZKHIP_BENCH_REPLICA_WITNESS") or ("host" if host_threads() // ranks_here >= 6 else "gpu")' in src
    assert bench.aggregator_expected_bits(9) == 3 and bench.aggregator_expected_bits(3) == 3 and bench.aggregator_expected_bits(1) == 3
    assert bench.aggregator_expected_bits(5) == 0                      # (any other count: the padded all-reject key of rounds 3-5)
    nvk, npr, nin, _ = bench.aggregator_inputs(9)
    assert nvk.shape == (60 + 12 * 10,) and npr.shape == (96,) and nin.shape == (18, 6)
    nvk5, npr5, nin5, _ = bench.aggregator_inputs(5)
    assert nvk5.shape == (60 + 12 * 6,) and nin5.shape == (10, 6)
