"""DESIGN.md section 1's table from two committed bench lines of the driver's command - profiles/r06_bench_line_driver_command.json (round 6)
and profiles/r05_bench_line_driver_command.json (round 5's final commit) - so that no number is typed by hand; the third column is what
the DRIVER's own round-5 run recorded (BENCH_r05.json and the tail it kept)."""
import json, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
a = json.load(open(os.path.join(root, "profiles", "r06_bench_line_driver_command.json")))
b = json.load(open(os.path.join(root, "profiles", "r05_bench_line_driver_command.json")))


def g(o, k):
    for part in k.split("."):
        o = o[part]
    return o


def both(fmt, *keys):
    return fmt.format(*[g(a, k) for k in keys]), fmt.format(*[g(b, k) for k in keys])


rows = [
    ("G1 MSM, 2^20 terms, uniform scalars, table-backed resident bases, ten in flight (round 5: eight) (BASELINE configs[1])", "84.17", "`value`, `ms_per_step`",
     both("**{} Mscalar/s**, {} ms per MSM", "value", "ms_per_step")),
    ("`k_accumulate<1>` ALONE on the chip", "11.07 ms", "`roofline.kernel_ms`, `.fq_mul_frac_vs_this_run_peak`, `.fq_mul_peak_this_run_g_per_s`",
     both("{} ms; {} of this run's measured multiplier peak ({} G Fq-mul/s)", "roofline.kernel_ms", "roofline.fq_mul_frac_vs_this_run_peak", "roofline.fq_mul_peak_this_run_g_per_s")),
    ("algorithmic bytes / kernel time against 8 TB/s (the contract's roofline): not HBM-bound (SURVEY §0.5)", "0.28 %", "`roofline.frac`, `.achieved`, `.traffic`",
     both("frac {} ({} GB/s); counter traffic {} B per launch", "roofline.frac", "roofline.achieved", "roofline.traffic")),
    ("the same stream without the window table / with the scalars uploaded from host memory per MSM", "-", "`plain_path`, `host_scalars`",
     both("{} / {} Mscalar/s", "plain_path.value", "host_scalars.value")),
    ("one 2^20 NTT (the seven of a proof, kernels alone)", "-", "`ntt_2_20`",
     both("**{} ms**, frac {} of 8 TB/s by algorithmic bytes, {} of the mad peak", "ntt_2_20.value", "ntt_2_20.roofline.frac", "ntt_2_20.roofline.mad_frac")),
    ("Groth16 proof over 2^20 - 8 constraints (configs[2]), five in flight, verified", "-", "`prover_2_20`",
     both("**{} proofs/s** ({} ms)", "prover_2_20.value", "prover_2_20.ms_per_step")),
    ("... its `k_accumulate<5>` alone (additions counted by the launch's sort)", "-", "`prover_2_20.roofline`",
     both("{} ms, {} of this run's peak", "prover_2_20.roofline.kernel_ms", "prover_2_20.roofline.fq_mul_frac_vs_this_run_peak")),
    ("... the C restatement proving the SAME system on 16 host cores, one whole proof timed, proof limb-identical", "-", "`prover_2_20.cpu_baseline`",
     both("{} proofs/s", "prover_2_20.cpu_baseline.value")),
    ("Groth16 proof over 2^22 - 8 constraints (body of configs[3]) on one GPU, four in flight (round 5: two), verified", "5.917", "`prover_2_22`",
     both("**{} proofs/s** ({} ms)", "prover_2_22.value", "prover_2_22.ms_per_proof")),
    ("the real batch-2 wrapping circuit (44,183 constraints, **65,536-point domain**), host witness, per-application constants, steady state", "(in the driver's record only as a key name)", "`wrapping_prover.value`, `.host_cores_busy`",
     both("**{} proofs/s** on {} host cores", "wrapping_prover.value", "wrapping_prover.host_cores_busy")),
    ("... the same pipeline timed with fill and drain inside the timed region (rounds 1-3's method)", "-", "`.value_fill_and_drain`",
     both("{}", "wrapping_prover.value_fill_and_drain")),
    ("... everything recomputed per proof (rounds 1-4's mode), steady state / fill and drain (round 3, same domain and method as the second: 324.5)", "-", "`.without_app_cache`",
     both("{} / {}", "wrapping_prover.without_app_cache.value", "wrapping_prover.without_app_cache.value_fill_and_drain")),
    ("... on the optional 49,152-point step domain, with / without the constants", "-", "`.step_domain`",
     both("{} / {}", "wrapping_prover.step_domain.value", "wrapping_prover.step_domain.without_app_cache.value")),
    ("... assignments generated on the GPU (application's own program), 256 batches outstanding; without the constants", "-", "`.gpu_witness`",
     both("**{} proofs/s on {} host cores**; {}", "wrapping_prover.gpu_witness.value", "wrapping_prover.gpu_witness.host_cores_busy", "wrapping_prover.gpu_witness.without_app_cache.value")),
    ("... hybrid: host generators beside the GPU generator", "-", "`.hybrid_witness`",
     both("{} proofs/s on {} host cores", "wrapping_prover.hybrid_witness.value", "wrapping_prover.hybrid_witness.host_cores_busy")),
    ("... `k_accumulate<5>` of one wrapping proof alone, full assignment (4,314,435 mixed additions counted by the launch's sort) / masked", "-", "`wrapping_prover.roofline`, `.one_proof_alone_with_app_cache_ms`",
     both("{} ms = {} of this run's peak / {} ms", "wrapping_prover.roofline.kernel_ms", "wrapping_prover.roofline.fq_mul_frac_vs_this_run_peak", "wrapping_prover.one_proof_alone_with_app_cache_ms.k_accumulate5_ms")),
    ("one wrapping proof alone: witness, five MSMs, QAP, host tail (with the constants: MSMs)", "-", "`.one_proof_alone_ms`",
     both("{}, {} ({}), {}, {} ms", "wrapping_prover.one_proof_alone_ms.witness_host", "wrapping_prover.one_proof_alone_ms.msm_sequence_all_five",
          "wrapping_prover.one_proof_alone_with_app_cache_ms.msm_sequence_all_five", "wrapping_prover.one_proof_alone_ms.qap", "wrapping_prover.one_proof_alone_ms.host_tail")),
    ("nine inputs per nested proof (Zeth-shaped, 92,055 constraints, **131,072-point domain**, configs[4] on one GPU), host witness; without the constants; step domain", "249.9 on 4.35 host cores; 191.5; 268.5 (nested proofs INVALID: result bits 0)", "`zeth_shaped`",
     both("**{} proofs/s** on {} host cores; {}; {}", "zeth_shaped.value", "zeth_shaped.host_cores_busy", "zeth_shaped.without_app_cache.value", "zeth_shaped.step_domain.value")),
    ("... assignments generated on the GPU / hybrid", "239.5 on 2.19 / 232.2 on 3.29", "`zeth_shaped.gpu_witness`, `.hybrid_witness`",
     both("**{} on {} cores** / {} on {}", "zeth_shaped.gpu_witness.value", "zeth_shaped.gpu_witness.host_cores_busy", "zeth_shaped.hybrid_witness.value", "zeth_shaped.hybrid_witness.host_cores_busy")),
    ("CPU restatement on the box's 16 host cores (a port, not libsnark): Mscalar/s; wrapping proofs/s; 2^20-proofs/s", "-", "the three `cpu_baseline` objects",
     both("{}; {}; {}", "cpu_baseline.value", "wrapping_prover.cpu_baseline.value", "prover_2_20.cpu_baseline.value")),
]
print("| what | round 6 (`r06_bench_line_driver_command.json`) | round 5's final commit, another box (`r05_bench_line_driver_command.json`) | round 5, the driver's run (`BENCH_r05.json`) | bench key |")
print("|---|---|---|---|---|")
for what, r4, key, (va, vb) in rows:
    print("| %s | %s | %s | %s | %s |" % (what, va, vb, r4, key))
za = a["zeth_shaped"]
print("| ... result bits of the nine-input streams' last proofs / the C restatement proving the identical nine-input batch (one whole proof, limb-identical) | %s (VALID nested proofs) / %s proofs/s | 0 (invalid by construction) / - | 0 / - | `zeth_shaped.result_bits`, `.cpu_baseline` |" % (za["result_bits"], za["cpu_baseline"]["value"]))
print("| N > 1 | `python bench.py --gpus N`: configs[3] / [4] legs with `leg_wall_s`, the partitioned leg set up by `zkhip_groth16_setup_slice` (§8); two ranks rehearsed on one GPU; **RCCL has still not seen more than one rank** | | | §8 |")
