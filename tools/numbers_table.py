"""DESIGN.md section 1's table from the two committed bench lines (profiles/r05_bench_line_driver_command.json: the final commit;
profiles/r05_bench_line_other_box.json: an earlier commit of the round on another box), so that no number is typed by hand."""
import json, os
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
a = json.load(open(os.path.join(root, "profiles", "r05_bench_line_driver_command.json")))
b = json.load(open(os.path.join(root, "profiles", "r05_bench_line_other_box.json")))


def g(o, k):
    for part in k.split("."):
        o = o[part]
    return o


def both(fmt, *keys):
    return fmt.format(*[g(a, k) for k in keys]), fmt.format(*[g(b, k) for k in keys])


rows = [
    ("G1 MSM, 2^20 terms, uniform scalars, table-backed resident bases, eight in flight (BASELINE configs[1])", "82.88", "`value`, `ms_per_step`",
     both("**{} Mscalar/s**, {} ms per MSM", "value", "ms_per_step")),
    ("`k_accumulate<1>` ALONE on the chip", "11.18 ms, 0.745", "`roofline.kernel_ms`, `.fq_mul_frac_vs_this_run_peak`, `.fq_mul_peak_this_run_g_per_s`",
     both("{} ms; {} of this run's measured multiplier peak ({} G Fq-mul/s)", "roofline.kernel_ms", "roofline.fq_mul_frac_vs_this_run_peak", "roofline.fq_mul_peak_this_run_g_per_s")),
    ("algorithmic bytes / kernel time against 8 TB/s (the contract's roofline): not HBM-bound (SURVEY §0.5)", "0.28 %", "`roofline.frac`, `.achieved`, `.traffic`",
     both("frac {} ({} GB/s); counter traffic {} B per launch", "roofline.frac", "roofline.achieved", "roofline.traffic")),
    ("the same stream without the window table / with the scalars uploaded from host memory per MSM", "67.6 / 78.3", "`plain_path`, `host_scalars`",
     both("{} / {} Mscalar/s", "plain_path.value", "host_scalars.value")),
    ("one 2^20 NTT (the seven of a proof, kernels alone)", "0.293 ms, 4.3 %, 0.46 (from the kernel trace)", "`ntt_2_20`",
     both("**{} ms**, frac {} of 8 TB/s by algorithmic bytes, {} of the mad peak", "ntt_2_20.value", "ntt_2_20.roofline.frac", "ntt_2_20.roofline.mad_frac")),
    ("Groth16 proof over 2^20 - 8 constraints (configs[2]), five in flight, verified", "20.93", "`prover_2_20`",
     both("**{} proofs/s** ({} ms)", "prover_2_20.value", "prover_2_20.ms_per_step")),
    ("... its `k_accumulate<5>` alone (additions counted by the launch's sort)", "37.6 ms", "`prover_2_20.roofline`",
     both("{} ms, {} of this run's peak", "prover_2_20.roofline.kernel_ms", "prover_2_20.roofline.fq_mul_frac_vs_this_run_peak")),
    ("... the C restatement proving the SAME system on 16 host cores, one whole proof timed, proof limb-identical", "0.0211", "`prover_2_20.cpu_baseline`",
     both("{} proofs/s", "prover_2_20.cpu_baseline.value")),
    ("Groth16 proof over 2^22 - 8 constraints (body of configs[3]) on one GPU, two in flight, verified", "5.81", "`prover_2_22`",
     both("**{} proofs/s** ({} ms)", "prover_2_22.value", "prover_2_22.ms_per_proof")),
    ("the real batch-2 wrapping circuit (44,183 constraints, **65,536-point domain**), host witness, per-application constants, steady state", "371.3 on the 49,152-point domain, nothing cached, 7.35 cores", "`wrapping_prover.value`, `.host_cores_busy`",
     both("**{} proofs/s** on {} host cores", "wrapping_prover.value", "wrapping_prover.host_cores_busy")),
    ("... the same pipeline timed with fill and drain inside the timed region (rounds 1-3's method)", "-", "`.value_fill_and_drain`",
     both("{}", "wrapping_prover.value_fill_and_drain")),
    ("... everything recomputed per proof (rounds 1-4's mode), steady state / fill and drain (round 3, same domain and method as the second: 324.5)", "-", "`.without_app_cache`",
     both("{} / {}", "wrapping_prover.without_app_cache.value", "wrapping_prover.without_app_cache.value_fill_and_drain")),
    ("... on the optional 49,152-point step domain, with / without the constants", "371.3 (without)", "`.step_domain`",
     both("{} / {}", "wrapping_prover.step_domain.value", "wrapping_prover.step_domain.without_app_cache.value")),
    ("... assignments generated on the GPU (application's own program), 256 batches outstanding; without the constants", "321.3 on 2.85", "`.gpu_witness`",
     both("**{} proofs/s on {} host cores**; {}", "wrapping_prover.gpu_witness.value", "wrapping_prover.gpu_witness.host_cores_busy", "wrapping_prover.gpu_witness.without_app_cache.value")),
    ("... hybrid: host generators beside the GPU generator", "-", "`.hybrid_witness`",
     both("{} proofs/s on {} host cores", "wrapping_prover.hybrid_witness.value", "wrapping_prover.hybrid_witness.host_cores_busy")),
    ("... `k_accumulate<5>` of one wrapping proof alone, full assignment (4,314,435 mixed additions counted by the launch's sort) / masked", "2.37 ms", "`wrapping_prover.roofline`, `.one_proof_alone_with_app_cache_ms`",
     both("{} ms = {} of this run's peak / {} ms", "wrapping_prover.roofline.kernel_ms", "wrapping_prover.roofline.fq_mul_frac_vs_this_run_peak", "wrapping_prover.one_proof_alone_with_app_cache_ms.k_accumulate5_ms")),
    ("one wrapping proof alone: witness, five MSMs, QAP, host tail (with the constants: MSMs)", "6.8, 5.3, 0.50, 0.71", "`.one_proof_alone_ms`",
     both("{}, {} ({}), {}, {} ms", "wrapping_prover.one_proof_alone_ms.witness_host", "wrapping_prover.one_proof_alone_ms.msm_sequence_all_five",
          "wrapping_prover.one_proof_alone_with_app_cache_ms.msm_sequence_all_five", "wrapping_prover.one_proof_alone_ms.qap", "wrapping_prover.one_proof_alone_ms.host_tail")),
    ("nine inputs per nested proof (Zeth-shaped, 92,055 constraints, **131,072-point domain**, configs[4] on one GPU), host witness; without the constants; step domain", "195.3 on the 98,304-point domain", "`zeth_shaped`",
     both("**{} proofs/s** on {} host cores; {}; {}", "zeth_shaped.value", "zeth_shaped.host_cores_busy", "zeth_shaped.without_app_cache.value", "zeth_shaped.step_domain.value")),
    ("... assignments generated on the GPU / hybrid", "-", "`zeth_shaped.gpu_witness`, `.hybrid_witness`",
     both("**{} on {} cores** / {} on {}", "zeth_shaped.gpu_witness.value", "zeth_shaped.gpu_witness.host_cores_busy", "zeth_shaped.hybrid_witness.value", "zeth_shaped.hybrid_witness.host_cores_busy")),
    ("CPU restatement on the box's 16 host cores (a port, not libsnark): Mscalar/s; wrapping proofs/s; 2^20-proofs/s", "0.119; 0.338; 0.0211", "the three `cpu_baseline` objects",
     both("{}; {}; {}", "cpu_baseline.value", "wrapping_prover.cpu_baseline.value", "prover_2_20.cpu_baseline.value")),
]
print("| what | final commit (`r05_bench_line_driver_command.json`) | earlier commit, another box (`r05_bench_line_other_box.json`) | round 4, driver's run | bench key |")
print("|---|---|---|---|---|")
for what, r4, key, (va, vb) in rows:
    print("| %s | %s | %s | %s | %s |" % (what, va, vb, r4, key))
print("| N > 1 | the N > 1 line now also measures configs[3] / [4] (§8); **RCCL has still not seen more than one rank** | | | §8 |")
