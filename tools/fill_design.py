#!/usr/bin/env python3
"""Fill the NUM_* / K_* placeholders of DESIGN.md.in from a bench line (the driver's command) and the committed r03 profiles, write DESIGN.md.
usage: python tools/fill_design.py gpurun_out/bench_line_final.json"""
import csv, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.loads([ln for ln in open(sys.argv[1]).read().splitlines() if ln.startswith("{")][-1])
rf, w, p20, z = d["roofline"], d["wrapping_prover"], d["prover_2_20"], d["zeth_shaped"]
serial = {}
n_acc = 1
for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "r03_bench_msm_serial_kernel_stats.csv"))):
    name = r["Name"].split("(")[0].replace("void ", "").replace("zkhip::", "")
    serial[name] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6)
n_acc = serial["k_accumulate<1>"][0]
per = lambda *names: sum(serial[n][1] for n in names if n in serial) / n_acc
traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))["k_accumulate_hbm_bytes_per_launch"]
serial_line = json.loads([ln for ln in open(os.path.join(ROOT, "gpurun_out", "prof", "msm_serial_trace.log")).read().splitlines() if ln.startswith("{")][-1])
one = w["one_proof_alone_ms"]
rep = {
    "NUM_MSM_STREAM": "%.1f" % d["value"], "NUM_MSM_STEP": "%.2f" % d["ms_per_step"],
    "NUM_MSM_SERIAL_MS": "%.1f" % serial_line["ms_per_step"], "NUM_MSM_SERIAL": "%.1f" % serial_line["value"],
    "NUM_ACC_ALONE": "%.2f" % rf["kernel_ms_alone"], "NUM_ACC_STREAM": "%.2f" % rf["kernel_ms"],
    "NUM_FRAC_ALONE": "%.2f" % rf["fq_mul_frac_alone"], "NUM_FRAC_STREAM": "%.2f" % rf["fq_mul_frac"], "NUM_FRAC_STEP": "%.2f" % rf["fq_mul_frac_whole_step"],
    "NUM_HBM_FRAC": "%.2f %%" % (100 * rf["frac"]), "NUM_HBM_GBS": "%.1f" % rf["achieved"],
    "NUM_TRAFFIC_X": "%.0f" % (traffic / 251658240.0), "NUM_TRAFFIC": "%.1f" % (traffic / 1e9),
    "NUM_PLAIN": "%.1f" % d["plain_path"]["value"],
    "NUM_P20_MSM": "%.1f" % p20["phase_ms_one_proof_alone"]["msm_sequence_all_five"], "NUM_P20_ACC": "%.1f" % p20["roofline"]["kernel_ms"],
    "NUM_P20_FRAC": "%.2f" % p20["roofline"]["fq_mul_frac"], "NUM_P20_MS": "%.1f" % p20["ms_per_step"], "NUM_P20": "%.1f" % p20["value"],
    "NUM_WRAP_GW_CORES": "%.1f" % w["gpu_witness"]["host_cores_busy"], "NUM_WRAP_GW": "%.0f" % w["gpu_witness"]["value"],
    "NUM_WRAP_CORES": "%.1f" % w["host_cores_busy"], "NUM_WRAP_MS": "%.2f" % w["ms_per_step"], "NUM_WRAP": "%.0f" % w["value"],
    "NUM_GW_GAP": "%.1f" % (100 * (1 - w["gpu_witness"]["value"] / w["value"])),
    "NUM_ONE_MSM": "%.2f" % one["msm_sequence_all_five"], "NUM_ONE_TAIL": "%.2f" % one["host_tail"], "NUM_ONE_W": "%.1f" % one["witness_host"],
    "NUM_ONE": "%.1f" % (one["witness_host"] + one["upload_z"] + one["qap"] + one["msm_sequence_all_five"] + one["host_tail"]),
    "NUM_ZETH_NESTED": "%.0f" % z["nested_proofs_per_s"], "NUM_ZETH": "%.0f" % z["value"],
    "NUM_CPU_MSM": "%.3f" % d["cpu_baseline"]["value"], "NUM_CPU_WRAP": "%.2f" % w["cpu_baseline"]["value"], "NUM_CPU_P20": "%.3f" % p20["cpu_baseline"]["value"],
    "K_DP0": "%.2f" % per("k_digit_pass<0, false>"), "K_DP1": "%.2f" % per("k_digit_pass<1, false>"), "K_BS": "%.2f" % per("k_bucket_sort"),
    "K_SCAN": "%.2f" % per("k_scan_local", "k_scan_tot", "k_scan_add"),
    "K_ACC_FRAC": "%.2f" % rf["fq_mul_frac_alone"], "K_ACC": "%.2f" % per("k_accumulate<1>"),
    "K_FOLD": "%.2f" % per("k_fixup_fold<false>", "k_fixup_fold<true>"), "K_FIX": "%.2f" % per("k_fixup"), "K_SUMLDS": "%.2f" % per("k_sum_lds"),
    "K_TAIL": "%.1f" % per("k_sum<true>", "k_seg<true>", "k_place_hilo", "k_window_combine", "k_hilo_combine"),
}
s = open(os.path.join(ROOT, "DESIGN.md.in")).read()
for k in sorted(rep, key=len, reverse=True):
    s = s.replace(k, rep[k])
left = [t for t in s.replace("(", " ").replace(")", " ").split() if t.startswith("NUM_") or t.startswith("K_")]
assert not left, left
open(os.path.join(ROOT, "DESIGN.md"), "w").write(s)
print("DESIGN.md written;", len(rep), "figures from", sys.argv[1])
