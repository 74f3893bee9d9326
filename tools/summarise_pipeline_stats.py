"""gpurun_out/pipeline_stats/*.log (tools/collect_pipeline_stats.sh) -> profiles/r05_pipeline_host_side.txt: the streaming prover's host
side before (round-4 tree) and after (this tree) on the same box - proofs/s, host cores, CPU seconds per thread name, the pipeline's own
per-proof statistics."""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", "pipeline_stats")
rows = [("r04_host", "round 4 tree (1b01b24): host witness generator, 49,152-point domain, nothing cached"),
        ("r04_gpu", "round 4 tree: GPU witness generator"),
        ("r05_host_nocache", "this tree: host witness generator, 65,536-point domain, per-application constants OFF"),
        ("r05_host", "this tree: host witness generator, per-application constants (the default)"),
        ("r05_gpu", "this tree: GPU witness generator, per-application constants")]
out = ["# Host side of the streaming prover, one box, `bench.py --workload aggregator --steps 2000` (tools/collect_pipeline_stats.sh)",
       "# (CPU seconds per thread name: threads alive at the end of the run, whole process life including set-up; the host witness generator's",
       "#  per-section helper threads end with their witness and are missing from it - host_cores_busy, process CPU time over the timed region, has them)", ""]
for name, what in rows:
    path = os.path.join(src, name + ".log")
    if not os.path.exists(path):
        continue
    line, stats = None, []
    for l in open(path):
        if l.startswith("{"):
            line = json.loads(l)
        elif l.startswith("zkhip pipeline:"):
            stats.append(l.strip())
    if line is None:
        out.append("%s: no bench line (see the log)" % name)
        continue
    out.append("## %s" % what)
    out.append("proofs/s %.1f   host_cores_busy %.2f   ms_per_step %.3f   last_proof_verifies %s" % (line["value"], line["host_cores_busy"], line["ms_per_step"], line.get("last_proof_verifies")))
    out.append("CPU ms per proof (process): %.2f" % (line["host_cores_busy"] / line["value"] * 1e3))
    if "thread_cpu_s" in line:
        out.append("CPU seconds per thread name, whole process life: " + ", ".join("%s %.1f" % kv for kv in line["thread_cpu_s"].items()))
    out += stats
    out.append("")
open(os.path.join(root, "profiles", "r05_pipeline_host_side.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out))
