"""Host-side wrapping circuit under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only: GPU sanitizers are not
available on the pool).  tools/sanitize/witness_check.cpp builds the batch-2 circuit, generates the witness of the reference
fixtures twice, checks all 44,183 constraints with the host field arithmetic and the key hash against primary input 0."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_witness_generator_is_clean_under_asan_ubsan(tmp_path):
    import bench
    nvk_l, npr, nin, _ = bench.aggregator_inputs()
    inp = tmp_path / "in.bin"
    np.concatenate([nvk_l, npr, nin.reshape(-1)]).astype(np.uint64).tofile(inp)
    exe = tmp_path / "witness_check"
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-fno-omit-frame-pointer", "-pthread", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "zecale_amd", "csrc"), os.path.join(ROOT, "tools", "sanitize", "witness_check.cpp"),
                           os.path.join(ROOT, "zecale_amd", "csrc", "aggregator.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe), str(inp)], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "constraints=44183 vars=44206 unsatisfied=0 hash_matches_input0=1" in out.stdout
    assert "ERROR" not in out.stderr and "runtime error" not in out.stderr


def test_gpu_witness_program_on_the_cpu_under_asan_ubsan(tmp_path):
    """The GPU witness generator's PROGRAM without a GPU (tools/sanitize/tape_check.cpp): the tape builder (recording, balanced
    sums, bounds for the lazy reduction, levels by kind, the key-hash chain) runs under ASan + UBSan; the tape's structure and the
    bounds the device relies on are re-derived independently; interpreted with the host field arithmetic it must reproduce the host
    generator's assignment limb for limb - for the valid batch and for a batch with a bumped (invalid) nested input."""
    import bench
    nvk_l, npr, nin, _ = bench.aggregator_inputs()
    exe = tmp_path / "tape_check"
    csrc = os.path.join(ROOT, "zecale_amd", "csrc")
    subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-fno-omit-frame-pointer", "-pthread", "-I", os.path.join(ROOT, "include"), "-I", csrc,
                           os.path.join(ROOT, "tools", "sanitize", "tape_check.cpp"), os.path.join(csrc, "aggregator.cpp"),
                           os.path.join(csrc, "witness_tape.cpp"), "-o", str(exe)])
    bumped = nin.copy()
    bumped[1][0] += 1
    for name, inputs in (("valid", nin), ("bumped", bumped)):
        inp = tmp_path / (name + ".bin")
        np.concatenate([nvk_l, npr, inputs.reshape(-1)]).astype(np.uint64).tofile(inp)
        out = subprocess.run([str(exe), str(inp)], capture_output=True, text=True, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1"))
        assert out.returncode == 0, out.stdout + out.stderr
        assert "differences=0" in out.stdout and "FAIL" not in out.stdout, out.stdout
        assert "ERROR" not in out.stderr and "runtime error" not in out.stderr, out.stderr
