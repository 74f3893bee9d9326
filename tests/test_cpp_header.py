"""include/groth16_snark_hip.hpp compiles stand-alone and its policy-class adapter instantiates against a
mock of the reference's CPU policy class (libsnark is not in this image).  CPU only, syntax + link check."""
import os
import subprocess
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_compiles_and_links(tmp_path):
    src = tmp_path / "t.cpp"
    src.write_text(textwrap.dedent(r'''
        #include "groth16_snark_hip.hpp"
        #include <cstdio>
        struct mock_cpu_snark { struct proving_key {}; struct proof { int tag; }; static const char* name() { return "GROTH16"; } };
        struct mock_pb {};
        struct mock_bridge {
          static zecale_amd::hip_proving_key* upload(const mock_cpu_snark::proving_key&) { return nullptr; }
          static void assignment(const mock_pb&, std::vector<uint64_t>& z) { z.assign(6, 0); }
          static void random_scalars(uint64_t r[6], uint64_t s[6]) { for (int i = 0; i < 6; i++) r[i] = s[i] = 0; }
          static mock_cpu_snark::proof proof_from_limbs(const zecale_amd::groth16_proof&) { return {1}; }
        };
        using wsnark = zecale_amd::groth16_snark_hip<mock_cpu_snark, mock_bridge>;
        int main() {
          // instantiate the template (never called: there is no device here)
          auto fn = &wsnark::generate_proof<mock_pb>;
          (void)fn;
          int rc = zkhip_init(0);
          std::printf("%s rc=%d %s\n", wsnark::name(), rc, zkhip_strerror(rc));
          try { zecale_amd::zk_check(ZKHIP_ERR_ARG, "probe"); } catch (const std::runtime_error& e) { std::printf("%s\n", e.what()); return 0; }
          return 1;
        }
    '''))
    exe = tmp_path / "t"
    subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", os.path.join(ROOT, "zecale_amd"), "-lzkhip", "-Wl,-rpath," + os.path.join(ROOT, "zecale_amd")])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GROTH16" in out.stdout and "probe: bad argument" in out.stdout
