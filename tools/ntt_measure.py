"""The NTT passes alone, for the counters and the roofline: zkhip_measure_ntt at 2^20 (or the sizes given), the four
modes, one vector and the QAP map's batch of three.  Under rocprofv3 this is the command whose k_ntt_pass dispatches are read:

    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o ntt -- python3 tools/ntt_measure.py 20
    ... --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE ... -- python3 tools/ntt_measure.py 20
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zecale_amd import zkhip  # noqa: E402

logs = [int(x) for x in sys.argv[1:]] or [20]
zkhip.init(0)
peak = zkhip.measure_fq_mul_rate()
print("multiplier peak of this box: %.2f G Fq-mul/s = %.1f G v_mad_u64_u32/s" % (peak / 1e9, peak * 1458 / 1e9))
for log_d in logs:
    d = 1 << log_d
    for batch in (1, 3):
        for inverse in (False, True):
            for coset in (False, True):
                ms = zkhip.measure_ntt(log_d, inverse, coset, batch, reps=10)
                gbs = 2 * d * 48 / (ms * 1e-3) / 1e9
                mads = (d // 2 * log_d + d) * 406                 # butterfly products + the inter-step twiddle
                print("2^%d batch %d %s%s: %.4f ms per transform, %.1f GB/s algorithmic (%.2f %% of 8 TB/s), %.3f of the mad peak"
                      % (log_d, batch, "i" if inverse else "", "cosetFFT" if coset else "FFT", ms, gbs, gbs / 80.0, mads / (ms * 1e-3) / (peak * 1458)))
