"""Time the NTT passes alone (nothing else on the GPU): run under rocprofv3 --kernel-trace --stats.

    cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d OUT -o ntt -- python3 tools/ntt_bench.py [log_d ...]

Each size runs the four modes (FFT, iFFT, cosetFFT, icosetFFT) several times through zkhip_ntt_dev on a resident buffer;
the per-kernel averages of k_ntt_pass<..> in the stats file are the pass times.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zecale_amd import zkhip  # noqa: E402

logs = [int(x) for x in sys.argv[1:]] or [16, 20]
zkhip.init(0)
for log_d in logs:
    d = 1 << log_d
    buf = torch.zeros((d, 6), dtype=torch.int64, device="cuda:0")
    buf[:, 0] = torch.arange(d, device="cuda:0")
    for _ in range(5):
        for inverse in (False, True):
            for coset in (False, True):
                zkhip.ntt_dev(buf.data_ptr(), log_d, inverse, coset)
    torch.cuda.synchronize()
    print("log_d", log_d, "done")
