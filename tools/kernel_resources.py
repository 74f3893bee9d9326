#!/usr/bin/env python3
"""VGPRs, spills, LDS and scratch of every kernel in zecale_amd/libzkhip.so (or another library / object given as argument): reads the
code objects embedded in .hip_fatbin (clang offload bundles) and prints their AMDGPU metadata.  Works without a GPU."""
import os, re, subprocess, sys, tempfile
LLVM = "/opt/rocm/lib/llvm/bin"
path = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zecale_amd", "libzkhip.so")
pat = sys.argv[2] if len(sys.argv) > 2 else ""
raw = open(path, "rb").read()
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
pos, n = 0, 0
while True:
    i = raw.find(MAGIC, pos)
    if i < 0:
        break
    import struct
    nb = struct.unpack_from("<Q", raw, i + 24)[0]
    off = i + 32
    for _ in range(nb):
        o, sz, tl = struct.unpack_from("<QQQ", raw, off)
        triple = raw[off + 24: off + 24 + tl].decode()
        off += 24 + tl
        if "amdgcn" in triple and sz:
            with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
                f.write(raw[i + o: i + o + sz])
            txt = subprocess.run([LLVM + "/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
            os.unlink(f.name)
            for blk in txt.split("- .agpr_count")[1:]:
                g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
                name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip().split("(")[0]
                if pat in name:
                    print("%-60s vgpr %s  sgpr %s  spill_v %s  spill_s %s  lds %s  scratch %s" % (name[:60], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("sgpr_spill_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
            n += 1
    pos = i + 24
