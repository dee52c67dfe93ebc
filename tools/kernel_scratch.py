#!/usr/bin/env python3
"""Registers / LDS / scratch of every kernel in the BUILT library (no recompilation): walks the clang offload bundles inside
libmvs_hip.so, reads each gfx950 code object's AMDGPU metadata note with llvm-readelf.
    python tools/kernel_scratch.py [libmvs_hip.so] [filter]   -> one line per kernel; kernels with scratch > 0 are marked"""
import os, re, struct, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".so") else os.path.join(ROOT, "self-supervised-mvs_amd", "libmvs_hip.so")
flt = sys.argv[-1] if len(sys.argv) > 1 and not sys.argv[-1].endswith(".so") else ""
data = open(lib, "rb").read()
magic = b"__CLANG_OFFLOAD_BUNDLE__"
pos, rows = 0, []
while True:
    i = data.find(magic, pos)
    if i < 0:
        break
    n = struct.unpack_from("<Q", data, i + 24)[0]
    p = i + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", data, p)
        triple = data[p + 24:p + 24 + tl].decode()
        p += 24 + tl
        if "gfx950" in triple and size:
            with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as fh:
                fh.write(data[i + off:i + off + size])
            out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", fh.name], capture_output=True, text=True).stdout
            os.unlink(fh.name)
            for blk in out.split("- .agpr_count:")[1:]:
                g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk)
                name = g("name").group(1)
                rows.append((name, int(g("vgpr_count").group(1)), int(blk.split()[0]), int(g("sgpr_count").group(1)),
                             int(g("group_segment_fixed_size").group(1)), int(g("private_segment_fixed_size").group(1))))
    pos = i + 24
names = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (name, vg, ag, sg, lds, sc), dn in zip(rows, names):
    if flt and flt not in dn:
        continue
    print("%s vgpr %3d agpr %3d sgpr %3d lds %6d scratch %4d  %s" % ("!!" if sc else "  ", vg, ag, sg, lds, sc, dn[:150]))
