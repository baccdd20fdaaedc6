"""VGPR / SGPR / LDS / scratch of every kernel in the built library, read from the code objects' metadata notes.
usage: python tools/kernel_regs.py [regex [library or object file]]"""
import os, re, struct, subprocess, sys, tempfile
LIB = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vcfdist_amd", "lib", "libvcfdist_pr.so")
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
data = open(LIB, "rb").read()
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
rows = []
pos = 0
while True:
    pos = data.find(MAGIC, pos)
    if pos < 0:
        break
    n = struct.unpack_from("<Q", data, pos + 24)[0]
    o = pos + 32
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", data, o)
        triple = data[o + 24:o + 24 + tl].decode()
        o += 24 + tl
        if "gfx" in triple and size:
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(data[pos + off:pos + off + size]); f.flush()
                txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
            for b in txt.split("- .agpr_count")[1:]:
                g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", b) or [None, "?"])[1]
                rows.append((g("name"), g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size")))
    pos += 24
pat = re.compile(sys.argv[1] if len(sys.argv) > 1 else ".")
for r in sorted(set(rows)):
    name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", r[0]], capture_output=True, text=True).stdout.strip() if False else r[0]
    if pat.search(name):
        print("%-100s vgpr %4s sgpr %4s lds %7s scratch %5s" % (name[:100], *r[1:]))
