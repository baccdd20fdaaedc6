"""Disassembly of one kernel of the built library (gfx950 code object inside the .so).
usage: python tools/kernel_isa.py <regex of the mangled name> [out.s]"""
import os, re, struct, subprocess, sys, tempfile
LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vcfdist_amd", "lib", "libvcfdist_pr.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
data = open(LIB, "rb").read()
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
pat = re.compile(sys.argv[1])
pos = 0
out = []
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
                txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True).stdout
            cur = None
            for line in txt.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
                if m:
                    cur = m.group(1) if pat.search(m.group(1)) else None
                    if cur:
                        out.append("=== " + cur)
                    continue
                if cur:
                    out.append(line)
    pos += 24
txt = "\n".join(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt)
else:
    print(txt)
