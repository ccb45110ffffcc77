"""Disassembly of one kernel of a built libcriminisi_hip.so (llvm-objdump on the embedded gfx950 code object).
python3 tools/disasm_kernel.py <mangled-name substring> [lib] > out.s"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
want = sys.argv[1]
lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "image-processing_amd", "lib", "libcriminisi_hip.so")
data = open(lib, "rb").read()
off = [m.start() for m in re.finditer(b"\x7fELF\x02\x01\x01\x40", data)][0]
with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
    f.write(data[off:])
    path = f.name
out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
os.unlink(path)
on = False
for line in out.splitlines():
    m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
    if m:
        on = want in m.group(1)
    if on:
        print(line)
