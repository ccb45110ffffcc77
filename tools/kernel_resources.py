"""Registers, scratch (spills) and LDS of every kernel in a built libcriminisi_hip.so, read from the
code object's metadata notes.  python3 tools/kernel_resources.py [lib] [name filter]"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] else os.path.join(ROOT, "image-processing_amd", "lib", "libcriminisi_hip.so")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
LLVM = "/opt/rocm/lib/llvm/bin"
data = open(lib, "rb").read()
# the fat binary holds one gfx950 code object: an ELF image inside the host library
offs = [m.start() for m in re.finditer(b"\x7fELF\x02\x01\x01\x40", data)]   # ELFOSABI_AMDGPU_HSA = 64
for off in offs:
    with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
        f.write(data[off:])
        path = f.name
    out = subprocess.run([LLVM + "/llvm-readelf", "--notes", path], capture_output=True, text=True).stdout
    os.unlink(path)
    cur = {}
    rows = []
    for line in out.splitlines():
        m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line)
        if not m:
            continue
        k, v = m.group(1), m.group(2).strip()
        if k == "name" and "symbol" not in cur and cur.get("_in_kernel"):
            cur["name"] = v
        if k in ("sgpr_count", "vgpr_count", "agpr_count", "private_segment_fixed_size", "group_segment_fixed_size",
                 "sgpr_spill_count", "vgpr_spill_count", "max_flat_workgroup_size"):
            cur[k] = v
        if k == "symbol":
            cur["symbol"] = v
        if k == "wavefront_size":      # the last key of a kernel's entry (keys are in alphabetical order)
            if "symbol" in cur:
                rows.append(cur)
            cur = {}
    for r in rows:
        sym = subprocess.run(["c++filt", r["symbol"].replace(".kd", "").strip("'")], capture_output=True, text=True).stdout.strip()
        if flt and flt not in sym:
            continue
        print("%-52s vgpr %3s agpr %3s sgpr %3s scratch %5s lds %6s spills v%s s%s" % (
            sym[:52], r.get("vgpr_count"), r.get("agpr_count", "0"), r.get("sgpr_count"), r.get("private_segment_fixed_size"),
            r.get("group_segment_fixed_size"), r.get("vgpr_spill_count", "0"), r.get("sgpr_spill_count", "0")))
