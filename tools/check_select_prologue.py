"""Build check for the pipelined search kernels that also select the target (k_search_pipe<P, CHROMA, true>).

Their prologue requests the target patch's four bytes per thread with register-destination loads issued from
inline assembly and waits for them much later with a hand-counted `s_waitcnt vmcnt(N)`: until that wait the
compiler believes the destination registers are already defined.  This check disassembles every such
instance and fails unless, between the four loads and the wait,
  * exactly N vector-memory instructions are issued (the ring loads the count stands for),
  * no instruction names one of the four destination registers, and
  * nothing is spilled to scratch.
python3 tools/check_select_prologue.py [lib]   (exit status 1 on a violation)"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "image-processing_amd", "lib", "libcriminisi_hip.so")
data = open(lib, "rb").read()
off = [m.start() for m in re.finditer(b"\x7fELF\x02\x01\x01\x40", data)][0]
with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
    f.write(data[off:])
    path = f.name
out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
os.unlink(path)
kernels, cur = {}, None
for line in out.splitlines():
    m = re.match(r"^[0-9a-f]+ <(.*)>:", line)
    if m:
        cur = m.group(1)
        kernels[cur] = []
    elif cur and line.strip():
        kernels[cur].append(line.split("//")[0].strip())
bad = checked = 0
for name, ins in kernels.items():
    m = re.match(r"_Z13k_search_pipeILi(\d+)ELb([01])ELb1EEv6Params", name)
    if not m:
        continue
    P = int(m.group(1))
    lrows = 32 + 2 * ((P + 1) // 2) - 1
    nk = (lrows * 17 + 255) // 256            # 16-byte chunks of a plane tile per thread
    want = 3 * nk + 1
    loads = [i for i, s in enumerate(ins) if re.match(r"global_load_[us]byte\s+v\d+,", s)]
    # the four patch loads are the first run of register-destination byte loads in the kernel
    run = [i for i in loads[:4]]
    if len(run) != 4 or run[-1] - run[0] > 12:
        print("%s: the four patch loads were not found together" % name); bad += 1; continue
    dests = [re.match(r"global_load_[us]byte\s+(v\d+),", ins[i]).group(1) for i in run]
    end = None
    for i in range(run[-1] + 1, len(ins)):
        w = re.match(r"s_waitcnt\s+vmcnt\((\d+)\)", ins[i])
        if w and int(w.group(1)) == want:
            end = i
            break
        if re.match(r"s_waitcnt\s+vmcnt\((\d+)\)", ins[i]) and int(re.match(r"s_waitcnt\s+vmcnt\((\d+)\)", ins[i]).group(1)) < want:
            end = i   # a stricter wait is also safe
            break
    if end is None:
        print("%s: no s_waitcnt vmcnt(%d) after the patch loads" % (name, want)); bad += 1; continue
    between = ins[run[-1] + 1:end]
    vmem = sum(1 for s in between if re.match(r"(global|buffer|flat|scratch)_", s))
    problems = []
    if vmem != want and not re.match(r"s_waitcnt\s+vmcnt\((\d+)\)", ins[end]).group(1) == "0":
        problems.append("%d vector-memory instructions between the loads and the wait, the wait counts %d" % (vmem, want))
    for s in between:
        if s.startswith("scratch_"):
            problems.append("scratch access before the wait: " + s)
        for d in dests:
            if re.search(r"\b%s\b" % d, s):
                problems.append("%s is touched before the wait: %s" % (d, s))
    checked += 1
    if problems:
        bad += 1
        print(name + ":\n  " + "\n  ".join(problems))
print("select prologue: %d instances checked, %d with problems" % (checked, bad))
sys.exit(1 if bad or not checked else 0)
