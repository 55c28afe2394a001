"""CPU-side ISA audit of tapgemm.hip (hipcc cross-compiles gfx950 without a GPU):  python tools/check_isa.py [-DVGEN_X ...]

For the build with the given defines: register spills per kernel, waterfall loops (a `buffer_load` whose resource /
scalar offset instruction selection found in a VGPR: v_readfirstlane + s_cbranch_execnz around it) and, for the
ping-pong 256x160 instantiation, the instruction mix of every loop block (VALU next to the MFMAs is what the K-step
probe says costs time).  With --uniformity also: LLVM's own uniformity analysis on the optimised IR of that
instantiation (`opt -passes='print<uniformity>'`) — private-memory allocas that survived (a load from one is divergent
by definition: r03 found two local arrays tail-merged into a pointer phi that way) and every LDS-DMA call whose resource
or scalar offset the analysis calls divergent (each becomes a waterfall loop)."""
import hashlib, os, re, subprocess, sys
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vgen_amd import build as b

defs = [a for a in sys.argv[1:] if a.startswith("-D")]
out = os.path.join("/tmp", "tapgemm_" + ("_".join(d[2:] for d in defs) or "product") + ".s")
flags = [f for f in b.FLAGS if f != "-fPIC"] + defs
r = subprocess.run([b._hipcc()] + flags + ["-S", "--cuda-device-only", os.path.join(b.CSRC, "tapgemm.hip"), "-o", out],
                   capture_output=True, text=True)
assert r.returncode == 0, r.stderr[-3000:]
t = open(out).read()


def isa_fingerprint(asm_text):
    """sha256 over the instruction stream of every kernel (labels, directives, comments and the per-compilation
    __hip_cuid symbol dropped): two sources with the same fingerprint ARE the same machine code.  The product value is
    committed in tests/golden/tapgemm_isa.sha256 and checked by tests/test_abi.py — a kernel edit has to refresh it
    (`python tools/check_isa.py --update-hash`) and, with it, rerun the GPU parity cases."""
    keep = []
    for l in asm_text.split("\n"):
        l = l.split(";")[0].strip()
        if not l or l.startswith(".") or l.endswith(":") or l.startswith("__hip_cuid"):
            continue
        keep.append(" ".join(l.split()))
    return hashlib.sha256("\n".join(keep).encode()).hexdigest(), len(keep)


fp, nins = isa_fingerprint(t)
HASH_FILE = os.path.join(ROOT, "tests", "golden", "tapgemm_isa.sha256")
print(f"ISA fingerprint: {fp} ({nins} instructions)")
if "--update-hash" in sys.argv and not defs:
    cc = subprocess.run([b._hipcc(), "--version"], capture_output=True, text=True).stdout.split("\n")[0].strip()
    open(HASH_FILE, "w").write(fp + "\nhipcc: " + cc + "\n")     # the compiler the stream belongs to (tests/test_abi.py)
    print("written to", HASH_FILE)
vs = [int(x) for x in re.findall(r"\.vgpr_spill_count:\s+(\d+)", t)]
ss = [int(x) for x in re.findall(r"\.sgpr_spill_count:\s+(\d+)", t)]
vg = [int(x) for x in re.findall(r"\.vgpr_count:\s+(\d+)", t)]
print(f"{out}: {len(vs)} kernels, VGPR spills {sum(v > 0 for v in vs)} kernels, max VGPRs {max(vg)}, max SGPR spills {max(ss)}")
parts = re.split(r"\n(_ZN12_GLOBAL__N_1\w+):", t)
wf_total = 0
for i in range(1, len(parts), 2):
    if "tapgemm_kernel" not in parts[i]:
        continue
    body = parts[i + 1].split("s_endpgm")[0]
    wf = sum(1 for blk in re.split(r"\n\.LBB\d+_\d+:", body)
             if "v_readfirstlane" in blk and "buffer_load" in blk and "s_cbranch_execnz" in blk)
    wf_total += wf
print("waterfall loops around DMA instructions:", wf_total)
KN = "_ZN12_GLOBAL__N_114tapgemm_kernelI3F16Li256ELi160ELi64ELi4ELi2ELi3ELb1ELb0EEE"
lines = t.split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(KN)][0]
name, blocks, order = None, {}, []
for l in lines[start:]:
    ls = l.strip()
    if ls.startswith("s_endpgm"):
        break
    m = re.match(r"(\.LBB\d+_\d+):", ls)
    if m:
        name = m.group(1) + (" LOOP" if "Loop" in ls else "")
        blocks[name] = []
        order.append(name)
        continue
    if name and ls and not ls.startswith((";", ".")):
        blocks[name].append(ls.split()[0])
print("loop blocks of the F16 256x160 ping-pong kernel:")
for n in order:
    if "LOOP" not in n or not blocks[n]:
        continue
    c = Counter(blocks[n])
    valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
    salu = sum(v for k, v in c.items() if k.startswith("s_"))
    keys = {k: v for k, v in c.items() if k.startswith(("v_mfma", "ds_read", "global_load", "buffer_load", "s_barrier", "v_readfirstlane", "v_readlane"))}
    print(f"  {n:18s} {len(blocks[n]):4d} instr  VALU {valu:3d}  SALU {salu:3d}  {keys}")

if "--uniformity" in sys.argv:
    import tempfile
    # common.h includes ../../include/vgen_hip.h: mirror that layout
    td = tempfile.mkdtemp()
    os.makedirs(os.path.join(td, "a", "b")); os.makedirs(os.path.join(td, "include"))
    import shutil
    shutil.copy(os.path.join(ROOT, "include", "vgen_hip.h"), os.path.join(td, "include"))
    for f in os.listdir(b.CSRC):
        if os.path.isfile(os.path.join(b.CSRC, f)):
            shutil.copy(os.path.join(b.CSRC, f), os.path.join(td, "a", "b"))
    ll = os.path.join(td, "a", "b", "tapgemm.ll")
    r = subprocess.run([b._hipcc()] + flags + ["-S", "-emit-llvm", "--cuda-device-only", os.path.join(td, "a", "b", "tapgemm.hip"), "-o", ll],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    ir = open(ll).read()
    print("allocas in the optimised IR (all kernels):", ir.count(" = alloca "))
    fn = KN + "v17vgen_tapgemm_argsiPfi"
    i = ir.index("@" + fn + "(")
    i = ir.rfind("\ndefine", 0, i) + 1
    j = ir.index("\n}\n", i) + 3
    decls = "\n".join(l for l in ir.split("\n") if l.startswith(("declare", "attributes", "!", "@", "target", "%")))
    one = os.path.join(td, "one.ll")
    open(one, "w").write(decls + "\n" + ir[i:j].replace(" comdat {", " {"))
    opt = os.path.join(os.path.dirname(os.path.realpath(b._hipcc())), "..", "lib", "llvm", "bin", "opt")
    if not os.path.exists(opt):
        opt = "/opt/rocm/lib/llvm/bin/opt"
    u = subprocess.run([opt, "-passes=print<uniformity>", "-disable-output", one], capture_output=True, text=True).stderr
    div = set(re.findall(r"DIVERGENT:\s+(%\d+) =", u))
    calls = [l for l in open(one).read().split("\n") if "raw.ptr.buffer.load.lds(" in l and "call" in l]
    bad = 0
    for c in calls:
        m = re.search(r"\(ptr addrspace\(8\)(?: \w+)* (%\d+), ptr addrspace\(3\)(?: \w+)* [^,]+, i32 16, i32 (%\d+|\d+), i32 (%\d+|\d+),", c)
        if m and (m.group(1) in div or m.group(3) in div):
            bad += 1
    print(f"uniformity: {len(div)} divergent values; {len(calls)} buffer LDS-DMA calls, {bad} with a divergent resource / scalar offset")
    shutil.rmtree(td)
