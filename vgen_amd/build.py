"""In-tree build of libvgen_hip.so (gfx950) with hipcc.

`python -m vgen_amd.build` or `vgen_amd.build.build()`.  hipcc cross-compiles without a GPU;
objects are cached by source mtime.  The shared object stays next to this file so that it
travels with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libvgen_hip.so")
SOURCES = ["cabi.cpp", "tapgemm.hip", "panelgemm.hip", "norms.hip", "attention.hip", "misc.hip", "stems.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "tapgemm_plans.inc"),
           os.path.join(HERE, "..", "include", "vgen_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
         "-x", "hip"]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libvgen_hip.so)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


TUNING_LIB = os.path.join(HERE, "libvgen_hip_tuning.so")
HOST_LIB = os.path.join(HERE, "libvgen_host.so")
# HOST code of the pack-time calibration (csrc/host_round.cpp): plain g++, no device part.  x86-64-v3 = AVX2 + F16C (the
# fp16 round-to-nearest-even conversion), which every EPYC host of an MI355X has; -ffp-contract=off keeps the loop's
# arithmetic identical to its torch restatement (no fused multiply-add).
HOST_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-Wall", "-pthread"]
HOST_ARCH = (["-march=x86-64-v3"], ["-mavx2", "-mf16c"])   # g++ >= 11 knows the level name; older ones the two features


def build_host(force: bool = False, verbose: bool = False) -> str:
    """libvgen_host.so next to libvgen_hip.so (git-ignored, travels with the snapshot).  An OPTIONAL accelerator of the
    pack-time calibration: vgen_amd/calibrate.py falls back to its (bit-identical, ~50x slower) torch loop without it, so
    callers that build everything (`build_all`, __graft_entry__.build) treat a failure here as a warning."""
    src = os.path.join(CSRC, "host_round.cpp")
    if force or _stale(HOST_LIB, [src]):
        gxx = os.environ.get("CXX") or shutil.which("g++") or shutil.which("c++")
        if not gxx:
            raise RuntimeError("g++ not found (need a host C++ compiler to build libvgen_host.so)")
        err = None
        for arch in HOST_ARCH:
            cmd = [gxx] + HOST_FLAGS + arch + [src, "-o", HOST_LIB]
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode == 0:
                err = None
                break
            err = "build failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr)
        if err:
            raise RuntimeError(err)
    return HOST_LIB


def build_host_optional(force: bool = False, verbose: bool = False):
    """build_host, with a toolchain problem (no g++, an old one, a host without AVX2 / F16C) reported as a warning: the
    device library and everything that needs it stay usable, calibration runs on its torch loop."""
    try:
        return build_host(force=force, verbose=verbose)
    except (RuntimeError, OSError) as exc:
        import warnings
        warnings.warn(f"libvgen_host.so not built ({str(exc).splitlines()[0]}); vgen_amd.calibrate uses its torch loop")
        return None


def build(force: bool = False, verbose: bool = False, tuning: bool = False, variant: str = "", defines=()) -> str:
    """tuning=True builds libvgen_hip_tuning.so with -DVGEN_TUNING: the tap-GEMM's phase-ablation switches and plan
    overrides (environment variables read per launch) exist only there — tools/ select it with VGEN_HIP_LIB; the product
    library cannot be told to skip work.  variant / defines: an experimental build libvgen_hip_<variant>.so with extra
    -D flags, for same-box A/B runs of two kernel versions (VGEN_HIP_LIB=...; tools/ab_libs.sh)."""
    if tuning:
        variant, defines = "tuning", tuple(defines) + ("VGEN_TUNING",)
    obj_dir = OBJ + ("_" + variant if variant else "")
    lib_path = os.path.join(HERE, f"libvgen_hip_{variant}.so") if variant else LIB
    flags = FLAGS + ["-D" + d for d in defines]
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = _hipcc()
    jobs = []
    objs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(obj_dir, os.path.splitext(src)[0] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + HEADERS):
            jobs.append([hipcc] + flags + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("build failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        return r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 2)) as ex:
            list(ex.map(run, jobs))
    if force or jobs or _stale(lib_path, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs)
    return lib_path


if __name__ == "__main__":
    _var = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--variant=")), "")
    print(build(force="--force" in sys.argv, verbose=True, tuning="--tuning" in sys.argv, variant=_var,
                defines=tuple(a[2:] for a in sys.argv if a.startswith("-D"))))
    print(build_host_optional(force="--force" in sys.argv, verbose=True))
