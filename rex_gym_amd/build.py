"""Build the HIP extension in-tree: rex_gym_amd/librexsim_hip.so (gfx950, hipcc cross-compiles without a GPU).

The library is compiled variant group by variant group -- one translation unit per group of rex_step_kernel /
rex_settle_kernel instantiations (csrc/rex_step_*.hip, rex_settle_*.hip) next to the C ABI (csrc/rexsim.hip) -- in
parallel, then linked: 40 s on 8 cores where the single translation unit took over 2 minutes.

Developer knobs (never needed for the product build):
  REX_LIB_PATH=<path>     write / load the library somewhere else (A/B builds)
  REX_BUILD_ONLY=base,arm compile only these variant groups' kernels; the other launchers become stubs that report an error
  build(defines=[...], unity=True)  one translation unit (tools/prof_sections.py: -DREX_PROF keeps its counters in one
                          device global)
"""
import os
import shutil
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.environ.get("REX_LIB_PATH") or os.path.join(PKG_DIR, "librexsim_hip.so")   # REX_LIB_PATH: developer A/B builds
# translation unit -> variant group name (REX_BUILD_ONLY) ; rexsim.hip (C ABI, reset and controller kernels) is always built
GROUPS = {"rex_step_base.hip": "base", "rex_step_arm.hip": "arm", "rex_step_mixed_base.hip": "mixed_base",
          "rex_step_mixed_arm.hip": "mixed_arm", "rex_step_body.hip": "body", "rex_settle_base.hip": "base", "rex_settle_arm.hip": "arm"}
SOURCES = ["rexsim.hip"] + sorted(GROUPS)
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h"))   # every header the sources can include
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the rexsim HIP extension cannot be built")


def needs_build(lib_path=None):
    lib_path = lib_path or LIB_PATH
    if not os.path.exists(lib_path):
        return True
    t = os.path.getmtime(lib_path)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(PKG_DIR, "..", "include", "rexsim.h")]
    return any(os.path.getmtime(d) > t for d in deps)


_STUB = """// stub of a variant group left out of a developer build (REX_BUILD_ONLY)
#include "rex_kernels.h"
#include <cstdio>
#include <cstdlib>
%s
"""


def _stub_source(tu):
    name = tu[:-4]
    if name.startswith("rex_step_"):
        sig = "void rex_launch_%s(RexSim*, int, hipStream_t, const float*, float*, float*, uint8_t*, float*)" % name[4:]
    else:
        sig = "void rex_launch_%s(RexSim*, int, hipStream_t, float*)" % name[4:]
    return _STUB % (sig + ' { fprintf(stderr, "librexsim_hip.so: variant group %s was left out of this developer build (REX_BUILD_ONLY)\\n"); abort(); }' % name)


def build(force=False, verbose=False, lib_path=None, defines=(), unity=False, only=None, jobs=None):
    """Compile csrc/*.hip -> librexsim_hip.so. Returns the library path."""
    lib_path = lib_path or LIB_PATH
    if not force and not needs_build(lib_path):
        return lib_path
    hipcc = _hipcc()
    only = only if only is not None else os.environ.get("REX_BUILD_ONLY")
    keep = set(only.split(",")) if only else None
    flags = HIPCC_FLAGS + list(defines) + ["-I", CSRC]
    with tempfile.TemporaryDirectory(prefix="rexsim_build_") as tmp:
        if unity:
            uni = os.path.join(tmp, "unity.hip")
            with open(uni, "w") as f:
                f.write("".join('#include "%s"\n' % os.path.join(CSRC, s) for s in SOURCES))
            cmd = [hipcc] + flags + ["-shared", uni, "-o", lib_path]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            return lib_path
        jobs_list = []
        for s in SOURCES:
            src = os.path.join(CSRC, s)
            if keep is not None and s in GROUPS and GROUPS[s] not in keep:
                src = os.path.join(tmp, "stub_" + s)
                with open(src, "w") as f:
                    f.write(_stub_source(s))
            obj = os.path.join(tmp, s[:-4] + ".o")
            jobs_list.append(([hipcc] + flags + ["-c", src, "-o", obj], obj))

        def run(job):
            if verbose:
                print(" ".join(job[0]), flush=True)
            subprocess.check_call(job[0])
            return job[1]

        with ThreadPoolExecutor(max_workers=jobs or min(len(jobs_list), os.cpu_count() or 4)) as pool:
            objs = list(pool.map(run, jobs_list))
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib_path]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return lib_path


if __name__ == "__main__":
    print(build(force=True, verbose=True))
