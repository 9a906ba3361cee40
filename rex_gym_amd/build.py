"""Build the HIP extension in-tree: rex_gym_amd/librexsim_hip.so (gfx950, hipcc cross-compiles without a GPU).

The library is compiled variant group by variant group -- one translation unit per group of rex_step_kernel /
rex_settle_kernel instantiations (csrc/rex_step_*.hip, rex_settle_*.hip) next to the C ABI (csrc/rexsim.hip) -- in
parallel, then linked: ~45 s on 8 cores (18 jobs: the step units are compiled three times, see TRACE_SOURCES / SEG_SOURCES) where the single translation
unit took over 2 minutes.

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
# the step translation units are compiled a second time with -DREX_TU_TRACE=1: the kernel instantiations with the event trace
# (rex_set_event_trace, a debug aid of the parity tests) compiled in -- the product kernels carry none of it
TRACE_SOURCES = sorted(f for f in GROUPS if f.startswith("rex_step_"))
# ... and a third time with -DREX_TU_SEG=1: the instantiations behind rex_step_segment (a loop over the steps of a rollout segment around
# the step; kept apart because the loop costs registers -- what the body forms from loop invariants is hoisted -- and rex_step's own
# kernels stay exactly what they were)
SEG_SOURCES = TRACE_SOURCES
# ... and the single-task toes-only units a fourth time with -DREX_TU_POL=1: the segment kernels with the reference's Gaussian MLP actor
# evaluated in front of every step (rex_step_policy / rex_step_segment_policy, csrc/rex_policy.h)
POL_SOURCES = ["rex_step_arm.hip", "rex_step_base.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h"))   # every header the sources can include
# -ffp-contract=on: a * b + c inside one expression is one fma and nothing else is fused -- the arithmetic of a kernel is fixed by its
# source and does not depend on what else is compiled into it (hipcc's default lets the backend fuse across statements by heuristics:
# the _trace instantiations then differ from the product kernels in the last bit).  csrc/rex_kernels.h repeats it as a pragma; the
# flag also covers the HIP headers' inline functions.  tools/kres.sh, kstat.sh and check_dpp_masks.py compile with COMPILE_FLAGS too.
COMPILE_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on"]
HIPCC_FLAGS = COMPILE_FLAGS + ["-fPIC", "-fvisibility=hidden"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the rexsim HIP extension cannot be built")


def _flags_stamp():
    # the compile flags fix the kernels' arithmetic (-ffp-contract=on): a flags-only edit of this file must rebuild too
    import hashlib
    return hashlib.sha256(" ".join(HIPCC_FLAGS + POL_SOURCES).encode()).hexdigest()[:16]


def needs_build(lib_path=None):
    lib_path = lib_path or LIB_PATH
    if not os.path.exists(lib_path):
        return True
    try:
        with open(lib_path + ".flags") as f:
            if f.read().strip() != _flags_stamp():
                return True
    except OSError:
        return True
    t = os.path.getmtime(lib_path)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(PKG_DIR, "..", "include", "rexsim.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def _stamp(lib_path, defines, only):
    if not defines and not only:          # (developer builds with extra defines / left-out groups are never "up to date")
        with open(lib_path + ".flags", "w") as f:
            f.write(_flags_stamp() + "\n")


_STUB = """// stub of a variant group left out of a developer build (REX_BUILD_ONLY)
#include "rex_kernels.h"
#include <cstdio>
#include <cstdlib>
%s
"""


def _stub_source(tu, trace=False):
    name = tu[:-4] + (trace if isinstance(trace, str) else ("_trace" if trace else ""))
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
                f.write("#undef REX_TU_TRACE\n#undef REX_STEP_LAUNCHER\n#undef REX_LAUNCH_STEP\n#define REX_TU_TRACE 1\n"
                        "#define REX_STEP_LAUNCHER(group) rex_launch_step_##group##_trace\n"
                        "#define REX_LAUNCH_STEP(EPW, ARM, MIXED, BODY) hipLaunchKernelGGL((rex::rex_step_kernel<EPW, ARM, MIXED, BODY, true>), "
                        "dim3(blocks), dim3(REX_WAVE), 0, st, s->dev, s->d_state, s->d_snap, a, o, r, d, m, rex::NoPol{})\n")
                f.write("".join('#include "%s"\n' % os.path.join(CSRC, s) for s in TRACE_SOURCES))
                f.write("#undef REX_STEP_LAUNCHER\n#undef REX_LAUNCH_STEP\n"
                        "#define REX_STEP_LAUNCHER(group) rex_launch_step_##group##_seg\n"
                        "#define REX_LAUNCH_STEP(EPW, ARM, MIXED, BODY) hipLaunchKernelGGL((rex::rex_step_kernel<EPW, ARM, MIXED, BODY, false, true>), "
                        "dim3(blocks), dim3(REX_WAVE), 0, st, s->dev, s->d_state, s->d_snap, a, o, r, d, m, rex::NoPol{})\n")
                f.write("".join('#include "%s"\n' % os.path.join(CSRC, s) for s in SEG_SOURCES))
                f.write("#undef REX_STEP_LAUNCHER\n#undef REX_LAUNCH_STEP\n#undef REX_TU_POL\n#define REX_TU_POL 1\n"
                        "#define REX_STEP_LAUNCHER(group) rex_launch_step_##group##_pol\n"
                        "#define REX_LAUNCH_STEP(EPW, ARM, MIXED, BODY) rex_launch_policy_kernel<EPW, ARM>(s, blocks, st, a, o, r, d, m)\n")
                f.write("".join('#include "%s"\n' % os.path.join(CSRC, s) for s in POL_SOURCES))
            cmd = [hipcc] + flags + ["-shared", uni, "-o", lib_path]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            _stamp(lib_path, defines, only)
            return lib_path
        jobs_list = []
        for s, trace in [(s, "") for s in SOURCES] + [(s, "_trace") for s in TRACE_SOURCES] + [(s, "_seg") for s in SEG_SOURCES] + [(s, "_pol") for s in POL_SOURCES]:
            src = os.path.join(CSRC, s)
            tag = trace
            if keep is not None and s in GROUPS and GROUPS[s] not in keep:
                src = os.path.join(tmp, "stub_" + s[:-4] + tag + ".hip")
                with open(src, "w") as f:
                    f.write(_stub_source(s, trace))
            obj = os.path.join(tmp, s[:-4] + tag + ".o")
            jobs_list.append(([hipcc] + flags + ({"_trace": ["-DREX_TU_TRACE=1"], "_seg": ["-DREX_TU_SEG=1"], "_pol": ["-DREX_TU_POL=1"]}.get(trace, [])) + ["-c", src, "-o", obj], obj))

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
        _stamp(lib_path, defines, only)
    return lib_path


DIAG_LIB_PATH = os.path.join(PKG_DIR, "librexsim_hip_diag.so")
DIAG_INSIDE = "0.3"      # rad; oracle/Makefile DIAG_INSIDE is the same number


def build_diag(force=False, verbose=False):
    """The DIAGNOSTIC twin of the library for one parity test (tests/test_gpu_parity.py::test_mark_arm_window_with_quiet_arm_rows): the mark-arm
    variant groups compiled with -DREX_DIAG_ARM_REST_INSIDE=0.3 (csrc/rex_arm_model_gen.h: the arm's rest targets 0.3 rad inside their
    bounds instead of 0.1 rad beyond them).  Never loaded by the product: rex_gym_amd._lib loads LIB_PATH; the test points REX_LIB_PATH
    at this file in a subprocess."""
    if not force and os.path.exists(DIAG_LIB_PATH) and \
            all(os.path.getmtime(os.path.join(CSRC, f)) <= os.path.getmtime(DIAG_LIB_PATH) for f in SOURCES + HEADERS):
        return DIAG_LIB_PATH
    return build(force=True, verbose=verbose, lib_path=DIAG_LIB_PATH, defines=[f"-DREX_DIAG_ARM_REST_INSIDE={DIAG_INSIDE}"], only="arm,mixed_arm")


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    if "--diag" in __import__("sys").argv:
        print(build_diag(force=True, verbose=True))
