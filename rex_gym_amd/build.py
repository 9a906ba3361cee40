"""Build the HIP extension in-tree: rex_gym_amd/librexsim_hip.so (gfx950, hipcc cross-compiles without a GPU)."""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.environ.get("REX_LIB_PATH") or os.path.join(PKG_DIR, "librexsim_hip.so")   # REX_LIB_PATH: developer A/B builds
SOURCES = ["rexsim.hip"]
HEADERS = sorted(f for f in os.listdir(CSRC) if f.endswith(".h"))   # every header rexsim.hip can include
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fvisibility=hidden"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the rexsim HIP extension cannot be built")


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.join(PKG_DIR, "..", "include", "rexsim.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile csrc/rexsim.hip -> librexsim_hip.so. Returns the library path."""
    if not force and not needs_build():
        return LIB_PATH
    cmd = [_hipcc()] + HIPCC_FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
