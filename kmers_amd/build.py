"""Build libkmers.so (gfx950) in-tree with hipcc.  `python -m kmers_amd.build` or __graft_entry__.build()."""
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIB_DIR, "libkmers.so")
SOURCES = ["kernels.hip", "model.hip", "simreads.hip", "ingest.cpp", "libkmers.cpp", "npz.cpp"]
HEADERS = [os.path.join(CSRC, h) for h in ("kernels.hpp", "store.hpp")] + \
          [os.path.join(ROOT, "include", h) for h in ("libkmers.h", "kmers_hip.h")]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the library can only be built with the ROCm toolchain")
    return exe


def up_to_date():
    if not os.path.exists(LIB):
        return False
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return all(os.path.getmtime(d) <= t for d in deps)


def build(force=False, verbose=False, stamps=False, defines=(), variant=None):
    """Compile kmers_amd/lib/libkmers.so for gfx950.  Cross-compiles without a GPU.
    stamps=True builds the diagnostic variant lib/libkmers_stamps.so (in-kernel phase timers; never the product)."""
    if variant:       # diagnostic A/B builds: lib/libkmers_<variant>.so with extra -D flags (never the product)
        out = os.path.join(LIB_DIR, f"libkmers_{variant}.so")
    elif stamps:
        out = os.path.join(LIB_DIR, "libkmers_stamps.so")
    else:
        out = LIB
        if not force and up_to_date():
            return LIB
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "-o", out]
    if stamps:
        cmd.append("-DKK_STAMPS")
    cmd += ["-D" + d for d in defines]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    cmd += ["-lz", "-lpthread"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    defs = [a[2:] for a in sys.argv[1:] if a.startswith("-D")]
    var = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--variant=")), None)
    print(build(force="--force" in sys.argv, verbose=True, stamps="--stamps" in sys.argv, defines=defs, variant=var))
