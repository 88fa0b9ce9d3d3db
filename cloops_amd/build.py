"""In-tree build of libcloops_hip.so with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "cloops_hip.hip")
HDR = os.path.join(os.path.dirname(HERE), "include", "cloops_hip.h")
OUT = os.path.join(HERE, "libcloops_hip.so")


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(p) > t for p in (SRC, HDR))


def build(force=False, verbose=False, devel=False):
    """`devel=True` builds libcloops_hip_devel.so with -DCLOOPS_DEVEL (ablation / shape knobs read from the
    environment; loaded only when CLOOPS_DEVEL_LIB=1) -- the shipped library has none of them."""
    out = OUT.replace(".so", "_devel.so") if devel else OUT
    if not devel and not force and not needs_build():
        return OUT
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-ffp-contract=off", "-Wall", "-Wno-unused-function", SRC, "-o", out]
    if devel:
        cmd.insert(1, "-DCLOOPS_DEVEL")
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, devel="--devel" in sys.argv))
