"""In-tree build of libcloops_hip.so with hipcc for gfx950 (cross-compiles without a GPU).

The library is several translation units (cloops_amd/csrc/*.hip over the shared header cl_common.h): each is
compiled to an object on its own (in parallel, only when it or a header changed), then linked -- an edit to one
kernel family recompiles one file."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
HDR = os.path.join(os.path.dirname(HERE), "include", "cloops_hip.h")
OUT = os.path.join(HERE, "libcloops_hip.so")
COMM_SRC = os.path.join(CSRC, "cloops_comm.cpp")
COMM_HDR = os.path.join(os.path.dirname(HERE), "include", "cloops_comm.h")
COMM_OUT = os.path.join(HERE, "libcloops_comm.so")
OBJDIR = os.path.join(os.path.dirname(HERE), "build", "obj")

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def headers():
    return sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [HDR]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(p) > t for p in sources() + headers())


def _compile(src, obj, devel, verbose):
    cmd = [hipcc()] + FLAGS + (["-DCLOOPS_DEVEL"] if devel else []) + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)


def build_comm(force=False, verbose=False):
    """libcloops_comm.so: the RCCL collectives of the multi-GPU path (include/cloops_comm.h), linked against librccl."""
    if not force and os.path.exists(COMM_OUT) and os.path.getmtime(COMM_OUT) >= max(os.path.getmtime(COMM_SRC), os.path.getmtime(COMM_HDR)):
        return COMM_OUT
    rocm_lib = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")
    cmd = [hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-x", "hip", COMM_SRC, "-o", COMM_OUT,
           "-L" + rocm_lib, "-lrccl", "-Wl,-rpath," + rocm_lib]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return COMM_OUT


def build(force=False, verbose=False, devel=False):
    """`devel=True` builds libcloops_hip_devel.so with -DCLOOPS_DEVEL (ablation / shape knobs read from the
    environment; loaded only when CLOOPS_DEVEL_LIB=1) -- the shipped library has none of them."""
    out = OUT.replace(".so", "_devel.so") if devel else OUT
    if not devel:
        try:
            build_comm(force, verbose)
        except (subprocess.CalledProcessError, OSError) as e:
            # the single-GPU path does not need RCCL: cloops_amd.comm.load() raises a clear ImportError when it is asked for
            print("cloops_amd.build: libcloops_comm.so not built (%s); the multi-GPU path is unavailable" % e, file=sys.stderr)
    if not devel and not force and not needs_build():
        return OUT
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_t = max(os.path.getmtime(p) for p in headers())
    jobs, objs = [], []
    import hashlib
    tag = hashlib.sha1(" ".join(FLAGS + (["-DCLOOPS_DEVEL"] if devel else [])).encode()).hexdigest()[:8]      # other flags: other objects
    for src in sources():
        obj = os.path.join(OBJDIR, "%s_%s.o" % (os.path.basename(src)[:-4] + ("_devel" if devel else ""), tag))
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append((src, obj))
    with ThreadPoolExecutor(max_workers=max(1, min(8, len(jobs)))) as pool:
        for f in [pool.submit(_compile, s, o, devel, verbose) for s, o in jobs]:
            f.result()
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, devel="--devel" in sys.argv))
