"""In-tree build of the native libraries (no JIT cache: the .so travels with the repo snapshot).

  python -m libagmv_amd.build          # libagmv_amd/libagmv_hip.so (+ libagmv.so once csrc/*.c exist)
"""
import fcntl
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROOT = os.path.dirname(HERE)
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _compile(cmd, target, verbose):
    """compile to a temporary name, then rename: a concurrent reader never sees a half-written library"""
    tmp = target + ".tmp.%d" % os.getpid()
    cmd = [tmp if c == target else c for c in cmd]
    if verbose:
        print(" ".join(cmd))
    try:
        subprocess.run(cmd, check=True)
        os.replace(tmp, target)
    finally:
        if os.path.exists(tmp):
            os.unlink(tmp)


def build(force=False, verbose=False):
    # one builder at a time (the ranks of a multi-GPU job all call this): the others wait, then find the result fresh
    with open(os.path.join(HERE, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            return _build(force, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build(force=False, verbose=False):
    hip_src = os.path.join(CSRC, "agmv_hip.hip")
    hdrs = glob.glob(os.path.join(ROOT, "include", "*.h"))
    hip_so = os.path.join(HERE, "libagmv_hip.so")
    if force or _stale(hip_so, [hip_src] + hdrs):
        _compile([HIPCC, "--offload-arch=" + ARCH, "-O3", "-fPIC", "-shared", "-std=c++17", hip_src, "-o", hip_so], hip_so, verbose)
    c_srcs = sorted(glob.glob(os.path.join(CSRC, "*.c")))
    if c_srcs:
        host_so = os.path.join(HERE, "libagmv.so")
        if force or _stale(host_so, c_srcs + hdrs + [hip_so]):
            _compile(["gcc", "-O2", "-fPIC", "-shared", "-std=gnu11", "-Wall", "-I" + os.path.join(ROOT, "include")] + c_srcs +
                     ["-o", host_so, "-L" + HERE, "-lagmv_hip", "-Wl,-rpath,$ORIGIN", "-lpthread", "-lm"], host_so, verbose)
    return hip_so


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
