"""In-tree build of the native libraries (no JIT cache: the .so travels with the repo snapshot).

  python -m libagmv_amd.build          # libagmv_amd/libagmv_hip.so (+ libagmv.so once csrc/*.c exist)
"""
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


def build(force=False, verbose=False):
    hip_src = os.path.join(CSRC, "agmv_hip.hip")
    hdrs = glob.glob(os.path.join(ROOT, "include", "*.h"))
    hip_so = os.path.join(HERE, "libagmv_hip.so")
    if force or _stale(hip_so, [hip_src] + hdrs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-O3", "-fPIC", "-shared", "-std=c++17", hip_src, "-o", hip_so]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True)
    c_srcs = sorted(glob.glob(os.path.join(CSRC, "*.c")))
    if c_srcs:
        host_so = os.path.join(HERE, "libagmv.so")
        if force or _stale(host_so, c_srcs + hdrs + [hip_so]):
            cmd = ["gcc", "-O2", "-fPIC", "-shared", "-std=gnu11", "-Wall", "-I" + os.path.join(ROOT, "include")] + c_srcs + \
                  ["-o", host_so, "-L" + HERE, "-lagmv_hip", "-Wl,-rpath,$ORIGIN", "-lpthread", "-lm"]
            if verbose:
                print(" ".join(cmd))
            subprocess.run(cmd, check=True)
    return hip_so


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
