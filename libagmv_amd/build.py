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


def _scratch_users(remarks):
    """kernels with a non-zero ScratchSize in hipcc's -Rpass-analysis=kernel-resource-usage remarks"""
    bad, name = [], None
    for line in remarks.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split("[")[0].strip()
        elif "ScratchSize [bytes/lane]:" in line:
            n = int(line.split("ScratchSize [bytes/lane]:")[1].split("[")[0])
            if n:
                bad.append("%s (%d bytes/lane)" % (name, n))
    return bad


def _compile(cmd, target, verbose, no_scratch=False):
    """compile to a temporary name, then rename: a concurrent reader never sees a half-written library.
    no_scratch: refuse a device build in which any kernel spills to scratch memory (round 3: a 12-byte spill in k_fp_walk
    gave sporadically wrong block counts on the MI355X; every kernel of this library is sized to stay in registers)"""
    tmp = target + ".tmp.%d" % os.getpid()
    cmd = [tmp if c == target else c for c in cmd]
    if no_scratch:
        cmd = cmd[:1] + ["-Rpass-analysis=kernel-resource-usage"] + cmd[1:]
    if verbose:
        print(" ".join(cmd))
    try:
        if no_scratch:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)
            keep, in_remark = [], False                      # the remarks (and the source lines quoted under them) are not shown
            for l in r.stderr.splitlines():
                if "kernel-resource-usage" in l:
                    in_remark = True
                elif in_remark and l.lstrip()[:1] in ("|", "") or in_remark and l.split("|")[0].strip().isdigit():
                    pass
                else:
                    in_remark = False
                    keep.append(l)
            rest = "\n".join(keep)
            if rest.strip():
                sys.stderr.write(rest + "\n")
            if r.returncode:
                raise subprocess.CalledProcessError(r.returncode, cmd)
            bad = _scratch_users(r.stderr)
            if bad:
                raise RuntimeError("kernels spilling to scratch memory: " + ", ".join(bad))
        else:
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
        _compile([HIPCC, "--offload-arch=" + ARCH, "-O3", "-fPIC", "-shared", "-std=c++17", hip_src, "-o", hip_so], hip_so, verbose, no_scratch=True)
    c_srcs = sorted(glob.glob(os.path.join(CSRC, "*.c")))
    if c_srcs:
        host_so = os.path.join(HERE, "libagmv.so")
        if force or _stale(host_so, c_srcs + hdrs + [hip_so]):
            _compile(["gcc", "-O2", "-fPIC", "-shared", "-std=gnu11", "-Wall", "-I" + os.path.join(ROOT, "include")] + c_srcs +
                     ["-o", host_so, "-L" + HERE, "-lagmv_hip", "-Wl,-rpath,$ORIGIN", "-lpthread", "-lm"], host_so, verbose)
    return hip_so


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
