"""Build libfemo_hip.so in-tree with hipcc for gfx950 (no CPU fallback exists).

Staleness is decided by content, not by time stamps: the digest of every source the library is compiled
from (csrc/*.hip, csrc/*.h, include/*.h) and of the compiler flags is stored beside the library
(``libfemo_hip.srchash``).  The pair travels to the GPU box together; ``_lib.load()`` refuses (or rebuilds,
when hipcc is there) a library whose digest does not match the sources it sits next to."""
from __future__ import annotations

import contextlib
import fcntl
import glob
import hashlib
import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
INCLUDE = os.path.normpath(os.path.join(CSRC, "..", "..", "include"))
LIB = os.path.join(CSRC, "libfemo_hip.so")
HASHFILE = os.path.join(CSRC, "libfemo_hip.srchash")
SOURCES = ["femo_hip.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
         "-Wno-unused-value"]


def dependencies():
    """Every file the library is compiled from: the translation units, all headers next to them, the C ABI header."""
    deps = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))
                  + [os.path.join(INCLUDE, "femo_hip.h")])
    return deps


def source_digest() -> str:
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for d in dependencies():
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


@contextlib.contextmanager
def _build_lock(path):
    """Exclusive lock for one library's build: every rank of a torchrun / mp.spawn job calls ``load()`` at the same time and
    would otherwise link into the same file while another rank dlopens it.  The first to take the lock builds; the others
    wait, re-check the digest under the lock and find the library fresh."""
    fd = os.open(path + ".lock", os.O_CREAT | os.O_RDWR, 0o644)
    try:
        fcntl.flock(fd, fcntl.LOCK_EX)
        yield
    finally:
        fcntl.flock(fd, fcntl.LOCK_UN)
        os.close(fd)


def _write_atomic(path, text):
    tmp = f"{path}.{os.getpid()}.tmp"
    with open(tmp, "w") as fh:
        fh.write(text)
    os.replace(tmp, path)


def needs_build() -> bool:
    if not os.path.exists(LIB) or not os.path.exists(HASHFILE):
        return True
    with open(HASHFILE) as fh:
        return fh.read().strip() != source_digest()


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP library if it is missing or stale; returns its path."""
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libfemo_hip.so cannot be built (there is no CPU fallback)")
    with _build_lock(LIB):
        if not force and not needs_build():          # another process built it while this one waited for the lock
            return LIB
        digest = source_digest()
        tmp = f"{LIB}.{os.getpid()}.tmp"
        cmd = [hipcc, *FLAGS, "-o", tmp, *SOURCES]
        res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
        if verbose or res.returncode:
            print(" ".join(cmd))
            print(res.stdout, res.stderr)
        if res.returncode:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError("hipcc failed building libfemo_hip.so:\n" + res.stderr[-4000:])
        os.replace(tmp, LIB)
        _write_atomic(HASHFILE, digest + "\n")
    return LIB


# ---- libfemo_symbolic.so: the analysis phase, host C++ (g++, OpenMP); same content-digest rule
SYM_LIB = os.path.join(CSRC, "libfemo_symbolic.so")
SYM_HASHFILE = os.path.join(CSRC, "libfemo_symbolic.srchash")
SYM_FLAGS = ["-O3", "-std=c++17", "-fPIC", "-shared", "-fopenmp"]


def symbolic_digest() -> str:
    h = hashlib.sha256(" ".join(SYM_FLAGS).encode())
    for d in (os.path.join(CSRC, "symbolic.cpp"), os.path.join(INCLUDE, "femo_symbolic.h")):
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def symbolic_needs_build() -> bool:
    if not os.path.exists(SYM_LIB) or not os.path.exists(SYM_HASHFILE):
        return True
    with open(SYM_HASHFILE) as fh:
        return fh.read().strip() != symbolic_digest()


def build_symbolic(force: bool = False) -> str:
    if not force and not symbolic_needs_build():
        return SYM_LIB
    gxx = shutil.which("g++")
    if not gxx:
        raise RuntimeError("g++ not found: libfemo_symbolic.so cannot be built")
    with _build_lock(SYM_LIB):
        if not force and not symbolic_needs_build():
            return SYM_LIB
        digest = symbolic_digest()
        tmp = f"{SYM_LIB}.{os.getpid()}.tmp"
        res = subprocess.run([gxx, *SYM_FLAGS, "-o", tmp, "symbolic.cpp"], cwd=CSRC, capture_output=True, text=True)
        if res.returncode:
            if os.path.exists(tmp):
                os.remove(tmp)
            raise RuntimeError("g++ failed building libfemo_symbolic.so:\n" + res.stderr[-4000:])
        os.replace(tmp, SYM_LIB)
        _write_atomic(SYM_HASHFILE, digest + "\n")
    return SYM_LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
    print(build_symbolic(force=True))
