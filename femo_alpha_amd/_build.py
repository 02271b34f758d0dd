"""Build libfemo_hip.so in-tree with hipcc for gfx950 (no CPU fallback exists)."""
from __future__ import annotations

import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libfemo_hip.so")
SOURCES = ["femo_hip.hip"]
HEADERS = ["shell_device.h", os.path.join("..", "..", "include", "femo_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-munsafe-fp-atomics",
         "-Wno-unused-value"]


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile the HIP library if it is missing or stale; returns its path."""
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: libfemo_hip.so cannot be built (there is no CPU fallback)")
    cmd = [hipcc, *FLAGS, "-o", LIB, *SOURCES]
    res = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if verbose or res.returncode:
        print(" ".join(cmd))
        print(res.stdout, res.stderr)
    if res.returncode:
        raise RuntimeError("hipcc failed building libfemo_hip.so:\n" + res.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
