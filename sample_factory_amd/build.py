"""Build libsf_hip.so for gfx950 with hipcc (cross-compiles without a GPU).  `python -m sample_factory_amd.build`."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsf_hip.so")

# (source, extra flags).  sf_rl keeps the reference's fp32 op order -> no FMA contraction; sf_nn is MFMA/FMA code.
SOURCES = [
    ("sf_rl.hip", ["-ffp-contract=off"]),
    ("sf_nn.hip", []),
    ("sf_rnn.hip", ["-ffp-contract=off"]),  # the fused cell arithmetic must equal k_rnn_cell_* of sf_rl.hip
    ("sf_dp.hip", []),                      # host code: RCCL gradient exchange (librccl resolved with dlopen at first use)
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
          f"-I{os.path.join(ROOT, 'include')}", f"-I{CSRC}"]


def source_sha16() -> str:
    """sha256[:16] over the kernel sources (csrc/*.hip, csrc/*.h, include/sf_hip.h, in name order): stamps measurements
    that are only valid for the code they were taken on (profiles/*traffic*.json, read back by bench.py)"""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h")))
    for path in files + [os.path.join(ROOT, "include", "sf_hip.h")]:
        h.update(os.path.basename(path).encode())
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "sf_hip.h"))
    objs = []
    for src, flags in SOURCES:
        s = os.path.join(CSRC, src)
        if not os.path.exists(s):
            continue
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + COMMON + flags + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
