"""Build libofdm_hip.so (the C-ABI library) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the authoring container and on the GPU box alike.
The built .so is git-ignored but travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libofdm_hip.so")
# the profile build (-DOFDM_PROFILE_BUILD=1): same sources with the kernels' ablation exits and s_memtime section timers compiled
# in; used by tools/ only (ofdm_amd._lib.use_profile_build()), never by the tests, the bench line or smoke()
LIB_PROFILE = os.path.join(HERE, "libofdm_hip_profile.so")
OBJ_PROFILE = os.path.join(CSRC, "_obj_profile")
SOURCES = ["kernels_sym.hip", "kernels_fast.hip", "kernels_rx1024.hip", "kernels_mid.hip", "kernels_sync.hip", "kernels_sc80.hip", "kernels_scstream.hip", "kernels_scbig.hip", "kernels_bytes.hip", "ofdm_abi.hip", "ofdm_host_path.hip", "outer_code.hip"]
HEADERS = ["device_common.hpp", "kernels.hpp", "ofdm_ctx.hpp", "ofdm_hip_tuning.h", os.path.join("..", "..", "include", "ofdm_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]
# per-file additions.  kernels_sync: the SLP vectoriser pairs f32 ops into v_pk_* and pays for it in v_mov (measured:
# 332 -> 246 VALU in the filter's phase 1, 16 fewer VGPRs)
EXTRA_FLAGS = {"kernels_sync.hip": ["-fno-slp-vectorize"], "kernels_scbig.hip": ["-fno-slp-vectorize"], "kernels_scstream.hip": ["-fno-slp-vectorize"], "kernels_sc80.hip": ["-fno-slp-vectorize"], "kernels_fast.hip": ["-fno-slp-vectorize"], "kernels_rx1024.hip": ["-fno-slp-vectorize"], "kernels_mid.hip": ["-fno-slp-vectorize"], "kernels_sym.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libofdm_hip.so cannot be built (ROCm 7.x with gfx950 support is required)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


class _BuildLock:
    """Exclusive advisory lock on csrc/_obj/.build.lock: N ranks importing the package at once (bench.py --gpus N,
    torchrun) must never run hipcc into the same _obj/*.o / libofdm_hip.so concurrently.  The first rank builds, the
    others block here and then find everything up to date."""

    def __enter__(self):
        import fcntl

        os.makedirs(OBJ, exist_ok=True)
        self.fd = os.open(os.path.join(OBJ, ".build.lock"), os.O_CREAT | os.O_RDWR, 0o644)
        fcntl.flock(self.fd, fcntl.LOCK_EX)
        return self

    def __exit__(self, *exc):
        import fcntl

        fcntl.flock(self.fd, fcntl.LOCK_UN)
        os.close(self.fd)
        return False


def build(force: bool = False, verbose: bool = False, profile: bool = False) -> str:
    with _BuildLock():
        return _build_locked(force, verbose, profile)


def _build_locked(force: bool, verbose: bool, profile: bool = False) -> str:
    hipcc = _hipcc()
    obj_dir, lib = (OBJ_PROFILE, LIB_PROFILE) if profile else (OBJ, LIB)
    flags = FLAGS + (["-DOFDM_PROFILE_BUILD=1"] if profile else [])
    os.makedirs(obj_dir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(obj_dir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            jobs.append([hipcc, *flags, *EXTRA_FLAGS.get(s, []), "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    objs = [os.path.join(obj_dir, s.replace(".hip", ".o")) for s in srcs]
    if force or jobs or _stale(lib, objs):
        tmp = lib + f".tmp{os.getpid()}"  # link beside the target, then rename: a reader never sees a half-written .so
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs])
        os.replace(tmp, lib)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, profile="--profile" in sys.argv))
