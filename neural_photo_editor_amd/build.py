"""Builds libian.so (hand-written HIP kernels + host runtime) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this also runs in the CPU-only container.
The .so stays in the package directory: it is git-ignored but travels to the GPU
box with the repo snapshot.
"""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["kernels_tapgemm.hip", "kernels_misc.hip", "kernels_head.hip", "kernels_npe.hip", "kernels_wgrad.hip", "kernels_train.hip", "ian_runtime.cpp", "ian_train_abi.cpp", "ian_trainer.cpp", "ian_comm_rccl.cpp"]
HEADERS = ["ian_internal.h", "ian_guard.h", "ian_rt_types.h", "ian_rt_util.inc", "ian_rt_pack.inc", "ian_rt_schedule.inc", "ian_rt_exec.inc",
           "ian_rt_autotune.inc", "ian_rt_io.inc", "ian_rt_backward.inc", "ian_rt_edit.inc", "ian_rt_api.inc", "ian_rt_layer.inc", os.path.join("..", "..", "include", "ian.h"), os.path.join("..", "..", "include", "ian_train.h")]
# IAN_ABLATION_BUILD=1 (tests/test_gpu_ablation.py, scripts/ablate_tapgemm.sh): a SEPARATE library, libian_ablation.so, with
# -DIAN_ABLATION: the measured NEGATIVE results kept runnable (tapgemm K-loop schedules 0 and 3, the in-launch split-K
# combine, the batch-1 streaming deconv of kernels_b1.hip, the 4-wave tapwgrad tile -- all bitwise / parity tested there)
# plus the timing-only tapgemm ablations.  The product library contains none of them and rejects their option values.
ABLATION = bool(os.environ.get("IAN_ABLATION_BUILD"))
if ABLATION:
    SOURCES = SOURCES[:1] + ["kernels_b1.hip"] + SOURCES[1:]
# IAN_SANITIZE=1: libian_asan.so -- the HOST translation units of the C-ABI layer (runtime, training ABI, trainer) under
# AddressSanitizer + UBSan with guard bands around every device allocation (csrc/ian_guard.h); the .hip objects are the
# product build's.  Loaded by tests through IAN_LIB=<path> with the ASan runtime preloaded (tests/test_sanitize.py).
SANITIZE = bool(os.environ.get("IAN_SANITIZE")) and not ABLATION
# IAN_NOFUSE_BUILD=1 (scripts/exp/tgfuse_ab.py): libian_nofuse.so = the product sources with -DIAN_NO_TG_FUSE, i.e. the small tap-GEMM
# tiles WITHOUT round 5's in-launch split-K combine epilogue (their round-4 object code), for an in-process A/B against libian.so.
NOFUSE = bool(os.environ.get("IAN_NOFUSE_BUILD")) and not ABLATION and not SANITIZE
LIB = os.path.join(HERE, "libian_ablation.so" if ABLATION else ("libian_asan.so" if SANITIZE else ("libian_nofuse.so" if NOFUSE else "libian.so")))
STAMP = os.path.join(HERE, ".libian_ablation.stamp" if ABLATION else (".libian_asan.stamp" if SANITIZE else (".libian_nofuse.stamp" if NOFUSE else ".libian.stamp")))
HOST_SOURCES = ("ian_runtime.cpp", "ian_train_abi.cpp", "ian_trainer.cpp", "ian_comm_rccl.cpp")
# The host units contain no device code (they call the launch_* wrappers of the .hip files), so the sanitized build compiles
# them with g++ against the HIP runtime API: GCC's ASan runtime, unlike ROCm clang's, does not intercept the HSA allocator
# (under the ROCm one every process that touches the GPU dies in hsa_amd_memory_pool_allocate on this image).
SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g", "-DIAN_SANITIZE", "-D__HIP_PLATFORM_AMD__"]


def _rocm_include():
    return os.path.join(os.path.dirname(os.path.dirname(os.path.realpath(_hipcc()))), "include")


def asan_runtime():
    """Path of the ASan runtime to LD_PRELOAD when a non-instrumented executable (python) loads libian_asan.so."""
    out = subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE).stdout.decode().strip()
    out = os.path.realpath(out)
    if os.path.exists(out) and os.path.isabs(out):
        return out
    raise RuntimeError("GCC's libasan.so not found")
ARCH = "gfx950"


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libian.so cannot be built")


# sources that only the training step uses: a change there cannot alter a reconstruction / brush kernel
TRAINING_ONLY = ("kernels_train.hip", "kernels_wgrad.hip", "ian_train_abi.cpp", "ian_trainer.cpp", "ian_comm_rccl.cpp", "ian_rt_layer.inc",
                 os.path.join("..", "..", "include", "ian_train.h"))


def _digest(scope=None):
    """sha256 over the sources.  scope="inference": without the training-only files -- what a committed rocprofv3 summary of
    the reconstruction workloads (profiles/r*_ian_*.json) is valid for; None: everything (the library stamp, the training profile)."""
    h = hashlib.sha256()
    for f in SOURCES + HEADERS:
        if scope == "inference" and f in TRAINING_ONLY:
            continue
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(fh.read())
    h.update(ARCH.encode())
    h.update(b"ablation" if ABLATION else b"")
    h.update(b"sanitize-g++-libasan" if SANITIZE else b"")
    h.update(b"no-tg-fuse" if NOFUSE else b"")

    return h.hexdigest()


def _fresh(dig):
    if os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as fh:
            return fh.read().strip() == dig
    return False


def have_hipcc():
    try:
        _hipcc()
        return True
    except RuntimeError:
        return False


def is_fresh():
    """The in-tree libian.so was built from exactly the sources now under csrc/."""
    return _fresh(_digest())


def build(force=False, verbose=False):
    """Compile every HIP source for gfx950 and link libian.so. Returns the library path.
    Safe to call from several processes at once (one rank per GPU): the build is serialised by a file lock and the
    late comers find a fresh library."""
    import fcntl
    dig = _digest()
    if not force and _fresh(dig):
        return LIB
    lock = open(os.path.join(HERE, ".libian.lock"), "w")
    try:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and _fresh(dig):
            return LIB
        return _build_locked(dig, verbose)
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _build_locked(dig, verbose):
    hipcc = _hipcc()
    objs = []
    bdir = os.path.join(HERE, "build_ablation" if ABLATION else ("build_nofuse" if NOFUSE else "build"))
    os.makedirs(bdir, exist_ok=True)
    procs = []
    for src in SOURCES:
        obj = os.path.join(bdir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if ABLATION:
            cmd.insert(1, "-DIAN_ABLATION")
        if NOFUSE:
            cmd.insert(1, "-DIAN_NO_TG_FUSE")
        if SANITIZE:
            if src not in HOST_SOURCES:
                if not os.path.exists(obj):
                    raise RuntimeError("IAN_SANITIZE=1 reuses the product build's kernel objects: build libian.so first")
                objs.append(obj)
                continue
            sdir = os.path.join(HERE, "build_asan")
            os.makedirs(sdir, exist_ok=True)
            obj = os.path.join(sdir, os.path.splitext(src)[0] + ".o")
            cmd = ["g++", "-O1", "-std=c++17", "-fPIC", "-I" + _rocm_include()] + SAN_FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode(errors="replace")))
    tmp = LIB + ".%d.tmp" % os.getpid()      # never link over a library other ranks may have mapped
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp] + objs
    if SANITIZE:
        gccdir = os.path.dirname(os.path.realpath(subprocess.run(["gcc", "-print-file-name=libasan.so"], stdout=subprocess.PIPE).stdout.decode().strip()))
        cmd += ["-L" + gccdir, "-lasan", "-lubsan"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stdout.decode(errors="replace"))
    os.replace(tmp, LIB)
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
