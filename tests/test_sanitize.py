"""The C-ABI layer under AddressSanitizer + UBSan (SURVEY section 5's build-hygiene note): `IAN_SANITIZE=1` builds
libian_asan.so from the SAME sources -- host translation units (ian_runtime.cpp, ian_train_abi.cpp, ian_trainer.cpp)
instrumented, every device allocation wrapped in NaN-filled guard bands that are verified at each free and at
ian_destroy / ian_layer_destroy / ian_trainer_destroy (csrc/ian_guard.h).  A child Python with the ASan runtime
preloaded runs real tests against that library (IAN_LIB): host-logic tests here on CPU, parity tests on the GPU."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _sanitized_env():
    from neural_photo_editor_amd import build
    if not build.have_hipcc():
        pytest.skip("hipcc is needed to build libian_asan.so")
    build.build()                                                  # the kernel objects are the product build's
    env = dict(os.environ, IAN_SANITIZE="1")
    r = subprocess.run([sys.executable, "-c", "from neural_photo_editor_amd import build as b; print(b.build()); print(b.asan_runtime())"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=1200)
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    lib, rt = r.stdout.decode().strip().splitlines()[-2:]
    assert lib.endswith("libian_asan.so") and os.path.exists(lib) and os.path.exists(rt)
    env = dict(os.environ)
    env.pop("IAN_SANITIZE", None)
    env.update(IAN_LIB=lib, LD_PRELOAD=rt, UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:protect_shadow_gap=0:detect_odr_violation=0:verify_asan_link_order=0")
    return env


def _run(env, args, timeout):
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    out = r.stdout.decode(errors="replace")
    assert "AddressSanitizer" not in out and "runtime error:" not in out and "IAN_SANITIZE:" not in out, out[-4000:]
    assert r.returncode == 0, out[-4000:]
    return out


def test_host_side_of_the_c_abi_under_asan_ubsan():
    env = _sanitized_env()
    out = _run(env, ["tests/test_host.py", "-k",
                     "library_builds or exports_every or without_gpu or headers_are_plain or reference_surface"], 900)
    assert " passed" in out


@pytest.mark.gpu
def test_parity_and_training_step_under_asan_with_guard_bands():
    """Golden encode/decode, brush gradients (both archs), ragged batches and the C training step against the sanitized
    library: no ASan / UBSan report, no guard band touched, and the numbers still pass their parity bars (a descriptor that
    read past a buffer would now return NaN patterns instead of zeros or a neighbour's values)."""
    env = _sanitized_env()
    out = _run(env, ["tests/test_gpu_parity.py", "-m", "gpu", "-k", "golden_encode_decode or brush_gradients_vs_golden or (ragged_batches and 5-)"], 1500)   # one ragged size per arch: the
    # descriptor arithmetic is the same at 1 / 3 / 5 / 33 images and the product library runs all four
    assert " passed" in out
    out = _run(env, ["tests/test_gpu_train_step.py", "-m", "gpu", "-k", "host_buffers_only"], 1500)
    assert " passed" in out
