"""AddressSanitizer + UndefinedBehaviorSanitizer over the HOST side of the C ABI (the reference has no
sanitizer story; GPU sanitizers are not available on the pool, so this is the CPU build the task prescribes): the
library's sources are compiled with the host pass instrumented (-Xarch_host -fsanitize=address,undefined) together with
tests/cabi/arg_errors.cpp, which drives every entry point through its argument checks, the workspace queries, the
thread-local tuning state and the exchange plan of the sharded transform.  No GPU is touched: every checked path
returns before the first HIP call."""

import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "quantum-systems_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_argument_paths_under_asan_and_ubsan(tmp_path):
    import __graft_entry__ as ge

    exe = tmp_path / "arg_errors"
    srcs = [os.path.join(CSRC, s) for s in ge.SOURCES] + [os.path.join(ROOT, "tests", "cabi", "arg_errors.cpp")]
    flags = ["-O1", "-g", "-std=c++17", "--offload-arch=gfx950", "-DQS_S4_ONLY=14", "-DQS_DEV_FEW_SHAPES",
             "-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-omit-frame-pointer",
             "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]

    # one object per source, a few at a time (the kernels' device passes are most of the build; in one hipcc call
    # the sources compile one after the other: 5 minutes)
    def compile_one(src):
        obj = tmp_path / (os.path.basename(src) + ".o")
        res = subprocess.run([HIPCC, *flags, "-c", src, "-o", str(obj)], capture_output=True, text=True, timeout=900)
        assert res.returncode == 0, res.stderr[-3000:]
        return str(obj)

    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=4) as pool:
        objs = list(pool.map(compile_one, srcs))
    host_link = ["-Xarch_host", "-fsanitize=address,undefined"]       # (the host pass only, as in the compile steps)
    res = subprocess.run([HIPCC, "--offload-arch=gfx950", *objs, *host_link, "-ldl", "-o", str(exe)],
                         capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-3000:]
    sym = subprocess.run(["nm", str(exe)], capture_output=True, text=True).stdout
    assert "__asan_init" in sym and "__ubsan_handle" in sym            # the host code really is instrumented
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    assert "all argument checks ok" in run.stdout
    assert "runtime error" not in run.stderr and "AddressSanitizer" not in run.stderr
    shutil.rmtree(tmp_path, ignore_errors=True)
