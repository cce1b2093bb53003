"""The C ABI driven by a plain C++ host program (tests/cabi/transform_demo.cpp): no Python and
no torch between the caller and libqs_amd.so -- the shape a non-Python host of the reference's
hot path would have.  Compiled here with hipcc against the in-tree library and run as a child
process."""

import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# (55, 55): BASELINE.json configs[1], the two fused passes of the balanced 4-wide kernel; (58, 57): ceil(l/4) = 15 on
# its instantiation for 16)
@pytest.mark.parametrize("L,M", [(14, 11), (9, 16), (32, 32), (55, 55), (58, 57)])
def test_native_host_program(tmp_path, L, M):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    libdir = os.path.join(ROOT, "quantum-systems_amd")
    assert os.path.exists(os.path.join(libdir, "libqs_amd.so")), "build the library first (__graft_entry__.build)"
    exe = str(tmp_path / "transform_demo")
    subprocess.run(
        [hipcc, "-O2", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"),
         os.path.join(ROOT, "tests", "cabi", "transform_demo.cpp"), "-L", libdir, "-l:libqs_amd.so",
         f"-Wl,-rpath,{libdir}", "-o", exe],
        check=True, capture_output=True, timeout=300,
    )
    # (the sharded-rows call of the demo loads RCCL at run time: ROCm's library directory on the loader path)
    env = dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    res = subprocess.run([exe, str(L), str(M)], capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0 and "CABI_DEMO_OK" in res.stdout, res.stdout + res.stderr
    assert "mixed_rel_err" in res.stdout and "sharded_rows=1" in res.stdout, res.stdout      # both round-3 entry points ran
