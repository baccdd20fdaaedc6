"""The container model behind the tie replay (vcfdist_amd/csrc/pr_tie.hip): libstdc++'s unordered_set iteration order as a
closed form, and the bucket counts it grows through, checked against the real std::unordered_set of this image with the
reference's hash (dist.h:42-50).  CPU only: tests/tie_order_model.cpp is compiled with g++ into a temporary directory."""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def test_order_model_and_bucket_sequence(tmp_path):
    exe = str(tmp_path / "tie_order_model")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(HERE, "tie_order_model.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True)
    lines = out.stdout.strip().splitlines()
    assert out.returncode == 0 and lines[-1] == "bad = 0", out.stdout[-400:]
    measured = [int(x) for x in lines[0].split()]
    src = open(os.path.join(HERE, "..", "vcfdist_amd", "csrc", "pr_tie.hip")).read()
    table = re.search(r"#define TIE_BUCKET_LIST \{(.*?)\}", src, re.S).group(1)
    table = [int(x) for x in re.findall(r"(\d+)u", table)]
    assert len(table) == int(re.search(r"#define TIE_N_BUCKETS (\d+)", src).group(1))
    assert table[:len(measured)] == measured and len(measured) >= 18      # 13, 29, 59, ... as this libstdc++ grows


def test_running_libstdcxx_matches_the_tie_model_table():
    """the same check the library makes at vpr_create (vpr_selfcheck_tie_model), on the libstdc++ this process runs with: CPU"""
    import ctypes as C
    from vcfdist_amd import api
    L = api.lib()
    L.vpr_selfcheck_tie_model.restype = C.c_int
    L.vpr_selfcheck_tie_model.argtypes = [C.c_int]
    assert L.vpr_selfcheck_tie_model(18) == 18


import pytest


@pytest.mark.gpu
def test_tie_model_on_the_gpu_box(tmp_path):
    """the container model against the GPU box's own libstdc++ and g++ (the CPU test above only sees the build container's)"""
    test_order_model_and_bucket_sequence(tmp_path)
    test_running_libstdcxx_matches_the_tie_model_table()
