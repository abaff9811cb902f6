"""include/granne.hpp -- the C++ mirror of the reference's Rust API. Compiles everywhere (g++,
host only); the program itself is the reference's index tests and runs on the GPU box."""
import os
import subprocess

import pytest

from granne_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "test_granne_api.cpp")


def compile_program(out):
    build.build_library()
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), SRC, "-L", build.LIB_DIR,
           "-lgranne_hip", "-Wl,-rpath," + build.LIB_DIR, "-Wl,--allow-shlib-undefined", "-o", out]
    subprocess.check_call(cmd)


def test_cpp_mirror_compiles(tmp_path):
    compile_program(str(tmp_path / "test_granne_api"))


@pytest.mark.gpu
def test_reference_index_tests_through_the_cpp_mirror(tmp_path):
    exe = str(tmp_path / "test_granne_api")
    compile_program(exe)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=600, env=env)
    print(out.stdout, out.stderr)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok")
