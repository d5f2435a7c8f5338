"""The C++ mirror of the halo2 interfaces (include/zkhip.hpp): it compiles standalone, its parity test binary links and
loads without a GPU, and on a GPU the binary's checks against the C oracle all pass."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "mirror_parity")


@pytest.fixture(scope="module")
def mirror_bin():
    """tests/cpp/mirror_parity is built together with the libraries; g++ rebuilds it in seconds where it did not travel."""
    if not os.path.exists(BIN):
        import __graft_entry__
        __graft_entry__.build_cpp_mirror_test()
    return BIN


def test_header_compiles_standalone():
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++",
                           os.path.join(ROOT, "include", "zkhip.hpp")])


def test_mirror_binary_links_and_loads(mirror_bin):
    out = subprocess.run([mirror_bin, "--list"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "evaluate_h" in out.stdout and "gfx950" in out.stdout


@pytest.mark.gpu
def test_cpp_mirror_parity_against_the_oracle(mirror_bin):
    out = subprocess.run([mirror_bin], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.splitlines()
    assert lines[-1].startswith("all ok") and not any(l.startswith("FAIL") for l in lines)
    assert sum(l.startswith("ok") for l in lines) >= 32
