"""The C++ mirror of the halo2 interfaces (include/zkhip.hpp): it compiles standalone, its parity test binary links and
loads without a GPU, and on a GPU the binary's checks against the C oracle all pass."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "cpp", "mirror_parity")


def test_header_compiles_standalone():
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Wextra", "-Werror", "-fsyntax-only", "-x", "c++",
                           os.path.join(ROOT, "include", "zkhip.hpp")])


def test_mirror_binary_links_and_loads():
    assert os.path.exists(BIN), "run __graft_entry__.build() first"
    out = subprocess.run([BIN, "--list"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and "evaluate_h" in out.stdout and "gfx950" in out.stdout


@pytest.mark.gpu
def test_cpp_mirror_parity_against_the_oracle():
    out = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    print(out.stdout, out.stderr)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.splitlines()
    assert lines[-1].startswith("all ok") and sum(l.startswith("ok") for l in lines) >= 21 and not any(l.startswith("FAIL") for l in lines)
