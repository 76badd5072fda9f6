"""The 5 x 5 median network of the depth-fill kernel (lmono_amd/csrc/median_net.hpp) is exact: all 2^25 binary inputs
(0-1 principle for networks of min / max / med3) plus random byte windows, on the host build of the same header."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_median_network_is_exact(tmp_path):
    exe = str(tmp_path / "median_net_check")
    subprocess.check_call(["g++", "-O2", "-fopenmp", os.path.join(ROOT, "tests", "median_net_check.cpp"), "-o", exe])
    out = subprocess.check_output([exe], text=True)
    assert out.strip() == "BINARY_FAILURES 0 RANDOM_FAILURES 0"
