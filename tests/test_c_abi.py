"""The header is plain C and the library is usable without Python/torch: a C99 client is compiled with gcc against
include/tinyntt.h and linked to tiny_ntt_amd/lib/libtinyntt.so."""
import os
import subprocess

import pytest

from conftest import ROOT, have_gpu

SRC = os.path.join(ROOT, "tests", "c_abi", "c_abi_client.c")
LIBDIR = os.path.join(ROOT, "tiny_ntt_amd", "lib")


def build(tmp_path):
    exe = str(tmp_path / "c_abi_client")
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), SRC,
           "-L", LIBDIR, "-ltinyntt", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


@pytest.mark.skipif(have_gpu(), reason="CPU-side check of the no-device path")
def test_header_is_c99_and_client_reports_no_device(tmp_path):
    exe = build(tmp_path)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stdout, r.stdout


@pytest.mark.gpu
def test_c_client_runs_on_gpu(tmp_path):
    exe = build(tmp_path)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "c abi ok" in r.stdout, r.stdout
