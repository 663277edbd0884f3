"""The header is plain C and the library is usable without Python/torch: a C99 client is compiled with gcc against
include/tinyntt.h and linked to tiny_ntt_amd/lib/libtinyntt.so; a C++17 client does the same through include/tinyntt.hpp,
the mirror of the reference's C++ benchmark functions."""
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


CPP_SRC = os.path.join(ROOT, "tests", "c_abi", "cpp_mirror_client.cpp")


def build_cpp(tmp_path):
    exe = str(tmp_path / "cpp_mirror_client")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), CPP_SRC,
           "-L", LIBDIR, "-ltinyntt", "-Wl,-rpath," + LIBDIR, "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


@pytest.mark.skipif(have_gpu(), reason="CPU-side check of the no-device path")
def test_cpp_mirror_compiles_and_reports_no_device(tmp_path):
    r = subprocess.run([build_cpp(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stdout, r.stdout


@pytest.mark.gpu
def test_cpp_mirror_reproduces_the_reference_binaries_checksums(tmp_path):
    """include/tinyntt.hpp: negacyclic_mul_ntt / forward_ntt_bench / ntt<Inverse> / negacyclic_mul_reference / make_poly / checksum
    with the reference's names (benchmark_ntt_60bit.cpp:79-188); both benchmark parameter sets print the reference's checksums."""
    r = subprocess.run([build_cpp(tmp_path)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "cpp mirror ok" in r.stdout, r.stdout
