"""The DEFLATE decoder of the file reader (2fast2q_amd/csrc/f2q_inflate.h) against zlib, under ASan + UBSan.
Host only; the checker is tests/emu/inflate_fuzz.cpp."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "emu", "inflate_fuzz.cpp")
BIN = os.path.join(HERE, "emu", "inflate_fuzz.bin")
HDR = os.path.join(os.path.dirname(HERE), "2fast2q_amd", "csrc", "f2q_inflate.h")


@pytest.fixture(scope="module")
def fuzz_bin():
    if not os.path.exists(BIN) or max(os.path.getmtime(SRC), os.path.getmtime(HDR)) > os.path.getmtime(BIN):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                               "-o", BIN, SRC, "-lz"])
    return BIN


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_inflater_agrees_with_zlib(fuzz_bin, seed):
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    res = subprocess.run([fuzz_bin, "250", str(seed)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=500)
    assert res.returncode == 0, res.stdout[-2000:]
    assert res.stdout.startswith("ok 250,"), res.stdout


PSRC = os.path.join(HERE, "emu", "pargz_fuzz.cpp")
PBIN = os.path.join(HERE, "emu", "pargz_fuzz.bin")
PHDR = os.path.join(os.path.dirname(HERE), "2fast2q_amd", "csrc", "f2q_pargz.h")


@pytest.fixture(scope="module")
def pargz_bin():
    if not os.path.exists(PBIN) or max(os.path.getmtime(PSRC), os.path.getmtime(PHDR), os.path.getmtime(HDR)) > os.path.getmtime(PBIN):
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                               "-o", PBIN, PSRC, "-lz", "-lpthread"])
    return PBIN


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_parallel_gunzip_agrees_with_zlib(pargz_bin, seed):
    """f2q_pargz.h (block search, decoding with an unknown window, chained resolution, 1..5 threads, chunks of 700 B ... 1 MiB)
    against zlib on intact and damaged streams, under ASan + UBSan"""
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    res = subprocess.run([pargz_bin, "40", str(seed)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:]
    assert res.stdout.startswith("ok 40,"), res.stdout
