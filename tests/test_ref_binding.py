"""INTEGRATION.md section 1 as a test: tests/ref_binding/gpu_float.h is the reference-side binding -- a
PolarCode::Decoding::Decoder subclass (reference: include/polarcode/decoding/decoder.h:40-180) whose decode() calls the C ABI
of libpolar_amd.so.  oracle/Makefile compiles it against the reference's own headers and links it with the reference
library built from /root/reference (oracle/_ref, test infrastructure) and with the product library.

CPU test: it compiles, links, constructs through the reference's base-class interface and its getters agree with the
reference's own decoder from Decoding::create(); both setSignal overloads land in the subclass's containers (GpuFloat: a
FloatContainer, GpuChar for type "char": a CharContainer).  GPU test (-m gpu): the reference's own loop --
Decoder::decode_vector(const float*, void*) (decoder.cpp:154-167), one codeword per call -- runs through the subclass
on the fixtures' LLRs and returns the bytes and verdicts the reference's CPU decoder produced."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "ref_binding_test")
HAVE_REF = os.path.isdir("/root/reference/src/polarcode")


def _binary():
    if HAVE_REF:  # (the build container: make sure it is current)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "polar-codes_amd")], stdout=subprocess.DEVNULL)
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL)
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/ref_binding_test was not built (needs /root/reference at build time)")
    return BIN


def test_reference_side_subclass_builds_and_agrees_with_create():
    r = subprocess.run([_binary(), "cpu"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "cpu ok" in r.stdout


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cfg3_n1024_k512_l8_crc24_awgn.bin", "cfg2_n1024_k512_sc_awgn.bin", "cfg4_n1024_k176_l8_crc11_awgn.bin",
                                  "nsys_cfg3_n1024_k512_l8_crc24.bin",
                                  # the non-virtual decode_vector(const char*) / setSignal(const char*) (decoder.h:161, decoder.cpp:140-142,
                                  # 169-181): int8 rows into the float decoders (GpuFloat's FloatContainer) and the int8 decoders (GpuChar)
                                  "i8_cfg3_n1024_k512_l8_crc24.bin", "i8_cfg2_n1024_k512_sc.bin", "char_cfg3_n1024_k512_l8_crc24.bin",
                                  "char_cfg2_n1024_k512_sc.bin"])
def test_reference_decode_vector_loop_runs_on_the_gpu(name):
    r = subprocess.run([_binary(), "gpu", os.path.join(ROOT, "tests", "golden", name)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "gpu ok" in r.stdout
