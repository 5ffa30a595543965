"""The `ctucopy` executable's G.711 decoders against the reference's own expander, all 256 codes of both laws.

src/io/amulaw.h is FFTW-free: oracle/Makefile compiles it where it lies into oracle/_ref/libref_amulaw.so.  The host
decoder (ctucopy_amd/host/g711.h) is compiled stand-alone here; no GPU involved.
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from oracle.oracle import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libref_amulaw.so")


def test_g711_tables_identical_to_reference(tmp_path):
    if not os.path.exists(REF_SO) and os.path.exists("/root/reference/src/io/amulaw.h"):
        build(force=True)
    if not os.path.exists(REF_SO):
        pytest.skip("oracle/_ref/libref_amulaw.so not built and /root/reference absent")
    src = tmp_path / "t.cc"
    src.write_text('#include "g711.h"\nextern "C" void host_table(int alaw, short *o) '
                   '{ for (int v = 0; v < 256; v++) o[v] = g711_to_linear((uint8_t)v, alaw != 0); }\n')
    so = tmp_path / "t.so"
    subprocess.run(["g++", "-O2", "-shared", "-fPIC", "-I", os.path.join(ROOT, "ctucopy_amd", "host"), str(src), "-o", str(so)], check=True)
    host, ref = ctypes.CDLL(str(so)), ctypes.CDLL(REF_SO)
    for mode in (0, 1):  # 0 = mu-law, 1 = A-law (src/io/amulaw.h:21)
        a, b = np.zeros(256, dtype=np.int16), np.zeros(256, dtype=np.int16)
        host.host_table(mode, a.ctypes.data_as(ctypes.c_void_p))
        ref.ref_amulaw_table(mode, b.ctypes.data_as(ctypes.c_void_p))
        assert np.array_equal(a, b), mode
        assert len(set(b.tolist())) > 200
