#!/usr/bin/env python3
"""Host check of the expf the device evaluates for the sigmoid activation (differt2d_amd/csrc/d2d_kernels.hpp: expf_libm): the
algorithm of glibc's expf (sysdeps/ieee754/flt-32/e_expf.c) restated in NumPy float64 -- one rounding per operation, no fused
multiply-add, as the device code is compiled -- against this host's libm, bit for bit; and the table of 2^(i/32) recomputed
from scratch (decimal arithmetic, correctly rounded) against the constants compiled into the kernels."""
import ctypes
import re
import struct
import sys
from decimal import Decimal, getcontext

import numpy as np

N = 32
getcontext().prec = 60
T = []
for i in range(N):
    bits = struct.unpack("<Q", struct.pack("<d", float(Decimal(2) ** (Decimal(i) / Decimal(N)))))[0]
    T.append((bits - (i << 47)) & 0xFFFFFFFFFFFFFFFF)
src = open(__file__.replace("scripts/check_expf_model.py", "differt2d_amd/csrc/d2d_kernels.hpp")).read()
tab = [int(v, 16) for v in re.findall(r"0x([0-9a-f]{16})ull", src[src.index("EXPF_TAB[32]"):src.index("expf_libm(float x)")])]
assert tab == T, "the table in d2d_kernels.hpp is not 2^(i/32) - (i << 47)"
libm = ctypes.CDLL("libm.so.6")
libm.expf.restype = ctypes.c_float
libm.expf.argtypes = [ctypes.c_float]
InvLn2N, SHIFT = float.fromhex("0x1.71547652b82fep+0") * N, float.fromhex("0x1.8p+52")
C = [float.fromhex("0x1.c6af84b912394p-5") / N / N / N, float.fromhex("0x1.ebfce50fac4f3p-3") / N / N, float.fromhex("0x1.62e42ff0c52d6p-1") / N]


def model(x):
    z = InvLn2N * x.astype(np.float64)
    kd = z + SHIFT
    ki = kd.view(np.uint64)
    r = z - (kd - SHIFT)
    s = (np.array(T, np.uint64)[(ki % N).astype(np.int64)] + (ki << np.uint64(47))).view(np.float64)
    with np.errstate(over="ignore"):
        y = (((C[0] * r + C[1]) * (r * r) + (C[2] * r + 1)) * s).astype(np.float32)
    y = np.where(x < np.float32(float.fromhex("-0x1.9d1d9ep6")), np.float32(2.0**-149), y)
    y = np.where(x > np.float32(float.fromhex("0x1.62e42ep6")), np.float32(np.inf), y)
    return np.where(x < np.float32(float.fromhex("-0x1.9fe368p6")), np.float32(0), y)


rng = np.random.default_rng(0)
bad = tot = 0
for lo, hi in ((-104.5, 89.5), (-20, 20), (-1, 1), (-104, -86)):
    x = (rng.random(200_000) * (hi - lo) + lo).astype(np.float32)
    ref = np.array([libm.expf(float(v)) for v in x], np.float32)
    nb = int((ref.view(np.uint32) != model(x).view(np.uint32)).sum())
    print(f"[{lo}, {hi}]: {nb} of {x.size} arguments differ from libm's expf")
    bad, tot = bad + nb, tot + x.size
print("table ok;", bad, "of", tot, "differ")
sys.exit(1 if bad else 0)
