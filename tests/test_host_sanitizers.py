"""CPU sanitizer runs (never on the GPU: the pool has no GPU ASan).

* the C oracle's sanitizer build (`make -C oracle asan`: -fsanitize=address,undefined) sweeps small and degenerate inputs;
* the host-only logic of libd2d.so -- differt2d_amd/csrc/d2d_host.hpp: candidate enumeration, parameter validation, the
  LDS / heavy-list size arithmetic with its 4 GiB guard -- compiled with g++ -fsanitize=address,undefined
  (tests/native/d2d_host_san.cpp) and driven through ctypes with edge cases.

Both run in a child process with libasan preloaded (an instrumented library cannot be loaded into a plain python
otherwise); any sanitizer report makes the child exit non-zero.
"""

import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _libasan():
    out = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return out if os.path.isabs(out) and os.path.exists(out) else None


def _run_child(code, preload):
    env = dict(os.environ, LD_PRELOAD=preload, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", OMP_NUM_THREADS="2", PYTHONPATH=ROOT)
    return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], capture_output=True, text=True, env=env, timeout=600)


@pytest.fixture(scope="module")
def asan():
    lib = _libasan()
    if lib is None:
        pytest.skip("gcc's libasan.so is not available")
    return lib


def test_oracle_under_asan_and_ubsan(asan):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "asan"])
    code = f"""
        import ctypes as C, sys
        import numpy as np
        sys.path.insert(0, {ROOT!r})
        from oracle import c_oracle as CO
        CO._LIB_PATH = {os.path.join(ROOT, "oracle", "libd2d_oracle_asan.so")!r}
        CO._lib = None
        rng = np.random.default_rng(0)
        F = np.float32
        cases = 0
        for n in (0, 1, 2, 5, 9):
            walls = rng.random((n, 2, 2), dtype=F)
            if n >= 2:
                walls[1] = walls[0]                      # two identical walls
                walls[0, 1] = walls[0, 0]                # a zero-length wall
            tx = rng.random(2, dtype=F)
            for shape in ((1, 1), (3, 5), (8, 8)):
                X, Y = rng.random(shape, dtype=F), rng.random(shape, dtype=F)
                X[0, 0], Y[0, 0] = tx                    # a cell on the transmitter
                for kw in (dict(approx=False), dict(approx=True), dict(approx=True, function="sigmoid", alpha=10.0)):
                    for lo, hi in ((0, 0), (0, 2), (3, 3) if n <= 5 else (1, 2)):
                        for role in ("rx", "tx"):
                            for prune in (False, True):
                                allowed = None if n < 3 else (np.arange(n) % 3 != 1).astype(np.uint8)
                                out, cnt = CO.power_and_count_maps(walls, tx, X, Y, allowed=allowed, min_order=lo, max_order=hi,
                                                                   prune=prune, grid_role=role, patch=0.01, **kw)
                                assert out.shape == shape and cnt.shape == shape
                                cases += 1
            if n:
                v, f, idx = CO.eval_candidates(walls, tx, tx + F(0.1), min_order=0, max_order=2)
                assert len(v) == CO.num_candidates(n, 0, 2)
        print("ORACLE-SAN-OK", cases)
    """
    out = _run_child(code, asan)
    assert out.returncode == 0 and "ORACLE-SAN-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-6000:]


def test_libd2d_host_logic_under_asan_and_ubsan(asan, tmp_path):
    so = str(tmp_path / "libd2d_host_san.so")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-shared", "-fPIC", "-Wall", "-Wextra", "-o", so, os.path.join(ROOT, "tests", "native", "d2d_host_san.cpp")])
    code = f"""
        import ctypes as C, itertools, sys
        import numpy as np
        sys.path.insert(0, {ROOT!r})
        from differt2d_amd._lib import Params
        L = C.CDLL({so!r})
        i64 = C.c_int64
        L.san_count.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(i64)]
        L.san_enumerate.argtypes = [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, i64]
        L.san_check_params.argtypes = [C.POINTER(Params), C.c_char_p, C.c_int]
        L.san_integer_pow.restype = C.c_float
        L.san_integer_pow.argtypes = [C.c_float, C.c_int]
        L.san_lds.argtypes = [C.c_int, C.c_int, C.c_int] + [C.POINTER(C.c_uint64)] * 3
        L.san_heavy_plan.argtypes = [C.c_longlong] * 4 + [C.POINTER(C.c_longlong)]

        def python_enum(n, allowed, lo, hi):
            ok = [j for j in range(n) if allowed is None or allowed[j]]
            out = []
            for k in range(lo, hi + 1):
                out += [t for t in itertools.product(ok, repeat=k) if all(a != b for a, b in zip(t, t[1:]))]
            return out

        # ---- enumeration: exact capacity (one element less must be refused, not overrun), filters, every order
        for n in (0, 1, 2, 3, 7):
            for allowed in (None, np.array([j % 2 == 0 for j in range(n)], np.uint8), np.zeros(n, np.uint8)):
                ap = None if allowed is None else allowed.ctypes.data_as(C.c_void_p)
                for lo, hi in ((0, 0), (0, 1), (0, 2), (2, 2), (0, 4), (3, 4), (2, 1)):
                    want = python_enum(n, allowed, lo, hi)
                    cnt = i64(-1)
                    assert L.san_count(n, ap, lo, hi, C.byref(cnt)) == 0 and cnt.value == len(want), (n, lo, hi, cnt.value, len(want))
                    cand = np.full((max(len(want), 1), 4), 99, np.int32)   # EXACTLY count rows: an overrun is an ASan report
                    order = np.full(max(len(want), 1), 99, np.int32)
                    rc = L.san_enumerate(n, ap, lo, hi, cand.ctypes.data_as(C.c_void_p), order.ctypes.data_as(C.c_void_p), len(want))
                    assert rc == 0
                    got = [tuple(int(v) for v in cand[i, : order[i]]) for i in range(len(want))]
                    assert got == want
                    assert all((cand[i, order[i]:] == -1).all() for i in range(len(want)))
                    if len(want) > 0:
                        assert L.san_enumerate(n, ap, lo, hi, cand.ctypes.data_as(C.c_void_p), order.ctypes.data_as(C.c_void_p), len(want) - 1) == -1
                    assert L.san_enumerate(n, ap, lo, hi, None, None, len(want)) == 0      # both outputs optional
        cnt = i64(0)
        assert L.san_enumerate(3, None, 0, 5, None, None, 10**9) == -1                   # order above D2D_MAX_ORDER
        assert L.san_enumerate(3, None, -1, 2, None, None, 10**9) == -1
        assert L.san_count(-1, None, 0, 1, C.byref(cnt)) == -1 and L.san_count(3, None, 0, 1, None) == -1
        assert L.san_count(2**31 - 1, None, 0, 2, C.byref(cnt)) == 0 and cnt.value == 1 + (2**31 - 1) + (2**31 - 1) * (2**31 - 2)
        assert L.san_count(2**31 - 1, None, 0, 4, C.byref(cnt)) == -1          # 2^124 candidates: refused, no signed overflow
        # ---- parameter validation
        def params(**kw):
            p = Params(min_order=0, max_order=2, approx=0, act=0, alpha=100.0, tol=1e-2, patch=0.0, seg_tol=0.005, fun_id=0,
                       r_coef=0.5, height=0.1, solver=0, steps=100, out_mode=0, grid_role=0, strict_nan=0, many=1)
            for k, v in kw.items():
                setattr(p, k, v)
            return p
        msg = C.create_string_buffer(8)   # tiny buffer: the message must be truncated, not overrun
        assert L.san_check_params(C.byref(params()), msg, 8) == 0
        for bad, rc in ((dict(min_order=-1), -1), (dict(max_order=5), -1), (dict(approx=1, alpha=0.0), -1), (dict(approx=1, alpha=float("nan")), -1),
                        (dict(approx=1, act=7), -4), (dict(fun_id=9), -4), (dict(fun_id=-1), -4), (dict(out_mode=2), -1), (dict(grid_role=-3), -1),
                        (dict(seg_tol=-0.1), -1), (dict(seg_tol=float("nan")), -1)):
            assert L.san_check_params(C.byref(params(**bad)), msg, 8) == rc, bad
            assert len(msg.value) <= 7
        assert L.san_check_params(None, msg, 8) == -1
        assert L.san_check_params(C.byref(params(max_order=5)), None, 0) == -1
        # ---- lax.integer_pow
        for x in (0.5, -1.25, 0.0, 3.0):
            ref = np.float32(1.0)
            for n in range(0, 9):
                got = L.san_integer_pow(x, n)
                assert got == np.float32(x) ** n or abs(got - float(np.float32(x)) ** n) <= 1e-6 * abs(got), (x, n, got)
        # ---- content hash of a grid (d2d_set_grid): every length incl. odd tails read exactly (ASan), any single bit matters
        L.san_hash_floats.restype = C.c_uint64
        L.san_hash_floats.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64]
        rng = np.random.default_rng(0)
        seen = set()
        for n in (0, 1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, 1000, 4099):
            a = rng.random(n, dtype=np.float32)   # EXACTLY n floats: reading past the end is an ASan report
            h = L.san_hash_floats(a.ctypes.data_as(C.c_void_p), n, 5)
            assert h == L.san_hash_floats(a.copy().ctypes.data_as(C.c_void_p), n, 5) and h not in seen
            seen.add(h)
            assert n == 0 or L.san_hash_floats(a.ctypes.data_as(C.c_void_p), n, 6) != h
            for i in ([0, n // 2, n - 1] if n else []):
                b = a.copy()
                b.view(np.uint32)[i] ^= np.uint32(1 << int(rng.integers(0, 32)))
                assert L.san_hash_floats(b.ctypes.data_as(C.c_void_p), n, 5) != h, (n, i)
            if n >= 2:
                b = a.copy(); b[0], b[1] = a[1], a[0]
                assert (a[0] == a[1]) or L.san_hash_floats(b.ctypes.data_as(C.c_void_p), n, 5) != h
        z, nz = np.zeros(4, np.float32), np.array([0.0, -0.0, 0.0, 0.0], np.float32)
        assert L.san_hash_floats(z.ctypes.data_as(C.c_void_p), 4, 0) != L.san_hash_floats(nz.ctypes.data_as(C.c_void_p), 4, 0)  # bit patterns
        # ---- LDS sizes: monotone, aligned, inside each other
        prev = 0
        for n in (0, 1, 50, 200, 1023, 1024, 5000):
            t, b, tot = C.c_uint64(), C.c_uint64(), C.c_uint64()
            L.san_lds(n, 4, 16, C.byref(t), C.byref(b), C.byref(tot))
            assert t.value == (4 * n + 1) * 16 + 512 and t.value > prev
            assert b.value % 16 == 0 and b.value >= t.value - 512 + 3 * 16 * 64 * 4 and tot.value == b.value + 4 * 512
            prev = t.value
        # ---- heavy-list plan: sizes consistent; the 4 GiB guard switches the cut off instead of overflowing
        out = (C.c_longlong * 4)()
        for tiles, Nc, hs, parts in ((16384, 50, 64, 4), (100, 50, 64, 4), (15, 50, 64, 4), (16384, 1, 64, 4), (16384, 50, 0, 4),
                                      (2**31 - 1, 256, 2**40, 4), (2**31 - 1, 2**20, 2**40, 4), (2**62, 2**31, 2**62, 4),
                                      (16384, 50, 64, 0), (-5, 50, 64, 4)):
            L.san_heavy_plan(tiles, Nc, hs, parts, out)
            H, cap, lf, ci = out[0], out[1], out[2], out[3]
            if H == 0:
                assert (cap, lf, ci) == (0, 0, 0)
                continue
            assert H == min(hs, tiles // 4, ((4 << 30) // 256) // (parts * cap)) and cap >= (Nc // parts + 1) * (Nc - 1) + Nc + 1   # covers a part's candidates
            assert lf == H * parts * cap * 64 and ci == H * parts * 65 and lf * 4 <= 4 << 30
        L.san_heavy_plan(16384, 50, 64, 4, out)
        assert out[0] == 64 and out[1] == -(-50 * 49 // 3) + 52       # the benchmark's own launch (parts by rank: a third each at most)
        L.san_heavy_plan(2**31 - 1, 256, 2**40, 4, out)
        assert 0 < out[0] == ((4 << 30) // 256) // (4 * out[1]) and out[2] * 4 <= 4 << 30   # as many as fit below 4 GiB
        # ---- region-list plan: every list owns a chunk, the levels nest, absurd sizes switch the lists off
        L.san_region_plan.argtypes = [C.c_int, C.c_int, C.c_longlong] + [C.c_int] * 5 + [C.c_longlong, C.c_int, C.POINTER(C.c_longlong)]
        L.san_region_plan.restype = None
        o = (C.c_longlong * 12)()
        for tx_, ty_, Nc, lo, hi, Rl, Rt, S, budget in ((128, 128, 50, 0, 2, 4, 16, 0, 512 << 20), (256, 256, 200, 0, 3, 4, 16, 0, 512 << 20),
                                                      (1, 1, 2, 2, 2, 4, 16, 0, 1 << 20), (128, 128, 50, 3, 4, 4, 6, 7, 512 << 20),
                                                      (128, 128, 1, 0, 2, 4, 16, 0, 512 << 20), (128, 128, 50, 0, 1, 4, 16, 0, 512 << 20),
                                                      (2**31 - 1, 2**31 - 1, 50, 0, 2, 1, 1, 0, 2**62), (128, 128, 50, 0, 2, 4, 16, 0, 1024),
                                                      (128, 128, 2**40, 0, 4, 4, 16, 0, 2**40), (0, 5, 50, 0, 2, 4, 16, 0, 1 << 30),
                                                      (128, 128, 50, 0, 2, 0, 16, 0, 1 << 30), (128, 128, 50, 0, 2, 4, 16, 5000, 1 << 30)):
            L.san_region_plan(tx_, ty_, Nc, lo, hi, Rl, Rt, S, budget, 128, o)
            if not o[0]:
                continue
            tR, tS, tr, ts, lR, lS, lr, ls, nst, mc, klo = [o[i] for i in range(1, 12)]
            assert lR == Rl and lS == 1 and tR % lR == 0 and tR >= lR and 1 <= tS <= 1024
            assert lr == -(-tx_ // lR) * -(-ty_ // lR) == ls and tr == -(-tx_ // tR) * -(-ty_ // tR) and ts == tr * tS
            assert klo == max(2, lo) <= hi and nst == (ls + ts) * (hi - klo + 1) and 2 * nst <= mc <= 2**31 - 1 and mc * 128 * 8 <= budget
        L.san_region_plan(128, 128, 50, 0, 2, 4, 16, 0, 512 << 20, 128, o)
        assert list(o) == [1, 16, 13, 64, 832, 4, 1, 1024, 1024, 1856, 524288, 2]   # the benchmark's own launch
        L.san_region_plan(128, 128, 50, 0, 1, 4, 16, 0, 512 << 20, 128, o)
        assert o[0] == 0                                                            # orders <= 1: no lists
        L.san_region_plan(2**31 - 1, 2**31 - 1, 50, 0, 2, 1, 1, 0, 2**62, 128, o)
        assert o[0] == 0                                                            # 2^62 regions: refused
        print("HOST-SAN-OK")
    """
    out = _run_child(code, asan)
    assert out.returncode == 0 and "HOST-SAN-OK" in out.stdout, out.stdout[-2000:] + out.stderr[-6000:]
