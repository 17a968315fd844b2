"""CPU tier for the DEVICE code: bp_pp_amd/csrc/{field,point,merlin,verify_core}.h are `__host__ __device__`, so the exact
functions the HIP kernels call are compiled here with g++ (tests/emul) and run thread by thread against the oracle.
This catches logic errors before any GPU minute is spent; the -m gpu tests then check the same code as compiled for
gfx950.  The emulation library is never loaded by the product."""
import ctypes as C
import json
import os
import random

import numpy as np
import pytest

import bppp_oracle as O
from emul.build import load

GOLD = os.path.join(os.path.dirname(__file__), "golden")
b32 = lambda v: int(v).to_bytes(32, "big")


@pytest.fixture(scope="module")
def L():
    return load()


def test_field_arithmetic(L):
    rnd = random.Random(7)
    edge = [0, 1, 2, O.P - 1, O.P - 2, 2**256 - 1 - O.P, 1 << 255, 2**32 + 977, O.N - 1, O.N, 2**128, 0xFFFFFFFF, 1 << 32,
            2**256 - 1, O.P + 1, (1 << 224) - 1]
    vals = edge + [rnd.getrandbits(256) for _ in range(400)]
    o1, o2, o3 = (C.create_string_buffer(32) for _ in range(3))
    for i, v in enumerate(vals):
        a, b = v % O.P, vals[(i * 7 + 3) % len(vals)] % O.P
        L.emul_fe_mul(b32(a), b32(b), o1)
        assert int.from_bytes(o1.raw, "big") == a * b % O.P
        L.emul_fe_addsub(b32(a), b32(b), o1, o2, o3)
        assert int.from_bytes(o1.raw, "big") == (a + b) % O.P
        assert int.from_bytes(o2.raw, "big") == (a - b) % O.P
        assert int.from_bytes(o3.raw, "big") == a * 21 % O.P
        a, b = v % O.N, vals[(i * 7 + 3) % len(vals)] % O.N
        L.emul_sc_mul(b32(a), b32(b), o1)
        assert int.from_bytes(o1.raw, "big") == a * b % O.N
        L.emul_sc_addsub(b32(a), b32(b), o1, o2)
        assert int.from_bytes(o1.raw, "big") == (a + b) % O.N and int.from_bytes(o2.raw, "big") == (a - b) % O.N
    for v in vals[:48]:
        a = v % O.P
        L.emul_fe_inv(b32(a), o1, o2)
        assert int.from_bytes(o1.raw, "big") == (pow(a, -1, O.P) if a else 0)
        assert int.from_bytes(o2.raw, "big") == pow(a, (O.P + 1) // 4, O.P)
        a = v % O.N
        L.emul_sc_inv(b32(a), o1)
        assert int.from_bytes(o1.raw, "big") == (pow(a, -1, O.N) if a else 0)
    assert L.emul_fe_canonical(b32(O.P - 1)) == 1 and L.emul_fe_canonical(b32(O.P)) == 0 and L.emul_fe_canonical(b32(2**256 - 1)) == 0
    assert L.emul_sc_canonical(b32(O.N - 1)) == 1 and L.emul_sc_canonical(b32(O.N)) == 0


def test_complete_point_formulas_and_straus(L):
    rnd = random.Random(11)
    pts = [None, O.G, O.pt_mul(O.G, 2), O.pt_neg(O.G)] + [O.pt_mul(O.G, rnd.getrandbits(256)) for _ in range(5)]
    out = C.create_string_buffer(64)
    for A in pts:
        for B in pts:       # includes identity operands, P + P and P + (-P): what complete formulas are for
            assert L.emul_pt_op(0, O.pt_to_xy64(A), O.pt_to_xy64(B), out) == 0
            assert out.raw == O.pt_to_xy64(O.pt_add(A, B))
            L.emul_pt_op(1, O.pt_to_xy64(A), O.pt_to_xy64(B), out)
            assert out.raw == O.pt_to_xy64(O.pt_add(A, B))
        L.emul_pt_op(2, O.pt_to_xy64(A), O.pt_to_xy64(A), out)
        assert out.raw == O.pt_to_xy64(O.pt_add(A, A))
    bad = bytearray(O.pt_to_xy64(O.G)); bad[63] ^= 1
    assert L.emul_pt_op(0, bytes(bad), O.pt_to_xy64(O.G), out) == -1
    for m in (1, 2, 5):
        for _ in range(3):
            P = [pts[rnd.randrange(len(pts))] for _ in range(m)]
            ks = [rnd.choice([0, 1, O.N - 1, 8, 2**255, 0x8888888888888888, rnd.getrandbits(256) % O.N]) for _ in range(m)]
            assert L.emul_straus(m, b"".join(map(O.pt_to_xy64, P)), b"".join(map(b32, ks)), out) == 0
            exp = None
            for p, k in zip(P, ks):
                exp = O.pt_add(exp, O.pt_mul(p, k))
            assert out.raw == O.pt_to_xy64(exp)


def test_glv_split_and_straus(L):
    """k = k1 + k2*lambda (mod n) with both halves below 2^128 (the lattice bound is (|a1| + |a2|) / 2 < 2^127.4; the signed 5-bit
    recoding of the u64 verifier needs |k_i| + OFF5 < 2^130, i.e. |k_i| < 1.93 * 2^128), then the full GLV shared-doubling MSM."""
    rnd = random.Random(13)
    OFF = int("8" * 33, 16)
    k1p, k2p = (C.c_uint32 * 5)(), (C.c_uint32 * 5)()
    n1, n2 = C.c_int(), C.c_int()
    # the basis of the decomposition lattice: the halves are largest near the corners of its fundamental cell
    a1, b1 = 0x3086D221A7D46BCDE86C90E49284EB15, -0xE4437ED6010E88286F547FA90ABFE4C3
    a2, b2 = 0x114CA50F7A8E2F3F657C1108D9D44CFD8, 0x3086D221A7D46BCDE86C90E49284EB15
    corners = [((u * a1 + v * a2) // 2 + (u * b1 + v * b2) // 2 * O.LAMBDA + d) % O.N for u in (-1, 1) for v in (-1, 1) for d in range(-3, 4)]
    structured = [(i * O.N) // 64 + j for i in range(64) for j in (-1, 0, 1)] + [(1 << i) % O.N for i in range(0, 256, 7)]
    worst = 0
    for k in [0, 1, 2, O.N - 1, O.N // 2, O.LAMBDA, O.N - O.LAMBDA, 2**128, 2**255] + corners + [x % O.N for x in structured] + \
            [rnd.getrandbits(256) % O.N for _ in range(1500)]:
        L.emul_glv_split(b32(k), k1p, k2p, C.byref(n1), C.byref(n2))
        a = sum(int(k1p[i]) << (32 * i) for i in range(5)) - OFF
        b = sum(int(k2p[i]) << (32 * i) for i in range(5)) - OFF
        assert 0 <= a < 2**128 and 0 <= b < 2**128
        worst = max(worst, a, b)
        a, b = (-a if n1.value else a), (-b if n2.value else b)
        assert (a + b * O.LAMBDA) % O.N == k
    assert 2**126 < worst < 2**128                 # the corners do get close to the bound, and stay under it
    pts = [None, O.G, O.pt_neg(O.G)] + [O.pt_mul(O.G, rnd.getrandbits(256)) for _ in range(5)]
    out = C.create_string_buffer(64)
    for m in (1, 2, 5):
        for _ in range(4):
            P = [pts[rnd.randrange(len(pts))] for _ in range(m)]
            ks = [rnd.choice([0, 1, O.N - 1, O.LAMBDA, 2**255, rnd.getrandbits(256) % O.N, rnd.getrandbits(256) % O.N]) for _ in range(m)]
            assert L.emul_straus_glv(m, b"".join(map(O.pt_to_xy64, P)), b"".join(map(b32, ks)), out) == 0
            exp = None
            for p, k in zip(P, ks):
                exp = O.pt_add(exp, O.pt_mul(p, k))
            assert out.raw == O.pt_to_xy64(exp)


def test_division_step_inversion(L):
    """fe_inv / sc_inv (Bernstein-Yang division steps, modinv.h) against big-integer inverses and the exponentiation forms."""
    rnd = random.Random(41)
    a, b = C.create_string_buffer(32), C.create_string_buffer(32)
    for which, m in ((0, O.P), (1, O.N)):
        vals = [0, 1, 2, m - 1, m - 2, (m + 1) // 2, 2**255 % m, 2**30, 2**30 - 1, 2**60 + 1, 0x3FFFFFFF << 30, (1 << 256) % m]
        vals += [rnd.getrandbits(256) % m for _ in range(40)] + [rnd.getrandbits(k) for k in (8, 31, 64, 129, 200)]
        for v in vals:
            assert L.emul_inv(which, b32(v), a, b) == 0
            exp = pow(v, -1, m) if v else 0
            assert int.from_bytes(a.raw, "big") == exp, (which, hex(v))
            assert b.raw == a.raw
    assert L.emul_inv(0, b32(O.P), a, b) == -1          # non-canonical encodings are still refused upstream


def test_affine_tables_and_jacobian_straus(L):
    """The u64 verifier's variable-base path: per-proof affine tables from one batched inversion (identity points included),
    Jacobian shared-doubling sum with deferred exception detection, complete-formula fallback."""
    rnd = random.Random(29)
    pts = [None, O.G, O.pt_neg(O.G)] + [O.pt_mul(O.G, rnd.getrandbits(256)) for _ in range(6)]
    out, fb = C.create_string_buffer(64), C.c_int(0)
    special = [0, 1, O.N - 1, O.LAMBDA, 2**255, 8, O.N - 8, int("8" * 64, 16) % O.N]
    n_fast = 0
    for m in (1, 2, 5):
        for _ in range(8):
            P = [pts[rnd.randrange(len(pts))] for _ in range(m)]
            ks = [rnd.choice(special + [rnd.getrandbits(256) % O.N] * 8) for _ in range(m)]
            assert L.emul_straus_affine(m, b"".join(map(O.pt_to_xy64, P)), b"".join(map(b32, ks)), out, C.byref(fb)) == 0
            exp = None
            for p, k in zip(P, ks):
                exp = O.pt_add(exp, O.pt_mul(p, k))
            assert out.raw == O.pt_to_xy64(exp)
            n_fast += 1 - fb.value
    assert n_fast >= 12
    # distinct random points and scalars: the fast law alone must do (no fallback)
    for m in (2, 5):
        P = [O.pt_mul(O.G, rnd.getrandbits(256)) for _ in range(m)]
        ks = [rnd.getrandbits(256) % O.N for _ in range(m)]
        assert L.emul_straus_affine(m, b"".join(map(O.pt_to_xy64, P)), b"".join(map(b32, ks)), out, C.byref(fb)) == 0
        assert fb.value == 0
    # exceptional additions: the same point twice with the same scalar (second stream adds an entry to itself); P and -P
    # (sum passes through the identity); both must be flagged and still come out right
    A = O.pt_mul(O.G, 0xABCDEF)
    for P, ks in (([A, A], [5, 5]), ([A, A], [rnd.getrandbits(120)] * 2), ([A, O.pt_neg(A)], [77, 77]),
                  ([A, O.pt_neg(A), A, A, A], [9, 9, 0, 0, 0])):
        assert L.emul_straus_affine(len(P), b"".join(map(O.pt_to_xy64, P)), b"".join(map(b32, ks)), out, C.byref(fb)) == 0
        exp = None
        for p, k in zip(P, ks):
            exp = O.pt_add(exp, O.pt_mul(p, k))
        assert out.raw == O.pt_to_xy64(exp)
        assert fb.value == 1


@pytest.mark.parametrize("parts", [2, 4])
def test_per_lane_tables_and_half_stream_sums(L, parts):
    """The small-call path of the u64 verifier (straus_core.h: affine_table_one, straus_split_lane): a lane builds ONE window table
    -- of P, or of 2^65 P (two parts per stream) / 2^35 P, 2^70 P, 2^100 P (four parts) for the later parts of a stream -- and a
    lane walks ONE part of a GLV stream; the lanes' shares added up must be sum k_j P_j, and every table entry must be the oracle's
    multiple.  Identity points, special scalars, exceptional additions."""
    rnd = random.Random(31 + parts)
    starts = (0, 13) if parts == 2 else (0, 7, 14, 20)
    pts = [None, O.G, O.pt_neg(O.G)] + [O.pt_mul(O.G, rnd.getrandbits(256)) for _ in range(6)]
    out, fb = C.create_string_buffer(64), C.c_int(0)
    tabs = C.create_string_buffer(4 * 13 * 16 * 64)
    special = [0, 1, O.N - 1, O.LAMBDA, 2**255, 16, O.N - 16, 2**35, 2**35 - 1, 2**65, 2**65 - 1, 2**70, 2**100 - 1, (1 << 130) - 1,
               int("8" * 64, 16) % O.N]
    n_fast = 0
    for m in (1, 2, 5):
        for it in range(8):
            P = [pts[rnd.randrange(len(pts))] for _ in range(m)]
            ks = [rnd.choice(special + [rnd.getrandbits(256) % O.N] * 8) for _ in range(m)]
            assert L.emul_straus_split(m, parts, b"".join(map(O.pt_to_xy64, P)), b"".join(map(b32, ks)), out, C.byref(fb),
                                       tabs if it == 0 else None) == 0
            exp = None
            for p, k in zip(P, ks):
                exp = O.pt_add(exp, O.pt_mul(p, k))
            assert out.raw == O.pt_to_xy64(exp)
            n_fast += 1 - fb.value
            if it == 0:
                for h in range(parts):
                    for j in range(13):
                        base = O.pt_mul(P[j], 2**(5 * starts[h])) if j < m else None
                        for e in range(16):
                            got = tabs.raw[64 * ((13 * h + j) * 16 + e):][:64]
                            assert got == O.pt_to_xy64(O.pt_mul(base, e + 1)), (m, h, j, e)
    assert n_fast >= 12
    for m in (2, 5):
        P = [O.pt_mul(O.G, rnd.getrandbits(256)) for _ in range(m)]
        ks = [rnd.getrandbits(256) % O.N for _ in range(m)]
        assert L.emul_straus_split(m, parts, b"".join(map(O.pt_to_xy64, P)), b"".join(map(b32, ks)), out, C.byref(fb), None) == 0
        assert fb.value == 0
    # an exceptional addition inside one lane's part of a stream (a table entry added to its own value) is flagged, the result still right
    A = O.pt_mul(O.G, 0xABCDEF)
    for P, ks in (([A, O.pt_neg(A)], [77, 77]), ([A], [33 * 32 + 1])):
        assert L.emul_straus_split(len(P), parts, b"".join(map(O.pt_to_xy64, P)), b"".join(map(b32, ks)), out, C.byref(fb), None) == 0
        exp = None
        for p, k in zip(P, ks):
            exp = O.pt_add(exp, O.pt_mul(p, k))
        assert out.raw == O.pt_to_xy64(exp)


def test_merlin_known_answer_on_device_code(L):
    kat = C.create_string_buffer(32)
    L.emul_merlin_kat(b"test protocol", 13, b"some data", 9, kat, 32)
    assert kat.raw.hex() == "d5a21972d0d5fe320c0d263fac7fffb8145aa640af6e9bca177c03c7efcf0615"


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(GOLD, "u64_golden.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("W", [4, 8, 16, 10, 18, 19, 20, 22])
def test_fixed_base_window_digits_recompose_the_scalar(L, W):
    """k == sum_w d_w 2^(W w) for the unsigned (4/8/16) and signed (10/18/19/20/22) window schemes, incl. the extreme scalars."""
    rnd = random.Random(W)
    idx, skip, neg = C.c_uint64(), C.c_int(), C.c_int()
    edge = [sum(1 << (W * i + W - 1) for i in range(256 // W)) % O.N, sum(((1 << (W - 1)) - 1) << (W * i) for i in range(256 // W)) % O.N]
    for k in [0, 1, O.N - 1, 2**255, 2**256 % O.N, int("8" * 64, 16) % O.N, int("7" * 64, 16)] + edge + [rnd.getrandbits(256) % O.N for _ in range(20)]:
        nwin = L.emul_fb_digit(W, b32(k), 0, C.byref(idx), C.byref(skip), C.byref(neg))
        total = 0
        for w in range(nwin):
            L.emul_fb_digit(W, b32(k), w, C.byref(idx), C.byref(skip), C.byref(neg))
            d = 0 if skip.value else (idx.value + 1)
            assert d <= (2**(W - 1) if W in (10, 18, 19, 20, 22) else 2**W - 1)
            total += (-d if neg.value else d) << (W * w)
        assert total == k


# window codes with two widths (fb_core.h: fb_wb): code = Wb + 100 ka -- ka windows of Wb + 1 bits, then windows of Wb bits
MIXED_CODES = [609, 310, 208, 523, 618, 1119]


def _shape(L, code, w=0, bits=0):
    out = (C.c_uint64 * 6)()
    L.emul_fb_shape(code, w, bits, out)
    return dict(nwin=out[0], pos=out[1], per_win=out[2], off=out[3], per_base=out[4], windows_for=out[5])


@pytest.mark.parametrize("code", MIXED_CODES)
def test_two_width_window_geometry_and_digits(L, code):
    """Tables whose windows are sized to the bit (round 5): the windows tile at least 258 bits without gaps, the entries of a generator
    are laid out window after window, a scalar below 2^bits reaches exactly the windows up to bit `bits`, and the signed digits
    recompose every scalar -- the extreme ones included (all digits at either end of their range, carries into the top window)."""
    Wb, ka = code % 100, code // 100
    nwin = _shape(L, code)["nwin"]
    assert nwin == -(-(258 - ka) // Wb) and Wb * nwin + ka >= 258
    pos = off = 0
    for w in range(nwin):
        sh = _shape(L, code, w)
        width = Wb + (1 if w < ka else 0)
        assert sh["pos"] == pos and sh["off"] == off and sh["per_win"] == 1 << (width - 1)
        pos += width
        off += sh["per_win"]
    assert _shape(L, code)["per_base"] == off
    if code == 523:
        assert nwin == 11 and off == 5 * 2**23 + 6 * 2**22      # 4.3 GB per generator
    for bits in (1, 4, 8, 63, 64, 65, 127, 128, 200, 255):
        need = next(n for n in range(1, nwin + 1) if _shape(L, code, n)["pos"] >= bits + 1 or n == nwin)
        assert _shape(L, code, 0, bits)["windows_for"] == need, bits
    rnd = random.Random(code)
    idx, skip, neg = C.c_uint64(), C.c_int(), C.c_int()
    tops = [_shape(L, code, w + 1)["pos"] - 1 for w in range(nwin)]
    edge = [sum(1 << t for t in tops if t < 256) % O.N, (sum(1 << t for t in tops if t < 255) - 1) % O.N, O.N - 2**234, O.N - 1 - 2**233]
    for k in [0, 1, O.N - 1, 2**255, 2**256 % O.N, int("8" * 64, 16) % O.N, int("7" * 64, 16)] + edge + [rnd.getrandbits(256) % O.N for _ in range(40)]:
        total = 0
        for w in range(nwin):
            assert L.emul_fb_digit(code, b32(k), w, C.byref(idx), C.byref(skip), C.byref(neg)) == nwin
            d = 0 if skip.value else (idx.value + 1)
            assert d <= _shape(L, code, w)["per_win"]
            total += (-d if neg.value else d) << _shape(L, code, w)["pos"]
        assert total == k


@pytest.mark.parametrize("W", [4, 8, 10, 609, 310])
def test_fixed_base_tables(L, gold, W):
    gens = bytes.fromhex(gold["generators"])
    g3 = gens[:128] + bytes(64)                  # two generators + the identity as a (degenerate but legal) generator
    ent = L.emul_fb_table_entries(3, W)
    tab = np.zeros(ent * 64, dtype=np.uint8)
    assert L.emul_fb_build(g3, 3, W, tab.ctypes.data) == 0
    out = C.create_string_buffer(64)
    for ks in ([0x1234567890ABCDEF1234567890ABCDEF, O.N - 1, 12345], [0, 0, 0], [1, 0, 5], [2**255 + 2**9, O.N - 2, 1 << 19],
               [int("7" * 64, 16) % O.N, int("8" * 63, 16), 511]):
        L.emul_fb_msm(tab.ctypes.data, W, 0, 3, b"".join(map(b32, ks)), out)
        exp = O.pt_add(O.pt_mul(O.pt_from_xy64(gens[:64]), ks[0]), O.pt_mul(O.pt_from_xy64(gens[64:128]), ks[1]))
        assert out.raw == O.pt_to_xy64(exp)


@pytest.mark.parametrize("W", [4, 8, 10, 609])
def test_fixed_base_fast_accumulator_and_its_fallback(L, gold, W):
    """The device's fixed-base lane sums use an incomplete (XYZZ) accumulator with deferred exception detection.  Independent
    generators never trip it.  A REPEATED generator with equal window digits makes a lane add a table entry to itself: the sums that
    start from an EMPTY accumulator (a wavefront per sum, nl = 64) must detect that and re-do it with the complete formulas; the sums
    that start from the offset point (nl = 1, 8) do not even see an exceptional addition there.  Every form gives the right sum."""
    gens = bytes.fromhex(gold["generators"])
    G0, G1 = O.pt_from_xy64(gens[:64]), O.pt_from_xy64(gens[64:128])
    out, fb = C.create_string_buffer(64), C.c_int(0)
    rng = np.random.default_rng(5)
    # (a) independent generators: fast path, no fallback
    g3 = gens[:192]
    G2 = O.pt_from_xy64(gens[128:192])
    ent = L.emul_fb_table_entries(3, W)
    tab = np.zeros(ent * 64, dtype=np.uint8)
    assert L.emul_fb_build(g3, 3, W, tab.ctypes.data) == 0
    for nl in (1, 8, 64):
        for _ in range(4):
            ks = [int.from_bytes(rng.bytes(32), "big") % O.N for _ in range(3)]
            assert L.emul_fb_msm_lanes_nl(tab.ctypes.data, W, 0, 3, b"".join(map(b32, ks)), out, C.byref(fb), nl) == 0
            exp = O.pt_add(O.pt_add(O.pt_mul(G0, ks[0]), O.pt_mul(G1, ks[1])), O.pt_mul(G2, ks[2]))
            assert out.raw == O.pt_to_xy64(exp) and fb.value == 0
        for ks in ([0, 0, 0], [1, 0, 0], [0, O.N - 1, 0], [5, 5, 5], [O.N - 1, O.N - 1, O.N - 1]):
            assert L.emul_fb_msm_lanes_nl(tab.ctypes.data, W, 0, 3, b"".join(map(b32, ks)), out, C.byref(fb), nl) == 0
            exp = O.pt_add(O.pt_add(O.pt_mul(G0, ks[0]), O.pt_mul(G1, ks[1])), O.pt_mul(G2, ks[2]))
            assert out.raw == O.pt_to_xy64(exp) and fb.value == 0
    # (b) generator list G0, G1, G2, G3, G0: a scalar with a single non-zero window on both copies of G0 makes the lane that holds both
    # add a table entry to itself (the doubling case of the incomplete law)
    gd = gens[:256] + gens[:64]
    G3 = O.pt_from_xy64(gens[192:256])
    ent = L.emul_fb_table_entries(5, W)
    tab = np.zeros(ent * 64, dtype=np.uint8)
    assert L.emul_fb_build(gd, 5, W, tab.ctypes.data) == 0
    Wn = W % 100
    cases = [[7, 0, 0, 0, 7], [3 << (2 * Wn + 2), 0, 0, 0, 3 << (2 * Wn + 2)], [5 << (8 * Wn + 6), 0, 0, 0, 5 << (8 * Wn + 6)],
             [7, 1, 2, 3, 7], [11, 0, 0, 0, 12], [int("5" * 64, 16) % O.N] * 5]
    for nl in (1, 8, 64):
        hits = 0
        for ks in cases:
            assert L.emul_fb_msm_lanes_nl(tab.ctypes.data, W, 0, 5, b"".join(map(b32, ks)), out, C.byref(fb), nl) == 0
            exp = O.pt_mul(G0, (ks[0] + ks[4]) % O.N)
            for Gi, k in ((G1, ks[1]), (G2, ks[2]), (G3, ks[3])):
                exp = O.pt_add(exp, O.pt_mul(Gi, k))
            assert out.raw == O.pt_to_xy64(exp)
            hits += fb.value
        if nl == 64 and W < 100 and (4 * (256 // W)) % 64 == 0:       # copies 0 and 4 land on one lane when 4 * windows is a multiple of 64 (W = 4, 8)
            assert hits >= 3          # the self-additions were detected (and re-done), not silently mis-added
        if nl < 64:
            assert hits == 0          # from the offset point the same sums are ordinary additions


@pytest.mark.parametrize("W_hi,W_lo", [(10, 8), (10, 4), (8, 10), (609, 10), (10, 310)])
def test_fixed_base_sums_over_a_table_in_two_regions(L, gold, W_hi, W_lo):
    """FbTable's second region (round 5: the 17 generators both fixed-base sums of a u64 verify run over get 24-bit windows, the other 32
    keep 22): the first hi_bases generators in one table at one width, the rest in another table at another width, counted from 0.
    Runs inside either region and across the boundary, on 1, 8 and 64 lanes, fast form == complete form == term-by-term == the oracle."""
    gens = bytes.fromhex(gold["generators"])
    NB, HI = 7, 3
    pts = [O.pt_from_xy64(gens[64 * i:64 * i + 64]) for i in range(NB)]
    thi = np.zeros(L.emul_fb_table_entries(HI, W_hi) * 64, dtype=np.uint8)
    tlo = np.zeros(L.emul_fb_table_entries(NB - HI, W_lo) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens[:64 * HI], HI, W_hi, thi.ctypes.data) == 0
    assert L.emul_fb_build(gens[64 * HI:64 * NB], NB - HI, W_lo, tlo.ctypes.data) == 0
    rng = np.random.default_rng(W_hi * 100 + W_lo)
    out = C.create_string_buffer(64)
    for first, count in ((0, NB), (0, HI), (HI, NB - HI), (1, 4), (2, 1), (HI - 1, 2), (4, 3)):
        for nl in (1, 8, 64):
            ks = [int.from_bytes(rng.bytes(32), "big") % O.N for _ in range(count)]
            if nl == 8:
                ks[0] = O.N - 1
                ks[-1] = 0 if count > 1 else 1
            assert L.emul_fb_msm_mixed(thi.ctypes.data, W_hi, HI, tlo.ctypes.data, W_lo, first, count, b"".join(map(b32, ks)), out, nl) == 0
            exp = O.pt_mul(pts[first], ks[0])
            for j in range(1, count):
                exp = O.pt_add(exp, O.pt_mul(pts[first + j], ks[j]))
            assert out.raw == O.pt_to_xy64(exp), (first, count, nl)


@pytest.mark.parametrize("G", [2, 4, 8, 16])
def test_shared_inversion_of_a_batch(L, G):
    """fe_batch_inv_lane (k_verify_shared_inv<G>): 1 / v for every element of a batch from one inversion per G elements -- equal to the
    inversion of each element alone, zeros (the identity's Z) staying zero without spoiling their group, for batch sizes that leave
    the last groups short or empty, in place and out of place."""
    p = 2**256 - 2**32 - 977
    rng = np.random.default_rng(G)
    for n in (1, 2, G - 1, G, G + 1, 3 * G + 2, 64, 65, 257):
        vals = [int.from_bytes(rng.bytes(32), "big") % p for _ in range(n)]
        for z in {0, n // 2, n - 1}:
            vals[z] = 0                                   # zeros at the ends and in the middle
        if n > 5:
            vals[3], vals[4] = 1, p - 1
        src = b"".join(v.to_bytes(32, "big") for v in vals)
        for in_place in (0, 1):
            out = np.zeros(32 * n, np.uint8)
            assert L.emul_fe_batch_inv(G, n, src, out.ctypes.data, in_place) == 0
            for t, v in enumerate(vals):
                got = int.from_bytes(out[32 * t:32 * t + 32].tobytes(), "big")
                assert got == (pow(v, p - 2, p) if v else 0), (n, t)


# shared: proofs per shared field inversion (plan_core.h: shared_inv -- the table build in five passes and the rounds take 1 / v from
# fe_batch_inv_lane instead of inverting per proof; the emulator also checks the tables bit for bit against the one-pass build)
@pytest.mark.parametrize("shared", [0, 2, 4, 8, 16])
@pytest.mark.parametrize("W", [4, 10, 609])       # 10 = the signed-window scheme (26 windows: pairs dealt round-robin over the lanes); 609 = windows of two widths (6 x 10 + 22 x 9 bits)
def test_full_verify_pipeline_against_golden(L, gold, oracle_c, W, shared):
    L.emul_set_shared_inv(shared)
    try:
        _full_verify_pipeline_against_golden(L, gold, oracle_c, W)
    finally:
        L.emul_set_shared_inv(0)


def _full_verify_pipeline_against_golden(L, gold, oracle_c, W):
    gens, label = bytes.fromhex(gold["generators"]), bytes.fromhex(gold["label"])
    ent = L.emul_fb_table_entries(49, W)
    tab = np.zeros(ent * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, 49, W, tab.ctypes.data) == 0
    items = [(c["commitment"], c["proof"], 1, 0) for c in gold["cases"]]
    items += [(c["commitment"], c["proof"], 0, c["status"]) for c in gold["negative_cases"]]
    n = len(items)
    V = np.frombuffer(b"".join(bytes.fromhex(i[0]) for i in items), dtype=np.uint8).reshape(n, 64).copy()
    P = np.frombuffer(b"".join(bytes.fromhex(i[1]) for i in items), dtype=np.uint8).reshape(n, 928).copy()
    acc, st, tr = np.zeros(n, np.uint8), np.zeros(n, np.int32), np.zeros((n, 704), np.uint8)
    assert 0 == L.emul_u64_verify_batch(tab.ctypes.data, W, label, len(label), n, V.ctypes.data, P.ctypes.data, acc.ctypes.data,
                            st.ctypes.data, tr.ctypes.data)
    assert acc.tolist() == [i[2] for i in items]
    assert st.tolist() == [i[3] for i in items]
    for k, c in enumerate(gold["cases"]):         # every challenge and every hashed commitment, byte for byte
        exp = bytes.fromhex(c["trace_challenges_and_points"])
        assert bytes(tr[k][:len(exp)]) == exp
    for k in range(n):                            # and the C oracle's full trace (incl. C4) wherever it decodes
        if items[k][3] == 0:
            rc, otr = oracle_c.u64_verify(gens, label, bytes(V[k]), bytes(P[k]), trace=True)
            assert rc == items[k][2] and bytes(tr[k]) == otr


@pytest.mark.parametrize("W", [4, 609])           # 609: windows of two widths -- the prover's short scalars (digits, multiplicities, u64 values) stop at the windows they reach
def test_full_prove_pipeline_is_byte_identical_to_the_oracle(L, gold, oracle_c, W):
    """The device prover code (prove_core.h) on CPU: proofs must equal the reference-shaped prover's bytes for the same
    (x, s, 52 random scalars), on the golden cases and on fresh seeded cases incl. x = 0 and x = 2^64 - 1."""
    import workload
    gens, label = bytes.fromhex(gold["generators"]), bytes.fromhex(gold["label"])
    ent = L.emul_fb_table_entries(49, W)
    tab = np.zeros(ent * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, 49, W, tab.ctypes.data) == 0
    cases = gold["cases"]
    n = len(cases) + 3
    x = np.array([c["x"] for c in cases] + [int(v) for v in workload.values(3, first=77)], dtype=np.uint64)
    s = np.concatenate([np.frombuffer(b"".join(bytes.fromhex(c["s"]) for c in cases), dtype=np.uint8).reshape(-1, 32),
                        workload.blindings(3, first=77)])
    rnd = np.concatenate([np.frombuffer(b"".join(bytes.fromhex(c["rnd"]) for c in cases), dtype=np.uint8).reshape(-1, 52 * 32),
                          workload.prover_randomness(3, first=77)])
    x, s, rnd = np.ascontiguousarray(x), np.ascontiguousarray(s), np.ascontiguousarray(rnd)
    proofs, V, st = np.zeros((n, 928), np.uint8), np.zeros((n, 64), np.uint8), np.zeros(n, np.int32)
    L.emul_u64_prove_batch(tab.ctypes.data, W, label, len(label), n, x.ctypes.data, s.ctypes.data, rnd.ctypes.data,
                           proofs.ctypes.data, V.ctypes.data, st.ctypes.data)
    assert not st.any()
    for i, c in enumerate(cases):
        assert bytes(V[i]).hex() == c["commitment"]
        assert bytes(proofs[i]).hex() == c["proof"], f"case {i}"
    op, ov = oracle_c.u64_prove_batch(gens, label, x, s, rnd, nthreads=2)
    assert (ov == V).all() and (op == proofs).all()


def test_secret_scalar_sums_in_the_constant_address_form(L, gold, oracle_c):
    """ "ct_prover" (fb_core.h: fb_lookup_add_ct): the prover's sums over the witness and its blindings read EVERY entry of every
    4-bit window and select by mask.  Same points as the digit-addressed gathers -- window by window, including the zero digit and
    an accumulator that equals the table entry (the doubling case of the complete law) -- and the same proof bytes as the oracle
    prover's, on the golden cases and at the edges of the inputs (x = 0, 2^64 - 1; zero and n - 1 blindings and draws)."""
    import workload
    gens, label = bytes.fromhex(gold["generators"]), bytes.fromhex(gold["label"])
    W = 4
    tab = np.zeros(L.emul_fb_table_entries(49, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, 49, W, tab.ctypes.data) == 0
    # one table addition, both forms: every digit of one window (0 .. 15), a few windows and bases, two accumulators
    G = gens[:64]
    for base, w in ((0, 0), (17, 5), (48, 63), (3, 31)):
        for d in range(16):
            k = (d << (4 * w)).to_bytes(32, "big")
            entry = oracle_c.point_mul(gens[64 * base:64 * base + 64], k) if d else None
            for acc in (G, entry) if entry else (G,):
                fast, ct = np.zeros(64, np.uint8), np.zeros(64, np.uint8)
                assert L.emul_fb_lookup_both(tab.ctypes.data, base, w, k, acc, fast.ctypes.data, ct.ctypes.data) == 0
                want = oracle_c.point_add(acc, entry) if entry else acc
                assert bytes(ct) == bytes(fast) == want, (base, w, d)
    ex, es, ernd = workload.edge_prover_inputs()
    cases = gold["cases"]
    x = np.array([c["x"] for c in cases] + [int(v) for v in ex], dtype=np.uint64)
    s = np.concatenate([np.frombuffer(b"".join(bytes.fromhex(c["s"]) for c in cases), dtype=np.uint8).reshape(-1, 32), es])
    rnd = np.concatenate([np.frombuffer(b"".join(bytes.fromhex(c["rnd"]) for c in cases), dtype=np.uint8).reshape(-1, 52 * 32), ernd])
    x, s, rnd = np.ascontiguousarray(x), np.ascontiguousarray(s), np.ascontiguousarray(rnd)
    n = len(x)
    proofs, V, st = np.zeros((n, 928), np.uint8), np.zeros((n, 64), np.uint8), np.zeros(n, np.int32)
    L.emul_set_prove_ct(1)
    try:
        assert 0 == L.emul_u64_prove_batch(tab.ctypes.data, W, label, len(label), n, x.ctypes.data, s.ctypes.data, rnd.ctypes.data,
                                           proofs.ctypes.data, V.ctypes.data, st.ctypes.data)
    finally:
        L.emul_set_prove_ct(0)
    assert not st.any()
    for i, c in enumerate(cases):
        assert bytes(V[i]).hex() == c["commitment"] and bytes(proofs[i]).hex() == c["proof"], f"case {i}"
    op, ov = oracle_c.u64_prove_batch(gens, label, x, s, rnd, nthreads=2)
    assert (ov == V).all() and (op == proofs).all()


def test_prover_next_commitments_as_fixed_base_sums(L, gold, oracle_c):
    """The small-call form of the u64 prover (ProveWs::next_by_msm): each WNLA level's commitment as wnla.commit(l_, n_) over the
    original generators (wnla.rs:186, :66-72: one more fixed-base sum) instead of com + y X + (y^2 - 1) R -- the same points, so
    the same proof bytes as the oracle prover's."""
    import workload
    gens, label = bytes.fromhex(gold["generators"]), bytes.fromhex(gold["label"])
    W = 4
    tab = np.zeros(L.emul_fb_table_entries(49, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, 49, W, tab.ctypes.data) == 0
    n = 5
    x = np.ascontiguousarray(workload.values(n, first=0))               # includes 0, 2^64 - 1, 123456
    s, rnd = np.ascontiguousarray(workload.blindings(n, first=0)), np.ascontiguousarray(workload.prover_randomness(n, first=0))
    proofs, V, st = np.zeros((n, 928), np.uint8), np.zeros((n, 64), np.uint8), np.zeros(n, np.int32)
    L.emul_set_prove_next_by_msm(1)
    try:
        assert 0 == L.emul_u64_prove_batch(tab.ctypes.data, W, label, len(label), n, x.ctypes.data, s.ctypes.data, rnd.ctypes.data,
                                           proofs.ctypes.data, V.ctypes.data, st.ctypes.data)
    finally:
        L.emul_set_prove_next_by_msm(0)
    assert not st.any()
    op, ov = oracle_c.u64_prove_batch(gens, label, x, s, rnd, nthreads=2)
    assert (ov == V).all() and (op == proofs).all()


@pytest.mark.parametrize("shared", [0, 16, 4])
def test_identity_and_repeated_points_in_a_proof(L, gold, oracle_c, shared):
    """(shared: with the inversions shared by that many proofs -- an identity's Z = 0 inside a group must neither poison the group's
    product nor come back as anything but 0.)"""
    L.emul_set_shared_inv(shared)
    try:
        _identity_and_repeated_points_in_a_proof(L, gold, oracle_c)
    finally:
        L.emul_set_shared_inv(0)


def _identity_and_repeated_points_in_a_proof(L, gold, oracle_c):
    """CPU twin of tests/test_gpu_verify.py::test_identity_points_in_proofs_vs_oracle on the device code: each of the 14 points of
    a golden proof replaced by the identity (64 zero bytes: well-formed, hashed as 33 zero bytes, the neutral element of every
    table and sum), all of them at once, and repeated points (P + P inside the window tables): accept, status and the full
    trace equal the C oracle's."""
    gens, label = bytes.fromhex(gold["generators"]), bytes.fromhex(gold["label"])
    W = 4
    tab = np.zeros(L.emul_fb_table_entries(49, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, 49, W, tab.ctypes.data) == 0
    V0 = np.frombuffer(bytes.fromhex(gold["cases"][1]["commitment"]), np.uint8)
    P0 = np.frombuffer(bytes.fromhex(gold["cases"][1]["proof"]), np.uint8)
    rows = []
    for j in range(14):
        V, P = V0.copy(), P0.copy()
        if j == 13:
            V[:] = 0
        else:
            P[64 * j:64 * j + 64] = 0
        rows.append((V, P))
    V, P = V0.copy(), P0.copy()
    V[:] = 0; P[:832] = 0
    rows.append((V, P))
    for a, b in ((8, 4), (0, 1), (12, 3)):
        V, P = V0.copy(), P0.copy()
        P[64 * a:64 * a + 64] = P[64 * b:64 * b + 64]
        rows.append((V, P))
        V, P = V0.copy(), P0.copy()                           # ... and the negated copy: P + (-P) inside the sums
        P[64 * a:64 * a + 32] = P[64 * b:64 * b + 32]
        y = int.from_bytes(P[64 * b + 32:64 * b + 64].tobytes(), "big")
        P[64 * a + 32:64 * a + 64] = np.frombuffer(((2**256 - 2**32 - 977) - y).to_bytes(32, "big"), np.uint8)
        rows.append((V, P))
    rows.append((V0.copy(), P0.copy()))
    n = len(rows)
    V, P = np.stack([r[0] for r in rows]), np.stack([r[1] for r in rows])
    acc, st, tr = np.zeros(n, np.uint8), np.zeros(n, np.int32), np.zeros((n, 704), np.uint8)
    assert 0 == L.emul_u64_verify_batch(tab.ctypes.data, W, label, len(label), n, V.ctypes.data, P.ctypes.data, acc.ctypes.data, st.ctypes.data, tr.ctypes.data)
    assert acc.tolist() == [0] * (n - 1) + [1] and not st.any()
    for k in range(n):
        rc, otr = oracle_c.u64_verify(gens, label, bytes(V[k]), bytes(P[k]), trace=True)
        assert rc == int(acc[k]) and bytes(tr[k]) == otr, k


def test_prover_at_the_edges_of_its_inputs(L, gold, oracle_c):
    """x = 0 and 2^64 - 1, blinding 0 and n - 1, prover draws all zero / all n - 1 / all one (workload.edge_prover_inputs): the
    device code's proofs equal the reference-shaped oracle prover's byte for byte and both verifiers accept them."""
    import workload
    gens, label = bytes.fromhex(gold["generators"]), bytes.fromhex(gold["label"])
    W = 4
    tab = np.zeros(L.emul_fb_table_entries(49, W) * 64, dtype=np.uint8)
    assert L.emul_fb_build(gens, 49, W, tab.ctypes.data) == 0
    x, s, rnd = workload.edge_prover_inputs()
    n = len(x)
    proofs, V, st = np.zeros((n, 928), np.uint8), np.zeros((n, 64), np.uint8), np.zeros(n, np.int32)
    L.emul_u64_prove_batch(tab.ctypes.data, W, label, len(label), n, x.ctypes.data, s.ctypes.data, rnd.ctypes.data, proofs.ctypes.data,
                           V.ctypes.data, st.ctypes.data)
    op, ov = oracle_c.u64_prove_batch(gens, label, x, s, rnd, nthreads=2)
    assert not st.any() and (ov == V).all() and (op == proofs).all()
    oacc, ost = oracle_c.u64_verify_batch(gens, label, V, proofs, nthreads=2)
    acc, vst, tr = np.zeros(n, np.uint8), np.zeros(n, np.int32), np.zeros((n, 704), np.uint8)
    assert 0 == L.emul_u64_verify_batch(tab.ctypes.data, W, label, len(label), n, V.ctypes.data, proofs.ctypes.data, acc.ctypes.data, vst.ctypes.data,
                            tr.ctypes.data)
    assert oacc.all() and acc.all() and not ost.any() and not vst.any()
