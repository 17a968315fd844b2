"""CPU oracle (Python big-int) for the Bulletproofs++ u64 range-proof hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (bp_pp_amd/, the C-ABI
library) may import or call this file; only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg use the oracle, and only as the checker.

This is a line-by-line restatement of the reference crate distributed-lab/bp-pp
0.1.1 (citations are `file:line` under /root/reference/src):

  util.rs          -> reduce, vector_*, weight_vector_mul, e, pow, diag_inv, ...
  transcript.rs    -> app_point, get_challenge
  wnla.rs          -> WeightNormLinearArgument.{commit, verify, prove}
  circuit.rs       -> ArithmeticCircuit.{commit, verify, prove} (+ collect_*)
  range_proof/reciprocal.rs -> ReciprocalRangeProofProtocol
  range_proof/u64_proof.rs  -> U64RangeProofProtocol

The reference's arithmetic lives in third-party crates that are NOT under
/root/reference: k256 0.13.3 (Cargo.lock:411) and merlin 3.0.0 (Cargo.lock:453).
Their published algorithms are restated here from the public specifications:
secp256k1 (SEC 2 v2 2.4.1), SEC1 compressed encoding, Keccak-f[1600] (FIPS 202),
STROBE-128 v1.0.2 and Merlin v1.0.

PARITY STATUS: **parity unpinned against the reference itself** -- the reference
has no golden vectors (src/tests.rs holds only OsRng round trips, tests.rs:13-171)
and cannot be built here (no cargo/rustc, k256/merlin not vendored).  What IS
pinned: secp256k1 public known answers (G, 2G, n*G = identity, lambda*G = (beta*x, y)),
Keccak-f via hashlib.sha3_256, the upstream Merlin known-answer test
("test protocol"/"some label"/"some data"/"challenge" ->
d5a21972...cf0615), the curve layer against an implementation that shares nothing
with this repository -- OpenSSL's secp256k1: k*G (affine and 33-byte SEC1) and x(k*P)
for 65 seeded and edge-case scalars, tests/golden/openssl_secp256k1.json,
tests/test_openssl_vectors.py -- and the reference's completeness property (honest
prove => verify true) on the three shapes of src/tests.rs.  See tests/test_oracle.py.
What stays unpinned is the PROTOCOL layer above k256 / merlin (the reference's own
Rust): facade/src/bin/gen_fixtures.rs is the hook that closes it on a machine with cargo.
"""
from __future__ import annotations

import hashlib
import struct
from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

# --------------------------------------------------------------------------
# secp256k1 (k256 0.13.3 semantics)
# --------------------------------------------------------------------------
P = 2**256 - 2**32 - 977
N = 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
B = 7
GX = 0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798
GY = 0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8
BETA = 0x7AE96A2B657C07106E64479EAC3434E99CF0497512F58995C1396C28719501EE
LAMBDA = 0x5363AD4CC05C30E0A5261C028812645A122E22EA20816678DF02967C1B23BD72

Point = Optional[Tuple[int, int]]  # affine (x, y); None = identity
IDENTITY: Point = None
G: Point = (GX, GY)


def on_curve(pt: Point) -> bool:
    if pt is None:
        return True
    x, y = pt
    return 0 <= x < P and 0 <= y < P and (y * y - x * x * x - B) % P == 0


def pt_neg(a: Point) -> Point:
    if a is None:
        return None
    return (a[0], (-a[1]) % P)


def pt_add(a: Point, b: Point) -> Point:
    """Textbook affine group law (ground truth; complete by case analysis)."""
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = (3 * x1 * x1) * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


def pt_sub(a: Point, b: Point) -> Point:
    return pt_add(a, pt_neg(b))


# Jacobian helpers: only an accelerator for pt_mul; cross-checked against the
# affine law in tests/test_oracle.py.
def _jac_dbl(p):
    X, Y, Z = p
    if Z == 0 or Y == 0:
        return (0, 1, 0)
    S = 4 * X * Y * Y % P
    M = 3 * X * X % P
    X3 = (M * M - 2 * S) % P
    Y3 = (M * (S - X3) - 8 * Y * Y * Y * Y) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def _jac_add_affine(p, q):
    """p Jacobian + q affine (q not identity)."""
    X1, Y1, Z1 = p
    x2, y2 = q
    if Z1 == 0:
        return (x2, y2, 1)
    Z1Z1 = Z1 * Z1 % P
    U2 = x2 * Z1Z1 % P
    S2 = y2 * Z1 * Z1Z1 % P
    if U2 == X1:
        if S2 == Y1:
            return _jac_dbl(p)
        return (0, 1, 0)
    H = (U2 - X1) % P
    R = (S2 - Y1) % P
    HH = H * H % P
    HHH = H * HH % P
    V = X1 * HH % P
    X3 = (R * R - HHH - 2 * V) % P
    Y3 = (R * (V - X3) - Y1 * HHH) % P
    Z3 = Z1 * H % P
    return (X3, Y3, Z3)


def _jac_to_affine(p) -> Point:
    X, Y, Z = p
    if Z == 0:
        return None
    zi = pow(Z, -1, P)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def pt_mul(a: Point, k: int) -> Point:
    """k*a, mathematically (k256 `ProjectivePoint * Scalar`; any algorithm gives the same group element)."""
    k %= N
    if a is None or k == 0:
        return None
    acc = (0, 1, 0)
    for bit in bin(k)[2:]:
        acc = _jac_dbl(acc)
        if bit == "1":
            acc = _jac_add_affine(acc, a)
    return _jac_to_affine(acc)


def pt_mul_affine_only(a: Point, k: int) -> Point:
    """Slow double-and-add using only the textbook affine law (used to validate pt_mul)."""
    k %= N
    acc = None
    for bit in bin(k)[2:] if k else "":
        acc = pt_add(acc, acc)
        if bit == "1":
            acc = pt_add(acc, a)
    return acc


def pt_to_bytes(a: Point) -> bytes:
    """k256 GroupEncoding::to_bytes: 33-byte SEC1 compressed; identity -> 33 zero bytes."""
    if a is None:
        return bytes(33)
    x, y = a
    return bytes([2 + (y & 1)]) + x.to_bytes(32, "big")


def pt_from_bytes(b: bytes) -> Point:
    """SEC1 compressed decode (sqrt via p = 3 mod 4)."""
    if b == bytes(33):
        return None
    if len(b) != 33 or b[0] not in (2, 3):
        raise ValueError("bad SEC1 compressed point")
    x = int.from_bytes(b[1:], "big")
    if x >= P:
        raise ValueError("x out of range")
    rhs = (x * x * x + B) % P
    y = pow(rhs, (P + 1) // 4, P)
    if y * y % P != rhs:
        raise ValueError("not on curve")
    if (y & 1) != (b[0] & 1):
        y = P - y
    return (x, y)


def pt_to_xy64(a: Point) -> bytes:
    """C-ABI point encoding: affine big-endian x||y, identity = 64 zero bytes (include/bppp.h)."""
    if a is None:
        return bytes(64)
    return a[0].to_bytes(32, "big") + a[1].to_bytes(32, "big")


def pt_from_xy64(b: bytes) -> Point:
    if b == bytes(64):
        return None
    pt = (int.from_bytes(b[:32], "big"), int.from_bytes(b[32:], "big"))
    if not on_curve(pt):
        raise ValueError("not on curve")
    return pt


def sc_to_bytes(s: int) -> bytes:
    return (s % N).to_bytes(32, "big")


def sc_from_bytes(b: bytes) -> int:
    v = int.from_bytes(b, "big")
    if v >= N:
        raise ValueError("non-canonical scalar")
    return v


def sc_inv(s: int) -> int:
    s %= N
    if s == 0:
        raise ZeroDivisionError("invert(0): the reference unwraps a None here")
    return pow(s, -1, N)


def wide_reduce(b64: bytes) -> int:
    """k256 Scalar::generate_biased: 64 big-endian bytes reduced mod n."""
    assert len(b64) == 64
    return int.from_bytes(b64, "big") % N


# --------------------------------------------------------------------------
# Keccak-f[1600], STROBE-128, Merlin (merlin 3.0.0)
# --------------------------------------------------------------------------
_M64 = (1 << 64) - 1
_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_ROT = [
    [0, 36, 3, 41, 18],
    [1, 44, 10, 45, 2],
    [62, 6, 43, 15, 61],
    [28, 55, 25, 21, 56],
    [27, 20, 39, 8, 14],
]


def _rol(v, r):
    r %= 64
    return ((v << r) | (v >> (64 - r))) & _M64 if r else v


def keccak_f1600(lanes: List[int]) -> List[int]:
    """lanes[x + 5*y], 25 little-endian u64 lanes (FIPS 202)."""
    A = [[lanes[x + 5 * y] for y in range(5)] for x in range(5)]
    for rnd in range(24):
        C = [A[x][0] ^ A[x][1] ^ A[x][2] ^ A[x][3] ^ A[x][4] for x in range(5)]
        D = [C[(x - 1) % 5] ^ _rol(C[(x + 1) % 5], 1) for x in range(5)]
        A = [[A[x][y] ^ D[x] for y in range(5)] for x in range(5)]
        Bm = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                Bm[y][(2 * x + 3 * y) % 5] = _rol(A[x][y], _ROT[x][y])
        A = [[Bm[x][y] ^ ((~Bm[(x + 1) % 5][y]) & Bm[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        A[0][0] ^= _RC[rnd]
    return [A[i % 5][i // 5] for i in range(25)]


def keccak_f1600_bytes(state: bytearray) -> None:
    lanes = list(struct.unpack("<25Q", bytes(state)))
    lanes = keccak_f1600(lanes)
    state[:] = struct.pack("<25Q", *lanes)


STROBE_R = 166
FLAG_I, FLAG_A, FLAG_C, FLAG_T, FLAG_M, FLAG_K = 1, 2, 4, 8, 16, 32


class Strobe128:
    """merlin::strobe::Strobe128 (merlin 3.0.0 src/strobe.rs)."""

    def __init__(self, protocol_label: bytes):
        st = bytearray(200)
        st[0:6] = bytes([1, STROBE_R + 2, 1, 0, 1, 96])
        st[6:18] = b"STROBEv1.0.2"
        keccak_f1600_bytes(st)
        self.state = st
        self.pos = 0
        self.pos_begin = 0
        self.cur_flags = 0
        self.meta_ad(protocol_label, False)

    def clone(self) -> "Strobe128":
        c = object.__new__(Strobe128)
        c.state = bytearray(self.state)
        c.pos, c.pos_begin, c.cur_flags = self.pos, self.pos_begin, self.cur_flags
        return c

    def _run_f(self):
        self.state[self.pos] ^= self.pos_begin
        self.state[self.pos + 1] ^= 0x04
        self.state[STROBE_R + 1] ^= 0x80
        keccak_f1600_bytes(self.state)
        self.pos = 0
        self.pos_begin = 0

    def _absorb(self, data: bytes):
        for byte in data:
            self.state[self.pos] ^= byte
            self.pos += 1
            if self.pos == STROBE_R:
                self._run_f()

    def _squeeze(self, n: int) -> bytes:
        out = bytearray(n)
        for i in range(n):
            out[i] = self.state[self.pos]
            self.state[self.pos] = 0
            self.pos += 1
            if self.pos == STROBE_R:
                self._run_f()
        return bytes(out)

    def _begin_op(self, flags: int, more: bool):
        if more:
            assert self.cur_flags == flags
            return
        assert flags & FLAG_T == 0
        old_begin = self.pos_begin
        self.pos_begin = self.pos + 1
        self.cur_flags = flags
        self._absorb(bytes([old_begin, flags]))
        force_f = (flags & (FLAG_C | FLAG_K)) != 0
        if force_f and self.pos != 0:
            self._run_f()

    def meta_ad(self, data: bytes, more: bool):
        self._begin_op(FLAG_M | FLAG_A, more)
        self._absorb(data)

    def ad(self, data: bytes, more: bool):
        self._begin_op(FLAG_A, more)
        self._absorb(data)

    def prf(self, n: int, more: bool) -> bytes:
        self._begin_op(FLAG_I | FLAG_A | FLAG_C, more)
        return self._squeeze(n)


class Transcript:
    """merlin::Transcript (merlin 3.0.0 src/transcript.rs)."""

    def __init__(self, label: bytes):
        self.strobe = Strobe128(b"Merlin v1.0")
        self.append_message(b"dom-sep", label)

    def clone(self) -> "Transcript":
        c = object.__new__(Transcript)
        c.strobe = self.strobe.clone()
        return c

    def append_message(self, label: bytes, message: bytes):
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(struct.pack("<I", len(message)), True)
        self.strobe.ad(message, False)

    def append_u64(self, label: bytes, x: int):
        self.append_message(label, struct.pack("<Q", x))

    def challenge_bytes(self, label: bytes, n: int) -> bytes:
        self.strobe.meta_ad(label, False)
        self.strobe.meta_ad(struct.pack("<I", n), True)
        return self.strobe.prf(n, False)


class DegenerateChallenge(Exception):
    """Raised where the reference would panic on `.unwrap()` (transcript.rs:13, circuit.rs:192,196, ...)."""


def app_point(label: bytes, p: Point, t: Transcript):
    """transcript.rs:6-8"""
    t.append_message(label, pt_to_bytes(p))


def get_challenge(label: bytes, t: Transcript) -> int:
    """transcript.rs:10-14: 32 PRF bytes, big-endian, from_repr(..).unwrap() (no reduction)."""
    v = int.from_bytes(t.challenge_bytes(label, 32), "big")
    if v >= N:
        raise DegenerateChallenge("challenge >= n (reference panics)")
    return v


# --------------------------------------------------------------------------
# util.rs -- generic vector helpers over "T in {Scalar, ProjectivePoint}"
# --------------------------------------------------------------------------
class _ScalarOps:
    zero = 0

    @staticmethod
    def add(a, b):
        return (a + b) % N

    @staticmethod
    def sub(a, b):
        return (a - b) % N

    @staticmethod
    def mul(a, s):
        return a * s % N


class _PointOps:
    zero = None

    @staticmethod
    def add(a, b):
        return pt_add(a, b)

    @staticmethod
    def sub(a, b):
        return pt_sub(a, b)

    @staticmethod
    def mul(a, s):
        return pt_mul(a, s)


SC = _ScalarOps
PT = _PointOps


def reduce(v: Sequence) -> Tuple[list, list]:
    """util.rs:7-22: (even-index entries, odd-index entries)."""
    return list(v[0::2]), list(v[1::2])


def vector_extend(v: Sequence, n: int, ops) -> list:
    """util.rs:24-26: right-pad with T::default() (also truncates when n < len, as the reference's map does)."""
    return [v[i] if i < len(v) else ops.zero for i in range(n)]


def weight_vector_mul(a: Sequence, b: Sequence[int], weight: int, ops=SC):
    """util.rs:28-44: sum a_i * (b_i * w^(i+1)); exponent starts at 1."""
    n = max(len(a), len(b))
    a_ext = vector_extend(a, n, ops)
    b_ext = vector_extend(b, n, SC)
    exp = 1
    result = ops.zero
    for a_val, b_val in zip(a_ext, b_ext):
        exp = exp * weight % N
        result = ops.add(result, ops.mul(a_val, b_val * exp % N))
    return result


def vector_mul(a: Sequence, b: Sequence[int], ops=SC):
    """util.rs:46-60: inner product; naive MSM when T = point."""
    n = max(len(a), len(b))
    a_ext = vector_extend(a, n, ops)
    b_ext = vector_extend(b, n, SC)
    result = ops.zero
    for a_val, b_val in zip(a_ext, b_ext):
        result = ops.add(result, ops.mul(a_val, b_val))
    return result


def vector_mul_on_scalar(a: Sequence, s: int, ops=SC) -> list:
    """util.rs:62-67"""
    return [ops.mul(x, s) for x in a]


def vector_add(a: Sequence, b: Sequence, ops=SC) -> list:
    """util.rs:69-76"""
    n = max(len(a), len(b))
    return [ops.add(x, y) for x, y in zip(vector_extend(a, n, ops), vector_extend(b, n, ops))]


def vector_sub(a: Sequence, b: Sequence, ops=SC) -> list:
    """util.rs:78-85"""
    n = max(len(a), len(b))
    return [ops.sub(x, y) for x, y in zip(vector_extend(a, n, ops), vector_extend(b, n, ops))]


def e_vec(v: int, n: int) -> List[int]:
    """util.rs:87-95 `e`: [1, v, v^2, ..., v^(n-1)]."""
    out, buf = [], 1
    for _ in range(n):
        out.append(buf)
        buf = buf * v % N
    return out


def sc_pow(s: int, n: int) -> int:
    """util.rs:97-99"""
    return pow(s, n, N)


def vector_tensor_mul(a: Sequence[int], b: Sequence[int]) -> List[int]:
    """util.rs:111-116: concat over x in b of a*x."""
    out: List[int] = []
    for x in b:
        out.extend(vector_mul_on_scalar(a, x))
    return out


def diag_inv(x: int, n: int) -> List[List[int]]:
    """util.rs:118-132: dense n x n, x^-(i+1) on the diagonal."""
    x_inv = sc_inv(x)
    val = 1
    rows = []
    for i in range(n):
        row = []
        for j in range(n):
            if i == j:
                val = val * x_inv % N
                row.append(val)
            else:
                row.append(0)
        rows.append(row)
    return rows


def vector_mul_on_matrix(a: Sequence[int], m: Sequence[Sequence[int]]) -> List[int]:
    """util.rs:134-142: row vector times dense matrix, column at a time."""
    return [vector_mul(a, [row[j] for row in m]) for j in range(len(m[0]))]


def minus(v: int) -> int:
    """util.rs:153-155"""
    return v * (N - 1) % N


# --------------------------------------------------------------------------
# wnla.rs
# --------------------------------------------------------------------------
@dataclass
class WnlaProof:
    """wnla.rs:25-30"""
    r: List[Point]
    x: List[Point]
    l: List[int]
    n: List[int]


@dataclass
class WeightNormLinearArgument:
    """wnla.rs:12-19"""
    g: Point
    g_vec: List[Point]
    h_vec: List[Point]
    c: List[int]
    rho: int
    mu: int

    def commit(self, l: Sequence[int], n: Sequence[int]) -> Point:
        """wnla.rs:66-72"""
        v = (vector_mul(self.c, l) + weight_vector_mul(n, n, self.mu)) % N
        return pt_add(pt_add(pt_mul(self.g, v), vector_mul(self.h_vec, l, PT)), vector_mul(self.g_vec, n, PT))

    def verify(self, commitment: Point, t: Transcript, proof: WnlaProof, trace: Optional[list] = None) -> bool:
        """wnla.rs:75-121 (recursion unrolled into a loop; same operations in the same order)."""
        w = self
        com = commitment
        r, x = list(proof.r), list(proof.x)
        while True:
            if len(x) != len(r):
                return False
            if not x:
                return com == w.commit(proof.l, proof.n)
            c0, c1 = reduce(w.c)
            g0, g1 = reduce(w.g_vec)
            h0, h1 = reduce(w.h_vec)
            app_point(b"wnla_com", com, t)
            app_point(b"wnla_x", x[-1], t)
            app_point(b"wnla_r", r[-1], t)
            t.append_u64(b"l.sz", len(w.h_vec))
            t.append_u64(b"n.sz", len(w.g_vec))
            y = get_challenge(b"wnla_challenge", t)
            if trace is not None:
                trace.append(("wnla_com", com))
                trace.append(("wnla_y", y))
            h_ = vector_add(h0, vector_mul_on_scalar(h1, y, PT), PT)
            g_ = vector_add(vector_mul_on_scalar(g0, w.rho, PT), vector_mul_on_scalar(g1, y, PT), PT)
            c_ = vector_add(c0, vector_mul_on_scalar(c1, y))
            com = pt_add(pt_add(com, pt_mul(x[-1], y)), pt_mul(r[-1], (y * y - 1) % N))
            w = WeightNormLinearArgument(g=w.g, g_vec=g_, h_vec=h_, c=c_, rho=w.mu, mu=w.mu * w.mu % N)
            r, x = r[:-1], x[:-1]

    def prove(self, commitment: Point, t: Transcript, l: List[int], n: List[int]) -> WnlaProof:
        """wnla.rs:125-190 (recursive, as the reference)."""
        if len(l) + len(n) < 6:
            return WnlaProof(r=[], x=[], l=list(l), n=list(n))
        rho_inv = sc_inv(self.rho)
        c0, c1 = reduce(self.c)
        l0, l1 = reduce(l)
        n0, n1 = reduce(n)
        g0, g1 = reduce(self.g_vec)
        h0, h1 = reduce(self.h_vec)
        mu2 = self.mu * self.mu % N
        vx = (weight_vector_mul(n0, n1, mu2) * (rho_inv * 2 % N) + vector_mul(c0, l1) + vector_mul(c1, l0)) % N
        vr = (weight_vector_mul(n1, n1, mu2) + vector_mul(c1, l1)) % N
        x = pt_mul(self.g, vx)
        x = pt_add(x, vector_mul(h0, l1, PT))
        x = pt_add(x, vector_mul(h1, l0, PT))
        x = pt_add(x, vector_mul(g0, vector_mul_on_scalar(n1, self.rho), PT))
        x = pt_add(x, vector_mul(g1, vector_mul_on_scalar(n0, rho_inv), PT))
        r = pt_mul(self.g, vr)
        r = pt_add(r, vector_mul(h1, l1, PT))
        r = pt_add(r, vector_mul(g1, n1, PT))
        app_point(b"wnla_com", commitment, t)
        app_point(b"wnla_x", x, t)
        app_point(b"wnla_r", r, t)
        t.append_u64(b"l.sz", len(l))
        t.append_u64(b"n.sz", len(n))
        y = get_challenge(b"wnla_challenge", t)
        h_ = vector_add(h0, vector_mul_on_scalar(h1, y, PT), PT)
        g_ = vector_add(vector_mul_on_scalar(g0, self.rho, PT), vector_mul_on_scalar(g1, y, PT), PT)
        c_ = vector_add(c0, vector_mul_on_scalar(c1, y))
        l_ = vector_add(l0, vector_mul_on_scalar(l1, y))
        n_ = vector_add(vector_mul_on_scalar(n0, rho_inv), vector_mul_on_scalar(n1, y))
        w = WeightNormLinearArgument(g=self.g, g_vec=g_, h_vec=h_, c=c_, rho=self.mu, mu=mu2)
        proof = w.prove(w.commit(l_, n_), t, l_, n_)
        proof.r.append(r)
        proof.x.append(x)
        return proof


# --------------------------------------------------------------------------
# circuit.rs
# --------------------------------------------------------------------------
LO, LL, LR, NO = "LO", "LL", "LR", "NO"  # circuit.rs:15-20 PartitionType


@dataclass
class CircuitProof:
    """circuit.rs:24-33"""
    c_l: Point
    c_r: Point
    c_o: Point
    c_s: Point
    r: List[Point]
    x: List[Point]
    l: List[int]
    n: List[int]


@dataclass
class CircuitWitness:
    """circuit.rs:80-91"""
    v: List[List[int]]
    s_v: List[int]
    w_l: List[int]
    w_r: List[int]
    w_o: List[int]


class ScalarRng:
    """Stand-in for `Scalar::generate_biased(rng)`: yields caller-supplied scalars in draw order."""

    def __init__(self, scalars: Sequence[int]):
        self._it = iter(scalars)
        self.drawn = 0

    def __call__(self) -> int:
        self.drawn += 1
        return next(self._it) % N


@dataclass
class ArithmeticCircuit:
    """circuit.rs:95-139"""
    dim_nm: int
    dim_no: int
    k: int
    dim_nl: int
    dim_nv: int
    dim_nw: int
    g: Point
    g_vec: List[Point]
    h_vec: List[Point]
    W_m: List[List[int]]
    W_l: List[List[int]]
    a_m: List[int]
    a_l: List[int]
    f_l: bool
    f_m: bool
    g_vec_: List[Point]
    h_vec_: List[Point]
    partition: Callable[[str, int], Optional[int]]

    def commit(self, v: Sequence[int], s: int) -> Point:
        """circuit.rs:146-151"""
        return pt_add(pt_add(pt_mul(self.g, v[0]), pt_mul(self.h_vec[0], s)), vector_mul(self.h_vec[9:], v[1:], PT))

    # -- private helpers, circuit.rs:559-653
    def linear_comb_coef(self, i: int, lam: int, mu: int) -> int:
        coef = 0
        if self.f_l:
            coef = (coef + sc_pow(lam, self.dim_nv * i)) % N
        if self.f_m:
            coef = (coef + sc_pow(mu, self.dim_nv * i + 1)) % N
        return coef

    def collect_cl0(self, lam: int, mu: int) -> List[int]:
        c_l0 = [0] * (self.dim_nv - 1)
        if self.f_l:
            c_l0 = e_vec(lam, self.dim_nv)[1:]
        if self.f_m:
            c_l0 = vector_sub(c_l0, vector_mul_on_scalar(e_vec(mu, self.dim_nv)[1:], mu))
        return c_l0

    def collect_lambda(self, lam: int, mu: int) -> List[int]:
        lambda_vec = e_vec(lam, self.dim_nl)
        if self.f_l and self.f_m:
            lambda_vec = vector_sub(
                lambda_vec,
                vector_add(
                    vector_tensor_mul(vector_mul_on_scalar(e_vec(lam, self.dim_nv), mu), e_vec(sc_pow(mu, self.dim_nv), self.k)),
                    vector_tensor_mul(e_vec(mu, self.dim_nv), e_vec(sc_pow(lam, self.dim_nv), self.k)),
                ),
            )
        return lambda_vec

    def collect_m_rl(self):
        nm = self.dim_nm
        M_lnL = [list(self.W_l[i][:nm]) for i in range(self.dim_nl)]
        M_mnL = [list(self.W_m[i][:nm]) for i in range(nm)]
        M_lnR = [list(self.W_l[i][nm:2 * nm]) for i in range(self.dim_nl)]
        M_mnR = [list(self.W_m[i][nm:2 * nm]) for i in range(nm)]
        return M_lnL, M_mnL, M_lnR, M_mnR

    def collect_m_o(self):
        nm = self.dim_nm
        W_lO = [list(self.W_l[i][2 * nm:]) for i in range(self.dim_nl)]
        W_mO = [list(self.W_m[i][2 * nm:]) for i in range(nm)]

        def map_f(isz, jsz, typ, W_x):
            out = []
            for i in range(isz):
                row = []
                for j in range(jsz):
                    j_ = self.partition(typ, j)
                    row.append(W_x[i][j_] if j_ is not None else 0)
                out.append(row)
            return out

        M_lnO = map_f(self.dim_nl, nm, NO, W_lO)
        M_llL = map_f(self.dim_nl, self.dim_nv, LL, W_lO)
        M_llR = map_f(self.dim_nl, self.dim_nv, LR, W_lO)
        M_llO = map_f(self.dim_nl, self.dim_nv, LO, W_lO)
        M_mnO = map_f(nm, nm, NO, W_mO)
        M_mlL = map_f(nm, self.dim_nv, LL, W_mO)
        M_mlR = map_f(nm, self.dim_nv, LR, W_mO)
        M_mlO = map_f(nm, self.dim_nv, LO, W_mO)
        return M_lnO, M_mnO, M_llL, M_mlL, M_llR, M_mlR, M_llO, M_mlO

    def collect_c(self, lambda_vec, mu_vec, mu):
        M_lnL, M_mnL, M_lnR, M_mnR = self.collect_m_rl()
        M_lnO, M_mnO, M_llL, M_mlL, M_llR, M_mlR, M_llO, M_mlO = self.collect_m_o()
        mu_diag_inv = diag_inv(mu, self.dim_nm)
        vmm = vector_mul_on_matrix
        c_nL = vmm(vector_sub(vmm(lambda_vec, M_lnL), vmm(mu_vec, M_mnL)), mu_diag_inv)
        c_nR = vmm(vector_sub(vmm(lambda_vec, M_lnR), vmm(mu_vec, M_mnR)), mu_diag_inv)
        c_nO = vmm(vector_sub(vmm(lambda_vec, M_lnO), vmm(mu_vec, M_mnO)), mu_diag_inv)
        c_lL = vector_sub(vmm(lambda_vec, M_llL), vmm(mu_vec, M_mlL))
        c_lR = vector_sub(vmm(lambda_vec, M_llR), vmm(mu_vec, M_mlR))
        c_lO = vector_sub(vmm(lambda_vec, M_llO), vmm(mu_vec, M_mlO))
        return c_nL, c_nR, c_nO, c_lL, c_lR, c_lO

    def verify(self, v: Sequence[Point], t: Transcript, proof: CircuitProof, trace: Optional[list] = None) -> bool:
        """circuit.rs:154-256"""
        app_point(b"commitment_cl", proof.c_l, t)
        app_point(b"commitment_cr", proof.c_r, t)
        app_point(b"commitment_co", proof.c_o, t)
        for v_val in v:
            app_point(b"commitment_v", v_val, t)
        rho = get_challenge(b"circuit_rho", t)
        lam = get_challenge(b"circuit_lambda", t)
        beta = get_challenge(b"circuit_beta", t)
        delta = get_challenge(b"circuit_delta", t)
        mu = rho * rho % N
        lambda_vec = self.collect_lambda(lam, mu)
        mu_vec = vector_mul_on_scalar(e_vec(mu, self.dim_nm), mu)
        c_nL, c_nR, c_nO, c_lL, c_lR, c_lO = self.collect_c(lambda_vec, mu_vec, mu)
        two = 2
        v_ = None
        for i in range(self.k):
            v_ = pt_add(v_, pt_mul(v[i], self.linear_comb_coef(i, lam, mu)))
        v_ = pt_mul(v_, two)
        app_point(b"commitment_cs", proof.c_s, t)
        tau = get_challenge(b"circuit_tau", t)
        tau_inv = sc_inv(tau)
        tau2 = tau * tau % N
        tau3 = tau2 * tau % N
        delta_inv = sc_inv(delta)
        pn_tau = vector_mul_on_scalar(c_nO, tau3 * delta_inv % N)
        pn_tau = vector_sub(pn_tau, vector_mul_on_scalar(c_nL, tau2))
        pn_tau = vector_add(pn_tau, vector_mul_on_scalar(c_nR, tau))
        ps_tau = (weight_vector_mul(pn_tau, pn_tau, mu)
                  + vector_mul(lambda_vec, self.a_l) * tau3 % N * two
                  - vector_mul(mu_vec, self.a_m) * tau3 % N * two) % N
        pt = pt_add(pt_mul(self.g, ps_tau), vector_mul(self.g_vec, pn_tau, PT))
        cr_tau = [
            1,
            tau_inv * beta % N,
            tau * beta % N,
            tau2 * beta % N,
            tau3 * beta % N,
            tau * tau3 % N * beta % N,
            tau2 * tau3 % N * beta % N,
            tau3 * tau3 % N * beta % N,
            tau3 * tau3 % N * tau % N * beta % N,
        ]
        c_l0 = self.collect_cl0(lam, mu)
        cl_tau = vector_mul_on_scalar(c_lO, tau3 * delta_inv % N)
        cl_tau = vector_sub(cl_tau, vector_mul_on_scalar(c_lL, tau2))
        cl_tau = vector_add(cl_tau, vector_mul_on_scalar(c_lR, tau))
        cl_tau = vector_mul_on_scalar(cl_tau, two)
        cl_tau = vector_sub(cl_tau, c_l0)
        c = cr_tau + cl_tau
        commitment = pt
        commitment = pt_add(commitment, pt_mul(proof.c_s, tau_inv))
        commitment = pt_sub(commitment, pt_mul(proof.c_o, delta))
        commitment = pt_add(commitment, pt_mul(proof.c_l, tau))
        commitment = pt_sub(commitment, pt_mul(proof.c_r, tau2))
        commitment = pt_add(commitment, pt_mul(v_, tau3))
        while len(c) < len(self.h_vec) + len(self.h_vec_):
            c.append(0)
        if trace is not None:
            trace.extend([("rho", rho), ("lambda", lam), ("beta", beta), ("delta", delta), ("tau", tau),
                          ("pn_tau", list(pn_tau)), ("ps_tau", ps_tau), ("c", list(c)), ("C0", commitment)])
        w = WeightNormLinearArgument(g=self.g, g_vec=self.g_vec + self.g_vec_, h_vec=self.h_vec + self.h_vec_,
                                     c=c, rho=rho, mu=mu)
        return w.verify(commitment, t, WnlaProof(r=proof.r, x=proof.x, l=proof.l, n=proof.n), trace)

    def prove(self, v: Sequence[Point], witness: CircuitWitness, t: Transcript, rng: Callable[[], int]) -> CircuitProof:
        """circuit.rs:260-556; `rng()` stands for Scalar::generate_biased(rng), same draw order."""
        ro = [rng(), rng(), rng(), rng(), 0, rng(), rng(), rng(), 0]
        rl = [rng(), rng(), rng(), 0, rng(), rng(), rng(), 0, 0]
        rr = [rng(), rng(), 0, rng(), rng(), rng(), 0, 0, 0]
        nl = list(witness.w_l)
        nr = list(witness.w_r)

        def part(typ, size):
            out = []
            for j in range(size):
                i = self.partition(typ, j)
                out.append(witness.w_o[i] if i is not None else 0)
            return out

        no = part(NO, self.dim_nm)
        lo = part(LO, self.dim_nv)
        ll = part(LL, self.dim_nv)
        lr = part(LR, self.dim_nv)
        co = pt_add(vector_mul(self.h_vec, ro + lo, PT), vector_mul(self.g_vec, no, PT))
        cl = pt_add(vector_mul(self.h_vec, rl + ll, PT), vector_mul(self.g_vec, nl, PT))
        cr = pt_add(vector_mul(self.h_vec, rr + lr, PT), vector_mul(self.g_vec, nr, PT))
        app_point(b"commitment_cl", cl, t)
        app_point(b"commitment_cr", cr, t)
        app_point(b"commitment_co", co, t)
        for v_val in v:
            app_point(b"commitment_v", v_val, t)
        rho = get_challenge(b"circuit_rho", t)
        lam = get_challenge(b"circuit_lambda", t)
        beta = get_challenge(b"circuit_beta", t)
        delta = get_challenge(b"circuit_delta", t)
        mu = rho * rho % N
        lambda_vec = self.collect_lambda(lam, mu)
        mu_vec = vector_mul_on_scalar(e_vec(mu, self.dim_nm), mu)
        c_nL, c_nR, c_nO, c_lL, c_lR, c_lO = self.collect_c(lambda_vec, mu_vec, mu)
        ls = [rng() for _ in range(self.dim_nv)]
        ns = [rng() for _ in range(self.dim_nm)]
        two = 2
        v_0 = 0
        for i in range(self.k):
            v_0 = (v_0 + witness.v[i][0] * self.linear_comb_coef(i, lam, mu)) % N
        v_0 = v_0 * two % N
        rv = [0] * 9
        for i in range(self.k):
            rv[0] = (rv[0] + witness.s_v[i] * self.linear_comb_coef(i, lam, mu)) % N
        rv[0] = rv[0] * two % N
        v_1 = [0] * (self.dim_nv - 1)
        for i in range(self.k):
            v_1 = vector_add(v_1, vector_mul_on_scalar(witness.v[i][1:], self.linear_comb_coef(i, lam, mu)))
        v_1 = vector_mul_on_scalar(v_1, two)
        c_l0 = self.collect_cl0(lam, mu)
        f_ = [0] * 8
        delta2 = delta * delta % N
        delta_inv = sc_inv(delta)
        wvm, vm, va = weight_vector_mul, vector_mul, vector_add
        # -2
        f_[0] = minus(wvm(ns, ns, mu))
        # -1
        f_[1] = (vm(c_l0, ls) + delta * two % N * wvm(ns, no, mu)) % N
        # 0
        f_[2] = (minus(vm(c_lR, ls) * two % N)
                 - vm(c_l0, lo) * delta
                 - wvm(ns, va(nl, c_nR), mu) * two
                 - wvm(no, no, mu) * delta2) % N
        # 1
        f_[3] = (vm(c_lL, ls) * two
                 + vm(c_lR, lo) * delta % N * two
                 + vm(c_l0, ll)
                 + wvm(ns, va(nr, c_nL), mu) * two
                 + wvm(no, va(nl, c_nR), mu) * two % N * delta) % N
        # 2
        f_[4] = (wvm(c_nR, c_nR, mu)
                 - vm(c_lO, ls) * delta_inv % N * two
                 - vm(c_lL, lo) * delta % N * two
                 - vm(c_lR, ll) * two
                 - vm(c_l0, lr)
                 - wvm(ns, c_nO, mu) * delta_inv % N * two
                 - wvm(no, va(nr, c_nL), mu) * delta % N * two
                 - wvm(va(nl, c_nR), va(nl, c_nR), mu)) % N
        # 4
        f_[5] = (wvm(c_nO, c_nR, mu) * delta_inv % N * two
                 + wvm(c_nL, c_nL, mu)
                 - vm(c_lO, ll) * delta_inv % N * two
                 - vm(c_lL, lr) * two
                 - vm(c_lR, v_1) * two
                 - wvm(va(nl, c_nR), c_nO, mu) * delta_inv % N * two
                 - wvm(va(nr, c_nL), va(nr, c_nL), mu)) % N
        # 5
        f_[6] = (minus(wvm(c_nO, c_nL, mu) * delta_inv % N * two % N)
                 + vm(c_nO, lr) * delta_inv % N * two
                 + vm(c_lL, v_1) * two
                 + wvm(va(nr, c_nL), c_nO, mu) * delta_inv % N * two) % N
        # 6
        f_[7] = minus(vm(c_lO, v_1) * delta_inv % N * two % N)
        beta_inv = sc_inv(beta)
        rs = [
            (f_[1] + ro[1] * delta % N * beta) % N,
            f_[0] * beta_inv % N,
            ((ro[0] * delta + f_[2]) % N * beta_inv - rl[1]) % N,
            ((f_[3] - rl[0]) % N * beta_inv + (ro[2] * delta + rr[1])) % N,
            ((f_[4] + rr[0]) % N * beta_inv + (ro[3] * delta - rl[2])) % N,
            minus(rv[0] * beta_inv % N),
            (f_[5] * beta_inv + ro[5] * delta + rr[3] - rl[4]) % N,
            (f_[6] * beta_inv + rr[4] + ro[6] * delta - rl[5]) % N,
            (f_[7] * beta_inv + ro[7] * delta - rl[6] + rr[5]) % N,
        ]
        cs = pt_add(vector_mul(self.h_vec, rs + ls, PT), vector_mul(self.g_vec, ns, PT))
        app_point(b"commitment_cs", cs, t)
        tau = get_challenge(b"circuit_tau", t)
        tau_inv = sc_inv(tau)
        tau2 = tau * tau % N
        tau3 = tau2 * tau % N
        vms = vector_mul_on_scalar
        l = vms(rs + ls, tau_inv)
        l = vector_sub(l, vms(ro + lo, delta))
        l = vector_add(l, vms(rl + ll, tau))
        l = vector_sub(l, vms(rr + lr, tau2))
        l = vector_add(l, vms(rv + v_1, tau3))
        pn_tau = vms(c_nO, tau3 * delta_inv % N)
        pn_tau = vector_sub(pn_tau, vms(c_nL, tau2))
        pn_tau = vector_add(pn_tau, vms(c_nR, tau))
        ps_tau = (wvm(pn_tau, pn_tau, mu)
                  + vm(lambda_vec, self.a_l) * tau3 % N * two
                  - vm(mu_vec, self.a_m) * tau3 % N * two) % N
        n_tau = vms(ns, tau_inv)
        n_tau = vector_sub(n_tau, vms(no, delta))
        n_tau = vector_add(n_tau, vms(nl, tau))
        n_tau = vector_sub(n_tau, vms(nr, tau2))
        n = vector_add(pn_tau, n_tau)
        cr_tau = [
            1,
            tau_inv * beta % N,
            tau * beta % N,
            tau2 * beta % N,
            tau3 * beta % N,
            tau * tau3 % N * beta % N,
            tau2 * tau3 % N * beta % N,
            tau3 * tau3 % N * beta % N,
            tau3 * tau3 % N * tau % N * beta % N,
        ]
        cl_tau = vms(c_lO, tau3 * delta_inv % N)
        cl_tau = vector_sub(cl_tau, vms(c_lL, tau2))
        cl_tau = vector_add(cl_tau, vms(c_lR, tau))
        cl_tau = vms(cl_tau, two)
        cl_tau = vector_sub(cl_tau, c_l0)
        c = cr_tau + cl_tau
        vv = (ps_tau + tau3 * v_0) % N
        commitment = pt_add(pt_add(pt_mul(self.g, vv), vector_mul(self.h_vec, l, PT)), vector_mul(self.g_vec, n, PT))
        while len(l) < len(self.h_vec) + len(self.h_vec_):
            l.append(0)
            c.append(0)
        while len(n) < len(self.g_vec) + len(self.g_vec_):
            n.append(0)
        w = WeightNormLinearArgument(g=self.g, g_vec=self.g_vec + self.g_vec_, h_vec=self.h_vec + self.h_vec_,
                                     c=c, rho=rho, mu=mu)
        pw = w.prove(commitment, t, l, n)
        return CircuitProof(c_l=cl, c_r=cr, c_o=co, c_s=cs, r=pw.r, x=pw.x, l=pw.l, n=pw.n)


# --------------------------------------------------------------------------
# range_proof/reciprocal.rs
# --------------------------------------------------------------------------
@dataclass
class ReciprocalProof:
    """reciprocal.rs:30-33"""
    circuit_proof: CircuitProof
    r: Point


@dataclass
class ReciprocalWitness:
    """reciprocal.rs:17-26"""
    x: int
    s: int
    m: List[int]
    digits: List[int]


@dataclass
class ReciprocalRangeProofProtocol:
    """reciprocal.rs:64-84"""
    dim_nd: int
    dim_np: int
    g: Point
    g_vec: List[Point]
    h_vec: List[Point]
    g_vec_: List[Point]
    h_vec_: List[Point]

    def commit_value(self, x: int, s: int) -> Point:
        """reciprocal.rs:88-90"""
        return pt_add(pt_mul(self.g, x), pt_mul(self.h_vec[0], s))

    def commit_poles(self, r: Sequence[int], s: int) -> Point:
        """reciprocal.rs:93-95"""
        return pt_add(pt_mul(self.h_vec[0], s), vector_mul(self.h_vec[9:], r, PT))

    def make_circuit(self, e: int) -> ArithmeticCircuit:
        """reciprocal.rs:150-214"""
        dim_nm = self.dim_nd
        dim_no = self.dim_np
        dim_nv = self.dim_nd + 1
        dim_nl = dim_nv
        dim_nw = self.dim_nd * 2 + self.dim_np
        a_m = [1] * dim_nm
        W_m = [[0] * dim_nw for _ in range(dim_nm)]
        for i in range(dim_nm):
            W_m[i][i + dim_nm] = minus(e)
        a_l = [0] * dim_nl
        base = self.dim_np % N
        W_l = [[0] * dim_nw for _ in range(dim_nl)]
        for i in range(dim_nm):
            W_l[0][i] = minus(sc_pow(base, i))
        for i in range(dim_nm):
            for j in range(dim_nm):
                W_l[i + 1][j + dim_nm] = 1
        for i in range(dim_nm):
            W_l[i + 1][i + dim_nm] = 0
        for i in range(dim_nm):
            for j in range(dim_no):
                W_l[i + 1][j + 2 * dim_nm] = minus(sc_inv((e + j) % N))
        np_ = self.dim_np

        def partition(typ, index):
            if typ == LL and index < np_:
                return index
            return None

        return ArithmeticCircuit(
            dim_nm=dim_nm, dim_no=dim_no, k=1, dim_nl=dim_nl, dim_nv=dim_nv, dim_nw=dim_nw,
            g=self.g, g_vec=list(self.g_vec), h_vec=list(self.h_vec), W_m=W_m, W_l=W_l, a_m=a_m, a_l=a_l,
            f_l=True, f_m=False, g_vec_=list(self.g_vec_), h_vec_=list(self.h_vec_), partition=partition)

    def verify(self, commitment: Point, proof: ReciprocalProof, t: Transcript, trace: Optional[list] = None) -> bool:
        """reciprocal.rs:98-107"""
        app_point(b"reciprocal_commitment", commitment, t)
        e = get_challenge(b"reciprocal_challenge", t)
        circuit = self.make_circuit(e)
        circuit_commitment = pt_add(commitment, proof.r)
        if trace is not None:
            trace.extend([("e", e), ("V+r", circuit_commitment)])
        return circuit.verify([circuit_commitment], t, proof.circuit_proof, trace)

    def prove(self, commitment: Point, witness: ReciprocalWitness, t: Transcript, rng: Callable[[], int]) -> ReciprocalProof:
        """reciprocal.rs:110-146"""
        app_point(b"reciprocal_commitment", commitment, t)
        e = get_challenge(b"reciprocal_challenge", t)
        r = [sc_inv((witness.digits[i] + e) % N) for i in range(self.dim_nd)]
        r_blind = rng()
        r_com = self.commit_poles(r, r_blind)
        v = [witness.x] + list(r)
        circuit = self.make_circuit(e)
        cw = CircuitWitness(v=[v], s_v=[(witness.s + r_blind) % N], w_l=list(witness.digits), w_r=list(r), w_o=list(witness.m))
        circuit_commitment = circuit.commit(cw.v[0], cw.s_v[0])
        return ReciprocalProof(circuit_proof=circuit.prove([circuit_commitment], cw, t, rng), r=r_com)


# --------------------------------------------------------------------------
# range_proof/u64_proof.rs
# --------------------------------------------------------------------------
G_VEC_FULL_SZ = 16       # u64_proof.rs:12
H_VEC_CIRCUIT_SZ = 26    # u64_proof.rs:13
H_VEC_FULL_SZ = 32       # u64_proof.rs:14
N_RNG_DRAWS_U64 = 52     # 1 + (7+6+5) + 17 + 16


def u64_to_hex(x: int) -> List[int]:
    """u64_proof.rs:84-90: little-endian hex digits."""
    out = []
    for _ in range(16):
        out.append(x % 16)
        x //= 16
    return out


def u64_to_hex_mapped(x: int) -> List[int]:
    """u64_proof.rs:92-102: digit multiplicities."""
    result = [0] * 16
    for _ in range(16):
        result[x % 16] += 1
        x //= 16
    return result


@dataclass
class U64RangeProofProtocol:
    """u64_proof.rs:19-28"""
    g: Point
    g_vec: List[Point]
    h_vec: List[Point]
    DIM_ND = 16
    DIM_NP = 16

    def _reciprocal(self) -> ReciprocalRangeProofProtocol:
        return ReciprocalRangeProofProtocol(
            dim_nd=self.DIM_ND, dim_np=self.DIM_NP, g=self.g, g_vec=list(self.g_vec),
            h_vec=list(self.h_vec[:H_VEC_CIRCUIT_SZ]), g_vec_=[], h_vec_=list(self.h_vec[H_VEC_CIRCUIT_SZ:]))

    def commit_value(self, x: int, s: int) -> Point:
        """u64_proof.rs:37-39"""
        return pt_add(pt_mul(self.g, x % N), pt_mul(self.h_vec[0], s))

    def verify(self, v: Point, proof: ReciprocalProof, t: Transcript, trace: Optional[list] = None) -> bool:
        """u64_proof.rs:42-54"""
        return self._reciprocal().verify(v, proof, t, trace)

    def prove(self, x: int, s: int, t: Transcript, rng: Callable[[], int]) -> ReciprocalProof:
        """u64_proof.rs:57-82"""
        digits = u64_to_hex(x)
        poles = u64_to_hex_mapped(x)
        rec = self._reciprocal()
        w = ReciprocalWitness(x=x % N, s=s, m=poles, digits=digits)
        return rec.prove(rec.commit_value(w.x, w.s), w, t, rng)


# --------------------------------------------------------------------------
# C-ABI wire layout helpers (include/bppp.h): u64 proof = 13 points + 3 scalars
#   points (64 B affine each): c_l, c_r, c_o, c_s, r[0..3], x[0..3], reciprocal r
#   scalars (32 B BE): l[0], l[1], n[0]                      => 13*64 + 3*32 = 928 B
# --------------------------------------------------------------------------
U64_PROOF_BYTES = 13 * 64 + 3 * 32


def u64_proof_to_bytes(p: ReciprocalProof) -> bytes:
    cp = p.circuit_proof
    assert len(cp.r) == 4 and len(cp.x) == 4 and len(cp.l) == 2 and len(cp.n) == 1
    pts = [cp.c_l, cp.c_r, cp.c_o, cp.c_s] + list(cp.r) + list(cp.x) + [p.r]
    return b"".join(pt_to_xy64(q) for q in pts) + b"".join(sc_to_bytes(s) for s in list(cp.l) + list(cp.n))


def u64_proof_from_bytes(b: bytes) -> ReciprocalProof:
    assert len(b) == U64_PROOF_BYTES
    pts = [pt_from_xy64(b[64 * i:64 * i + 64]) for i in range(13)]
    sc = [sc_from_bytes(b[832 + 32 * i:832 + 32 * i + 32]) for i in range(3)]
    cp = CircuitProof(c_l=pts[0], c_r=pts[1], c_o=pts[2], c_s=pts[3], r=pts[4:8], x=pts[8:12], l=sc[0:2], n=sc[2:3])
    return ReciprocalProof(circuit_proof=cp, r=pts[12])


# --------------------------------------------------------------------------
# Synthetic workload: SHAKE256 XOF, seed b"bppp-bench-v1"
# --------------------------------------------------------------------------
SEED = b"bppp-bench-v1"
LABEL = b"u64 range proof"  # benches/range_proof.rs:32


def xof(tag: bytes, idx: int, n: int, seed: bytes = SEED) -> bytes:
    return hashlib.shake_256(seed + tag + struct.pack("<Q", idx)).digest(n)


def synth_generator_scalar(i: int, seed: bytes = SEED) -> int:
    """Discrete log of synthetic generator i w.r.t. G (mirrors ProjectivePoint::random = random scalar * G)."""
    k = wide_reduce(xof(b"gen", i, 64, seed))
    return k if k else 1


def synth_generators(seed: bytes = SEED) -> Tuple[Point, List[Point], List[Point]]:
    pts = [pt_mul(G, synth_generator_scalar(i, seed)) for i in range(49)]
    return pts[0], pts[1:17], pts[17:49]


def synth_value(j: int, seed: bytes = SEED) -> int:
    forced = {0: 0, 1: 2**64 - 1, 2: 123456}
    if j in forced:
        return forced[j]
    return struct.unpack("<Q", xof(b"val", j, 8, seed))[0]


def synth_blinding(j: int, seed: bytes = SEED) -> int:
    return wide_reduce(xof(b"bld", j, 64, seed))


def synth_rng_scalars(j: int, seed: bytes = SEED) -> List[int]:
    raw = xof(b"rng", j, 64 * N_RNG_DRAWS_U64, seed)
    return [wide_reduce(raw[64 * i:64 * i + 64]) for i in range(N_RNG_DRAWS_U64)]


# --------------------------------------------------------------------------
# Setup step: reproducible generators with unknown discrete logarithms -- restatement of
# bppp_derive_generators (include/bppp.h) for the parity test; the reference itself draws random points
# (benches/range_proof.rs:18-20), so this has no reference counterpart to follow.
# --------------------------------------------------------------------------
def derive_generator(seed: bytes, index: int) -> Point:
    ctr = 0
    while True:
        xb = hashlib.shake_256(seed + b"bppp-gen" + struct.pack("<II", index, ctr)).digest(32)
        x = int.from_bytes(xb, "big")
        ctr += 1
        if x >= P:
            continue
        rhs = (x * x * x + 7) % P
        y = pow(rhs, (P + 1) // 4, P)
        if y * y % P != rhs:
            continue
        return (x, y if y % 2 == 0 else P - y)
