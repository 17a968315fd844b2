"""The size -> launch-sequence decision of the u64 verifier and prover (csrc/plan_core.h, exported as bppp_u64_plan / bppp_plan_describe):
a pure function, so the CPU tier can check that it is total and that it changes exactly at the documented sizes.  The GPU twin,
tests/test_gpu_plan_boundaries.py, runs T-1, T, T+1 proofs for every threshold T against the oracle and asserts which plan was taken."""
import os

import pytest


@pytest.fixture(scope="module")
def L():
    from bp_pp_amd import _build, _capi
    if not os.path.exists(_build.SO):
        pytest.skip("libbppp_hip.so not built yet")
    return _capi.lib()


def _changes(L, prove, S, flags, upto):
    prev, out = None, []
    for n in range(1, upto + 1):
        code = L.bppp_u64_plan(prove, n, S, flags)
        assert code >= 0
        if prev is not None and code != prev:
            out.append(n)
        prev = code
    return out


def test_verify_plan_changes_only_at_the_documented_sizes(L):
    # first size of each regime on an MI355X (S = 1,024 SIMDs): S+1, 4S+1, 16S+1, 32S+1, 64S+1 (from here to 128 S the one-lane sums
    # pace their wave priority), 128S (one lane per fixed-base sum; the batch as two half-batch chains: twin), 128S+1 (no pacing)
    assert _changes(L, 0, 1024, 0, 140000) == [1025, 4097, 16385, 32769, 65537, 131072, 131073]
    # RLC mode: the same regimes (the last round is never split there and there is no twin form, which does not move a boundary)
    assert _changes(L, 0, 1024, 1, 140000) == [1025, 4097, 16385, 32769, 65537, 131072, 131073]
    # per-kernel timing on: nothing runs beside anything, so the regimes that differ only by that collapse
    assert _changes(L, 0, 1024, 2, 140000) == [1025, 4097, 16385, 32769, 65537, 131072, 131073]
    # the thresholds scale with the device
    assert _changes(L, 0, 256, 0, 40000) == [257, 1025, 4097, 8193, 16385, 32768, 32769]
    # beyond 128 S: the twin form while the chip is filled once or twice -- up to 2.25 x 128 S, except where the last generation of
    # wavefronts would be 30 .. 70 % full (2,663 .. 3,481 workgroups of 64: off from 170,369, on again from 222,785 proofs) --, then the
    # number of proofs that share a field inversion: 8 beyond the twin range (from 256 S without it), 16 from 1,024 S
    assert _changes(L, 0, 1024, 0, 1100000) == [1025, 4097, 16385, 32769, 65537, 131072, 131073, 170369, 222785, 294913, 1048576]
    assert _changes(L, 0, 1024, 2, 1100000) == [1025, 4097, 16385, 32769, 65537, 131072, 131073, 262144, 1048576]       # (timing on: no twin)
    assert _changes(L, 0, 64, 0, 70000) == [65, 257, 1025, 2049, 4097, 8192, 8193, 10625, 13889, 18433, 65536]


def test_prove_plan_changes_only_at_the_documented_sizes(L):
    S = 1024
    expect = sorted({S + 1,                  # a wavefront per sum, wide round scalars
                     4 * S + 1,              # 16-lane stages and folds
                     128 * S // 16,          # fused launches of 4 jobs on 4 lanes per sum (4 * 4 n >= 128 S)
                     -(-128 * S // 12),      # ... of 3 jobs
                     16 * S - 63,            # the variable-base next commitment would go to the helper stream (4 * ceil(n / 64) >= S)
                     128 * S // 8,           # ... of 2 jobs
                     16 * S + 1,             # 4-lane stages and folds
                     128 * S // 4,           # every launch on 4 lanes per sum
                     32 * S + 1,             # next commitment by the variable-base kernel
                     64 * S + 1,             # 256-register builds of the one-lane kernels
                     128 * S,                # one lane per sum
                     128 * S + 1})           # round scalars in one part
    assert _changes(L, 1, S, 0, 140000) == expect
    assert _changes(L, 1, S, 1, 140000) == expect          # "ct_prover" changes the secret sums' kernels, not the regimes


def test_plan_is_total_and_describable(L):
    import ctypes as C
    from bp_pp_amd import _capi
    from bp_pp_amd.range_proof import plan_for
    buf = C.create_string_buffer(256)
    for prove in (0, 1):
        for S in (1, 4, 304, 1024, 4096):
            for n in [0, 1, 2, 63, 64, 65] + [S * k + d for k in (1, 4, 16, 32, 64, 128, 1000) for d in (-1, 0, 1)] + [2**31, 2**40]:
                for flags in range(4):
                    code = L.bppp_u64_plan(prove, max(n, 0), S, flags)
                    assert 0 <= code < 2**32
                    assert L.bppp_plan_describe(code, prove, buf, len(buf)) > 0 and b"?" not in buf.value
    E = _capi.ERR_INVALID_ARG
    assert L.bppp_u64_plan(2, 10, 1024, 0) == E and L.bppp_u64_plan(0, 10, 0, 0) == E and L.bppp_u64_plan(0, 10, 1024, 4) == E
    assert L.bppp_plan_describe(-1, 0, buf, len(buf)) == E
    # what the regimes look like (the strings the GPU tier asserts on)
    assert plan_for(1024) == "phase1=g16 tables=aside/4 fb=l64 c0var=g64 round=g16 tail_beside=0 small=1 split=1 twin=1 pace=0 shared_inv=0"
    assert plan_for(4096) == "phase1=g16 tables=aside/2 fb=l64 c0var=g32 round=g8 tail_beside=0 small=1 split=1 twin=1 pace=0 shared_inv=0"
    assert plan_for(16384) == "phase1=small tables=aside/1 fb=l8 c0var=g4 round=g4 tail_beside=0 small=1 split=0 twin=1 pace=0 shared_inv=0"
    assert plan_for(32768) == "phase1=wg4 tables=beside/1 fb=l8 c0var=small round=g2 tail_beside=0 small=1 split=0 twin=1 pace=0 shared_inv=0"
    assert plan_for(65536) == "phase1=wg4 tables=beside/1 fb=l8 c0var=small round=small tail_beside=1 small=1 split=0 twin=1 pace=0 shared_inv=0"
    assert plan_for(65537) == "phase1=full tables=inline/1 fb=l8 c0var=full round=full tail_beside=0 small=0 split=0 twin=1 pace=1 shared_inv=0"
    # (a twin call is described by what each of its two chains runs)
    assert plan_for(131072) == "phase1=full tables=inline/1 fb=l1 c0var=full round=full tail_beside=0 small=0 split=0 twin=2 pace=1 shared_inv=0"
    assert plan_for(131072, timing=True) == "phase1=full tables=inline/1 fb=l1 c0var=full round=full tail_beside=0 small=0 split=0 twin=1 pace=1 shared_inv=0"
    assert plan_for(1 << 18) == "phase1=full tables=inline/1 fb=l1 c0var=full round=full tail_beside=0 small=0 split=0 twin=2 pace=0 shared_inv=0"
    assert plan_for(1 << 18, timing=True) == "phase1=full tables=inline/1 fb=l1 c0var=full round=full tail_beside=0 small=0 split=0 twin=1 pace=0 shared_inv=8"
    assert plan_for(196608) == "phase1=full tables=inline/1 fb=l1 c0var=full round=full tail_beside=0 small=0 split=0 twin=1 pace=0 shared_inv=0"
    assert plan_for(1 << 19) == "phase1=full tables=inline/1 fb=l1 c0var=full round=full tail_beside=0 small=0 split=0 twin=1 pace=0 shared_inv=8"
    assert plan_for(1 << 20) == "phase1=full tables=inline/1 fb=l1 c0var=full round=full tail_beside=0 small=0 split=0 twin=1 pace=0 shared_inv=16"
    assert plan_for(1 << 14, prove=True) == "fb=l8 fb4_from_jobs=2 stage=g4 fold=g4 scalars=parts next_by_msm=1 w2=0 overlap_next=1 next_g4=1 ct=0"
