"""bp_pp_amd: MI355X-native batch engine for the Bulletproofs++ u64 range-proof hot path of distributed-lab/bp-pp.

The product is libbppp_hip.so (hand-written HIP kernels for gfx950 behind the C ABI of include/bppp.h); this package is
the thin Python mirror of the reference's `U64RangeProofProtocol` used by tests/ and bench.py.  (The directory is named
bp_pp_amd because `bp-pp_amd` is not an importable Python identifier.)"""
from .range_proof import U64RangeProofProtocol, U64_PROOF_BYTES, G_VEC_FULL_SZ, H_VEC_FULL_SZ, derive_generators  # noqa: F401
from ._capi import BpppError  # noqa: F401
