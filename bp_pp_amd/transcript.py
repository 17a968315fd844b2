"""`merlin::Transcript` as the 203 serialized bytes the C ABI exchanges (include/bppp.h: bppp_transcript_*): the host-side mirror
of what the reference's callers hold when they pass `t: &mut Transcript` (u64_proof.rs:42, tests.rs:34,40).  Host code of
libbppp_hip.so only -- no GPU is touched."""
from __future__ import annotations

import ctypes as C
import struct

from . import _capi

STATE_BYTES = 203


class Transcript:
    def __init__(self, label: bytes = None, state: bytes = None):
        self._buf = C.create_string_buffer(STATE_BYTES)
        if state is not None:
            if len(state) != STATE_BYTES:
                raise ValueError("a serialized transcript is 203 bytes")
            self._buf.raw = bytes(state)
        else:
            _capi.check(_capi.lib().bppp_transcript_new(label, len(label), self._buf))

    @property
    def state(self) -> bytes:
        return self._buf.raw

    def clone(self) -> "Transcript":
        return Transcript(state=self.state)

    def append_message(self, label: bytes, message: bytes) -> None:
        _capi.check(_capi.lib().bppp_transcript_append_message(self._buf, label, len(label), message, len(message)))

    def append_u64(self, label: bytes, x: int) -> None:
        self.append_message(label, struct.pack("<Q", x))

    def challenge_bytes(self, label: bytes, n: int) -> bytes:
        out = C.create_string_buffer(n)
        _capi.check(_capi.lib().bppp_transcript_challenge_bytes(self._buf, label, len(label), out, n))
        return out.raw
