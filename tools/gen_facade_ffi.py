#!/usr/bin/env python3
"""Generate facade/src/ffi.rs -- the Rust `extern "C"` block -- from include/bppp.h, so that the facade declares every exported
symbol with the header's exact parameter list.  tests/test_facade_tree.py re-runs this and fails when ffi.rs is out of date."""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TMAP = [(r'const uint8_t\s*\*\s*const\s*\*', '*const *const u8'), (r'const void\s*\*\s*const\s*\*', '*const *const c_void'),
        (r'void\s*\*\s*const\s*\*', '*const *mut c_void'),
        (r'const uint8_t\s*\*', '*const u8'), (r'uint8_t\s*\*', '*mut u8'), (r'const int32_t\s*\*', '*const i32'), (r'int32_t\s*\*', '*mut i32'),
        (r'const uint64_t\s*\*', '*const u64'), (r'uint64_t\s*\*', '*mut u64'), (r'const size_t\s*\*', '*const usize'), (r'size_t\s*\*', '*mut usize'),
        (r'const void\s*\*', '*const c_void'), (r'void\s*\*', '*mut c_void'), (r'const char\s*\*\s*\*', '*mut *const c_char'),
        (r'const char\s*\*', '*const c_char'), (r'char\s*\*', '*mut c_char'), (r'double\s*\*', '*mut f64'), (r'int64_t\s*\*', '*mut i64'), (r'const int\s*\*', '*const c_int'),
        (r'bppp_ctx\s*\*\s*\*', '*mut *mut BpppCtx'), (r'const bppp_ctx\s*\*', '*const BpppCtx'), (r'bppp_ctx\s*\*', '*mut BpppCtx'),
        (r'bppp_circuit\s*\*\s*\*', '*mut *mut BpppCircuit'), (r'const bppp_circuit\s*\*', '*const BpppCircuit'), (r'bppp_circuit\s*\*', '*mut BpppCircuit'),
        (r'bppp_group\s*\*\s*\*', '*mut *mut BpppGroup'), (r'const bppp_group\s*\*', '*const BpppGroup'), (r'bppp_group\s*\*', '*mut BpppGroup'),
        (r'size_t', 'usize'), (r'uint64_t', 'u64'), (r'int32_t', 'i32'), (r'int', 'c_int'), (r'long', 'c_long')]
RET = {'int': ' -> c_int', 'long': ' -> c_long', 'void': '', 'size_t': ' -> usize', 'const char*': ' -> *const c_char', 'const char *': ' -> *const c_char',
       'bppp_ctx*': ' -> *mut BpppCtx', 'bppp_ctx *': ' -> *mut BpppCtx'}
HEAD = '''//! `extern "C"` declarations of EVERY entry point of include/bppp.h (libbppp_hip.so).  Generated from the header by
//! tools/gen_facade_ffi.py and checked against it by tests/test_facade_tree.py.  UNCOMPILED (no Rust toolchain in the build
//! image); the identical prototypes are exercised through ctypes in bp_pp_amd/_capi.py.
#![allow(dead_code)]
use std::os::raw::{c_char, c_int, c_long, c_void};

#[repr(C)] pub struct BpppCtx { _private: [u8; 0] }
#[repr(C)] pub struct BpppCircuit { _private: [u8; 0] }
#[repr(C)] pub struct BpppGroup { _private: [u8; 0] }

pub const BPPP_OK: c_int = 0;
pub const BPPP_ERR_NO_DEVICE: c_int = -1;
pub const BPPP_ERR_INVALID_ARG: c_int = -2;
pub const BPPP_ERR_HIP: c_int = -3;
pub const BPPP_ERR_ENCODING: c_int = -4;
pub const BPPP_ERR_NOMEM: c_int = -5;
pub const BPPP_ERR_RCCL: c_int = -6;
pub const BPPP_ST_BAD_ENCODING: i32 = 1;
pub const BPPP_ST_DEGENERATE: i32 = 2;
pub const POINT_BYTES: usize = 64;
pub const SCALAR_BYTES: usize = 32;
pub const U64_PROOF_BYTES: usize = 928;
pub const U64_PROOF_SEC1_BYTES: usize = 525;
pub const TRANSCRIPT_STATE_BYTES: usize = 203;

extern "C" {
'''


def conv_type(t):
    t = t.strip()
    for pat, rep in TMAP:
        if re.fullmatch(pat, t):
            return rep
    raise SystemExit('unmapped C type: ' + repr(t))


def generate():
    h = open(os.path.join(ROOT, 'include', 'bppp.h')).read()
    hc = re.sub(r'/\*.*?\*/', '', h, flags=re.S)
    hc = '\n'.join(l for l in hc.split('\n') if not l.lstrip().startswith('#'))
    out, names = [], []
    for d in re.findall(r'BPPP_API\s+(.*?);', hc, flags=re.S):
        d = ' '.join(d.split())
        m = re.match(r'(.*?)\b(bppp_\w+)\s*\((.*)\)$', d)
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        params = []
        if args and args != 'void':
            for a in args.split(','):
                a = a.strip()
                arr = re.match(r'(.*?)(\w+)\[(\d*)\]$', a)      # `const uint8_t g[64]` is a pointer parameter
                if arr:
                    ty, nm = arr.group(1).strip() + ' *', arr.group(2)
                else:
                    mm = re.match(r'(.*?)(\w+)$', a)
                    ty, nm = mm.group(1), mm.group(2)
                if nm in ('type', 'ref', 'in', 'fn', 'mod', 'use', 'move', 'box'):
                    nm += '_'
                params.append(f'{nm}: {conv_type(ty)}')
        out.append(f'    pub fn {name}(' + ', '.join(params) + f'){RET[ret]};')
        names.append(name)
    return HEAD + '\n'.join(out) + '\n}\n', names


if __name__ == '__main__':
    src, names = generate()
    path = os.path.join(ROOT, 'facade', 'src', 'ffi.rs')
    if '--check' in sys.argv:
        sys.exit(0 if open(path).read() == src else 1)
    open(path, 'w').write(src)
    print(len(names), 'externs ->', path)
