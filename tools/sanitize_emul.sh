#!/bin/bash
# AddressSanitizer + UndefinedBehaviorSanitizer over the device code compiled for the host (tests/emul): every __host__ __device__
# function of bp_pp_amd/csrc/*.h runs under the sanitizers through the emulation tests.  GPU sanitizers are not available on this
# pool, so this is where out-of-bounds workspace indexing, misaligned accesses, signed overflow and bad shifts would show.
# usage: tools/sanitize_emul.sh [pytest args]     (default: all emulation test files)
set -e
cd "$(dirname "$0")/.."
export BPPP_EMUL_SANITIZE=1
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export LD_PRELOAD="$(g++ -print-file-name=libasan.so):$(g++ -print-file-name=libubsan.so)"
if [ $# -eq 0 ]; then set -- tests/test_coalesce_emul.py tests/test_core_emul.py tests/test_wnla_emul.py tests/test_recip_emul.py tests/test_circuit_emul.py tests/test_rlc_emul.py tests/test_transcript_state.py tests/test_multi_rank.py tests/test_ct_trace.py tests/test_group_emul.py; fi
exec python -m pytest -x -q -m "not gpu" "$@"
