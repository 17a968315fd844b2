// Single-translation-unit build of libbppp_hip.so for diagnostic variants that need a device-global shared by the host
// side and the kernels (-DBPPP_PHASE_TIMING: the phase-stamp buffer read by tools/probes/phase_probe.py).  The product build
// compiles the units separately and in parallel (bp_pp_amd/_build.py); this file only includes them.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=hidden -DBPPP_PHASE_TIMING \
//         -o bp_pp_amd/libbppp_hip_pt.so bp_pp_amd/csrc/unity/bppp_unity.hip
#include "../bppp_ctx.hip"
#include "../bppp_u64.hip"
#include "../bppp_generic.hip"
#include "../bppp_group.hip"
#include "../bppp_coalesce.hip"
#include "../k_verify_misc.hip"
#include "../k_verify_var.hip"
#include "../k_verify_fixed.hip"
#include "../k_verify_bucket.hip"
#include "../k_prove.hip"
#include "../k_prove_w2.hip"
#include "../k_generic.hip"
#include "../k_gprove.hip"
