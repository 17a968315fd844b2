// libbppp_hip.so, host side: one batch over the GPUs of a node (bppp_group_*): one context, stream and host thread per device, RCCL through dlopen.
#include "host.h"

#include <dlfcn.h>

#include <functional>
#include <thread>

#include "group_core.h"

extern "C" {

// ---------------------------------------------------------------- one batch over the GPUs of a node
// RCCL through dlopen: no link-time dependency, and whichever librccl the process already holds (e.g. torch's) serves.
namespace {
typedef void* rcclComm;
struct RcclApi {
    void* handle = nullptr;
    int (*CommInitAll)(rcclComm*, int, const int*) = nullptr;
    int (*CommDestroy)(rcclComm) = nullptr;
    int (*CommAbort)(rcclComm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, rcclComm, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
const int kRcclInt32 = 2, kRcclSum = 0;   // ncclInt32, ncclSum (rccl.h)
bool rccl_load(RcclApi& a) {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        a.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (a.handle) break;
    }
    if (!a.handle) { g_last_error = std::string("dlopen(librccl): ") + (dlerror() ? dlerror() : "not found"); return false; }
    a.CommInitAll = (int (*)(rcclComm*, int, const int*))dlsym(a.handle, "ncclCommInitAll");
    a.CommDestroy = (int (*)(rcclComm))dlsym(a.handle, "ncclCommDestroy");
    a.CommAbort = (int (*)(rcclComm))dlsym(a.handle, "ncclCommAbort");
    a.AllReduce = (int (*)(const void*, void*, size_t, int, int, rcclComm, hipStream_t))dlsym(a.handle, "ncclAllReduce");
    a.GroupStart = (int (*)())dlsym(a.handle, "ncclGroupStart");
    a.GroupEnd = (int (*)())dlsym(a.handle, "ncclGroupEnd");
    a.GetErrorString = (const char* (*)(int))dlsym(a.handle, "ncclGetErrorString");
    if (!a.CommInitAll || !a.CommDestroy || !a.AllReduce || !a.GroupStart || !a.GroupEnd) {
        g_last_error = "librccl lacks a required symbol";
        return false;
    }
    return true;
}
}  // namespace

struct bppp_group {
    std::mutex mu;                       // one sharded call at a time: two callers interleaving their collectives on different ranks
                                         // in different orders would deadlock (the contexts' own locks do not order them)
    std::vector<int> devices;
    std::vector<bppp_ctx*> ctx;
    std::vector<rcclComm> comm;          // empty when the accept-reduce needs no collective (one device)
    std::vector<int*> d_rej;             // per device: int32 reject counter of the host-buffer entry points
    RcclApi rccl;
    bool broken = false;                 // a collective failed and the communicators were aborted: sharded calls return BPPP_ERR_RCCL
    int fault_rank = -1;                 // testing aid (bppp_group_set_option "inject_fault_rank"): that rank's next prepare fails
};

void bppp_shard_range(size_t n_total, int rank, int world, size_t* lo, size_t* hi) {
    if (world <= 0 || rank < 0 || rank >= world) { if (lo) *lo = 0; if (hi) *hi = 0; return; }
    // n_total * rank may exceed 64 bits only for absurd sizes; use the quotient / remainder form
    const size_t q = n_total / (size_t)world, r = n_total % (size_t)world;
    auto at = [&](size_t k) { return q * k + (r * k) / (size_t)world; };   // floor(n_total * k / world)
    if (lo) *lo = at((size_t)rank);
    if (hi) *hi = at((size_t)rank + 1);
}

int bppp_group_size(const bppp_group* grp) { return grp ? (int)grp->devices.size() : 0; }
bppp_ctx* bppp_group_ctx(bppp_group* grp, int rank) { return (grp && rank >= 0 && rank < (int)grp->ctx.size()) ? grp->ctx[rank] : nullptr; }

void bppp_group_destroy(bppp_group* grp) {
    if (!grp) return;
    for (size_t r = 0; r < grp->comm.size(); r++)
        if (grp->comm[r]) { (void)hipSetDevice(grp->devices[r]); (void)grp->rccl.CommDestroy(grp->comm[r]); }
    for (size_t r = 0; r < grp->ctx.size(); r++) {
        if (r < grp->d_rej.size() && grp->d_rej[r]) { (void)hipSetDevice(grp->devices[r]); (void)hipFree(grp->d_rej[r]); }
        bppp_ctx_destroy(grp->ctx[r]);
    }
    delete grp;
}

int bppp_group_set_option(bppp_group* grp, const char* name, long value) {
    if (!grp || !name) return BPPP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(grp->mu);
    if (std::strcmp(name, "inject_fault_rank") == 0) {
        if (value < -1 || value >= (long)grp->devices.size()) return BPPP_ERR_INVALID_ARG;
        grp->fault_rank = (int)value;
        return BPPP_OK;
    }
    // every other option goes to each rank's context (bppp_ctx_set_option: rlc_superchunk, max_batch, host_chunk)
    for (bppp_ctx* c : grp->ctx) {
        int rc = bppp_ctx_set_option(c, name, value);
        if (rc != BPPP_OK) return rc;
    }
    return BPPP_OK;
}

int bppp_wnla_group_create(bppp_group** out, const uint8_t g[64], const uint8_t* g_vec, size_t ng, const uint8_t* h_vec, size_t nh,
                           const int* devices, int n_devices, int fb_window_bits) {
    if (!out || !g || (!g_vec && ng) || (!h_vec && nh) || !devices || n_devices <= 0 || n_devices > 64) return BPPP_ERR_INVALID_ARG;
    *out = nullptr;
    for (int i = 0; i < n_devices; i++)
        for (int j = 0; j < i; j++)
            if (devices[i] == devices[j]) return BPPP_ERR_INVALID_ARG;
    bppp_group* grp = new (std::nothrow) bppp_group();
    if (!grp) return BPPP_ERR_NOMEM;
    grp->devices.assign(devices, devices + n_devices);
    grp->ctx.assign(n_devices, nullptr);
    grp->d_rej.assign(n_devices, nullptr);
    // contexts (fixed-base tables) are built concurrently, one host thread per device
    std::vector<int> rcs(n_devices, BPPP_OK);
    std::vector<std::string> errs(n_devices);
    {
        std::vector<std::thread> th;
        for (int r = 0; r < n_devices; r++)
            th.emplace_back([&, r]() {
                rcs[r] = bppp_wnla_ctx_create(&grp->ctx[r], g, g_vec, ng, h_vec, nh, devices[r], fb_window_bits);
                if (rcs[r] == BPPP_OK && hipMalloc(&grp->d_rej[r], sizeof(int)) != hipSuccess) rcs[r] = BPPP_ERR_NOMEM;
                if (rcs[r] != BPPP_OK) errs[r] = g_last_error;
            });
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < n_devices; r++)
        if (rcs[r] != BPPP_OK) { g_last_error = errs[r]; int rc = rcs[r]; bppp_group_destroy(grp); return rc; }
    if (n_devices > 1 || std::getenv("BPPP_FORCE_RCCL")) {
        if (!rccl_load(grp->rccl)) { bppp_group_destroy(grp); return BPPP_ERR_RCCL; }
        grp->comm.assign(n_devices, nullptr);
        const int e = grp->rccl.CommInitAll(grp->comm.data(), n_devices, grp->devices.data());
        if (e != 0) {
            g_last_error = std::string("ncclCommInitAll: ") + (grp->rccl.GetErrorString ? grp->rccl.GetErrorString(e) : "failed");
            grp->comm.clear();
            bppp_group_destroy(grp);
            return BPPP_ERR_RCCL;
        }
    }
    *out = grp;
    return BPPP_OK;
}
int bppp_group_create(bppp_group** out, const uint8_t g[64], const uint8_t* g_vec, const uint8_t* h_vec, const int* devices, int n_devices,
                      int fb_window_bits) {
    if (!g_vec || !h_vec) return BPPP_ERR_INVALID_ARG;
    return bppp_wnla_group_create(out, g, g_vec, 16, h_vec, 32, devices, n_devices, fb_window_bits);
}

}  // extern "C"

// ---------------------------------------------------------------- the sharded call
namespace {
// One array of a sharded call: per proof `stride` bytes.  Host form: `host` is the whole batch's array (input when !is_output),
// rank r stages rows [lo, hi) in its context's I/O buffer; `shared_rows` > 0 marks an input every rank receives whole (one
// transcript shared by the batch).  Device form: dev[r] is rank r's resident shard.
struct ShardArray {
    const void* host_in = nullptr;
    void* host_out = nullptr;
    size_t stride = 0;
    size_t shared_rows = 0;
    bool scratch = false;                 // host form: stage it even when the caller passed no host array (a required device buffer)
    const void* const* dev = nullptr;     // device form: per-rank pointers (entries may be null where the form allows it)
};
struct ShardedCall {
    bppp_group* grp;
    size_t n;
    std::vector<ShardArray> arrays;
    bool device_form = false;
    void* const* d_reject_count = nullptr;   // device form: per-rank int32[1], receives the global count
    int32_t* reject_count = nullptr;         // host form (optional)
    bool exchange = true;                    // false: a call with no exchange step (prove) -- no reject counter, no collective
    // enqueue rank r's shard: m proofs, p[i] = device pointer of array i, d_rej = device reject counter
    std::function<int(int r, bppp_ctx* c, size_t m, void* const* p, void* d_rej)> enqueue;
};

int run_call(ShardedCall& call) {
    bppp_group* grp = call.grp;
    std::lock_guard<std::mutex> group_lock(grp->mu);
    if (grp->broken) { g_last_error = "group unusable: an earlier collective failed and its communicators were aborted"; return BPPP_ERR_RCCL; }
    const int G = (int)grp->devices.size();
    const size_t NA = call.arrays.size();
    std::vector<int> counts(G, 0);
    std::vector<std::vector<void*>> ptrs(G, std::vector<void*>(NA, nullptr));
    std::vector<void*> rej(G, nullptr);
    const int fault_rank = grp->fault_rank;
    grp->fault_rank = -1;
    auto range = [&](int r, size_t& lo, size_t& m) { size_t hi; bppp_shard_range(call.n, r, G, &lo, &hi); m = hi - lo; };
    // every rank thread holds its context's lock from prepare to finish: the staging buffer and the stream belong to this call
    auto prepare = [&](int r) -> int {
        bppp_ctx* c = grp->ctx[r];
        size_t lo, m;
        range(r, lo, m);
        HIP_TRY(hipSetDevice(grp->devices[r]));
        if (r == fault_rank) { g_last_error = "injected fault (bppp_group_set_option inject_fault_rank)"; return BPPP_ERR_NOMEM; }
        if (call.device_form) {
            for (size_t i = 0; i < NA; i++) ptrs[r][i] = const_cast<void*>(call.arrays[i].dev ? call.arrays[i].dev[r] : nullptr);
            rej[r] = call.exchange ? call.d_reject_count[r] : nullptr;
        } else {
            // the shard goes through the context's persistent I/O staging, exactly as the single-device host entry points do
            size_t off = 0;
            std::vector<size_t> offs(NA);
            for (size_t i = 0; i < NA; i++) {
                const ShardArray& a = call.arrays[i];
                offs[i] = off;
                off = align16(off + (a.shared_rows ? a.shared_rows : m) * a.stride);
            }
            const size_t need = off + 16;
            {
                const int rc_io = ensure_io(c, need);
                if (rc_io != BPPP_OK) return rc_io;
            }
            for (size_t i = 0; i < NA; i++) {
                const ShardArray& a = call.arrays[i];
                if (!a.host_in && !a.host_out && !a.scratch) continue;   // an optional array the caller did not pass
                ptrs[r][i] = c->d_io + offs[i];
                if (!a.host_in) continue;
                const size_t rows = a.shared_rows ? a.shared_rows : m, first = a.shared_rows ? 0 : lo;
                if (rows) HIP_TRY(hipMemcpyAsync(ptrs[r][i], (const uint8_t*)a.host_in + first * a.stride, rows * a.stride, hipMemcpyHostToDevice, c->stream));
            }
            rej[r] = call.exchange ? grp->d_rej[r] : nullptr;
        }
        if (m == 0) {
            if (rej[r]) HIP_TRY(hipMemsetAsync(rej[r], 0, sizeof(int), c->stream));
            return BPPP_OK;
        }
        return call.enqueue(r, c, m, ptrs[r].data(), rej[r]);
    };
    auto collective = [&](int r) -> int {
        if (grp->comm.empty() || !call.exchange) return BPPP_OK;
        const int e = grp->rccl.AllReduce(rej[r], rej[r], 1, kRcclInt32, kRcclSum, grp->comm[r], grp->ctx[r]->stream);
        if (e != 0) {
            g_last_error = std::string("ncclAllReduce: ") + (grp->rccl.GetErrorString ? grp->rccl.GetErrorString(e) : "failed");
            return BPPP_ERR_RCCL;
        }
        return BPPP_OK;
    };
    auto drain = [&](int r) -> int {
        bppp_ctx* c = grp->ctx[r];
        (void)hipSetDevice(grp->devices[r]);
        const hipError_t e1 = hipStreamSynchronize(c->stream), e2 = hipStreamSynchronize(c->aux_stream);     // both, whatever the first says
        if (e1 != hipSuccess || e2 != hipSuccess) {
            g_last_error = std::string("draining rank ") + std::to_string(r) + ": " + hipGetErrorString(e1 != hipSuccess ? e1 : e2);
            (void)hipGetLastError();
            return BPPP_ERR_HIP;
        }
        return BPPP_OK;
    };
    auto finish = [&](int r) -> int {
        bppp_ctx* c = grp->ctx[r];
        size_t lo, m;
        range(r, lo, m);
        if (!call.device_form) {
            for (size_t i = 0; i < NA && m; i++) {
                const ShardArray& a = call.arrays[i];
                if (a.host_out) HIP_TRY(hipMemcpyAsync((uint8_t*)a.host_out + lo * a.stride, ptrs[r][i], m * a.stride, hipMemcpyDeviceToHost, c->stream));
            }
            if (rej[r]) HIP_TRY(hipMemcpyAsync(&counts[r], rej[r], sizeof(int), hipMemcpyDeviceToHost, c->stream));
        }
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipStreamSynchronize(c->aux_stream));
        return BPPP_OK;
    };
    auto abort_comm = [&](int r) {
        if (!grp->comm.empty() && grp->comm[r] && grp->rccl.CommAbort) { (void)grp->rccl.CommAbort(grp->comm[r]); grp->comm[r] = nullptr; }
    };
    // a group of one device runs on the CALLER's thread: its current device is put back afterwards
    int caller_device = -1;
    if (G == 1 && hipGetDevice(&caller_device) != hipSuccess) { caller_device = -1; (void)hipGetLastError(); }
    bppp_host::ShardedResult res = bppp_host::run_sharded(
        G, [&](int r) { grp->ctx[r]->mu.lock(); }, prepare, collective, finish, drain, abort_comm, [&](int r) { grp->ctx[r]->mu.unlock(); },
        []() { return g_last_error; });
    if (caller_device >= 0) (void)hipSetDevice(caller_device);
    if (res.collective_failed) grp->broken = true;
    if (res.code != BPPP_OK) { g_last_error = res.error; return res.code; }
    if (call.reject_count) {
        // with a communicator every device already holds the global count; without one (a single device) it is the local one
        int total = 0;
        if (!grp->comm.empty()) total = counts[0];
        else for (int r = 0; r < G; r++) total += counts[r];
        *call.reject_count = total;
    }
    return BPPP_OK;
}
}  // namespace

namespace {
ShardArray host_in(const void* p, size_t stride, size_t shared_rows = 0) { ShardArray a; a.host_in = p; a.stride = stride; a.shared_rows = shared_rows; return a; }
ShardArray host_out(void* p, size_t stride, bool scratch = false) { ShardArray a; a.host_out = p; a.stride = stride; a.scratch = scratch; return a; }
ShardArray dev_arr(const void* const* p, size_t stride) { ShardArray a; a.dev = p; a.stride = stride; return a; }
// device form: d_reject_count[r] everywhere, and the required arrays wherever the rank's shard is not empty
int check_device_form(const bppp_group* grp, size_t n, void* const* d_reject_count, std::initializer_list<const void* const*> required) {
    if (!d_reject_count) return BPPP_ERR_INVALID_ARG;
    const int G = (int)grp->devices.size();
    for (const void* const* a : required)
        if (!a) return BPPP_ERR_INVALID_ARG;
    for (int r = 0; r < G; r++) {
        size_t lo, hi;
        bppp_shard_range(n, r, G, &lo, &hi);
        if (!d_reject_count[r]) return BPPP_ERR_INVALID_ARG;
        if (hi > lo)
            for (const void* const* a : required)
                if (!a[r]) return BPPP_ERR_INVALID_ARG;
    }
    return BPPP_OK;
}
const void* const* cv(void* const* p) { return (const void* const*)p; }
bool u64_shape(const bppp_group* grp) { return grp->ctx[0]->ng == 16 && grp->ctx[0]->nh == 32; }
}  // namespace

extern "C" {

// ---- U64RangeProofProtocol::verify, exact and RLC, host and device buffers
static int u64_sharded_host(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments, const uint8_t* proofs,
                            uint8_t* accept, int32_t* status, int32_t* reject_count, const uint8_t* seed) {
    if (!grp || !label_ok(label, label_len) || !commitments || !proofs || !accept || !u64_shape(grp)) return BPPP_ERR_INVALID_ARG;
    if (reject_count) *reject_count = 0;
    if (n == 0) return BPPP_OK;
    ShardedCall call;
    call.grp = grp; call.n = n; call.reject_count = reject_count;
    call.arrays = {host_in(commitments, 64), host_in(proofs, BPPP_U64_PROOF_BYTES), host_out(accept, 1), host_out(status, 4)};
    call.enqueue = [=](int, bppp_ctx* c, size_t m, void* const* p, void* d_rej) {
        return verify_device_impl(c, label, label_len, m, p[0], p[1], p[2], p[3], nullptr, d_rej, seed, nullptr);
    };
    return run_call(call);
}
static int u64_sharded_device(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, const void* const* d_commitments,
                              const void* const* d_proofs, void* const* d_accept, void* const* d_status, void* const* d_reject_count,
                              const uint8_t* seed) {
    if (!grp || !label_ok(label, label_len) || !u64_shape(grp)) return BPPP_ERR_INVALID_ARG;
    int rc = check_device_form(grp, n, d_reject_count, {d_commitments, d_proofs, cv(d_accept)});
    if (rc != BPPP_OK) return rc;
    ShardedCall call;
    call.grp = grp; call.n = n; call.device_form = true; call.d_reject_count = d_reject_count;
    call.arrays = {dev_arr(d_commitments, 64), dev_arr(d_proofs, BPPP_U64_PROOF_BYTES), dev_arr(cv(d_accept), 1), dev_arr(cv(d_status), 4)};
    call.enqueue = [=](int, bppp_ctx* c, size_t m, void* const* p, void* d_rej) {
        return verify_device_impl(c, label, label_len, m, p[0], p[1], p[2], p[3], nullptr, d_rej, seed, nullptr);
    };
    return run_call(call);
}
int bppp_u64_verify_batch_sharded(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments,
                                  const uint8_t* proofs, uint8_t* accept, int32_t* status, int32_t* reject_count) {
    return u64_sharded_host(grp, label, label_len, n, commitments, proofs, accept, status, reject_count, nullptr);
}
int bppp_u64_verify_batch_rlc_sharded(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments,
                                      const uint8_t* proofs, uint8_t* accept, int32_t* status, int32_t* reject_count, const uint8_t seed[32]) {
    if (!seed) return BPPP_ERR_INVALID_ARG;
    return u64_sharded_host(grp, label, label_len, n, commitments, proofs, accept, status, reject_count, seed);
}
int bppp_u64_verify_batch_sharded_device(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, const void* const* d_commitments,
                                         const void* const* d_proofs, void* const* d_accept, void* const* d_status,
                                         void* const* d_reject_count) {
    return u64_sharded_device(grp, label, label_len, n, d_commitments, d_proofs, d_accept, d_status, d_reject_count, nullptr);
}
int bppp_u64_verify_batch_rlc_sharded_device(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, const void* const* d_commitments,
                                             const void* const* d_proofs, void* const* d_accept, void* const* d_status,
                                             void* const* d_reject_count, const uint8_t seed[32]) {
    if (!seed) return BPPP_ERR_INVALID_ARG;
    return u64_sharded_device(grp, label, label_len, n, d_commitments, d_proofs, d_accept, d_status, d_reject_count, seed);
}

// ---- SEC1-compressed inputs (33-byte commitments, 525-byte proofs), expanded on each device
int bppp_u64_verify_batch_sec1_sharded(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, const uint8_t* commitments33,
                                       const uint8_t* proofs525, uint8_t* accept, int32_t* status, int32_t* reject_count) {
    if (!grp || !label_ok(label, label_len) || !commitments33 || !proofs525 || !accept || !u64_shape(grp)) return BPPP_ERR_INVALID_ARG;
    if (reject_count) *reject_count = 0;
    if (n == 0) return BPPP_OK;
    ShardedCall call;
    call.grp = grp; call.n = n; call.reject_count = reject_count;
    call.arrays = {host_in(commitments33, 33), host_in(proofs525, BPPP_U64_PROOF_SEC1_BYTES), host_out(accept, 1), host_out(status, 4)};
    call.enqueue = [=](int, bppp_ctx* c, size_t m, void* const* p, void* d_rej) {
        return verify_sec1_device_impl(c, label, label_len, m, p[0], p[1], p[2], p[3], nullptr, d_rej);
    };
    return run_call(call);
}
int bppp_u64_verify_batch_sec1_sharded_device(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n,
                                              const void* const* d_commitments33, const void* const* d_proofs525, void* const* d_accept,
                                              void* const* d_status, void* const* d_reject_count) {
    if (!grp || !label_ok(label, label_len) || !u64_shape(grp)) return BPPP_ERR_INVALID_ARG;
    int rc = check_device_form(grp, n, d_reject_count, {d_commitments33, d_proofs525, cv(d_accept)});
    if (rc != BPPP_OK) return rc;
    ShardedCall call;
    call.grp = grp; call.n = n; call.device_form = true; call.d_reject_count = d_reject_count;
    call.arrays = {dev_arr(d_commitments33, 33), dev_arr(d_proofs525, BPPP_U64_PROOF_SEC1_BYTES), dev_arr(cv(d_accept), 1), dev_arr(cv(d_status), 4)};
    call.enqueue = [=](int, bppp_ctx* c, size_t m, void* const* p, void* d_rej) {
        return verify_sec1_device_impl(c, label, label_len, m, p[0], p[1], p[2], p[3], nullptr, d_rej);
    };
    return run_call(call);
}

// ---- the caller's transcripts (`t: &mut Transcript`, u64_proof.rs:42): one state shared by the batch or one per proof
int bppp_u64_verify_batch_transcript_sharded(bppp_group* grp, size_t n, const uint8_t* states, size_t n_states, const uint8_t* commitments,
                                             const uint8_t* proofs, uint8_t* accept, int32_t* status, uint8_t* states_out,
                                             int32_t* reject_count) {
    if (!grp || !states || !commitments || !proofs || !accept || (n_states != 1 && n_states != n) || !u64_shape(grp)) return BPPP_ERR_INVALID_ARG;
    if (reject_count) *reject_count = 0;
    if (n == 0) return BPPP_OK;
    for (size_t i = 0; i < n_states; i++)
        if (states[203 * i + 200] >= BPPP_STROBE_R || states[203 * i + 201] > BPPP_STROBE_R) return BPPP_ERR_INVALID_ARG;
    const size_t SB = BPPP_TRANSCRIPT_STATE_BYTES;
    const bool shared = n_states == 1;
    ShardedCall call;
    call.grp = grp; call.n = n; call.reject_count = reject_count;
    call.arrays = {host_in(commitments, 64), host_in(proofs, BPPP_U64_PROOF_BYTES), host_out(accept, 1), host_out(status, 4),
                   host_in(states, SB, shared ? 1 : 0), host_out(states_out, SB)};
    call.enqueue = [=](int, bppp_ctx* c, size_t m, void* const* p, void* d_rej) {
        VerifyTranscripts tx = {p[4], shared ? (size_t)1 : m, p[5]};
        return verify_device_impl(c, nullptr, 0, m, p[0], p[1], p[2], p[3], nullptr, d_rej, nullptr, &tx);
    };
    return run_call(call);
}
int bppp_u64_verify_batch_transcript_sharded_device(bppp_group* grp, size_t n, const void* const* d_states, size_t n_states,
                                                    const void* const* d_commitments, const void* const* d_proofs, void* const* d_accept,
                                                    void* const* d_status, void* const* d_reject_count, void* const* d_states_out) {
    if (!grp || (n_states != 1 && n_states != n) || !u64_shape(grp)) return BPPP_ERR_INVALID_ARG;
    int rc = check_device_form(grp, n, d_reject_count, {d_states, d_commitments, d_proofs, cv(d_accept)});
    if (rc != BPPP_OK) return rc;
    const bool shared = n_states == 1;
    ShardedCall call;
    call.grp = grp; call.n = n; call.device_form = true; call.d_reject_count = d_reject_count;
    call.arrays = {dev_arr(d_commitments, 64), dev_arr(d_proofs, BPPP_U64_PROOF_BYTES), dev_arr(cv(d_accept), 1), dev_arr(cv(d_status), 4),
                   dev_arr(d_states, BPPP_TRANSCRIPT_STATE_BYTES), dev_arr(cv(d_states_out), BPPP_TRANSCRIPT_STATE_BYTES)};
    call.enqueue = [=](int, bppp_ctx* c, size_t m, void* const* p, void* d_rej) {
        VerifyTranscripts tx = {p[4], shared ? (size_t)1 : m, p[5]};
        return verify_device_impl(c, nullptr, 0, m, p[0], p[1], p[2], p[3], nullptr, d_rej, nullptr, &tx);
    };
    return run_call(call);
}

// ---- ReciprocalRangeProofProtocol::verify (reciprocal.rs:98-107) over a group built by bppp_wnla_group_create: BASELINE
//      configs[4] (2^18 instances of the 256-digit shape over 8 GPUs).  Same contiguous split, same 4-byte reduce.
static int recip_sharded_host(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                              const uint8_t* commitments, const uint8_t* proofs, size_t rounds, size_t nl, size_t nn, uint8_t* accept,
                              int32_t* status, int32_t* reject_count, const uint8_t* seed) {
    if (!grp || !label_ok(label, label_len) || !commitments || !proofs || !accept || rounds > 12) return BPPP_ERR_INVALID_ARG;
    if (reject_count) *reject_count = 0;
    if (n == 0) return BPPP_OK;
    const size_t proof_bytes = 64 * (5 + 2 * rounds) + 32 * (nl + nn);
    ShardedCall call;
    call.grp = grp; call.n = n; call.reject_count = reject_count;
    call.arrays = {host_in(commitments, 64), host_in(proofs, proof_bytes), host_out(accept, 1), host_out(status, 4, /*scratch=*/true)};
    call.enqueue = [=](int, bppp_ctx* c, size_t m, void* const* p, void* d_rej) {
        return recip_verify_device_entry(c, label, label_len, m, dim_nd, dim_np, p[0], p[1], rounds, nl, nn, p[2], p[3], seed, d_rej);
    };
    return run_call(call);
}
static int recip_sharded_device(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                                const void* const* d_commitments, const void* const* d_proofs, size_t rounds, size_t nl, size_t nn,
                                void* const* d_accept, void* const* d_status, void* const* d_reject_count, const uint8_t* seed) {
    if (!grp || !label_ok(label, label_len) || rounds > 12) return BPPP_ERR_INVALID_ARG;
    int rc = check_device_form(grp, n, d_reject_count, {d_commitments, d_proofs, cv(d_accept), cv(d_status)});
    if (rc != BPPP_OK) return rc;
    const size_t proof_bytes = 64 * (5 + 2 * rounds) + 32 * (nl + nn);
    ShardedCall call;
    call.grp = grp; call.n = n; call.device_form = true; call.d_reject_count = d_reject_count;
    call.arrays = {dev_arr(d_commitments, 64), dev_arr(d_proofs, proof_bytes), dev_arr(cv(d_accept), 1), dev_arr(cv(d_status), 4)};
    call.enqueue = [=](int, bppp_ctx* c, size_t m, void* const* p, void* d_rej) {
        return recip_verify_device_entry(c, label, label_len, m, dim_nd, dim_np, p[0], p[1], rounds, nl, nn, p[2], p[3], seed, d_rej);
    };
    return run_call(call);
}
int bppp_reciprocal_verify_batch_sharded(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                                         const uint8_t* commitments, const uint8_t* proofs, size_t rounds, size_t nl, size_t nn, uint8_t* accept,
                                         int32_t* status, int32_t* reject_count) {
    return recip_sharded_host(grp, label, label_len, n, dim_nd, dim_np, commitments, proofs, rounds, nl, nn, accept, status, reject_count, nullptr);
}
int bppp_reciprocal_verify_batch_rlc_sharded(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                                             const uint8_t* commitments, const uint8_t* proofs, size_t rounds, size_t nl, size_t nn,
                                             uint8_t* accept, int32_t* status, int32_t* reject_count, const uint8_t seed[32]) {
    if (!seed) return BPPP_ERR_INVALID_ARG;
    return recip_sharded_host(grp, label, label_len, n, dim_nd, dim_np, commitments, proofs, rounds, nl, nn, accept, status, reject_count, seed);
}
int bppp_reciprocal_verify_batch_sharded_device(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd, size_t dim_np,
                                                const void* const* d_commitments, const void* const* d_proofs, size_t rounds, size_t nl,
                                                size_t nn, void* const* d_accept, void* const* d_status, void* const* d_reject_count) {
    return recip_sharded_device(grp, label, label_len, n, dim_nd, dim_np, d_commitments, d_proofs, rounds, nl, nn, d_accept, d_status,
                                d_reject_count, nullptr);
}
int bppp_reciprocal_verify_batch_rlc_sharded_device(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, size_t dim_nd,
                                                    size_t dim_np, const void* const* d_commitments, const void* const* d_proofs, size_t rounds,
                                                    size_t nl, size_t nn, void* const* d_accept, void* const* d_status,
                                                    void* const* d_reject_count, const uint8_t seed[32]) {
    if (!seed) return BPPP_ERR_INVALID_ARG;
    return recip_sharded_device(grp, label, label_len, n, dim_nd, dim_np, d_commitments, d_proofs, rounds, nl, nn, d_accept, d_status,
                                d_reject_count, seed);
}

// ---- U64RangeProofProtocol::prove, sharded: independent proofs, no exchange step -- the ranks only vote on their return codes
int bppp_u64_prove_batch_sharded(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, const uint64_t* x, const uint8_t* s,
                                 const uint8_t* rnd, uint8_t* proofs, uint8_t* commitments, int32_t* status) {
    if (!grp || !label_ok(label, label_len) || !x || !s || !rnd || !proofs || !commitments || !u64_shape(grp)) return BPPP_ERR_INVALID_ARG;
    if (n == 0) return BPPP_OK;
    ShardedCall call;
    call.grp = grp; call.n = n; call.exchange = false;
    call.arrays = {host_in(x, 8), host_in(s, 32), host_in(rnd, 52 * 32), host_out(proofs, BPPP_U64_PROOF_BYTES), host_out(commitments, 64),
                   host_out(status, 4)};
    call.enqueue = [=](int, bppp_ctx* c, size_t m, void* const* p, void*) {
        return prove_device_impl(c, label, label_len, m, p[0], p[1], p[2], p[3], p[4], p[5], nullptr);
    };
    return run_call(call);
}
int bppp_u64_prove_batch_sharded_device(bppp_group* grp, const uint8_t* label, size_t label_len, size_t n, const void* const* d_x,
                                        const void* const* d_s, const void* const* d_rnd, void* const* d_proofs, void* const* d_commitments,
                                        void* const* d_status) {
    if (!grp || !label_ok(label, label_len) || !u64_shape(grp)) return BPPP_ERR_INVALID_ARG;
    const int G = (int)grp->devices.size();
    for (const void* const* a : {d_x, d_s, d_rnd, cv(d_proofs), cv(d_commitments)}) {
        if (!a) return BPPP_ERR_INVALID_ARG;
        for (int r = 0; r < G; r++) {
            size_t lo, hi;
            bppp_shard_range(n, r, G, &lo, &hi);
            if (hi > lo && !a[r]) return BPPP_ERR_INVALID_ARG;
        }
    }
    ShardedCall call;
    call.grp = grp; call.n = n; call.device_form = true; call.exchange = false;
    call.arrays = {dev_arr(d_x, 8), dev_arr(d_s, 32), dev_arr(d_rnd, 52 * 32), dev_arr(cv(d_proofs), BPPP_U64_PROOF_BYTES),
                   dev_arr(cv(d_commitments), 64), dev_arr(cv(d_status), 4)};
    call.enqueue = [=](int, bppp_ctx* c, size_t m, void* const* p, void*) {
        return prove_device_impl(c, label, label_len, m, p[0], p[1], p[2], p[3], p[4], p[5], nullptr);
    };
    return run_call(call);
}

}  // extern "C"
