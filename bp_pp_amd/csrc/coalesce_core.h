// Request coalescing for the reference's own calling pattern -- ONE proof per call, many callers (u64_proof.rs:42, :57; the types are
// Send + Sync, so N threads call verify / prove concurrently) -- as host-side control flow with no HIP in it.
//
// A call of one proof costs the GPU a full dependent chain (2.3 ms for a verify) however empty the chip is, and calls from different
// host threads queue behind each other on the hardware queues: 64 threads of single-proof calls got 601 verifies/s in round 3, below
// the CPU.  The same 64 proofs in ONE batch cost 2.2 ms.  So the single-proof entry points do not launch anything themselves:
//
//   submit()     a caller claims the next slot of the OPEN batch with one atomic add on a ticket word (no lock, no system call: a
//                thousand callers waking at once from the previous batch must not convoy on a mutex -- on a box whose cgroup grants 16
//                CPUs that convoy alone exhausted the CPU quota and stalled everything for the rest of the 100 ms period), copies its
//                request into that slot of the staging arrays, and sleeps on one of the batch's completion words (futexes);
//   dispatcher   `lanes` threads, each owning one GPU context.  An open batch is sealed when it is full or `wait_us` after its first
//                request; a dispatcher runs the sealed batch as ONE batched call (backend.run), publishes the return code and wakes the
//                batch's callers, each of which copies its own outputs from its slot.  While a batch runs, the next one fills: the
//                batch size follows the load by itself, and up to `lanes` batches overlap on the GPU (a batch of a few hundred proofs
//                leaves most SIMDs idle).
//
// Every request gets exactly the result the batched entry point gives that row: the rows of a batch are independent (own transcript
// state, own status), so a malformed request flags itself and nobody else; only a failure of the batched call as a whole (e.g. out of
// memory) is shared by the batch's callers, as its return code.
//
// shutdown() drains: requests already submitted complete normally, later submissions return `closed_code`, and it returns only when
// no caller is left inside submit() -- which is what lets bppp_ctx_destroy be called while other threads still sit in a *_one call.
//
// The same template drives libbppp_hip.so (bppp_coalesce.hip: pinned staging, HIP streams, the device verify / prove sequences) and
// the CPU tier's emulated front end (tests/emul: malloc staging, the device code compiled for the host), which is how the ring, the
// deadline and the shutdown are tested without a GPU.
#pragma once
#include <atomic>
#include <chrono>
#include <climits>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include <linux/futex.h>
#include <sys/syscall.h>
#include <unistd.h>

namespace bppp_host {

static const int COALESCE_MAX_ARRAYS = 4;
struct CoalesceShape {
    int n_in = 0, n_out = 0;
    size_t in_stride[COALESCE_MAX_ARRAYS] = {0}, out_stride[COALESCE_MAX_ARRAYS] = {0};   // bytes per request in each staging array
};
struct CoalesceStats {
    uint64_t requests = 0, batches = 0, largest_batch = 0, sealed_full = 0, sealed_deadline = 0;
    uint64_t run_us = 0, fill_wait_us = 0;      // dispatcher time inside the batched calls / waiting for the last caller's row copy
};

static inline void futex_wait_u32(std::atomic<uint32_t>* w, uint32_t seen) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0);
}
static inline void futex_wake_all_u32(std::atomic<uint32_t>* w) {
    (void)syscall(SYS_futex, reinterpret_cast<uint32_t*>(w), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
}

// Backend:  void* alloc_staging(size_t bytes);  void free_staging(void* p);      host memory the batched call reads / writes
//           int   run(int lane, size_t n, uint8_t* const in[], uint8_t* const out[]);   ONE batched call over rows 0 .. n-1, synchronous
//           bool  start_lane(int lane);  void stop_lane(int lane);                bracket a dispatcher thread's life (device binding)
//           std::string last_error();  void set_last_error(const std::string&);   the calling thread's error text: a failed batched
//                                                                                 call's message travels from the dispatcher to its callers
template <class Backend>
class Coalescer {
public:
    Coalescer(Backend* be, const CoalesceShape& shape, size_t max_batch, long wait_us, int lanes, int closed_code, int nomem_code)
        : be_(be), shape_(shape), max_(max_batch < 1 ? 1 : (max_batch > 0x7fffffffu ? 0x7fffffffu : max_batch)), wait_us_(wait_us < 0 ? 0 : wait_us),
          lanes_(lanes < 1 ? 1 : lanes), closed_code_(closed_code), nomem_code_(nomem_code) {}
    ~Coalescer() { shutdown(); }
    Coalescer(const Coalescer&) = delete;
    Coalescer& operator=(const Coalescer&) = delete;

    // staging + dispatcher threads; 0 or nomem_code  (called once by the owner, before any submit)
    int start() {
        if (started_.load()) return 0;
        const int nb = lanes_ + 2;          // `lanes` running, one filling, one being read out by its callers
        try {
            for (int i = 0; i < nb; i++) batches_.emplace_back(new Batch());
        } catch (...) { free_all(); return nomem_code_; }
        for (auto& bp : batches_) {
            Batch& b = *bp;
            for (int k = 0; k < shape_.n_in; k++)
                if (!(b.in[k] = (uint8_t*)be_->alloc_staging(shape_.in_stride[k] * max_))) { free_all(); return nomem_code_; }
            for (int k = 0; k < shape_.n_out; k++)
                if (!(b.out[k] = (uint8_t*)be_->alloc_staging(shape_.out_stride[k] * max_))) { free_all(); return nomem_code_; }
        }
        {
            std::lock_guard<std::mutex> lk(mu_);
            open_locked(0);
        }
        try {
            for (int l = 0; l < lanes_; l++) threads_.emplace_back([this, l] { dispatcher(l); });
        } catch (...) {               // could not spawn every dispatcher: stop the ones that run (nothing has been submitted)
            { std::lock_guard<std::mutex> lk(mu_); stopping_.store(true); ticket_.store(NO_BATCH); }
            cv_disp_.notify_all();
            for (auto& t : threads_)
                if (t.joinable()) t.join();
            threads_.clear();
            free_all();
            stopping_.store(false);
            return nomem_code_;
        }
        started_.store(true, std::memory_order_release);
        return 0;
    }

    // One request: in[k] points to shape.in_stride[k] bytes, out[k] receives shape.out_stride[k] bytes (an out pointer may be null: that
    // output is dropped).  Blocks until the request's batch has run; returns the batched call's code (0 = outputs written).
    int submit(const void* const in[], void* const out[]) {
        if (!started_.load(std::memory_order_acquire)) return closed_code_;
        callers_inside_.fetch_add(1, std::memory_order_acq_rel);       // from here on shutdown() waits for this caller
        Batch* b = nullptr;
        uint32_t slot = 0;
        int bidx = -1;
        for (;;) {
            if (stopping_.load(std::memory_order_acquire)) { callers_inside_.fetch_sub(1, std::memory_order_release); return closed_code_; }
            uint64_t t = ticket_.load(std::memory_order_acquire);
            if (!is_none(t) && (uint32_t)t < max_) {
                t = ticket_.fetch_add(1, std::memory_order_acq_rel);
                if (!is_none(t) && (uint32_t)t < max_) { bidx = (int)(t >> 32); b = batches_[bidx].get(); slot = (uint32_t)t; break; }
                // overshoot: the batch filled up (or was sealed) between the load and the add -- its sealer ignores counts beyond
                // max, and whoever opens the next batch stores a whole new ticket
            }
            wait_for_open_batch(t);
        }
        const uint32_t gen = b->done_gen[slot % DONE_WORDS].load(std::memory_order_acquire);   // cannot advance before this row is filled
        if (slot == 0) {            // the batch's first request starts its deadline and makes sure a dispatcher is watching it
            b->t0_ns.store(now_ns(), std::memory_order_release);
            { std::lock_guard<std::mutex> lk(mu_); }
            cv_disp_.notify_one();
        }
        if (slot + 1 == max_) {     // ... and its last one seals it
            std::lock_guard<std::mutex> lk(mu_);
            seal_locked(bidx, true);
            cv_disp_.notify_one();
        }
        for (int k = 0; k < shape_.n_in; k++) std::memcpy(b->in[k] + (size_t)slot * shape_.in_stride[k], in[k], shape_.in_stride[k]);
        b->filled.fetch_add(1, std::memory_order_release);
        std::atomic<uint32_t>& word = b->done_gen[slot % DONE_WORDS];
        if (test_bump_delay_us_) std::this_thread::sleep_for(std::chrono::microseconds(test_bump_delay_us_ / 2));     // (a caller late to its wait)
        while (word.load(std::memory_order_acquire) == gen) futex_wait_u32(&word, gen);
        const int rc = b->rc;
        if (rc != 0) be_->set_last_error(b->err);
        if (rc == 0)
            for (int k = 0; k < shape_.n_out; k++)
                if (out[k]) std::memcpy(out[k], b->out[k] + (size_t)slot * shape_.out_stride[k], shape_.out_stride[k]);
        // only the batch's last reader takes the lock again
        if (b->readers.fetch_sub(1, std::memory_order_acq_rel) == 1) {
            std::lock_guard<std::mutex> lk(mu_);
            release_locked(bidx);
        }
        callers_inside_.fetch_sub(1, std::memory_order_release);      // the caller's last touch of this object (shutdown polls the counter)
        return rc;
    }

    // Drain and stop (idempotent): what is submitted runs, new submissions are refused, returns when no caller is inside submit().
    void shutdown() {
        {
            std::unique_lock<std::mutex> lk(mu_);
            if (!started_.load()) return;
            stopping_.store(true, std::memory_order_release);
            // close the open batch for good: whoever claimed a slot before this exchange is drained, whoever comes later sees NO_BATCH
            const uint64_t t = ticket_.exchange(NO_BATCH, std::memory_order_acq_rel);
            if (!is_none(t)) {
                const int idx = (int)(t >> 32);
                if (batches_[idx]->state == OPEN) {
                    const uint32_t cnt = (uint32_t)t < max_ ? (uint32_t)t : (uint32_t)max_;
                    if (cnt) seal_counted_locked(idx, cnt, false);
                    else batches_[idx]->state = FREE;
                }
            }
            cv_disp_.notify_all();
            cv_space_.notify_all();
        }
        for (auto& t : threads_)
            if (t.joinable()) t.join();
        threads_.clear();
        // callers woken by the last batches are copying their rows out: wait for the last of them to leave (microseconds)
        while (callers_inside_.load(std::memory_order_acquire) != 0) std::this_thread::sleep_for(std::chrono::microseconds(50));
        {
            std::unique_lock<std::mutex> lk(mu_);
            free_all();
            started_.store(false);
        }
    }

    CoalesceStats stats() {
        std::lock_guard<std::mutex> lk(mu_);
        return stats_;
    }
    // testing aid (tests/emul/coalesce_stress.cpp): a dispatcher pre-empted between two completion words, made certain instead of rare
    void set_test_bump_delay_us(int us) { test_bump_delay_us_ = us; }
    size_t max_batch() const { return max_; }
    int lanes() const { return lanes_; }

private:
    // ticket_ = (index of the OPEN batch << 32) | slots handed out so far; NO_BATCH = every staging set is busy, or shutting down.
    // Callers add 1 and own the slot they read; counts beyond max_ are overshoots (their callers retry).  Only a thread holding mu_
    // replaces the whole word (sealing / opening).
    static constexpr uint64_t NO_BATCH = (uint64_t)0xFFFFFFFFu << 32;      // (adds by late callers only touch its low half)
    static bool is_none(uint64_t t) { return (t >> 32) == 0xFFFFFFFFu; }
    static const int DONE_WORDS = 8;       // completion futexes per batch: waiters spread over 8 kernel hash buckets
    enum State { FREE = 0, OPEN, READY, RUNNING };      // RUNNING lasts until the batch's last caller has copied its outputs out
    struct Batch {
        uint8_t* in[COALESCE_MAX_ARRAYS] = {nullptr};
        uint8_t* out[COALESCE_MAX_ARRAYS] = {nullptr};
        State state = FREE;                 // under mu_
        size_t count = 0;                   // fixed when sealed
        std::atomic<size_t> filled{0}, readers{0};
        std::atomic<int64_t> t0_ns{0};      // first request's arrival (0: not stamped yet)
        alignas(64) std::atomic<uint32_t> done_gen[DONE_WORDS];
        int rc = 0;
        std::string err;                    // the dispatcher's error text when rc != 0 (published with rc)
        Batch() { for (auto& w : done_gen) w.store(0); }
    };

    static int64_t now_ns() { return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
    int index_of(const Batch* b) const {
        for (size_t i = 0; i < batches_.size(); i++)
            if (batches_[i].get() == b) return (int)i;
        return -1;
    }
    int find_free() {
        for (size_t i = 0; i < batches_.size(); i++)
            if (batches_[i]->state == FREE) return (int)i;
        return -1;
    }
    void open_locked(int idx) {
        Batch& b = *batches_[idx];
        b.state = OPEN;
        b.count = 0;
        b.filled.store(0, std::memory_order_relaxed);
        b.t0_ns.store(0, std::memory_order_relaxed);
        ticket_.store((uint64_t)(uint32_t)idx << 32, std::memory_order_release);
    }
    // Seal the batch the ticket points to (if it still is batch idx and OPEN) and open the next free staging set in the same step, so
    // that callers keep claiming slots without ever touching the lock.
    void seal_locked(int idx, bool full) {
        const uint64_t t = ticket_.load(std::memory_order_acquire);
        if (is_none(t) || (int)(t >> 32) != idx || batches_[idx]->state != OPEN) return;     // somebody else sealed it first
        const int nxt = stopping_.load() ? -1 : find_free();
        uint64_t old;
        if (nxt >= 0) {
            Batch& nb = *batches_[nxt];
            nb.state = OPEN; nb.count = 0;
            nb.filled.store(0, std::memory_order_relaxed);
            nb.t0_ns.store(0, std::memory_order_relaxed);
            old = ticket_.exchange((uint64_t)(uint32_t)nxt << 32, std::memory_order_acq_rel);
        } else
            old = ticket_.exchange(NO_BATCH, std::memory_order_acq_rel);
        const uint32_t cnt = (uint32_t)old < max_ ? (uint32_t)old : (uint32_t)max_;
        seal_counted_locked(idx, cnt, full);
    }
    void seal_counted_locked(int idx, uint32_t cnt, bool full) {
        Batch& b = *batches_[idx];
        b.count = cnt;
        b.state = READY;
        // the rows' callers + the dispatcher that will run the batch: the staging set cannot be re-opened before the dispatcher has
        // bumped ALL of its completion words (a caller of the next use must never snapshot a word the previous use still has to bump)
        b.readers.store((size_t)cnt + 1, std::memory_order_release);
        ready_.push_back(idx);
        stats_.requests += cnt;
        stats_.batches++;
        if (cnt > stats_.largest_batch) stats_.largest_batch = cnt;
        if (full) stats_.sealed_full++; else stats_.sealed_deadline++;
    }
    // No slot to be had from ticket t: the open batch is full and its sealer has not published the next one yet (a moment), or every
    // staging set is busy (back-pressure: sleep until a batch's last reader frees one).
    void wait_for_open_batch(uint64_t t) {
        if (!is_none(t)) {
            for (int spin = 0; spin < 200; spin++) {
                if (ticket_.load(std::memory_order_acquire) >> 32 != t >> 32 || stopping_.load(std::memory_order_relaxed)) return;
                if (spin > 20) std::this_thread::yield();
            }
        }
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            if (stopping_.load()) return;
            const uint64_t cur = ticket_.load(std::memory_order_acquire);
            if (!is_none(cur)) {
                if ((uint32_t)cur < max_) return;
                // a full batch nobody has sealed yet can only be one whose last claimer is on its way to the lock we hold
                lk.unlock();
                std::this_thread::yield();
                lk.lock();
                continue;
            }
            const int f = find_free();
            if (f >= 0) { open_locked(f); return; }
            cv_space_.wait(lk);
        }
    }
    // the batch's last reference is gone (mu_ held): the staging set is free, and becomes the open batch if there is none
    void release_locked(int idx) {
        batches_[idx]->state = FREE;
        if (is_none(ticket_.load(std::memory_order_relaxed)) && !stopping_.load()) open_locked(idx);
        cv_space_.notify_all();
    }
    void free_all() {
        for (auto& bp : batches_) {
            Batch& b = *bp;
            for (int k = 0; k < COALESCE_MAX_ARRAYS; k++) {
                if (b.in[k]) { be_->free_staging(b.in[k]); b.in[k] = nullptr; }
                if (b.out[k]) { be_->free_staging(b.out[k]); b.out[k] = nullptr; }
            }
        }
        batches_.clear();
    }
    void dispatcher(int lane) {
        const bool lane_ok = be_->start_lane(lane);
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            int idx = -1;
            const uint64_t t = ticket_.load(std::memory_order_acquire);
            if (!ready_.empty()) { idx = ready_.front(); ready_.pop_front(); }
            else if (!is_none(t) && (uint32_t)t > 0) {
                const int open = (int)(t >> 32);
                int64_t t0 = batches_[open]->t0_ns.load(std::memory_order_acquire);
                const int64_t now = now_ns();
                if (t0 == 0) t0 = now;                   // claimed a moment ago, not stamped yet
                const int64_t deadline = t0 + (int64_t)wait_us_ * 1000;
                if (stopping_.load() || now >= deadline) {
                    seal_locked(open, false);
                    continue;                            // it is in the ready queue now (or another thread sealed it)
                }
                wait_ns(lk, deadline - now);
                continue;
            } else if (stopping_.load()) break;
            else { cv_disp_.wait(lk); continue; }
            Batch& b = *batches_[idx];
            b.state = RUNNING;
            const size_t n = b.count;
            lk.unlock();
            // callers copy their rows in outside the lock: wait for the last of them (a 1.2 KB copy)
            const auto t_seal = std::chrono::steady_clock::now();
            for (unsigned spin = 0; b.filled.load(std::memory_order_acquire) != n; spin++)
                if (spin > 64) std::this_thread::yield();
            const auto t_run = std::chrono::steady_clock::now();
            int rc;
            try { rc = lane_ok ? be_->run(lane, n, b.in, b.out) : nomem_code_; } catch (...) { rc = nomem_code_; }
            const auto t_done = std::chrono::steady_clock::now();
            if (rc != 0) { try { b.err = be_->last_error(); } catch (...) { b.err.clear(); } }
            b.rc = rc;
            for (int w = 0; w < DONE_WORDS; w++) {
                b.done_gen[w].fetch_add(1, std::memory_order_release);
                if (test_bump_delay_us_ && w + 1 < DONE_WORDS) std::this_thread::sleep_for(std::chrono::microseconds(test_bump_delay_us_));
            }
            for (int w = 0; w < DONE_WORDS && (size_t)w < n; w++) futex_wake_all_u32(&b.done_gen[w]);
            lk.lock();
            if (b.readers.fetch_sub(1, std::memory_order_acq_rel) == 1) release_locked(idx);      // the dispatcher's own reference
            stats_.fill_wait_us += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(t_run - t_seal).count();
            stats_.run_us += (uint64_t)std::chrono::duration_cast<std::chrono::microseconds>(t_done - t_run).count();
        }
        lk.unlock();
        be_->stop_lane(lane);
    }
    void wait_ns(std::unique_lock<std::mutex>& lk, int64_t ns) {
#if defined(__SANITIZE_THREAD__)
        // gcc 11's libtsan does not intercept pthread_cond_clockwait (what a steady_clock wait compiles to) and then reports the mutex
        // as never released inside the wait; the race-detector build waits on the system clock instead
        cv_disp_.wait_until(lk, std::chrono::system_clock::now() + std::chrono::nanoseconds(ns));
#else
        cv_disp_.wait_for(lk, std::chrono::nanoseconds(ns));
#endif
    }

    Backend* be_;
    CoalesceShape shape_;
    size_t max_;
    long wait_us_;
    int lanes_, closed_code_, nomem_code_;
    std::mutex mu_;
    std::condition_variable cv_disp_, cv_space_;
    std::vector<std::unique_ptr<Batch>> batches_;
    std::deque<int> ready_;
    alignas(64) std::atomic<uint64_t> ticket_{NO_BATCH};
    std::atomic<bool> started_{false}, stopping_{false};
    alignas(64) std::atomic<size_t> callers_inside_{0};
    std::vector<std::thread> threads_;
    CoalesceStats stats_;
    int test_bump_delay_us_ = 0;
};

}  // namespace bppp_host
