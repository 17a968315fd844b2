// One sharded call over the ranks of a device group, as host-side control flow with no HIP in it: one host thread per rank, and
// the rule that makes the group FAIL CLOSED instead of hanging --
//
//   phase 1  prepare(r)     fallible: stage the shard, enqueue its verification on rank r's stream
//   vote                    every rank thread publishes its return code and waits for the others (host-side latch)
//   phase 2  collective(r)  only if EVERY rank's phase 1 succeeded: enqueue the accept-reduce (ncclAllReduce) on rank r's stream
//   vote                    as above, on the collectives' return codes
//   phase 3  finish(r)      copy results back, wait for rank r's stream
//
// A rank that failed phase 1 never reaches the collective, and because of the vote neither does any other rank: nobody is left
// blocked in an all-reduce whose peer will not come (round 2's group returned the failing rank early and hung the rest).  If a
// collective call itself fails on some rank, the ranks that did enqueue theirs would wait for it forever: abort(r) (ncclCommAbort)
// releases them and the group is marked unusable by the caller.  After a failed vote every rank still runs drain(r)
// (wait for the work it enqueued) so no kernel is left reading caller memory when the call returns.
//
// The same template drives libbppp_hip.so's groups (bppp_group.hip: HIP streams + RCCL) and the CPU tier's emulated group
// (tests/emul: the device code compiled for the host + a blocking in-process all-reduce), which is how the vote is tested without
// a GPU: with the vote removed, the emulated all-reduce of a two-rank group with one failing rank never returns.
#pragma once
#include <condition_variable>
#include <new>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace bppp_host {

// Reusable vote: arrive(code) blocks until all `parties` have arrived and returns the first (lowest rank) non-zero code, or 0.
class Vote {
public:
    explicit Vote(int parties) : parties_(parties), codes_(parties, 0) {}
    int arrive(int rank, int code) {
        std::unique_lock<std::mutex> lk(mu_);
        codes_[rank] = code;
        const unsigned gen = generation_;
        if (++arrived_ == parties_) {
            result_ = 0;
            for (int c : codes_)
                if (c != 0) { result_ = c; break; }
            arrived_ = 0;
            generation_++;
            cv_.notify_all();
            return result_;
        }
        cv_.wait(lk, [&] { return generation_ != gen; });
        return result_;
    }
    // A vote cast on behalf of a party that does not exist as a thread (a rank whose thread could not be started): recorded without
    // waiting for the others, so one thread can cast several -- arrive() from the main thread blocked at the first of two missing ranks,
    // which only the same thread could have voted in.
    void post(int rank, int code) {
        std::lock_guard<std::mutex> lk(mu_);
        codes_[rank] = code;
        if (++arrived_ == parties_) {
            result_ = 0;
            for (int c : codes_)
                if (c != 0) { result_ = c; break; }
            arrived_ = 0;
            generation_++;
            cv_.notify_all();
        }
    }

private:
    std::mutex mu_;
    std::condition_variable cv_;
    int parties_, arrived_ = 0, result_ = 0;
    unsigned generation_ = 0;
    std::vector<int> codes_;
};

// what a phase that threw is reported as (the values of BPPP_ERR_NOMEM / BPPP_ERR_HIP in include/bppp.h)
static const int kShardErrNoMem = -5, kShardErrInternal = -3;

struct ShardedResult {
    int code = 0;              // 0, or the failing rank's code (lowest failing rank)
    int failed_rank = -1;
    bool collective_failed = false;   // a collective call failed after the phase-1 vote passed: communicators were aborted
    std::string error;         // the failing rank's message
};

// prepare / collective / finish / drain return 0 on success; last_error() returns the calling thread's error text.  enter(r) /
// leave(r) bracket everything rank r's thread does (the caller takes and releases rank r's context lock there: a recursive mutex
// has to be released by the thread that took it).
template <class Enter, class Prepare, class Collective, class Finish, class Drain, class Abort, class Leave, class LastError>
ShardedResult run_sharded(int G, Enter&& enter, Prepare&& prepare, Collective&& collective, Finish&& finish, Drain&& drain, Abort&& abort,
                          Leave&& leave, LastError&& last_error, int fail_thread_start_at = -1 /* testing aid: that rank's thread "cannot be started" */) {
    ShardedResult res;
    std::vector<int> rcs(G, 0);
    std::vector<std::string> errs(G);
    Vote vote1(G), vote2(G);
    // A phase must never leave its thread by an exception: a rank that dies before a vote would leave the others blocked in it for
    // ever (holding the group's and the contexts' locks).  Whatever a phase throws becomes that rank's return code, and the rank
    // still arrives at both votes.
    auto guarded = [&](auto&& phase, int r) -> int {
        try {
            const int rc = phase(r);
            if (rc != 0) errs[r] = last_error();
            return rc;
        } catch (const std::bad_alloc&) {
            errs[r] = "out of host memory in a rank thread";
            return kShardErrNoMem;
        } catch (...) {
            errs[r] = "exception in a rank thread";
            return kShardErrInternal;
        }
    };
    // wait for what this rank enqueued; a failure there is reported unless the rank already has an error of its own
    auto drained = [&](int r, int rc) -> int {
        std::string keep = errs[r];
        const int d = guarded(drain, r);
        if (rc != 0) { errs[r] = keep; return rc; }
        return d;
    };
    auto rank_body = [&](int r) {
        int rc = guarded(prepare, r);
        const int v1 = vote1.arrive(r, rc);
        if (v1 != 0) {                       // somebody failed before the collective: nobody enters it
            rcs[r] = drained(r, rc);         // whatever this rank did enqueue finishes before the call returns
            return;
        }
        rc = guarded(collective, r);
        const int v2 = vote2.arrive(r, rc);
        if (v2 != 0) {                       // a collective failed somewhere: release every rank that is (or will be) waiting in one
            try { abort(r); } catch (...) {}
            rcs[r] = drained(r, rc);
            if (r == 0) res.collective_failed = true;
            return;
        }
        rcs[r] = guarded(finish, r);
    };
    auto rank_main = [&](int r) { enter(r); rank_body(r); leave(r); };
    if (G == 1) rank_main(0);
    else {
        std::vector<std::thread> th;
        int started = 0;
        try {
            th.reserve(G);
            for (; started < G; started++) {
                if (started == fail_thread_start_at) throw std::bad_alloc();
                th.emplace_back(rank_main, started);
            }
        } catch (...) {
            // could not start every rank: the missing ranks vote "failed" from here (without waiting: there may be several of them and
            // this is the only thread that can cast their votes), so the ranks that run see a failed first vote, drain and leave
            for (int r = started; r < G; r++) {
                errs[r] = "could not start a rank thread";
                rcs[r] = kShardErrNoMem;
                vote1.post(r, kShardErrNoMem);
            }
        }
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < G; r++)
        if (rcs[r] != 0) { res.code = rcs[r]; res.failed_rank = r; res.error = errs[r]; break; }
    return res;
}

}  // namespace bppp_host
