// Which kernels a u64 verify / prove call of n proofs runs -- as ONE pure function of (n, SIMDs of the device, switches), with no HIP in
// it.  bppp_u64.hip's launch sequences take their decisions from the struct this returns and from nowhere else; the library exports the
// function (bppp_u64_plan, include/bppp.h) and remembers the plan of a context's last call ("last_verify_plan" / "last_prove_plan"
// of bppp_ctx_get_option), so tests can (a) check on the CPU that the plan is total and changes exactly at the documented sizes and
// (b) assert on the GPU that a batch of T-1, T, T+1 proofs really took the regime it was meant to exercise.
//
// The regimes exist because one lane per proof fills an MI355X (S = 1,024 SIMDs, 64 lanes each, two wavefronts per SIMD under the
// 256-register cap) only from 2^17 proofs up; below that a call's dependent chains are cut across lanes, and kernels whose data flow
// allows it share the SIMDs.  With S = 1,024:
//
//   verify  n <= S          (1,024)    "split4": a wavefront per fixed-base sum, 64 lanes per C0 sum, 16 per round, 4 table sets per proof
//           n <= 4 S        (4,096)    "split2": the same on 32 / 8 lanes, 2 table sets
//           n <= 16 S       (16,384)   lane groups of 4 for the variable-base sums; tables (a lane per point) on the helper stream
//           n <= 32 S       (32,768)   rounds on 2 lanes per proof; one-lane table kernel beside phase 1
//           n <= 64 S       (65,536)   one lane per proof in uncapped builds; tables beside phase 1; last round's sum beside the final sum
//           n <  128 S      (131,072)  the 256-register builds, 8 lanes per fixed-base sum
//           n >= 128 S                 one lane per fixed-base sum; up to 288 S (a chip filled once or twice) as TWO half-batch chains of
//                                      kernels that leapfrog on the SIMDs (twin), and up to 128 S with progress-paced wave priority (pace)
//   prove   n <= S: a wavefront per sum, wide round scalars | n <= 4 S: 16-lane stages and folds | n <= 16 S: 4-lane stages and folds |
//           n <= 32 S: next commitment from fixed-base sums | n <= 128 S: round scalars in four parts | n >= 128 S: one lane per sum;
//           the 256-register builds from more than 64 S values (one wavefront per SIMD) on
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>

namespace bppp_host {

static const unsigned PLAN_BLOCK = 64;          // BPPP_BLOCK: lanes of a one-lane-per-proof workgroup (checked against kernels.h in bppp_u64.hip)

struct PlanKnobs {
    int n_simds = 1024;
    bool no_small = false, no_lane_groups = false, no_split = false, timing = false;
    int tables_beside = -1, tail_beside = -1, fb_one_lane_mode = -1, next_overlap = -1;      // diagnostics: -1 = by size
    long next_msm_max = -1, lane_forms_max = -1, lane4_max = -1, scal_parts_max = -1;       // diagnostics: -1 = by n_simds
    int shared_inv = -1;         // diagnostics: proofs per shared inversion (0 none, 2 4 8 16) wherever the one-lane kernels run; -1 = by size
    int twin = -1;               // diagnostics: 1 = two half-batch sequences wherever the form allows it, 0 = never; -1 = by size
    int pace = -1;               // diagnostics: 1 = progress-paced wave priority in the one-lane sums always, 0 = never; -1 = by size
};

enum PlanTables { TABLES_INLINE = 0, TABLES_ASIDE, TABLES_BESIDE };        // on the main stream | lane-per-point kernel on the helper stream | one-lane kernel beside phase 1
enum PlanPhase1 { P1_FULL = 0, P1_SMALL, P1_WG4, P1_G16 };                 // 256-register build | uncapped | 256-thread workgroups (beside the tables) | 16 lanes per proof
enum PlanFb { FB_L8 = 0, FB_L1, FB_L64, FB_L4 };                           // lanes per fixed-base sum (FB_L1 also means: C0's two halves one after the other, not side by side)
enum PlanC0Var { C0V_FULL = 0, C0V_SMALL, C0V_G4, C0V_G32, C0V_G64 };
enum PlanRound { R_FULL = 0, R_SMALL, R_G2, R_G4, R_G8, R_G16 };

struct VerifyPlan {
    bool split = false;          // at most 4 proofs per SIMD: chains cut across lanes, `parts` table sets per proof
    int parts = 1;
    bool small = false;          // the grid is at most one wavefront per SIMD: uncapped register builds
    int tables = TABLES_INLINE, tparts = 1;
    int phase1 = P1_FULL, fb = FB_L8, c0var = C0V_FULL, round = R_FULL;
    bool tail_beside = false;    // round 4 as head + tail, final scalars and final sum on the helper stream
    bool final_scalars_g16 = false;
    size_t vtab_sets = 1;        // window-table sets per proof the call needs room for
    int shared_inv = 0;          // G = 2, 4, 8, 16: the field inversions of the table build and of the rounds are taken once per G proofs by
                                 // kernels of their own between the passes (straus_core.h: fe_batch_inv_lane); 0: every lane inverts for itself
    int twin = 1;                // 2: the batch runs as TWO half-batch launch sequences on two stream pairs (each half on the plan plan_verify_half
                                 // gives it); the other fields then describe a half.  1: one sequence
    int pace = 0;                // 1: the one-lane variable-base sums lower their wave priority as they advance (straus_core.h: straus_pace); 2: the one-lane fixed-base sums too (fb_core.h: fb_pace)
    uint32_t code() const {
        const uint32_t lg = shared_inv >= 16 ? 4 : shared_inv >= 8 ? 3 : shared_inv >= 4 ? 2 : shared_inv >= 2 ? 1 : 0;
        return (uint32_t)phase1 | (uint32_t)tables << 4 | (uint32_t)tparts << 8 | (uint32_t)fb << 12 | (uint32_t)c0var << 16 | (uint32_t)round << 20 |
               (uint32_t)(tail_beside ? 1 : 0) << 24 | (uint32_t)(small ? 1 : 0) << 25 | (uint32_t)(split ? 1 : 0) << 26 | lg << 27 |
               (uint32_t)(twin == 2 ? 1 : 0) << 30 | (uint32_t)(pace ? 1 : 0) << 31;        // (pace 2 is a diagnostic: described as pace=1)
    }
};

inline VerifyPlan plan_verify(size_t n, const PlanKnobs& k, bool rlc) {
    VerifyPlan p;
    const size_t S = (size_t)(k.n_simds > 0 ? k.n_simds : 1);
    const size_t blocks = (n + PLAN_BLOCK - 1) / PLAN_BLOCK;
    const bool lanes_ok = !k.no_small && !k.no_lane_groups && !k.no_split;
    p.split = lanes_ok && n <= 4 * S;
    // four parts per GLV stream up to one proof per SIMD, two beyond (the extra lanes start to queue: 4,096 proofs 4.3 ms in two parts, 5.1 in four)
    p.parts = p.split ? (n <= S ? 4 : 2) : 1;
    p.vtab_sets = (size_t)p.parts;
    p.small = !k.no_small && blocks <= S;
    // batches the lane groups serve (up to 16 proofs per SIMD) build their tables a lane per point on the helper stream while phase 1 runs
    const bool aside = p.split || (lanes_ok && 4 * blocks <= S);
    // ... the sizes above, while the one-lane kernels are a lone wavefront per SIMD, run the ONE-lane table kernel beside phase 1
    const bool beside = !aside && !k.timing && (k.tables_beside >= 0 ? k.tables_beside == 1 : (!k.no_split && !k.no_small && blocks <= S));
    p.tables = aside ? TABLES_ASIDE : beside ? TABLES_BESIDE : TABLES_INLINE;
    p.tparts = p.split ? p.parts : 1;
    p.phase1 = p.split ? P1_G16 : beside ? P1_WG4 : p.small ? P1_SMALL : P1_FULL;
    const bool fb_one_lane = k.fb_one_lane_mode >= 0 ? k.fb_one_lane_mode == 1 : n >= 128 * S;
    p.fb = p.split ? FB_L64 : fb_one_lane ? FB_L1 : FB_L8;
    const bool grouped = !k.no_lane_groups && 4 * blocks <= S;
    const bool pairs = !k.no_lane_groups && 2 * blocks <= S;
    p.c0var = p.split ? (p.parts == 4 ? C0V_G64 : C0V_G32) : grouped ? C0V_G4 : p.small ? C0V_SMALL : C0V_FULL;
    p.round = p.split ? (p.parts == 4 ? R_G16 : R_G8) : grouped ? R_G4 : pairs ? R_G2 : p.small ? R_SMALL : R_FULL;
    const bool one_lane_rounds = !p.split && !grouped && !pairs;
    p.tail_beside = !rlc && !k.timing && one_lane_rounds && (k.tail_beside >= 0 ? k.tail_beside == 1 : (p.small && !k.no_split));
    p.final_scalars_g16 = p.split;
    // Shared inversions: where the one-lane kernels run with four or more wavefronts per SIMD, a separate launch of n / G lanes that
    // inverts for G proofs each replaces 8 of the 9 field inversions a proof costs (4 in the table build, 4 in the rounds).  Such a
    // launch takes what ONE inversion takes (0.15 ms, a lone wavefront per SIMD) however many proofs there are, the inversions it
    // replaces 0.33 ms per 2^20 proofs: 2^20 proofs 147.6 -> 144.8 ms, 2^19 75.8 -> 74.8, 2^18 38.0 -> 37.6, nothing at 2^17
    // (profiles/r05/r05_q_ab_shared_inv.txt), so it starts at 256 S.
    const bool one_lane_all = p.round == R_FULL && p.tables == TABLES_INLINE && p.c0var == C0V_FULL;
    const int by_size = n >= 1024 * S ? 16 : n >= 256 * S ? 8 : 0;
    const int want = k.shared_inv >= 0 ? k.shared_inv : by_size;
    p.shared_inv = !one_lane_all ? 0 : want >= 16 ? 16 : want >= 8 ? 8 : want >= 4 ? 4 : want >= 2 ? 2 : 0;
    // A launch that fills the chip exactly once or twice ends in a long tail: the arbiter serves the older wavefront of a SIMD's pair
    // first, so one finishes early and its partner runs on alone at a lone wavefront's issue rate -- 41 % of the SIMD-time of the rounds
    // at 2^17 proofs (profiles/r06/r06_a_wave_timeline.txt), 9 % over the saturated rate for the whole call.  Two remedies, both plan fields:
    //   twin: the batch as two halves, each a launch sequence of its own on its own stream pair.  A SIMD then holds one wavefront of each
    //         half, of DIFFERENT kernels; when the older one ends, its sequence's next kernel moves in: the halves leapfrog and no SIMD
    //         is left with one wavefront except at the very end.  The halves run the 256-register one-lane kernels (plan_verify_half);
    //   pace: the sums lower their own wave priority as they advance, so that a pair ends together (straus_core.h: straus_pace).
    //   Where each pays (one box, child contexts timed in turns, profiles/r06/r06_j_twin_sizes.txt; g = wavefronts / wavefront slots):
    //   twin  g = 1 (2^17 proofs) -1.4 % on top of pace, 1.06 .. 1.25 -3 .. -6 %, 1.75 -11 %, 2 (2^18) -3.5 %, 2.25 -6 %; but +4 % at g = 1.5,
    //         +2 .. +4 % from 2.5 up (the halves' own part-filled generations collide): so up to g = 2.25, except a last generation that is
    //         30 .. 70 % full;
    //   pace  g <= 1 (one generation of pairs): -1.5 .. -2.3 %; +1.5 % beyond (a new wavefront must not outrank a half-done one).
    const size_t G = 2 * S, rem = blocks % G;
    const bool fills_once = blocks > S && blocks <= G;            // more than one wavefront per SIMD, at most two: ONE generation of pairs
    const bool twin_by_size = blocks >= G && 4 * blocks <= 9 * G && !(10 * rem > 3 * G && 10 * rem < 7 * G);
    const bool twin_ok = !rlc && !k.timing && one_lane_all && fb_one_lane && n >= 2 * PLAN_BLOCK;
    p.twin = twin_ok && (k.twin >= 0 ? k.twin == 1 : twin_by_size) ? 2 : 1;
    p.pace = !one_lane_all ? 0 : k.pace >= 0 ? k.pace : fills_once ? 1 : 0;
    return p;
}
// the plan of ONE half of a twin call (n_half proofs): the 256-register one-lane kernels and one lane per fixed-base sum whatever the
// half's size -- a half shares every SIMD with a wavefront of the other half
inline VerifyPlan plan_verify_half(size_t n_half, const PlanKnobs& k, int pace) {
    PlanKnobs h = k;
    h.no_small = true; h.fb_one_lane_mode = 1; h.tables_beside = 0; h.tail_beside = 0; h.twin = 0; h.pace = pace;
    VerifyPlan p = plan_verify(n_half, h, false);
    p.twin = 2;
    return p;
}
inline size_t twin_first_half(size_t n) { return (n / 2 + PLAN_BLOCK - 1) / PLAN_BLOCK * PLAN_BLOCK; }

enum PlanStage { ST_FULL = 0, ST_W2, ST_G4, ST_G4_W2, ST_G16, ST_G16_W2 };      // one lane (uncapped | 256 registers), 4 lanes, 16 lanes per value
enum PlanScalars { SC_ONE = 0, SC_PARTS, SC_WIDE };

struct ProvePlan {
    int fb = FB_L8;              // lanes per fixed-base sum of a single-job launch (FB_L4 is chosen per launch: fb4_from_jobs)
    int fb4_from_jobs = 0;       // a fused launch of at least this many jobs runs on 4 lanes per sum (0: never)
    bool next_by_msm = false;    // a level's commitment as E + R from fixed-base sums (else the variable-base kernel)
    bool w2 = false;             // 256-register builds of the one-lane stage / fold kernels
    int stage = ST_FULL, fold = ST_FULL;
    int scalars = SC_ONE;
    bool fold_leaves_scalars = false;      // the lane-form fold computes the next round's scalars itself
    bool overlap_next = false;   // the variable-base next commitment on the helper stream
    bool next_g4 = false;
    bool ct = false;
    uint32_t code() const {
        return (uint32_t)fb | (uint32_t)fb4_from_jobs << 4 | (uint32_t)stage << 8 | (uint32_t)fold << 12 | (uint32_t)scalars << 16 |
               (uint32_t)(next_by_msm ? 1 : 0) << 20 | (uint32_t)(w2 ? 1 : 0) << 21 | (uint32_t)(overlap_next ? 1 : 0) << 22 |
               (uint32_t)(next_g4 ? 1 : 0) << 23 | (uint32_t)(ct ? 1 : 0) << 24;
    }
};

inline ProvePlan plan_prove(size_t n, const PlanKnobs& k, bool ct) {
    ProvePlan p;
    p.ct = ct;
    const size_t S = (size_t)(k.n_simds > 0 ? k.n_simds : 1);
    const size_t blocks = (n + PLAN_BLOCK - 1) / PLAN_BLOCK;
    const bool fb_one_lane = k.fb_one_lane_mode >= 0 ? k.fb_one_lane_mode == 1 : n >= 128 * S;
    // a call of at most one value per SIMD: a wavefront per sum (6 additions per lane and a 6-step tree instead of 44 and 3)
    const bool fb_wave = !k.no_small && !k.no_split && n <= S;
    p.fb = fb_wave ? FB_L64 : fb_one_lane ? FB_L1 : FB_L8;
    // otherwise 8, 4 or 1 lanes: the fewest that still give every SIMD two wavefronts in the launch -- NJ * 4 n >= 128 S
    if (!fb_wave && !fb_one_lane) {
        for (int nj = 1; nj <= 4; nj++)
            if ((size_t)nj * 4 * n >= 128 * S) { p.fb4_from_jobs = nj; break; }
    }
    const size_t next_msm_max = k.next_msm_max >= 0 ? (size_t)k.next_msm_max : 32 * S;
    p.next_by_msm = fb_wave || (!k.no_split && !k.no_lane_groups && n <= next_msm_max);
    p.w2 = k.no_small || blocks > S;
    const size_t lane_forms_max = k.lane_forms_max >= 0 ? (size_t)k.lane_forms_max : 4 * S;
    const bool stage_lanes = !k.no_lane_groups && !k.no_split && !k.no_small && n <= lane_forms_max;
    const size_t g16_blocks = (16 * n + PLAN_BLOCK - 1) / PLAN_BLOCK, g4_blocks = (4 * n + PLAN_BLOCK - 1) / PLAN_BLOCK;
    const bool g16_w2 = g16_blocks > S;
    const bool stage_lanes4 = !stage_lanes && !k.no_lane_groups && !k.no_split && n <= (k.lane4_max >= 0 ? (size_t)k.lane4_max : 16 * S);
    const bool g4_w2 = k.no_small || g4_blocks > S;
    p.stage = stage_lanes ? (g16_w2 ? ST_G16_W2 : ST_G16) : stage_lanes4 ? (g4_w2 ? ST_G4_W2 : ST_G4) : p.w2 ? ST_W2 : ST_FULL;
    const bool fold_lanes = stage_lanes && p.next_by_msm, fold_lanes4 = stage_lanes4 && p.next_by_msm;
    p.fold = fold_lanes ? (g16_w2 ? ST_G16_W2 : ST_G16) : fold_lanes4 ? (g4_w2 ? ST_G4_W2 : ST_G4) : p.w2 ? ST_W2 : ST_FULL;
    p.fold_leaves_scalars = fold_lanes || fold_lanes4;
    const bool scal_parts = !fb_wave && !k.no_split && n <= (k.scal_parts_max >= 0 ? (size_t)k.scal_parts_max : 128 * S);
    p.scalars = fb_wave ? SC_WIDE : scal_parts ? SC_PARTS : SC_ONE;
    p.overlap_next = !k.timing && (k.next_overlap >= 0 ? k.next_overlap == 1 : 4 * blocks >= S);
    p.next_g4 = !k.no_lane_groups && 4 * blocks <= S;
    return p;
}

// ---- RLC mode: how the final checks of a batch are grouped, from what the previous RLC call on the context rejected.
// The stages (docs/design/09): bucket stage over superchunks of M proofs (Pippenger, ~ 4 ms per 2^20 proofs at M = 4096, more for
// smaller M) -> chunks of C proofs for the superchunks that failed (per proof a 64-doubling weighted commitment, ~ 7 ms per 2^20, plus
// the chunk's 49-base sum: 588 table additions shared by C proofs) -> the exact per-proof check for the chunks that failed.  A group
// passes only if every proof in it is valid: with a reject rate r a superchunk passes with probability e^(-r M), a chunk with
// e^(-r C).  Round 4 always took M = 4096 and C = 8: at r = 1/1024 every superchunk failed (its 4 ms wasted) and the chunks of 8 paid
// 74 table additions per proof where chunks of 32 pay 18.  With the rate of the previous call in hand:
//   M: the automatic size, halved while r M > 0.3 (down to 1,024: smaller bucket stages cost more than they save), no bucket stage at
//      all when even that one would fail more often than not (r M > 0.45);
//   C: 32 while a chunk of 32 still passes nine times in ten (r <= 1/256), else 8.
// rate < 0 = no history (first call, or the previous one still in flight): round 4's choice.
struct RlcPlan { unsigned super_m, chunk; };
inline RlcPlan plan_rlc(unsigned auto_super_m, bool super_is_auto, int chunk_option /* 0 auto, 8, 32 */, double rate) {
    RlcPlan p;
    p.super_m = auto_super_m;
    if (super_is_auto && auto_super_m && rate > 0) {
        unsigned m = auto_super_m;
        while (m > 1024 && rate * m > 0.3) m /= 2;
        p.super_m = rate * m > 0.45 ? 0u : m;
    }
    if (chunk_option == 8 || chunk_option == 32) p.chunk = (unsigned)chunk_option;
    else p.chunk = (rate >= 0 && rate * 256 <= 1.0) ? 32u : 8u;
    if (p.super_m % 32) p.chunk = 8;      // a superchunk is made of whole chunks (explicit sizes are multiples of 8)
    return p;
}

// "phase1=wg4 tables=beside/1 fb=l8 c0var=small round=small tail_beside=1 small=1 split=0" -- what tests assert on
inline int plan_describe(uint32_t code, bool prove, char* buf, size_t cap) {
    static const char* const P1[] = {"full", "small", "wg4", "g16"};
    static const char* const TB[] = {"inline", "aside", "beside"};
    static const char* const FB[] = {"l8", "l1", "l64", "l4"};
    static const char* const CV[] = {"full", "small", "g4", "g32", "g64"};
    static const char* const RD[] = {"full", "small", "g2", "g4", "g8", "g16"};
    static const char* const ST[] = {"full", "w2", "g4", "g4_w2", "g16", "g16_w2"};
    static const char* const SC[] = {"one", "parts", "wide"};
    auto pick = [](const char* const* t, size_t nt, uint32_t i) { return i < nt ? t[i] : "?"; };
    if (!prove)
        return std::snprintf(buf, cap, "phase1=%s tables=%s/%u fb=%s c0var=%s round=%s tail_beside=%u small=%u split=%u twin=%u pace=%u shared_inv=%u", pick(P1, 4, code & 15),
                             pick(TB, 3, (code >> 4) & 15), (code >> 8) & 15, pick(FB, 4, (code >> 12) & 15), pick(CV, 5, (code >> 16) & 15),
                             pick(RD, 6, (code >> 20) & 15), (code >> 24) & 1, (code >> 25) & 1, (code >> 26) & 1, ((code >> 30) & 1) + 1u, (code >> 31) & 1,
                             ((code >> 27) & 7) ? 1u << ((code >> 27) & 7) : 0u);
    return std::snprintf(buf, cap, "fb=%s fb4_from_jobs=%u stage=%s fold=%s scalars=%s next_by_msm=%u w2=%u overlap_next=%u next_g4=%u ct=%u",
                         pick(FB, 4, code & 15), (code >> 4) & 15, pick(ST, 6, (code >> 8) & 15), pick(ST, 6, (code >> 12) & 15), pick(SC, 3, (code >> 16) & 15),
                         (code >> 20) & 1, (code >> 21) & 1, (code >> 22) & 1, (code >> 23) & 1, (code >> 24) & 1);
}

}  // namespace bppp_host
