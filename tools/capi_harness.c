/* capi_harness.c -- a caller of libbppp_hip.so written in plain C, with nothing but include/bppp.h: what a cgo / Rust-FFI / JNI
 * binding sees.  (The Python host layer binds the same symbols through ctypes; this program shows that the header is valid C, that
 * the signatures in it are the ones the library exports, and that the entry points behave the same without Python, torch or a HIP
 * header on the caller's side.)
 *
 *   gcc -std=c99 -Wall -Wextra -Werror -Iinclude tools/capi_harness.c -o capi_harness -Lbp_pp_amd -lbppp_hip -Wl,-rpath,$PWD/bp_pp_amd
 *   ./capi_harness fixture.bin [fb_window_bits]
 *
 * fixture.bin (little-endian, written by tests/test_gpu_capi_harness.py from tests/golden/u64_golden.json):
 *   "BPPPFIX1" | u32 label_len | label | u32 n | generators 49 x 64 (g, g_vec[16], h_vec[32])
 *   | commitments n x 64 | proofs n x 928 | x n x u64 | s n x 32 | rnd n x 52 x 32
 * Output: one `key value` line per step; the test compares them with what it knows about the fixture.  Exit code 0 unless a call
 * returned an error code (a rejected proof is a result, not an error).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bppp.h"

#define CHECK(call)                                                                                     \
    do {                                                                                                \
        int rc_ = (call);                                                                               \
        if (rc_ != BPPP_OK) {                                                                           \
            fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, bppp_last_error());                          \
            return 10 - rc_;                                                                            \
        }                                                                                               \
    } while (0)

static void print_bits(const char* key, const uint8_t* a, size_t n) {
    size_t i;
    printf("%s ", key);
    for (i = 0; i < n; i++) putchar(a[i] ? '1' : '0');
    putchar('\n');
}
static void print_status(const char* key, const int32_t* s, size_t n) {
    size_t i;
    printf("%s", key);
    for (i = 0; i < n; i++) printf(" %d", (int)s[i]);
    putchar('\n');
}
static void* xread(FILE* f, size_t bytes) {
    void* p = malloc(bytes ? bytes : 1);
    if (!p || fread(p, 1, bytes, f) != bytes) {
        fprintf(stderr, "short fixture\n");
        exit(3);
    }
    return p;
}

int main(int argc, char** argv) {
    FILE* f;
    char magic[8];
    uint32_t label_len, n32;
    size_t n, i;
    uint8_t *label, *gens, *V, *P, *s, *rnd, *accept, *P2, *V2, *states_out;
    uint64_t* x;
    int32_t *status, reject = -1;
    bppp_ctx *ctx = NULL, *clone = NULL;
    bppp_group* grp = NULL;
    uint8_t st0[BPPP_TRANSCRIPT_STATE_BYTES], st1[BPPP_TRANSCRIPT_STATE_BYTES], ch[32];
    int wbits = argc > 2 ? atoi(argv[2]) : 8, dev0 = 0;
    size_t lo = 1, hi = 0;

    if (argc < 2 || !(f = fopen(argv[1], "rb"))) {
        fprintf(stderr, "usage: %s fixture.bin [fb_window_bits]\n", argv[0]);
        return 2;
    }
    if (fread(magic, 1, 8, f) != 8 || memcmp(magic, "BPPPFIX1", 8) || fread(&label_len, 4, 1, f) != 1) return 3;
    label = xread(f, label_len);
    if (fread(&n32, 4, 1, f) != 1) return 3;
    n = n32;
    gens = xread(f, 49 * BPPP_POINT_BYTES);
    V = xread(f, n * BPPP_POINT_BYTES);
    P = xread(f, n * BPPP_U64_PROOF_BYTES);
    x = xread(f, n * 8);
    s = xread(f, n * BPPP_SCALAR_BYTES);
    rnd = xread(f, n * 52 * BPPP_SCALAR_BYTES);
    fclose(f);
    accept = calloc(n + 1, 1);
    status = calloc(n + 1, sizeof(int32_t));
    P2 = malloc(n * BPPP_U64_PROOF_BYTES);
    V2 = malloc(n * BPPP_POINT_BYTES);
    states_out = malloc(n * BPPP_TRANSCRIPT_STATE_BYTES);
    if (!accept || !status || !P2 || !V2 || !states_out) return 4;

    /* error behaviour needs no device: null arguments are refused */
    printf("null_ctx_rc %d\n", bppp_u64_verify_batch(NULL, label, label_len, n, V, P, accept, status));

    CHECK(bppp_ctx_create(&ctx, gens, gens + 64, gens + 64 * (1 + BPPP_G_VEC_FULL_SZ), 0, wbits));

    /* U64RangeProofProtocol::verify over the batch */
    CHECK(bppp_u64_verify_batch(ctx, label, label_len, n, V, P, accept, status));
    print_bits("verify", accept, n);
    print_status("status", status, n);

    /* an empty batch is a no-op */
    printf("empty_rc %d\n", bppp_u64_verify_batch(ctx, label, label_len, 0, V, P, accept, status));

    /* U64RangeProofProtocol::prove + commit_value with the fixture's witnesses and randomness: the same bytes */
    CHECK(bppp_u64_prove_batch(ctx, label, label_len, n, x, s, rnd, P2, V2, status));
    printf("prove_same_proofs %d\n", memcmp(P2, P, n * BPPP_U64_PROOF_BYTES) == 0);
    printf("prove_same_commitments %d\n", memcmp(V2, V, n * BPPP_POINT_BYTES) == 0);

    /* the same proofs in the crate's wire format (SEC1-compressed points), straight back into the SEC1 verifier */
    {
        uint8_t* P525 = malloc(n * BPPP_U64_PROOF_SEC1_BYTES);
        uint8_t* V33 = malloc(n * 33);
        if (!P525 || !V33) return 4;
        CHECK(bppp_u64_prove_batch_sec1(ctx, label, label_len, n, x, s, rnd, P525, V33, status));
        printf("sec1_same_x %d\n", memcmp(P525 + 1, P, 32) == 0 && memcmp(V33 + 1, V, 32) == 0 && (P525[0] == 2 || P525[0] == 3));
        CHECK(bppp_u64_verify_batch_sec1(ctx, label, label_len, n, V33, P525, accept, status));
        print_bits("verify_sec1", accept, n);
        free(P525);
        free(V33);
    }

    /* the caller's own transcript: Transcript::new(label) by hand gives the same verdicts; one with extra context does not */
    CHECK(bppp_transcript_new(label, label_len, st0));
    CHECK(bppp_u64_verify_batch_transcript(ctx, n, st0, 1, V, P, accept, status, states_out));
    print_bits("verify_transcript", accept, n);
    memcpy(st1, st0, sizeof st1);
    CHECK(bppp_transcript_append_message(st1, (const uint8_t*)"ctx", 3, (const uint8_t*)"harness", 7));
    CHECK(bppp_u64_verify_batch_transcript(ctx, n, st1, 1, V, P, accept, status, NULL));
    print_bits("verify_other_transcript", accept, n);
    /* the transcript the verifier hands back keeps working as a transcript */
    CHECK(bppp_transcript_challenge_bytes(states_out, (const uint8_t*)"next", 4, ch, sizeof ch));
    printf("next_challenge ");
    for (i = 0; i < sizeof ch; i++) printf("%02x", ch[i]);
    putchar('\n');

    /* one flipped bit in every other proof's n[0] */
    memcpy(P2, P, n * BPPP_U64_PROOF_BYTES);
    for (i = 0; i < n; i += 2) P2[i * BPPP_U64_PROOF_BYTES + BPPP_U64_PROOF_BYTES - 1] ^= 1;
    CHECK(bppp_u64_verify_batch(ctx, label, label_len, n, V, P2, accept, status));
    print_bits("verify_flipped", accept, n);

    /* a second context sharing the first one's tables, and the one-device group (the sharded entry point's degenerate case) */
    CHECK(bppp_ctx_create_shared(&clone, ctx));
    CHECK(bppp_u64_verify_batch(clone, label, label_len, n, V, P2, accept, status));
    print_bits("verify_clone", accept, n);
    bppp_ctx_destroy(clone);
    bppp_shard_range(n, 0, 1, &lo, &hi);
    printf("shard_range %lu %lu\n", (unsigned long)lo, (unsigned long)hi);
    CHECK(bppp_group_create(&grp, gens, gens + 64, gens + 64 * (1 + BPPP_G_VEC_FULL_SZ), &dev0, 1, wbits));
    printf("group_size %d\n", bppp_group_size(grp));
    CHECK(bppp_u64_verify_batch_sharded(grp, label, label_len, n, V, P2, accept, status, &reject));
    print_bits("verify_group", accept, n);
    printf("reject_count %d\n", (int)reject);
    bppp_group_destroy(grp);
    bppp_ctx_destroy(ctx);
    printf("done 1\n");
    return 0;
}
