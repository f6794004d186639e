/* Two proofs in flight on ONE GPU through the C ABI alone (no Python, no torch): two host threads, one plk_ctx each on device 0
 * (the second borrows the first's key and MSM table: plk_ctx_share_srs), ONE plk_setup shared by both, a different witness of the
 * same circuit per thread.  What a Rust host that calls SetupForProver::prove(&self) from two threads would do (src/plonk.rs:132-176).
 * Prints "OK <proofs> <ms per proof sequential> <ms per proof concurrent>" when every concurrent proof equals, byte for byte, the
 * proof of the same witness made alone.  Built and run by tests/test_gpu_throughput.py:
 *   gcc -std=c99 -O2 -pthread -I include tests/host/two_provers.c -L plonkit_amd/lib -lplonkit_amd -Wl,-rpath,$PWD/plonkit_amd/lib */
#define _POSIX_C_SOURCE 200809L
#include "plonkit_amd.h"
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define CK(x) do { int32_t rc_ = (x); if (rc_ != PLK_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, rc_, plk_last_error()); exit(1); } } while (0)
static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec / 1e6; }

typedef struct { plk_ctx *ctx; const plk_setup *setup; const plk_circuit *circ; const uint8_t *want; uint64_t want_len; int reps, bad; } job_t;
static pthread_barrier_t gate;
static void *worker(void *arg) {
    job_t *j = (job_t *)arg;
    uint8_t buf[1 << 16]; uint64_t len = 0;
    pthread_barrier_wait(&gate);
    for (int r = 0; r < j->reps; r++) {
        if (plk_prove(j->ctx, j->setup, j->circ, buf, sizeof buf, &len) != PLK_OK) { fprintf(stderr, "prove: %s\n", plk_last_error()); j->bad++; return 0; }
        if (len != j->want_len || memcmp(buf, j->want, len) != 0) j->bad++;
    }
    return 0;
}

int main(int argc, char **argv) {
    const unsigned log_n = argc > 1 ? (unsigned)atoi(argv[1]) : 14;
    const int reps = argc > 2 ? atoi(argv[2]) : 6;
    const uint64_t n = 1ull << log_n;
    plk_ctx *a = 0, *b = 0;
    CK(plk_create(0, &a));
    CK(plk_srs_generate(a, n, 0, 42));
    CK(plk_create(0, &b));
    CK(plk_ctx_share_srs(b, a));                                   /* workspace only: key and table are a's */
    plk_circuit *c0 = 0, *c1 = 0;
    CK(plk_circuit_synthetic_ex(n - 2, 7, 0, 0, &c0));             /* same R1CS ... */
    CK(plk_circuit_synthetic_ex(n - 2, 7, 99, 0, &c1));            /* ... another witness */
    plk_setup *s = 0;
    CK(plk_setup_prepare(a, c0, &s));
    static uint8_t want0[1 << 16], want1[1 << 16], tmp[1 << 16]; uint64_t l0 = 0, l1 = 0, lt = 0;
    CK(plk_prove(a, s, c0, want0, sizeof want0, &l0));             /* the reference proofs, one at a time (also warms both contexts) */
    CK(plk_prove(a, s, c1, want1, sizeof want1, &l1));
    CK(plk_prove(b, s, c1, tmp, sizeof tmp, &lt));
    if (lt != l1 || memcmp(tmp, want1, l1) != 0 || (l0 == l1 && memcmp(want0, want1, l0) == 0)) { fprintf(stderr, "reference proofs inconsistent\n"); return 1; }
    double t0 = now_ms();
    for (int r = 0; r < reps; r++) { CK(plk_prove(a, s, c0, tmp, sizeof tmp, &lt)); CK(plk_prove(a, s, c1, tmp, sizeof tmp, &lt)); }
    const double seq = (now_ms() - t0) / (2.0 * reps);
    job_t j[2] = {{a, s, c0, want0, l0, reps, 0}, {b, s, c1, want1, l1, reps, 0}};
    pthread_t th[2];
    pthread_barrier_init(&gate, 0, 3);
    for (int k = 0; k < 2; k++) pthread_create(&th[k], 0, worker, &j[k]);
    pthread_barrier_wait(&gate);
    t0 = now_ms();
    for (int k = 0; k < 2; k++) pthread_join(th[k], 0);
    const double par = (now_ms() - t0) / (2.0 * reps);
    /* ownership rules: the lender keeps its key while it is on loan */
    if (plk_srs_generate(a, n, 0, 42) != PLK_ERR_ARG) { fprintf(stderr, "the lender replaced a key that is on loan\n"); return 1; }
    plk_setup_free(s); plk_circuit_free(c0); plk_circuit_free(c1);
    plk_destroy(b); plk_destroy(a);                                /* borrower first */
    if (j[0].bad || j[1].bad) { fprintf(stderr, "concurrent proofs differ from the sequential ones\n"); return 1; }
    printf("OK %d %.3f %.3f\n", 2 * reps, seq, par);
    return 0;
}
