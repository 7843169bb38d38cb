/* Plain-C consumer of the drop-in boundary (include/ripp_hip.h): no C++, no Python, no torch.
 *   gcc -O2 -Iinclude examples/c_abi_demo.c -Lripp_amd/lib -lripp_hip -Wl,-rpath,$PWD/ripp_amd/lib -o c_abi_demo && ./c_abi_demo [log2 n]
 * Synthesises a statement on the device, computes the claimed product (sipp/src/lib.rs:184-217), proves (sipp/src/lib.rs:42-106),
 * verifies (sipp/src/lib.rs:109-180), checks that a tampered proof is rejected, and exercises the reference's length error. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ripp_hip.h"

#define CHECK(call) do { int32_t rc_ = (call); if (rc_ != RIPP_OK) { fprintf(stderr, "%s -> status %d: %s\n", #call, rc_, ripp_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 10;
    const size_t n = (size_t)1 << lg;
    if (ripp_device_count() < 1) { fprintf(stderr, "no HIP device: libripp_hip has no CPU fallback\n"); return 2; }
    CHECK(ripp_init(0));
    ripp_g1a* a = malloc(n * sizeof *a); ripp_g2a* b = malloc(n * sizeof *b); ripp_fr* r = malloc(n * sizeof *r);
    ripp_gt* proof = malloc(2 * (size_t)lg * sizeof *proof); ripp_fr* ch = malloc((size_t)lg * sizeof *ch);
    CHECK(ripp_synth_g1(1000, 0, 1, n, a)); CHECK(ripp_synth_g2(2000, 0, 1, n, b)); CHECK(ripp_synth_fr(0, 0, 1, n, r));
    ripp_gt value; ripp_stats st;
    CHECK(ripp_pairing_product_coeffs_a(a, b, r, n, &value));
    CHECK(ripp_sipp_prove(a, b, r, n, &value, proof, ch, &st));
    int32_t ok = 0, bad = 1;
    CHECK(ripp_sipp_verify(a, b, r, n, &value, proof, (size_t)lg, &ok));
    ripp_gt saved = proof[0]; proof[0] = proof[1];
    CHECK(ripp_sipp_verify(a, b, r, n, &value, proof, (size_t)lg, &bad));
    proof[0] = saved;
    ripp_g1j l1[2]; ripp_g2j r1[1]; ripp_gt out; memset(l1, 0, sizeof l1); memset(r1, 0, sizeof r1);
    const int32_t len_rc = ripp_pairing_product_j(l1, 2, r1, 1, &out);      /* InnerProductError::MessageLengthInvalid(2, 1) */
    printf("n = 2^%d: prove %.1f ms (products %.1f, folds %.1f, host %.1f), verify %s, tampered proof %s, length error status %d (\"%s\")\n",
           lg, st.total_ms, st.miller_products_ms, st.fold_ms, st.host_ms, ok ? "accepts" : "REJECTS", bad ? "ACCEPTED" : "rejected", len_rc, ripp_last_error());
    ripp_shutdown();
    free(a); free(b); free(r); free(proof); free(ch);
    return (ok == 1 && bad == 0 && len_rc == RIPP_ERR_LENGTH) ? 0 : 1;
}
