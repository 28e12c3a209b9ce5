/* Builds the CPU oracle with AddressSanitizer + UBSan and drives every entry point once (tests/test_sanitizers.py).
 * Sanitizers run on the CPU build only; the GPU pool offers neither ASan nor XNACK. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int sw_oracle_score(const uint8_t *, const uint8_t *, const int8_t *, int);
void sw_oracle_batch_st(const uint8_t *, const uint8_t *, size_t, const int8_t *, int, int32_t *);
void sw_oracle_unpack(const uint8_t *, uint8_t *);
void sw_oracle_pack(const uint8_t *, uint8_t *);
void sw_oracle_generate(uint8_t *, uint8_t *, size_t, uint64_t, uint64_t);
int sw_oracle_banded_affine(const uint8_t *, const uint8_t *, int, const int8_t *, int, int);
int sg_oracle_xdrop(const uint8_t *, const uint8_t *, int32_t *, int32_t *, size_t, size_t *, int *);

int main(void)
{
    enum { N = 64 };
    uint8_t *a = malloc(N * 128), *b = malloc(N * 128);
    int32_t scores[N];
    const int8_t sm[16] = {10, -30, -30, -30, -30, 10, -30, -30, -30, -30, 10, -30, -30, -30, -30, 10};
    sw_oracle_generate(a, b, N, 10000, 0);
    sw_oracle_batch_st(a, b, N, sm, 15, scores);
    long sum = 0;
    for (int k = 0; k < N; ++k) sum += scores[k];
    if (sw_oracle_score(a, a, sm, 15) != 1280) return 1;
    uint8_t packed[32], back[128];
    sw_oracle_pack(a, packed);
    sw_oracle_unpack(packed, back);
    if (memcmp(a, back, 128) != 0) return 2;
    /* banded affine on a 256-mer built from two generated pairs */
    if (sw_oracle_banded_affine(a, a, 256, sm, 20, 5) != 2560) return 3;
    /* semi-global: 16384-mers = 128 generated 128-mers; one identical pair, one unrelated pair */
    uint8_t *l1 = malloc(16384), *l2 = malloc(16384);
    sw_oracle_generate(l1, l2, 128, 7, 0);
    int32_t *tb = malloc(sizeof(int32_t) * 2 * 32769);
    int32_t score = 0;
    size_t len = 0;
    int oob = 0;
    if (sg_oracle_xdrop(l1, l1, &score, tb, 32769, &len, &oob) != 0 || score != 16384 || len != 16385) return 4;
    if (sg_oracle_xdrop(l1, l2, &score, tb, 100, &len, &oob) != 0) return 5;      /* small cap: must not write past it */
    printf("oracle selftest ok (checksum %ld, unrelated semi-global score %d, path %zu)\n", sum, score, len);
    free(a); free(b); free(l1); free(l2); free(tb);
    return 0;
}
