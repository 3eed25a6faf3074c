/*
 * Pure-C client of libmemb_hip.so: builds a tiny `trained` storage by hand (two
 * symbols, codes "0" and "1"), looks up a few rows through the host-buffer
 * entry point and prints them. Compiled and run by tests/test_cabi.py (GPU).
 *   cc -std=c99 -I include tests/cabi/client.c -L memb_amd -lmemb_hip -Wl,-rpath,memb_amd -o client
 */
#include <stdio.h>
#include <string.h>
#include "memb_hip.h"

int main(void)
{
    /* two rows, dim 4: row 0 = 0101b, row 1 = 1100b, one byte per stream */
    const uint8_t packed[2] = {0x50, 0xC0};
    const uint32_t offsets[2] = {0, 1};
    const uint8_t keys[2] = {0, 1};
    const uint32_t size_offsets[2] = {0, 2};       /* both symbols have 1-bit codes */
    const float centroids[2] = {-1.5f, 2.25f};
    memb_hip_trained_desc desc;
    memb_hip_ctx* ctx = NULL;
    const uint32_t rows[4] = {1, MEMB_HIP_MISSING_ROW, 0, 1};
    float out[4][6];
    int i, j;

    memset(&desc, 0, sizeof desc);
    desc.dim = 4;
    desc.n_rows = 2;
    desc.packed_values = packed;
    desc.packed_values_bytes = sizeof packed;
    desc.value_offsets = offsets;
    desc.keys = keys;
    desc.n_keys = 2;
    desc.size_offsets = size_offsets;
    desc.n_size_offsets = 2;
    desc.centroids = centroids;
    desc.n_centroids = 2;

    if (memb_hip_ctx_create_trained(&ctx, 0, &desc) != MEMB_HIP_OK) {
        printf("create failed: %s\n", memb_hip_last_error());
        return 2;
    }
    for (i = 0; i < 4; ++i) {
        for (j = 0; j < 6; ++j) {
            out[i][j] = 9.0f;
        }
    }
    /* rows go to columns 1..4 of a 6-wide matrix */
    if (memb_hip_decode_rows(ctx, rows, 4, &out[0][0], 6, 1) != MEMB_HIP_OK) {
        printf("decode failed: %s\n", memb_hip_last_error());
        return 3;
    }
    for (i = 0; i < 4; ++i) {
        for (j = 0; j < 6; ++j) {
            printf("%g%c", out[i][j], j == 5 ? '\n' : ' ');
        }
    }
    if (memb_hip_decode_rows(ctx, rows, 4, &out[0][0], 4, 1) == MEMB_HIP_OK) {   /* ld < col_off + dim */
        printf("expected an error\n");
        return 4;
    }
    printf("error: %s\n", memb_hip_last_error());
    memb_hip_ctx_destroy(ctx);
    return 0;
}
