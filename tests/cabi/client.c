/*
 * Pure-C client of libmemb_hip.so: builds a tiny `trained` storage by hand (two
 * symbols, codes "0" and "1"), looks up a few rows through the host-buffer
 * entry point and prints them. Compiled and run by tests/test_cabi.py (GPU).
 *   cc -std=c99 -I include tests/cabi/client.c -L memb_amd -lmemb_hip -Wl,-rpath,memb_amd -o client
 */
#include <stddef.h>
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
    {
        /* memb_hip_ctx_get_info fills no more than the caller's struct_size: a client built against an older, shorter
           struct (here: up to tiles_per_wavefront) must find the bytes behind it untouched */
        union { memb_hip_ctx_info info; unsigned char bytes[sizeof(memb_hip_ctx_info)]; } whole, older;
        const size_t old_size = offsetof(memb_hip_ctx_info, tiles_per_wavefront);
        size_t k;
        memset(&whole, 0, sizeof whole);
        whole.info.struct_size = sizeof(memb_hip_ctx_info);
        if (memb_hip_ctx_get_info(ctx, &whole.info) != MEMB_HIP_OK || whole.info.struct_size != sizeof(memb_hip_ctx_info) ||
            memb_hip_abi_version() != MEMB_HIP_ABI_VERSION) {
            printf("get_info failed: %s\n", memb_hip_last_error());
            return 5;
        }
        memset(&older, 0xAB, sizeof older);
        older.info.struct_size = (uint32_t)old_size;
        older.info.batch_words = 0;
        if (memb_hip_ctx_get_info(ctx, &older.info) != MEMB_HIP_OK || older.info.struct_size != old_size) {
            printf("get_info with an older struct failed: %s\n", memb_hip_last_error());
            return 6;
        }
        for (k = old_size; k < sizeof older; ++k) {
            if (older.bytes[k] != 0xAB) {
                printf("get_info wrote behind the caller's struct_size (byte %u)\n", (unsigned)k);
                return 7;
            }
        }
        printf("info: kernel %s, union_kernel '%s'\n", whole.info.kernel, whole.info.union_kernel);
    }
    {
        /* word -> row on the device and words in / host rows out, as a C caller binds them: the two rows get keys,
           five C strings are packed, looked up and decoded in one call */
        static const char keys_packed[] = "alpha\0beta";        /* row 0 = "alpha", row 1 = "beta"; sizeof counts the final NUL */
        const uint32_t key_offsets[2] = {0, 6};
        const char* queries[5] = {"beta", "gamma", "alpha", "", "beta"};
        memb_hip_words* batch = NULL;
        size_t count = 0;
        float rows_out[5][4];
        if (memb_hip_ctx_stage_words(ctx, keys_packed, sizeof keys_packed, key_offsets, 2) != MEMB_HIP_OK ||
            memb_hip_words_create(&batch, 0) != MEMB_HIP_OK ||
            memb_hip_words_pack(batch, queries, NULL, 5) != MEMB_HIP_OK ||
            memb_hip_words_count(batch, &count) != MEMB_HIP_OK || count != 5 ||
            memb_hip_decode_words(ctx, batch, &rows_out[0][0], 4, 0) != MEMB_HIP_OK) {
            printf("word search failed: %s\n", memb_hip_last_error());
            return 8;
        }
        for (i = 0; i < 5; ++i) {
            printf("words:");
            for (j = 0; j < 4; ++j) {
                printf(" %g", rows_out[i][j]);
            }
            printf("\n");
        }
        memb_hip_words_destroy(batch);
    }
    memb_hip_ctx_destroy(ctx);
    return 0;
}
