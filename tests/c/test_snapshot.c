/*
 * tests/c/test_snapshot.c -- the snapshot reader / writer (include/clapgpu_snapshot.h) driven from C and
 * built with -fsanitize=address,undefined by tests/test_snapshot.py (CPU only; sanitizers are not
 * available on the GPU pool): round trip, lookups, and every way a damaged file is refused.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "clapgpu_snapshot.h"

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "FAIL line %d: %s\n", __LINE__, #cond); return 1; } } while (0)

static long file_size(const char *path) { FILE *f = fopen(path, "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fclose(f); return n; }

static int write_blob(const char *path, const unsigned char *b, long n)
{
    FILE *f = fopen(path, "wb");
    if (!f) return 1;
    fwrite(b, 1, (size_t)n, f);
    fclose(f);
    return 0;
}

int main(int argc, char **argv)
{
    if (argc != 2) return 2;
    char path[512], bad[512];
    snprintf(path, sizeof(path), "%s/c_snap.clps", argv[1]);
    snprintf(bad, sizeof(bad), "%s/c_snap_bad.clps", argv[1]);

    float pos[7][4];
    double body[5][3];
    uint8_t flags8[3] = { 1, 2, 3 };
    for (int i = 0; i < 7; i++) for (int k = 0; k < 4; k++) pos[i][k] = (float)(i * 10 + k);
    for (int i = 0; i < 5; i++) for (int k = 0; k < 3; k++) body[i][k] = i + 0.25 * k;

    clapgpu_snapshot_writer *w = NULL;
    CHECK(clapgpu_snapshot_create(&w, path) == 0);
    uint64_t d2[2] = { 7, 4 }, d1[1] = { 3 }, d3[2] = { 5, 3 }, d0[1] = { 0 };
    CHECK(clapgpu_snapshot_add(w, "entities.pos_scale", CLAPGPU_DT_F32, 2, d2, pos) == 0);
    CHECK(clapgpu_snapshot_add(w, "misc.flags", CLAPGPU_DT_U8, 1, d1, flags8) == 0);
    CHECK(clapgpu_snapshot_add(w, "bodies.pos", CLAPGPU_DT_F64, 2, d3, body) == 0);
    CHECK(clapgpu_snapshot_add(w, "empty", CLAPGPU_DT_I32, 1, d0, NULL) == 0);
    CHECK(clapgpu_snapshot_add(w, "entities.pos_scale", CLAPGPU_DT_F32, 2, d2, pos) != 0);      /* duplicate */
    CHECK(clapgpu_snapshot_add(w, "", CLAPGPU_DT_F32, 2, d2, pos) != 0);
    CHECK(clapgpu_snapshot_add(w, "x", 42, 2, d2, pos) != 0);
    CHECK(clapgpu_snapshot_add(w, "y", CLAPGPU_DT_F32, 2, d2, NULL) != 0);                       /* bytes but no data */
    CHECK(clapgpu_snapshot_finish(w) == 0);

    clapgpu_snapshot *s = NULL;
    clapgpu_snapshot_array a;
    CHECK(clapgpu_snapshot_open(&s, path) == 0);
    CHECK(clapgpu_snapshot_count(s) == 4);
    CHECK(clapgpu_snapshot_find(s, "bodies.pos", &a) == 0 && a.dtype == CLAPGPU_DT_F64 && a.ndim == 2 && a.count == 15);
    CHECK(((uintptr_t)a.data & 63) == 0 && memcmp(a.data, body, sizeof(body)) == 0);
    CHECK(clapgpu_snapshot_find(s, "entities.pos_scale", &a) == 0 && memcmp(a.data, pos, sizeof(pos)) == 0);
    CHECK(clapgpu_snapshot_find(s, "empty", &a) == 0 && a.count == 0);
    CHECK(clapgpu_snapshot_find(s, "nope", &a) != 0);
    CHECK(clapgpu_snapshot_at(s, 1, &a) == 0 && !strcmp(a.name, "misc.flags") && memcmp(a.data, flags8, 3) == 0);
    CHECK(clapgpu_snapshot_at(s, 4, &a) != 0);
    clapgpu_snapshot_close(s);

    /* damaged copies: every byte of the header and the table flipped in turn must either be refused or
     * still describe arrays that lie inside the file (the reader never trusts an offset or a size) */
    long n = file_size(path);
    unsigned char *blob = malloc((size_t)n);
    FILE *f = fopen(path, "rb");
    CHECK(fread(blob, 1, (size_t)n, f) == (size_t)n);
    fclose(f);
    uint64_t table_off;
    memcpy(&table_off, blob + 16, 8);
    int refused = 0, accepted = 0;
    for (long at = 0; at < n; at++) {
        if (at >= 32 && (uint64_t)at < table_off) continue;                 /* payload bytes are not validated */
        for (int bit = 0; bit < 8; bit += 7) {
            blob[at] ^= (unsigned char)(1u << bit);
            CHECK(write_blob(bad, blob, n) == 0);
            clapgpu_snapshot *t = NULL;
            if (clapgpu_snapshot_open(&t, bad) == 0) {
                for (uint32_t k = 0; k < clapgpu_snapshot_count(t); k++) {
                    clapgpu_snapshot_array q;
                    CHECK(clapgpu_snapshot_at(t, k, &q) == 0);
                    volatile unsigned char sink = 0;
                    size_t bytes = (size_t)q.count * clapgpu_dtype_size(q.dtype);
                    for (size_t b = 0; b < bytes; b++) sink ^= ((const unsigned char *)q.data)[b];   /* ASan checks the bounds */
                    (void)sink;
                }
                clapgpu_snapshot_close(t);
                accepted++;
            } else {
                refused++;
            }
            blob[at] ^= (unsigned char)(1u << bit);
        }
    }
    for (long cut = 0; cut < n; cut += 7) {                                 /* truncations */
        CHECK(write_blob(bad, blob, cut) == 0);
        clapgpu_snapshot *t = NULL;
        CHECK(clapgpu_snapshot_open(&t, bad) != 0);
    }
    free(blob);
    CHECK(refused > 100);
    printf("PASS (%d damaged files refused, %d accepted and read in bounds)\n", refused, accepted);
    return 0;
}
