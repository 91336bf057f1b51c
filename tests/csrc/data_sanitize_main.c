/* Sanitizer driver for the C ABI of libdb1_data.so ("compile the C-ABI library with -fsanitize=address,undefined").
 * Built by tests/test_data_cpu.py together with bdm_db1_amd/csrc_host/db1_data.cpp under AddressSanitizer + UBSan and run on the
 * reference-written token store tests/golden/data_fixture.{idx,bin}: every entry point, including the error paths (bad index,
 * out-of-range slice, missing file, truncated index), so that an out-of-bounds read of the mapping or an overflow in the index
 * builders aborts the test.  Prints a checksum line the test compares with the values the Python binding returns. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/db1_data.h"

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #c, db1_data_last_error()); return 1; } } while (0)

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    db1_idx* h = NULL;
    CHECK(db1_idx_open(argv[1], &h) == 0 && h);
    const int64_t n = db1_idx_len(h), es = db1_idx_elem_size(h);
    CHECK(n > 0 && es > 0 && db1_idx_doc_count(h) > 0 && db1_idx_dtype_code(h) > 0);
    int32_t* sizes = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);   /* the mapped array is only 2-byte aligned: copy, do not cast */
    memcpy(sizes, db1_idx_sizes(h), sizeof(int32_t) * (size_t)n);
    uint64_t sum = 0, tokens = 0;
    for (int64_t i = 0; i < n; i++) {
        const void* p; int64_t m;
        CHECK(db1_idx_get(h, i, 0, -1, &p, &m) == 0 && m == sizes[i]);
        for (int64_t k = 0; k < m * es; k++) sum += ((const unsigned char*)p)[k] * (uint64_t)(k % 251 + 1);   /* touches every byte of the item */
        tokens += (uint64_t)m;
        if (m > 2) { CHECK(db1_idx_get(h, i, 1, m - 2, &p, &m) == 0); }
    }
    const void* p; int64_t m;
    CHECK(db1_idx_get(h, n, 0, -1, &p, &m) != 0);          /* item out of range */
    CHECK(db1_idx_get(h, -1, 0, -1, &p, &m) != 0);
    CHECK(db1_idx_get(h, 0, sizes[0] + 1, -1, &p, &m) != 0); /* offset beyond the item */
    CHECK(db1_idx_get(h, 0, 0, sizes[0] + 5, &p, &m) != 0);  /* length beyond the item */
    (void)db1_idx_pointers(h); (void)db1_idx_doc_idx(h);
    /* index builders on the store's own sizes (two-call protocol) */
    const int64_t docs = db1_idx_doc_count(h) - 1;
    int32_t* doc_idx = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n * 3));
    for (int e = 0; e < 3; e++) for (int64_t i = 0; i < n; i++) doc_idx[e * n + i] = (int32_t)((i * 7 + e) % n);
    int64_t rows = 0;
    CHECK(db1_build_sample_idx(sizes, doc_idx, 16, 3, (int64_t)tokens, NULL, &rows) == 0 && rows > 0);
    int32_t* sidx = (int32_t*)malloc(sizeof(int32_t) * (size_t)rows * 2);
    CHECK(db1_build_sample_idx(sizes, doc_idx, 16, 3, (int64_t)tokens, sidx, &rows) == 0);
    for (int64_t i = 0; i < rows * 2; i++) sum += (uint64_t)sidx[i] * 3;
    int32_t pl[5] = {9, 4, 13, 2, 7};
    CHECK(db1_build_rl_sample_idx(pl, 5, 5, NULL, &rows) == 0 && rows == 8 + 3 + 12 + 1 + 6);
    int32_t* ridx = (int32_t*)malloc(sizeof(int32_t) * (size_t)rows * 3);
    CHECK(db1_build_rl_sample_idx(pl, 5, 5, ridx, &rows) == 0);
    for (int64_t i = 0; i < rows * 3; i++) sum += (uint64_t)ridx[i];
    uint8_t di[1000]; int64_t dsi[1000]; double w[4] = {0.5, 0.25, 0.15, 0.1};
    CHECK(db1_build_blending_indices(di, dsi, w, 4, 1000) == 0);
    for (int i = 0; i < 1000; i++) sum += di[i] + (uint64_t)dsi[i];
    free(doc_idx); free(sidx); free(ridx); free(sizes);
    db1_idx_close(h);
    db1_idx* bad = NULL;
    CHECK(db1_idx_open("/nonexistent/prefix", &bad) != 0 && bad == NULL);
    if (argc > 2) { CHECK(db1_idx_open(argv[2], &bad) != 0 && bad == NULL); }   /* a truncated copy of the index */
    printf("ok items=%lld tokens=%llu docs=%lld checksum=%llu version=%s\n", (long long)n, (unsigned long long)tokens, (long long)docs,
           (unsigned long long)sum, db1_data_version());
    return 0;
}
