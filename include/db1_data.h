/* db1_data.h -- C ABI of libdb1_data.so: the host-side data ingest in front of the DB1 hot path.
 *
 * Native code in the reference (pybind11 module src/data/helpers.cpp, numpy-mmap reader src/data/indexed_dataset.py), native
 * here: plain C++ behind extern "C", no Python, no torch types.  Integer / byte work, results bit-identical to the reference's.
 * Return value 0 = ok, negative = error (db1_data_last_error() gives the text).
 */
#ifndef DB1_DATA_H
#define DB1_DATA_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- ".idx" + ".bin" memory-mapped token store (indexed_dataset.py:351-563, MMapIndexedDataset).
 * .idx = magic "MMIDIDX\0\0", u64 version (1), u8 dtype code (1 u8, 2 i8, 3 i16, 4 i32, 5 i64, 6 f32(*), 7 f64, 8 u16),
 *        u64 len, u64 doc_count, i32 sizes[len], i64 byte pointers[len], i64 doc_idx[doc_count];   .bin = the items back to back.
 * (*) code 6 is numpy's `np.float` in the reference, i.e. float64 (indexed_dataset.py:109). */
typedef struct db1_idx db1_idx;
int db1_idx_open(const char* path_prefix, db1_idx** out);      /* maps <prefix>.idx and <prefix>.bin read-only */
void db1_idx_close(db1_idx* h);
int64_t db1_idx_len(const db1_idx* h);                         /* number of items (sequences) */
int64_t db1_idx_doc_count(const db1_idx* h);
int db1_idx_dtype_code(const db1_idx* h);
int db1_idx_elem_size(const db1_idx* h);                       /* bytes per element */
/* Views into the mapping, valid until close: int32 sizes[len], int64 pointers[len], int64 doc_idx[doc_count].  The format puts them
 * behind a 34-byte header, so they are only 2-BYTE ALIGNED: read them with memcpy (C) / numpy.frombuffer (Python), never through a
 * typed pointer (found by the UBSan run of tests/csrc/data_sanitize_main.c; the library's own reads are memcpy-based). */
const void* db1_idx_sizes(const db1_idx* h);
const void* db1_idx_pointers(const db1_idx* h);
const void* db1_idx_doc_idx(const db1_idx* h);
/* MMapIndexedDataset.get(idx, offset, length) (indexed_dataset.py:522-536): pointer to `*n_elems` elements of item idx starting
 * at element `offset`; length < 0 = to the end of the item.  Zero-copy (points into the mapping). */
int db1_idx_get(const db1_idx* h, int64_t idx, int64_t offset, int64_t length, const void** data, int64_t* n_elems);

/* ---- index builders (helpers.cpp).  Two-call protocol where the size is data dependent: out == NULL returns the row count. */
/* build_sample_idx (helpers.cpp:117-203; caller gpt_dataset.py:287): rows (index into doc_idx, offset in that document) of the
 * num_samples + 1 sample boundaries over the flattened token stream; num_samples = (num_epochs*tokens_per_epoch - 1) / seq_length.
 * out: int32 [(num_samples + 1) * 2]. */
int db1_build_sample_idx(const int32_t* sizes, const int32_t* doc_idx, int32_t seq_length, int32_t num_epochs,
                         int64_t tokens_per_epoch, int32_t* out, int64_t* n_rows);
/* build_rl_sample_idx (helpers.cpp:82-115; caller rl_dataset.py:275): for every path i and start j < len_i - 1 the row
 * (i, j, min(j + transition_num, len_i)).  out: int32 [n_rows * 3]. */
int db1_build_rl_sample_idx(const int32_t* path_lengths, int64_t n_paths, int32_t transition_num, int32_t* out, int64_t* n_rows);
/* build_blending_indices (helpers.cpp:20-80; blendable_dataset.py:100): greedy largest-deficit assignment of `size` samples to
 * `num_datasets` datasets with the given weights. */
int db1_build_blending_indices(uint8_t* dataset_index, int64_t* dataset_sample_index, const double* weights, int32_t num_datasets,
                               int64_t size);

const char* db1_data_last_error(void);
const char* db1_data_version(void);

#ifdef __cplusplus
}
#endif
#endif
