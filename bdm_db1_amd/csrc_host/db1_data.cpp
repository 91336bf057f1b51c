// libdb1_data.so: host-side data ingest in front of the DB1 hot path -- memory-mapped token store + index builders.
// Plain C++17 behind the C ABI of include/db1_data.h; integer / byte work only, results identical to the reference's
// src/data/indexed_dataset.py (MMapIndexedDataset) and src/data/helpers.cpp.
#include "../../include/db1_data.h"

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
extern "C" const char* db1_data_last_error(void) { return g_err; }
extern "C" const char* db1_data_version(void) { return "db1_data 0.1"; }

// ------------------------------------------------------------------------------------------------ mmap token store
struct Mapping {
    void* base = nullptr;
    size_t bytes = 0;
    int open_ro(const std::string& path) {
        const int fd = ::open(path.c_str(), O_RDONLY);
        if (fd < 0) return fail(-1, "cannot open %s", path.c_str());
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); return fail(-1, "cannot stat %s", path.c_str()); }
        bytes = (size_t)st.st_size;
        if (bytes) {
            base = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
            if (base == MAP_FAILED) { base = nullptr; ::close(fd); return fail(-1, "cannot mmap %s", path.c_str()); }
        }
        ::close(fd);
        return 0;
    }
    void release() {
        if (base) munmap(base, bytes);
        base = nullptr;
        bytes = 0;
    }
};

struct db1_idx {
    Mapping idx, bin;
    int dtype_code = 0, elem = 0;
    int64_t len = 0, docs = 0;
    const int32_t* sizes = nullptr;
    const int64_t* pointers = nullptr;
    const int64_t* doc_idx = nullptr;
};

static int elem_size_of(int code) {
    switch (code) {
        case 1: case 2: return 1;
        case 3: case 8: return 2;
        case 4: return 4;
        case 5: case 6: case 7: return 8;  // 6 = np.float = float64 in the reference's table
        default: return 0;
    }
}

extern "C" int db1_idx_open(const char* prefix, db1_idx** out) {
    if (!prefix || !out) return fail(-2, "idx_open: null argument");
    db1_idx* h = new db1_idx();
    int rc = h->idx.open_ro(std::string(prefix) + ".idx");
    if (rc == 0) rc = h->bin.open_ro(std::string(prefix) + ".bin");
    if (rc) { h->idx.release(); h->bin.release(); delete h; return rc; }
    static const unsigned char magic[9] = {'M', 'M', 'I', 'D', 'I', 'D', 'X', 0, 0};
    const unsigned char* p = (const unsigned char*)h->idx.base;
    const size_t header = 9 + 8 + 1 + 8 + 8;
    auto bad = [&](const char* why) { h->idx.release(); h->bin.release(); delete h; return fail(-3, "%s.idx: %s", prefix, why); };
    if (h->idx.bytes < header || memcmp(p, magic, 9) != 0) return bad("not an MMIDIDX index");
    uint64_t version, len, docs;
    memcpy(&version, p + 9, 8);
    if (version != 1) return bad("unsupported index version");
    h->dtype_code = p[17];
    h->elem = elem_size_of(h->dtype_code);
    if (!h->elem) return bad("unknown dtype code");
    memcpy(&len, p + 18, 8);
    memcpy(&docs, p + 26, 8);
    h->len = (int64_t)len;
    h->docs = (int64_t)docs;
    const size_t need = header + (size_t)len * 4 + (size_t)len * 8 + (size_t)docs * 8;
    if (h->idx.bytes < need) return bad("truncated index");
    h->sizes = (const int32_t*)(p + header);   // NOTE: only 2-byte aligned (34-byte header): never dereferenced as typed pointers, see the memcpy reads below
    h->pointers = (const int64_t*)(p + header + (size_t)len * 4);   // (the reference reads these unaligned arrays the same way)
    h->doc_idx = (const int64_t*)(p + header + (size_t)len * 12);
    *out = h;
    return 0;
}
extern "C" void db1_idx_close(db1_idx* h) {
    if (!h) return;
    h->idx.release();
    h->bin.release();
    delete h;
}
extern "C" int64_t db1_idx_len(const db1_idx* h) { return h->len; }
extern "C" int64_t db1_idx_doc_count(const db1_idx* h) { return h->docs; }
extern "C" int db1_idx_dtype_code(const db1_idx* h) { return h->dtype_code; }
extern "C" int db1_idx_elem_size(const db1_idx* h) { return h->elem; }
extern "C" const void* db1_idx_sizes(const db1_idx* h) { return h->sizes; }
extern "C" const void* db1_idx_pointers(const db1_idx* h) { return h->pointers; }
extern "C" const void* db1_idx_doc_idx(const db1_idx* h) { return h->doc_idx; }

extern "C" int db1_idx_get(const db1_idx* h, int64_t idx, int64_t offset, int64_t length, const void** data, int64_t* n_elems) {
    if (!h || !data || !n_elems) return fail(-2, "idx_get: null argument");
    if (idx < 0 || idx >= h->len) return fail(-4, "idx_get: item %lld out of range [0, %lld)", (long long)idx, (long long)h->len);
    int64_t ptr, size32;
    memcpy(&ptr, (const char*)h->pointers + idx * 8, 8);
    int32_t sz;
    memcpy(&sz, (const char*)h->sizes + idx * 4, 4);
    size32 = sz;
    if (length < 0) length = size32 - offset;
    if (offset < 0 || length < 0 || offset + length > size32)
        return fail(-4, "idx_get: [%lld, %lld) outside item %lld of %lld elements", (long long)offset, (long long)(offset + length), (long long)idx, (long long)size32);
    const int64_t byte0 = ptr + offset * h->elem;
    if (byte0 < 0 || (uint64_t)(byte0 + length * h->elem) > h->bin.bytes) return fail(-3, "idx_get: item %lld points outside the .bin file", (long long)idx);
    *data = (const char*)h->bin.base + byte0;
    *n_elems = length;
    return 0;
}

// ------------------------------------------------------------------------------------------------ index builders
extern "C" int db1_build_sample_idx(const int32_t* sizes, const int32_t* doc_idx, int32_t seq_length, int32_t num_epochs,
                                    int64_t tokens_per_epoch, int32_t* out, int64_t* n_rows) {
    if (!sizes || !doc_idx || !n_rows) return fail(-2, "build_sample_idx: null argument");
    if (seq_length <= 1 || num_epochs <= 0 || tokens_per_epoch <= 1) return fail(-4, "build_sample_idx: seq_length > 1, num_epochs > 0, tokens_per_epoch > 1 required");
    const int64_t num_samples = ((int64_t)num_epochs * tokens_per_epoch - 1) / seq_length;
    *n_rows = num_samples + 1;
    if (!out) return 0;
    int64_t pos = 0;       // index into doc_idx
    int32_t offset = 0;    // first unread token of that document
    out[0] = 0;
    out[1] = 0;
    for (int64_t s = 1; s <= num_samples; s++) {
        // a sample is seq_length + 1 tokens; consecutive samples overlap by one token (the label shift)
        int32_t want = seq_length + 1;
        while (want != 0) {
            const int32_t avail = sizes[doc_idx[pos]] - offset;
            want -= avail;
            if (want <= 0) {
                offset += want + avail - 1;  // stop ON the last token taken: the next sample starts there
                want = 0;
            } else {
                ++pos;
                offset = 0;
            }
        }
        out[2 * s] = (int32_t)pos;
        out[2 * s + 1] = offset;
    }
    return 0;
}

extern "C" int db1_build_rl_sample_idx(const int32_t* path_lengths, int64_t n_paths, int32_t transition_num, int32_t* out, int64_t* n_rows) {
    if (!path_lengths || !n_rows) return fail(-2, "build_rl_sample_idx: null argument");
    int64_t rows = 0;
    for (int64_t i = 0; i < n_paths; i++) rows += path_lengths[i] - 1;
    *n_rows = rows;
    if (!out) return 0;
    int64_t r = 0;
    for (int64_t i = 0; i < n_paths; i++) {
        const int32_t len = path_lengths[i];
        for (int32_t j = 0; j < len - 1; j++, r++) {
            out[3 * r] = (int32_t)i;
            out[3 * r + 1] = j;
            out[3 * r + 2] = j + transition_num < len ? j + transition_num : len;
        }
    }
    return 0;
}

extern "C" int db1_build_blending_indices(uint8_t* dataset_index, int64_t* dataset_sample_index, const double* weights, int32_t num_datasets,
                                          int64_t size) {
    if (!dataset_index || !dataset_sample_index || !weights) return fail(-2, "build_blending_indices: null argument");
    if (num_datasets <= 0 || num_datasets > 255) return fail(-4, "build_blending_indices: 1..255 datasets");
    int64_t taken[256] = {0};
    for (int64_t s = 0; s < size; s++) {
        const double n = s > 0 ? (double)s : 1.0;
        int32_t best = 0;
        double best_err = weights[0] * n - (double)taken[0];
        for (int32_t d = 1; d < num_datasets; d++) {
            const double err = weights[d] * n - (double)taken[d];
            if (err > best_err) { best_err = err; best = d; }
        }
        dataset_index[s] = (uint8_t)best;
        dataset_sample_index[s] = taken[best]++;
    }
    return 0;
}
