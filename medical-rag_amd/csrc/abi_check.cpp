// Host-side exerciser of the C ABI for the sanitizer build (make asan): every
// entry point is called with invalid and boundary arguments, and, when a GPU is
// present, a small index goes through create / search / read back / destroy.
// It needs no GPU to be useful: without one every device call must fail with an
// error code and a message, never with a crash or a leak.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/medtutor_retrieval.h"
#include "mtr_device.h"   // launch plans (host functions): their invariants are checked below

static int failures = 0;
#define EXPECT(cond)                                                       \
    do {                                                                   \
        if (!(cond)) {                                                     \
            ++failures;                                                    \
            fprintf(stderr, "abi_check: %s failed at line %d (last error: %s)\n", #cond, __LINE__, mtr_last_error()); \
        }                                                                  \
    } while (0)

// The index-file reader works on mapped text that ends without a terminator:
// every prefix of a sample, and the sample with each byte damaged in turn, goes
// through it in a heap block of exactly that size, so that a read past the end
// is a sanitizer report.
static void json_reader_checks() {
    const char* sample =
        " [ {\"index\": 1, \"text\": \"a \\\" ] } \\\\\", \"embedding\": [1, -2.5e-3, 12345678901234567890.125, NaN, -Infinity],"
        " \"more\": {\"embedding\": [\"x\"], \"l\": [1, [2]]}},\n {\"embedding\":[0,0.1,1E5,Infinity,-0], \"index\": 2} ] \n";
    const size_t full = strlen(sample);
    mtr_json_index_t* js = nullptr;
    EXPECT(mtr_json_index_scan(nullptr, 0, &js) == MTR_ERR_INVALID && js == nullptr);
    EXPECT(mtr_json_index_scan(sample, (int64_t)full, nullptr) == MTR_ERR_INVALID);
    EXPECT(mtr_json_index_info(nullptr, nullptr, nullptr, nullptr) == MTR_ERR_INVALID);
    EXPECT(mtr_json_index_read(nullptr, nullptr, MTR_F64, nullptr, 1) == MTR_ERR_INVALID);
    mtr_json_index_free(nullptr);
    int accepted = 0;
    for (int damage = -1; damage < (int)full; ++damage) {
        for (size_t len = damage < 0 ? 0 : full; len <= full; ++len) {
            char* text = static_cast<char*>(malloc(len ? len : 1));
            memcpy(text, sample, len);
            if (damage >= 0) text[damage] = "]\"\\,9e-"[damage % 7];
            if (mtr_json_index_scan(text, (int64_t)len, &js) == MTR_OK) {
                int64_t n = 0, rest_len = 0;
                int32_t d = 0;
                EXPECT(mtr_json_index_info(js, &n, &d, &rest_len) == MTR_OK);
                std::vector<double> matrix((size_t)n * d + 1);
                std::vector<char> rest((size_t)rest_len + 1);
                EXPECT(mtr_json_index_read(js, matrix.data(), 9, rest.data(), 2) == MTR_ERR_INVALID);
                if (mtr_json_index_read(js, matrix.data(), MTR_F64, rest.data(), 2) == MTR_OK && damage < 0) {
                    ++accepted;
                    EXPECT(len == full || len == full - 1 || len == full - 2);   // only the trailing blanks may go
                    EXPECT(n == 2 && d == 5 && matrix[1] == -2.5e-3 && matrix[2] == 12345678901234567890.125 &&
                           matrix[3] != matrix[3] && matrix[7] == 1e5 && matrix[8] > 1e308);
                    std::vector<float> as_f32((size_t)n * d);
                    EXPECT(mtr_json_index_read(js, as_f32.data(), MTR_F32, nullptr, 64) == MTR_OK && as_f32[6] == 0.1f);
                }
                mtr_json_index_free(js);
            } else {
                EXPECT(js == nullptr && strlen(mtr_last_error()) > 0);
            }
            free(text);
        }
    }
    EXPECT(accepted == 3);
}

// What the matrix-core kernels assume of a launch plan, over many shapes, for the
// full launch and for sampling launches over a part of the rows: the padded batch
// covers the batch, the LDS fits a CU, a tile's appends fit a candidate buffer, the
// tile width exists for the storage type, the row ranges cover every tile.
static void plan_checks() {
    const int bs[] = {1, 2, 5, 9, 16, 17, 32, 33, 34, 48, 49, 64, 65, 80, 81, 96, 97, 112, 113, 128, 129, 130, 200, 257, 1000, 4096};
    const long long ns[] = {1, 127, 128, 129, 1000, 4032, 20000, 60001, 131072, 1000000, 12500000};
    const int ks[] = {1, 10, 33, 100, 129, 256};
    const int lds[] = {32, 64, 128, 192, 768, 1024, 4096};
    long long checked = 0;
    for (int dtype : {0, 2})
        for (long long n : ns)
            for (int b : bs)
                for (int k : ks)
                    for (int ld : lds) {
                        mtr::MfmaPlan p;
                        if (mtr::plan_mfma(dtype, n, ld, b, k, 256, 0, &p) != 0) continue;
                        ++checked;
                        const bool sixteen = p.tnq % 32 != 0 || p.tnq == 96;
                        EXPECT(p.n_qtiles * p.tnq >= b && p.lds <= 160 * 1024 && p.cap >= 2 * p.tm && k + p.tm + p.slack <= p.cap &&
                               p.slack >= 0 && p.n_ranges >= 1 && p.tiles_per_range * p.n_ranges >= p.n_tiles &&
                               p.grid == p.n_qtiles * p.n_ranges && p.tnq % 16 == 0 && p.tnq >= 16 && p.tnq <= 128 &&
                               (dtype == 0 || !sixteen) && (p.tm == 128 || (p.tm == 256 && p.stages == 3 && dtype == 2)) &&
                               (!sixteen || (p.stages == 2 && p.tm == 128)));
                        // rows through registers: tall ring form only, whole 512-byte rows, never together with a claimed tail
                        EXPECT(!p.via_regs || (p.tm == 256 && p.stages == 3 && dtype == 2 && p.stream_rows && !p.dynamic_tail &&
                                               (p.tnq == 64 || p.tnq == 128) && (ld * 2) % 512 == 0));
                        mtr::WidePlan w;
                        if (mtr::plan_wide(dtype, n, ld, b, k, 40 * 64, 256, &w) == 0)
                            EXPECT((w.tq == 128 || w.tq == 192 || w.tq == 256) && w.n_qtiles * w.tq >= b &&
                                   (w.n_qtiles - 1) * 256 < b && w.grid == w.n_qtiles * w.n_lists && w.n_lists >= 1 &&
                                   (w.cap & (w.cap - 1)) == 0 && w.key_bytes == (size_t)w.grid * w.tq * mtr::kWideSubs * w.cap * 8 &&
                                   w.count_words == (size_t)w.grid * w.tq * mtr::kWideSubs && w.lds <= 160 * 1024);
                        for (int stride : {2, 13, 200}) {
                            const long long stiles = (n + 127) / 128, n_samp = (stiles + stride - 1) / stride;
                            mtr::MfmaPlan s;
                            EXPECT(mtr::plan_mfma(dtype, n_samp * 128, ld, b, k, 256, 1, &s) == 0 && s.tm == 128 && s.stages == 2 &&
                                   s.n_qtiles * s.tnq >= b && s.lds <= 160 * 1024);
                        }
                    }
    EXPECT(checked > 10000);
    // The forms the benchmark shapes are measured on (default switches, 256 CUs).
    {
        mtr::MfmaPlan p;
        EXPECT(mtr::plan_mfma(2, 10000000, 1024, 128, 10, 256, 0, &p) == 0 && p.tnq == 128 && p.tm == 256 && p.stages == 3 &&
               p.stream_rows && p.interleave && p.via_regs == 2 && !p.dynamic_tail && p.n_qtiles == 1);
        EXPECT(mtr::plan_mfma(2, 10000000, 1024, 96, 10, 256, 0, &p) == 0 && p.tnq == 128 && p.via_regs == 2);
        EXPECT(mtr::plan_mfma(2, 10000000, 1024, 64, 10, 256, 0, &p) == 0 && p.tnq == 64 && p.tm == 256 && p.via_regs == 2 && !p.dynamic_tail);
        EXPECT(mtr::plan_mfma(2, 12500000, 768, 64, 10, 256, 0, &p) == 0 && p.tnq == 64 && p.tm == 256 && p.dynamic_tail && !p.via_regs);
        EXPECT(mtr::plan_mfma(2, 10000000, 1024, 32, 10, 256, 0, &p) == 0 && p.tnq == 32 && p.dynamic_tail && !p.via_regs);
        EXPECT(mtr::plan_mfma(2, 12500000, 768, 128, 10, 256, 0, &p) == 0 && p.tnq == 128 && p.via_regs == 2);
        EXPECT(mtr::plan_mfma(2, 4032, 1024, 100, 10, 256, 0, &p) == 0 && !p.stream_rows && !p.via_regs);   // a shard within the caches
        EXPECT(mtr::plan_mfma(0, 1000000, 768, 34, 10, 256, 0, &p) == 0 && p.tnq == 48 && !p.via_regs);     // fp32: 16 x 16 tiles, two buffers
        mtr::WidePlan w;
        EXPECT(mtr::plan_wide(2, 10000000, 1024, 129, 10, 2000, 256, &w) == 0 && w.tq == 192 && w.n_qtiles == 1);
        EXPECT(mtr::plan_wide(2, 10000000, 1024, 256, 10, 2000, 256, &w) == 0 && w.tq == 256 && w.n_qtiles == 1);
        EXPECT(mtr::plan_wide(2, 1000000, 1024, 384, 10, 2000, 256, &w) == 0 && w.tq == 192 && w.n_qtiles == 2);
        EXPECT(mtr::plan_wide(2, 1000000, 1024, 4096, 100, 8000, 256, &w) == 0 && w.tq == 256 && w.n_qtiles == 16 && w.n_lists == 16);
    }
}

int main() {
    printf("%s\n", mtr_version());
    json_reader_checks();
    plan_checks();
    mtr_index_t* ix = nullptr;
    EXPECT(mtr_index_alloc(0, 8, MTR_F32, 1, 0, 0, &ix) == MTR_ERR_INVALID && ix == nullptr);
    EXPECT(strlen(mtr_last_error()) > 0);
    EXPECT(mtr_index_alloc(8, 8, 7, 1, 0, 0, &ix) == MTR_ERR_INVALID);
    EXPECT(mtr_index_alloc(8, 8, MTR_F32, 1, 0, 0, nullptr) == MTR_ERR_INVALID);
    EXPECT(mtr_index_search(nullptr, nullptr, 0, 0, 1, 1, 0, nullptr, nullptr) == MTR_ERR_INVALID);
    EXPECT(mtr_index_search_device(nullptr, nullptr, 0, 1, 1, 0, nullptr, nullptr, nullptr) == MTR_ERR_INVALID);
    EXPECT(mtr_index_search_device_noalloc(nullptr, nullptr, 0, 1, 1, 0, nullptr, nullptr, nullptr) == MTR_ERR_INVALID);
    EXPECT(mtr_index_info(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == MTR_ERR_INVALID);
    EXPECT(mtr_index_load_rows(nullptr, 0, nullptr, 0, 0, 0) == MTR_ERR_INVALID);
    EXPECT(mtr_index_read_rows(nullptr, 0, 0, nullptr) == MTR_ERR_INVALID);
    EXPECT(mtr_index_nonfinite(nullptr, nullptr) == MTR_ERR_INVALID);
    EXPECT(mtr_index_profile(nullptr, 1) == MTR_ERR_INVALID);
    EXPECT(mtr_index_read_ceiling(nullptr, 1, nullptr) == MTR_ERR_INVALID);
    {
        char name[8];
        EXPECT(mtr_index_last_kernel(nullptr, name, 8) == MTR_ERR_INVALID);
    }
    EXPECT(mtr_merge_topk_device(nullptr, nullptr, 1, 0, 0, 1, 1, 1, nullptr, nullptr, 0, nullptr) == MTR_ERR_INVALID);
    EXPECT(mtr_pool_normalize(nullptr, 0, nullptr, 1, 1, 1, nullptr, 0, nullptr) == MTR_ERR_INVALID);
    EXPECT(mtr_comm_create(nullptr, 0, 1, 0, nullptr) == MTR_ERR_INVALID);
    mtr_comm_t* comm = nullptr;
    EXPECT(mtr_comm_create(nullptr, 0, 1, 0, &comm) == MTR_ERR_INVALID && comm == nullptr);
    EXPECT(mtr_comm_exchange_merge_device(nullptr, nullptr, nullptr, 1, 1, 1, nullptr, nullptr, nullptr) == MTR_ERR_INVALID);
    EXPECT(mtr_sharded_search_device(nullptr, nullptr, nullptr, 0, 1, 1, 0, nullptr, nullptr, nullptr) == MTR_ERR_INVALID);
    mtr_comm_destroy(nullptr);
    mtr_index_destroy(nullptr);
    // one index over several GPUs, driven by this process
    mtr_multi_index_t* mi = nullptr;
    const int32_t devs[3] = {0, 0, 0}, bad_devs[2] = {0, -1};
    EXPECT(mtr_multi_index_alloc(10, 8, MTR_F32, 1, devs, 3, MTR_EXCHANGE_COPY, nullptr) == MTR_ERR_INVALID);
    EXPECT(mtr_multi_index_alloc(10, 8, MTR_F32, 1, nullptr, 3, MTR_EXCHANGE_COPY, &mi) == MTR_ERR_INVALID && mi == nullptr);
    EXPECT(mtr_multi_index_alloc(10, 8, MTR_F32, 1, devs, 0, MTR_EXCHANGE_COPY, &mi) == MTR_ERR_INVALID);
    EXPECT(mtr_multi_index_alloc(10, 8, MTR_F32, 1, devs, 65, MTR_EXCHANGE_COPY, &mi) == MTR_ERR_INVALID);
    EXPECT(mtr_multi_index_alloc(0, 8, MTR_F32, 1, devs, 3, MTR_EXCHANGE_COPY, &mi) == MTR_ERR_INVALID);
    EXPECT(mtr_multi_index_alloc(10, 8, MTR_F32, 1, devs, 3, 9, &mi) == MTR_ERR_INVALID);
    EXPECT(mtr_multi_index_alloc(10, 8, MTR_F32, 1, bad_devs, 2, MTR_EXCHANGE_COPY, &mi) != MTR_OK && mi == nullptr);
    EXPECT(mtr_multi_index_load_rows(nullptr, 0, nullptr, 0, MTR_F32) == MTR_ERR_INVALID);
    EXPECT(mtr_multi_index_search(nullptr, nullptr, 0, 1, 1, 0, nullptr, nullptr) == MTR_ERR_INVALID);
    EXPECT(mtr_multi_index_search_device(nullptr, nullptr, 0, 1, 1, 0, nullptr, nullptr, nullptr) == MTR_ERR_INVALID);
    EXPECT(mtr_multi_index_nonfinite(nullptr, nullptr) == MTR_ERR_INVALID);
    EXPECT(mtr_multi_index_info(nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) == MTR_ERR_INVALID);
    EXPECT(mtr_multi_index_block(nullptr, 0, nullptr, nullptr, nullptr, nullptr) == MTR_ERR_INVALID);
    EXPECT(mtr_multi_index_read_rows(nullptr, 0, 0, nullptr) == MTR_ERR_INVALID);
    mtr_multi_index_destroy(nullptr);
    int32_t v = -7;
    EXPECT(mtr_tuning_get("wide_min_batch", &v) == MTR_OK && v == 129);
    EXPECT(mtr_tuning_set("wide_min_batch", 200) == MTR_OK && mtr_tuning_get("wide_min_batch", &v) == MTR_OK && v == 200);
    EXPECT(mtr_tuning_set("wide_min_batch", 129) == MTR_OK);
    EXPECT(mtr_tuning_set("no_such_switch", 1) == MTR_ERR_INVALID);
    EXPECT(mtr_tuning_get(nullptr, &v) == MTR_ERR_INVALID);

    // With a GPU: one small index end to end; without: a clean error.
    const int n = 300, d = 40, b = 3, k = 5;
    std::vector<float> rows((size_t)n * d), q((size_t)b * d);
    for (size_t i = 0; i < rows.size(); ++i) {   // a hash of i: no two rows alike
        uint32_t x = (uint32_t)i * 2654435761u + 0x9e3779b9u;
        x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
        rows[i] = (float)(x % 2001) / 1000.f - 1.f;
    }
    for (int j = 0; j < b; ++j) memcpy(&q[(size_t)j * d], &rows[(size_t)(17 * j + 5) * d], d * sizeof(float));
    int rc = mtr_index_create(rows.data(), n, d, MTR_F32, MTR_F32, 1, 0, 0, &ix);
    if (rc == MTR_OK) {
        std::vector<int64_t> idx((size_t)b * k);
        std::vector<float> score((size_t)b * k);
        EXPECT(mtr_index_search(ix, q.data(), MTR_F32, 0, b, k, MTR_COSINE, idx.data(), score.data()) == MTR_OK);
        for (int j = 0; j < b; ++j) EXPECT(idx[(size_t)j * k] == 17 * j + 5);
        EXPECT(mtr_index_search(ix, q.data(), MTR_F32, 0, b, MTR_MAX_K + 1, MTR_COSINE, idx.data(), score.data()) == MTR_ERR_INVALID);
        std::vector<float> back((size_t)2 * d);
        EXPECT(mtr_index_read_rows(ix, 1, 2, back.data()) == MTR_OK);
        EXPECT(mtr_index_read_rows(ix, n - 1, 2, back.data()) == MTR_ERR_INVALID);
        q[3] = 1.0f / 0.0f;
        EXPECT(mtr_index_search(ix, q.data(), MTR_F32, 0, b, k, MTR_COSINE, idx.data(), score.data()) == MTR_ERR_NONFINITE);
        mtr_index_destroy(ix);
        // the same rows as three blocks on this GPU (100 rows each), and as five blocks of 2 rows: one stays empty
        q[3] = rows[(size_t)5 * d + 3];
        for (int blocks : {3, 5}) {
            const int32_t five[5] = {0, 0, 0, 0, 0};
            const int64_t rows_here = blocks == 3 ? n : 4;
            EXPECT(mtr_multi_index_create(rows.data(), rows_here, d, MTR_F32, MTR_F32, 1, five, blocks, MTR_EXCHANGE_AUTO, &mi) == MTR_OK);
            if (!mi) break;
            int32_t n_blocks = 0, form = -1, dev = -1;
            int64_t lo = -1, hi = -1;
            mtr_index_t* blk = nullptr;
            EXPECT(mtr_multi_index_info(mi, nullptr, nullptr, nullptr, nullptr, &n_blocks, &form) == MTR_OK && n_blocks == blocks &&
                   form == MTR_EXCHANGE_COPY);
            EXPECT(mtr_multi_index_block(mi, blocks - 1, &blk, &dev, &lo, &hi) == MTR_OK && dev == 0 && hi == rows_here &&
                   (blocks == 3 ? (lo == 200 && blk != nullptr) : (lo == 4 && blk == nullptr)));
            EXPECT(mtr_multi_index_block(mi, blocks, nullptr, nullptr, nullptr, nullptr) == MTR_ERR_INVALID);
            EXPECT(mtr_multi_index_search(mi, q.data(), MTR_F32, b, k, MTR_COSINE, idx.data(), score.data()) == MTR_OK);
            if (blocks == 3)
                for (int j = 0; j < b; ++j) EXPECT(idx[(size_t)j * k] == 17 * j + 5);
            else
                EXPECT(idx[4] == -1 && idx[3] >= 0);   // four rows, five result slots
            EXPECT(mtr_multi_index_read_rows(mi, rows_here - 2, 2, back.data()) == MTR_OK);
            EXPECT(mtr_multi_index_read_rows(mi, rows_here - 1, 2, back.data()) == MTR_ERR_INVALID);
            EXPECT(mtr_multi_index_load_rows(mi, rows_here - 1, rows.data(), 2, MTR_F32) == MTR_ERR_INVALID);
            q[3] = 0.0f / 0.0f;
            EXPECT(mtr_multi_index_search(mi, q.data(), MTR_F32, b, k, MTR_COSINE, idx.data(), score.data()) == MTR_ERR_NONFINITE);
            q[3] = rows[(size_t)5 * d + 3];
            mtr_multi_index_destroy(mi);
            mi = nullptr;
        }
        printf("abi_check: GPU path exercised\n");
    } else {
        EXPECT(ix == nullptr && strlen(mtr_last_error()) > 0);
        printf("abi_check: no usable GPU (%s); argument checks only\n", mtr_last_error());
    }
    printf("abi_check: %d failure(s)\n", failures);
    return failures ? 1 : 0;
}
