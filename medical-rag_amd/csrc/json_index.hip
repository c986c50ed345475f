// Reader of the reference's index file: one JSON list of page objects, each
// with an "embedding" array of d numbers next to its other keys
// (written at pipeline/utils/embed.py:74-79,110-113; the reference parses it
// with json.load into Python lists, pipeline/utils/local_rag.py:96-98, ten
// seconds for 4032 x 4096).  Host code only.
//
// Two passes over the text, which the caller keeps mapped:
//   scan   one thread walks the structure, notes where every page's
//          "embedding" array starts and ends, and builds the "rest": the same
//          text with each of those arrays replaced by null, a few KB a page,
//          which the caller hands to its own JSON parser for the page dicts;
//   read   the arrays are converted to doubles (or floats) straight into the
//          caller's matrix, pages split between threads.
// Numbers follow Python's json module: the JSON grammar plus NaN, Infinity and
// -Infinity, converted with correct rounding, so the matrix equals what
// json.load + numpy.array(float64) give.  Anything else this reader does not
// expect (not a list of objects, a page without the key or with it twice, an
// array that holds something other than numbers, rows of different lengths,
// trailing text) makes it decline with MTR_ERR_INVALID: the caller then takes
// the general parser and gets that parser's behaviour for the odd file.
#include <locale.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <cmath>
#include <limits>
#include <string>
#include <thread>
#include <vector>

#include "../../include/medtutor_retrieval.h"

namespace mtr {
int set_error(int code, const char* fmt, ...);  // mtr_api.hip: thread-local message behind mtr_last_error()
}

struct mtr_json_index {
    const char* text = nullptr;
    int64_t len = 0;
    int32_t d = 0;
    std::vector<std::pair<int64_t, int64_t>> arrays;  // [start, end) of each page's array, brackets included
    std::string rest;
};

namespace {

constexpr int kMaxDepth = 256;

inline bool is_ws(char c) { return c == ' ' || c == '\n' || c == '\r' || c == '\t'; }

struct Scanner {
    const char* p;
    const char* end;

    void ws() {
        while (p < end && is_ws(*p)) ++p;
    }
    // p at the opening quote; leaves p behind the closing one
    bool string() {
        ++p;
        while (p < end) {
            const char* q = static_cast<const char*>(memchr(p, '"', (size_t)(end - p)));
            if (!q) return false;
            // an odd run of backslashes before the quote escapes it
            const char* b = q;
            while (b > p && b[-1] == '\\') --b;
            p = q + 1;
            if (((q - b) & 1) == 0) return true;
        }
        return false;
    }
    // Skips any value; its content is checked later by the caller's parser.
    bool value(int depth) {
        if (depth > kMaxDepth || p >= end) return false;
        const char c = *p;
        if (c == '"') return string();
        if (c == '{' || c == '[') {
            const char close = c == '{' ? '}' : ']';
            ++p;
            ws();
            if (p < end && *p == close) { ++p; return true; }
            for (;;) {
                ws();
                if (c == '{') {
                    if (p >= end || *p != '"' || !string()) return false;
                    ws();
                    if (p >= end || *p != ':') return false;
                    ++p;
                    ws();
                }
                if (!value(depth + 1)) return false;
                ws();
                if (p >= end) return false;
                if (*p == ',') { ++p; continue; }
                if (*p == close) { ++p; return true; }
                return false;
            }
        }
        // number or literal: up to the next delimiter
        const char* s = p;
        while (p < end && !is_ws(*p) && *p != ',' && *p != ']' && *p != '}') ++p;
        return p > s;
    }
};

const char* scan_index(mtr_json_index* ix) {
    Scanner s{ix->text, ix->text + ix->len};
    const char* const base = ix->text;
    const char* copied = base;
    s.ws();
    if (s.p >= s.end || *s.p != '[') return "the file is not a JSON list";
    ++s.p;
    s.ws();
    if (s.p < s.end && *s.p == ']') {
        ++s.p;
    } else {
        for (;;) {
            s.ws();
            if (s.p >= s.end || *s.p != '{') return "a list element is not an object";
            ++s.p;
            bool have = false;
            s.ws();
            if (s.p < s.end && *s.p == '}') return "a page has no \"embedding\"";
            for (;;) {
                s.ws();
                if (s.p >= s.end || *s.p != '"') return "malformed object";
                const char* key = s.p + 1;
                if (!s.string()) return "unterminated string";
                const bool is_embedding = (s.p - 1 - key) == 9 && memcmp(key, "embedding", 9) == 0;
                s.ws();
                if (s.p >= s.end || *s.p != ':') return "malformed object";
                ++s.p;
                s.ws();
                if (is_embedding) {
                    if (have) return "a page has two \"embedding\" keys";
                    if (s.p >= s.end || *s.p != '[') return "an \"embedding\" is not an array";
                    const char* close = static_cast<const char*>(memchr(s.p, ']', (size_t)(s.end - s.p)));
                    if (!close) return "unterminated array";
                    ix->arrays.emplace_back(s.p - base, close + 1 - base);
                    ix->rest.append(copied, (size_t)(s.p - copied));
                    ix->rest.append("null");
                    copied = s.p = close + 1;
                    have = true;
                } else if (!s.value(2)) {
                    return "malformed value";
                }
                s.ws();
                if (s.p >= s.end) return "unterminated object";
                if (*s.p == ',') { ++s.p; continue; }
                if (*s.p == '}') { ++s.p; break; }
                return "malformed object";
            }
            if (!have) return "a page has no \"embedding\"";
            s.ws();
            if (s.p >= s.end) return "unterminated list";
            if (*s.p == ',') { ++s.p; continue; }
            if (*s.p == ']') { ++s.p; break; }
            return "malformed list";
        }
    }
    s.ws();
    if (s.p != s.end) return "text after the list";
    ix->rest.append(copied, (size_t)(s.end - copied));
    return nullptr;
}

locale_t c_locale() {
    static locale_t loc = newlocale(LC_ALL_MASK, "C", (locale_t)0);
    return loc;
}

const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                           1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

// One number of Python's json grammar at p (which lies inside "[...]", so a
// delimiter follows every token).  Leaves p behind it.
bool number(const char*& p, const char* end, double* out) {
    const char* s = p;
    if (end - p >= 3 && memcmp(p, "NaN", 3) == 0) {
        p += 3;
        *out = std::numeric_limits<double>::quiet_NaN();
        return true;
    }
    bool neg = false;
    if (p < end && *p == '-') { neg = true; ++p; }
    if (end - p >= 8 && memcmp(p, "Infinity", 8) == 0) {
        p += 8;
        *out = neg ? -std::numeric_limits<double>::infinity() : std::numeric_limits<double>::infinity();
        return true;
    }
    if (p >= end || *p < '0' || *p > '9') return false;
    uint64_t w = 0;      // the significant digits while they fit
    int digits = 0;      // significant digits seen (leading zeros do not count)
    int dropped = 0;     // integer digits that no longer fit w
    int frac = 0;        // fraction digits taken into w
    bool exact = true;   // w holds every significant digit
    if (*p == '0') {
        ++p;
        if (p < end && *p >= '0' && *p <= '9') return false;   // no leading zeros
    } else {
        while (p < end && *p >= '0' && *p <= '9') {
            if (digits < 19) { w = w * 10 + (uint64_t)(*p - '0'); ++digits; }
            else { ++dropped; exact = false; }
            ++p;
        }
    }
    if (p < end && *p == '.') {
        ++p;
        if (p >= end || *p < '0' || *p > '9') return false;
        while (p < end && *p >= '0' && *p <= '9') {
            if (digits < 19) {
                if (digits > 0 || *p != '0') { w = w * 10 + (uint64_t)(*p - '0'); ++digits; }
                ++frac;
            } else if (*p != '0') {
                exact = false;
            }
            ++p;
        }
    }
    long e10 = 0;
    if (p < end && (*p == 'e' || *p == 'E')) {
        ++p;
        bool eneg = false;
        if (p < end && (*p == '+' || *p == '-')) { eneg = *p == '-'; ++p; }
        if (p >= end || *p < '0' || *p > '9') return false;
        while (p < end && *p >= '0' && *p <= '9') {
            if (e10 < 100000) e10 = e10 * 10 + (*p - '0');
            ++p;
        }
        if (eneg) e10 = -e10;
    }
    e10 += dropped - frac;
    // Exact when the digits fit a double's 53 bits and the power of ten is itself
    // exact: then one correctly rounded multiply or divide gives the nearest double.
    if (exact && w <= (1ull << 53) && e10 >= -22 && e10 <= 22) {
        double v = (double)w;
        v = e10 < 0 ? v / kPow10[-e10] : v * kPow10[e10];
        *out = neg ? -v : v;
        return true;
    }
    char* stop = nullptr;
    *out = strtod_l(s, &stop, c_locale());   // correctly rounded; the token's grammar was checked above
    return stop == p;
}

// Converts one page's array into `d` values at out; false when it is not
// exactly d numbers.
template <typename T>
bool read_array(const char* p, const char* end, int d, T* out) {
    ++p;        // '['
    --end;      // ']'
    for (int i = 0; i < d; ++i) {
        while (p < end && is_ws(*p)) ++p;
        double v;
        if (!number(p, end, &v)) return false;
        out[i] = (T)v;
        while (p < end && is_ws(*p)) ++p;
        if (i + 1 < d) {
            if (p >= end || *p != ',') return false;
            ++p;
        }
    }
    return p == end;
}

int count_numbers(const char* p, const char* end) {
    int commas = 0;
    bool any = false;
    for (const char* q = p + 1; q < end - 1; ++q) {
        if (*q == ',') ++commas;
        else if (!is_ws(*q)) any = true;
    }
    return any ? commas + 1 : 0;
}

template <typename T>
bool read_all(const mtr_json_index* ix, T* matrix, int threads) {
    const int64_t n = (int64_t)ix->arrays.size();
    if (threads < 1) threads = 1;
    if ((int64_t)threads > n) threads = (int)n;
    std::atomic<bool> ok{true};
    auto work = [&](int64_t first, int64_t last) {
        for (int64_t r = first; r < last && ok.load(std::memory_order_relaxed); ++r) {
            const auto& span = ix->arrays[(size_t)r];
            if (!read_array<T>(ix->text + span.first, ix->text + span.second, ix->d, matrix + r * ix->d))
                ok.store(false, std::memory_order_relaxed);
        }
    };
    if (threads <= 1) {
        work(0, n);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < threads; ++t) pool.emplace_back(work, n * t / threads, n * (t + 1) / threads);
        for (auto& th : pool) th.join();
    }
    return ok.load();
}

}  // namespace

extern "C" {

int mtr_json_index_scan(const char* text, int64_t len, mtr_json_index_t** out) {
    if (!out) return mtr::set_error(MTR_ERR_INVALID, "out is NULL");
    *out = nullptr;
    if (!text || len < 0) return mtr::set_error(MTR_ERR_INVALID, "bad text");
    mtr_json_index* ix = nullptr;
    try {
        ix = new mtr_json_index;
        ix->text = text;
        ix->len = len;
        if (const char* why = scan_index(ix)) {
            delete ix;
            return mtr::set_error(MTR_ERR_INVALID, "not an index file this reader takes: %s", why);
        }
        if (!ix->arrays.empty()) {
            const auto& first = ix->arrays.front();
            ix->d = count_numbers(text + first.first, text + first.second);
            if (ix->d <= 0) {
                delete ix;
                return mtr::set_error(MTR_ERR_INVALID, "not an index file this reader takes: empty \"embedding\"");
            }
        }
    } catch (const std::bad_alloc&) {
        delete ix;
        return mtr::set_error(MTR_ERR_INVALID, "out of host memory while scanning the index file");
    }
    *out = ix;
    return MTR_OK;
}

int mtr_json_index_info(const mtr_json_index_t* ix, int64_t* n_pages, int32_t* d, int64_t* rest_len) {
    if (!ix) return mtr::set_error(MTR_ERR_INVALID, "handle is NULL");
    if (n_pages) *n_pages = (int64_t)ix->arrays.size();
    if (d) *d = ix->d;
    if (rest_len) *rest_len = (int64_t)ix->rest.size();
    return MTR_OK;
}

int mtr_json_index_read(const mtr_json_index_t* ix, void* matrix, int32_t dtype, char* rest, int32_t threads) {
    if (!ix) return mtr::set_error(MTR_ERR_INVALID, "handle is NULL");
    if (dtype != MTR_F32 && dtype != MTR_F64) return mtr::set_error(MTR_ERR_INVALID, "matrix dtype must be MTR_F32 or MTR_F64");
    if (rest) memcpy(rest, ix->rest.data(), ix->rest.size());
    if (ix->arrays.empty()) return MTR_OK;
    if (!matrix) return mtr::set_error(MTR_ERR_INVALID, "matrix is NULL");
    bool ok = false;
    try {
        ok = dtype == MTR_F64 ? read_all<double>(ix, static_cast<double*>(matrix), threads)
                              : read_all<float>(ix, static_cast<float*>(matrix), threads);
    } catch (const std::exception& e) {
        return mtr::set_error(MTR_ERR_INVALID, "reading the index file: %s", e.what());
    }
    if (!ok)
        return mtr::set_error(MTR_ERR_INVALID,
                              "not an index file this reader takes: an \"embedding\" is not %d numbers", ix->d);
    return MTR_OK;
}

void mtr_json_index_free(mtr_json_index_t* ix) { delete ix; }

}  // extern "C"
