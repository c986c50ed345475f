// Wave-level exact selection of the k largest 64-bit keys held in registers.
//
// A key packs one candidate:  (order-preserving score bits << 32) | ~row.
// Larger key == better candidate: higher score first, lower row on equal
// scores -- the reference's np.argmax / stable-argsort rule
// (pipeline/utils/local_rag.py:78).  Rows are unique, so keys are unique and the
// k-th largest key is a sharp cut.  Key 0 marks an empty slot.
//
// Selection is a most-significant-bit-first radix walk: 32 steps over the
// score word (one ballot per held key and step), and 32 more over the row word
// only when equal scores straddle the cut.
#pragma once
#include <hip/hip_runtime.h>

namespace mtr {

__device__ __forceinline__ unsigned int order_f32(float f) {
    const unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unorder_f32(unsigned int o) {
    const unsigned int u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}
__device__ __forceinline__ unsigned long long pack_key(float score, int row) {
    return ((unsigned long long)order_f32(score) << 32) | (unsigned int)(~row);
}
__device__ __forceinline__ float key_score(unsigned long long key) { return unorder_f32((unsigned int)(key >> 32)); }
__device__ __forceinline__ int key_row(unsigned long long key) { return (int)(~(unsigned int)key); }

// Returns the k-th largest of the wave's non-zero keys (E per lane), or 0 when
// there are at most k of them (then every key is among the k largest).
template <int E>
__device__ __forceinline__ unsigned long long wave_kth_largest(const unsigned long long (&key)[E], int k) {
    unsigned int hi[E], lo[E];
    int nvalid = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        hi[e] = (unsigned int)(key[e] >> 32);
        lo[e] = (unsigned int)key[e];
        nvalid += __popcll(__ballot(key[e] != 0ull));
    }
    if (nvalid <= k) return 0ull;
    unsigned int hp = 0;
    int need = k;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned int want = (hp | (1u << bit)) >> bit;  // prefix with this bit set (odd, so empty slots never match)
        int c = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) c += __popcll(__ballot((hi[e] >> bit) == want));
        if (c >= need) hp |= 1u << bit;
        else need -= c;
    }
    int group = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) group += __popcll(__ballot(hi[e] == hp));
    if (group == need) return (unsigned long long)hp << 32;  // the whole equal-score group is inside the cut
    unsigned int lp = 0;
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned int want = (lp | (1u << bit)) >> bit;
        int c = 0;
#pragma unroll
        for (int e = 0; e < E; ++e) c += __popcll(__ballot(hi[e] == hp && (lo[e] >> bit) == want));
        if (c >= need) lp |= 1u << bit;
        else need -= c;
    }
    return ((unsigned long long)hp << 32) | lp;
}

}  // namespace mtr
