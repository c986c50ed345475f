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

// Histogram form of the same selection: eight passes of 8 key bits, most
// significant first.  Each pass counts the still-undecided keys per byte value
// in a 256-word LDS histogram (`hist`, private to the wave), finds the byte
// value that holds the k-th largest, and narrows to it.
//
// slack > 0 allows an early stop: as soon as keeping the whole current bin
// would keep at most k + slack keys, the cut is the bin's lower bound.
// Returns the cut (keep keys >= cut; 0 when there are at most k keys);
// *passes_done tells how many key bytes of the cut are exact.
template <int E>
__device__ __forceinline__ unsigned long long wave_kth_largest_hist(const unsigned long long (&key)[E], int k,
                                                                    int slack, unsigned int* hist, int lane,
                                                                    int* passes_done) {
    bool act[E];
    int nact = 0;
#pragma unroll
    for (int e = 0; e < E; ++e) {
        act[e] = key[e] != 0ull;
        nact += __popcll(__ballot(act[e]));
    }
    *passes_done = 8;
    if (nact <= k) return 0ull;
    int need = k;
    unsigned long long prefix = 0ull;
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        reinterpret_cast<u32x4*>(hist)[lane] = u32x4{0u, 0u, 0u, 0u};
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < E; ++e)
            if (act[e]) atomicAdd(&hist[(unsigned int)(key[e] >> shift) & 255u], 1u);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const u32x4 h = reinterpret_cast<const u32x4*>(hist)[lane];  // bins 4*lane .. 4*lane+3
        const unsigned int mine = h[0] + h[1] + h[2] + h[3];
        unsigned int incl = mine;  // inclusive sum over lanes >= this one (higher lanes hold higher bins)
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned int t = __shfl_down(incl, off, 64);
            if (lane + off < 64) incl += t;
        }
        const unsigned int a3 = incl - mine, a2 = a3 + h[3], a1 = a2 + h[2], a0 = a1 + h[1];
        const unsigned int un = (unsigned int)need;
        int bin = -1;
        unsigned int above = 0, count = 0;
        if (a3 < un && un <= a3 + h[3]) { bin = 4 * lane + 3; above = a3; count = h[3]; }
        else if (a2 < un && un <= a2 + h[2]) { bin = 4 * lane + 2; above = a2; count = h[2]; }
        else if (a1 < un && un <= a1 + h[1]) { bin = 4 * lane + 1; above = a1; count = h[1]; }
        else if (a0 < un && un <= a0 + h[0]) { bin = 4 * lane; above = a0; count = h[0]; }
        const unsigned long long who = __ballot(bin >= 0);  // exactly one lane
        const int src = __ffsll((long long)who) - 1;
        const unsigned int b = (unsigned int)__builtin_amdgcn_readlane(bin, src);
        need -= __builtin_amdgcn_readlane((int)above, src);
        nact = __builtin_amdgcn_readlane((int)count, src);
        prefix |= (unsigned long long)b << shift;
#pragma unroll
        for (int e = 0; e < E; ++e) act[e] = act[e] && (((unsigned int)(key[e] >> shift) & 255u) == b);
        if (nact - need <= slack) {
            *passes_done = pass + 1;
            break;
        }
    }
    return prefix;
}

}  // namespace mtr
