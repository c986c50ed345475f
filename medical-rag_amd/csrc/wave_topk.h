// Wave-level exact top-k selection over a stream of (score, row id) pairs.
//
// One 64-lane wave owns a list of at most k entries in LDS, sorted by the
// total order  "a beats b  <=>  a.score > b.score || (a.score == b.score &&
// a.id < b.id)".  Row ids are unique, so the order is strict and every merge is
// deterministic: the final list is the exact top-k under the reference's rule
// (np.argmax / stable argsort: highest score first, lowest row on ties,
// pipeline/utils/local_rag.py:78) whatever order the pairs arrive in.
//
// push(): each lane offers at most one pair.  A pair is inserted only when it
// beats the current k-th entry, so after the list has filled most pushes cost
// one compare and one ballot.  Insertions compute, for every surviving list
// entry and every passing pair, its rank in the union and move it there in
// place (entries only ever move towards the tail, so sweeping the list from
// the tail forwards never overwrites an entry that has not been read).
#pragma once
#include <hip/hip_runtime.h>
#include <limits.h>

namespace mtr {

template <typename IdxT>
__device__ __forceinline__ bool beats(float s1, IdxT i1, float s2, IdxT i2) {
    return s1 > s2 || (s1 == s2 && i1 < i2);
}

template <typename IdxT> struct IdxMax;
template <> struct IdxMax<int> { static constexpr int value = INT_MAX; };
template <> struct IdxMax<long long> { static constexpr long long value = LLONG_MAX; };

// Broadcast of one lane's value; `lane` must be wave-uniform (it always comes
// from a ballot mask here), which makes this a v_readlane, not an LDS permute.
__device__ __forceinline__ float bcast_lane(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ int bcast_lane(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ long long bcast_lane(long long v, int lane) {
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), lane);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), lane);
    return ((long long)hi << 32) | (unsigned int)lo;
}

template <typename IdxT>
struct WaveTopK {
    float* ls;   // LDS, k floats
    IdxT* li;    // LDS, k ids
    int k;
    int cnt;     // wave-uniform
    float thr_s; // key of the k-th entry once cnt == k, else (-inf, max id)
    IdxT thr_i;

    __device__ __forceinline__ void init(float* s, IdxT* i, int k_) {
        ls = s; li = i; k = k_; cnt = 0;
        thr_s = -__builtin_huge_valf();
        thr_i = IdxMax<IdxT>::value;
    }

    // True when this lane's pair would enter the list.
    __device__ __forceinline__ bool passes(float s, IdxT i) const { return beats<IdxT>(s, i, thr_s, thr_i); }

    // Wave-collective.  `want`: this lane offers (s, i).  Pairs whose score is
    // -inf never enter (they cannot beat the initial threshold).
    __device__ __forceinline__ void push(float s, IdxT i, bool want) {
        const bool pass = want && passes(s, i);
        const unsigned long long mask = __ballot(pass);
        if (mask == 0ull) return;  // wave-uniform
        insert(s, i, pass, mask);
    }

    __device__ __forceinline__ void insert(float s, IdxT i, bool pass, unsigned long long mask) {
        const int lane = threadIdx.x & 63;
        int my_rank = 0;  // passing pair: number of union members that beat it
        const int nblk = (cnt + 63) >> 6;
        for (int t = nblk - 1; t >= 0; --t) {
            const int p = t * 64 + lane;
            const bool have = p < cnt;
            float es = -__builtin_huge_valf();
            IdxT ei = IdxMax<IdxT>::value;
            if (have) { es = ls[p]; ei = li[p]; }
            int shift = 0;
            unsigned long long m = mask;
            while (m) {  // wave-uniform loop over the passing lanes
                const int j = __ffsll((long long)m) - 1;
                m &= m - 1;
                const float js = bcast_lane(s, j);
                const IdxT ji = bcast_lane(i, j);
                const bool jb = have && beats<IdxT>(js, ji, es, ei);  // pair j beats this entry
                shift += jb ? 1 : 0;
                const unsigned long long ahead = __ballot(have && !jb);  // entries that beat pair j
                if (lane == j) my_rank += __popcll(ahead);
            }
            __builtin_amdgcn_wave_barrier();
            if (have) {
                const int np = p + shift;
                if (np < k) { ls[np] = es; li[np] = ei; }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // rank of each passing pair among the passing pairs
        {
            unsigned long long m = mask;
            while (m) {
                const int j = __ffsll((long long)m) - 1;
                m &= m - 1;
                const float js = bcast_lane(s, j);
                const IdxT ji = bcast_lane(i, j);
                if (pass && beats<IdxT>(js, ji, s, i)) my_rank += 1;
            }
        }
        if (pass && my_rank < k) { ls[my_rank] = s; li[my_rank] = i; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int nc = cnt + __popcll(mask);
        cnt = nc < k ? nc : k;
        if (cnt == k) {
            // every lane reads the same LDS word (broadcast)
            thr_s = ls[k - 1];
            thr_i = li[k - 1];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
};

}  // namespace mtr
