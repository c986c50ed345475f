// Candidate merge: the best k of several sorted-or-not candidate lists per
// query, ordered by (score descending, row id ascending).
//
// Used twice on the path: (1) after the row scan, to merge the per-workgroup
// lists of one shard, and (2) after the all-gather of the row-sharded search,
// to merge the per-GPU lists.  The reference's counterpart is the single
// np.argmax at pipeline/utils/local_rag.py:78.
//
// One workgroup per query.  Each wave streams a strided slice of the
// candidates through its own top-k list (wave_topk.h), then wave 0 folds the
// other waves' lists into its own and writes the result.
#include "mtr_device.h"
#include "wave_topk.h"

namespace mtr {

namespace {

struct MergeArgs {
    const float* cs;
    const int* ci32;
    const long long* ci64;
    int n_lists;
    long long list_stride;
    long long query_stride;
    int k_in;
    int k_out;
    long long idx_offset;
    long long* out_i;
    float* out_s;
};

__global__ void merge_topk_kernel(MergeArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nwaves = blockDim.x >> 6;
    long long* list_i = reinterpret_cast<long long*>(smem);
    float* list_s = reinterpret_cast<float*>(list_i + (size_t)nwaves * a.k_out);
    int* cnt_lds = reinterpret_cast<int*>(list_s + (size_t)nwaves * a.k_out);

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q = blockIdx.x;

    WaveTopK<long long> tk;
    tk.init(list_s + (size_t)wave * a.k_out, list_i + (size_t)wave * a.k_out, a.k_out);

    const long long total = (long long)a.n_lists * a.k_in;
    const long long qbase = (long long)q * a.query_stride;
    for (long long m0 = (long long)wave * 64; m0 < total; m0 += (long long)nwaves * 64) {
        const long long m = m0 + lane;
        bool ok = m < total;
        float s = 0.f;
        long long id = -1;
        if (ok) {
            const long long l = m / a.k_in;
            const long long j = m - l * a.k_in;
            const long long off = l * a.list_stride + qbase + j;
            s = a.cs[off];
            id = a.ci32 ? (long long)a.ci32[off] : a.ci64[off];
            ok = id >= 0;
            if (a.ci32) id += a.idx_offset;
        }
        tk.push(s, id, ok);
    }

    if (lane == 0) cnt_lds[wave] = tk.cnt;
    __syncthreads();
    if (wave == 0) {
        for (int w = 1; w < nwaves; ++w) {
            const int c = cnt_lds[w];
            const float* ws = list_s + (size_t)w * a.k_out;
            const long long* wi = list_i + (size_t)w * a.k_out;
            for (int j0 = 0; j0 < c; j0 += 64) {
                const int j = j0 + lane;
                const bool ok = j < c;
                const float s = ok ? ws[j] : 0.f;
                const long long id = ok ? wi[j] : 0;
                tk.push(s, id, ok);
            }
        }
        float* os = a.out_s + (size_t)q * a.k_out;
        long long* oi = a.out_i + (size_t)q * a.k_out;
        for (int j = lane; j < a.k_out; j += 64) {
            const bool ok = j < tk.cnt;
            os[j] = ok ? tk.ls[j] : -__builtin_huge_valf();
            oi[j] = ok ? tk.li[j] : -1ll;
        }
    }
}

}  // namespace

hipError_t launch_merge_topk(const float* cand_s, const int* idx32, const long long* idx64, int n_lists,
                             long long list_stride, long long query_stride, int k_in, int b, int k_out,
                             long long idx_offset, long long* out_idx, float* out_score, hipStream_t stream) {
    if (b < 1 || k_out < 1 || k_in < 1 || n_lists < 1) return hipErrorInvalidValue;
    MergeArgs a{cand_s, idx32, idx64, n_lists, list_stride, query_stride, k_in, k_out, idx_offset, out_idx, out_score};
    const long long total = (long long)n_lists * k_in;
    int nwaves = 16;
    while (nwaves > 1 && (long long)nwaves * 64 * 2 > total) nwaves /= 2;            // no idle waves on short inputs
    while (nwaves > 1 && (size_t)nwaves * k_out * 12 + 64 > 60 * 1024) nwaves /= 2;  // LDS budget
    const size_t lds = (size_t)nwaves * k_out * 12 + (size_t)nwaves * 4;
    hipLaunchKernelGGL(merge_topk_kernel, dim3(b), dim3(nwaves * 64), lds, stream, a);
    return hipGetLastError();
}

}  // namespace mtr
