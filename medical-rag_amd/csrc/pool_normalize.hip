// Encoder epilogue: masked mean over tokens followed by L2 normalisation, one
// pass over the hidden states, written as fp32 rows ready for the search.
//
// Replaces the five torch ops at pipeline/utils/local_rag.py:61-65 (and
// pipeline/utils/embed.py:68-72 on the index-build side):
//     expanded_mask = mask.unsqueeze(-1).expand(h.size()).float()
//     sum = (h * expanded_mask).sum(1);  cnt = clamp(expanded_mask.sum(1), min=1e-9)
//     v = sum / cnt;  v = F.normalize(v, p=2, dim=1)          (eps 1e-12)
// The encoder forward itself stays a PyTorch(-ROCm) call.
//
// Two forms.  Short sequences (keyword queries: tens of tokens) take ONE
// kernel, one workgroup per sequence: threads own columns (coalesced over d)
// and walk the T tokens; masked-out tokens are skipped without loading;
// accumulation is fp32 in token order.  Long sequences (index pages: up to 8192
// tokens of width 4096, 64 MB of bf16 each) would leave all but `batch` CUs
// idle that way, so they are cut over tokens AND columns: a first kernel, one
// workgroup per (sequence, token slice, column block) with 16-byte loads, writes
// fp32 partial sums, and a second one, one workgroup per sequence, adds the
// slices in order, divides, normalises and writes.  Both forms agree with the
// torch expression to ~1e-6 (fp32 sums in another order).
//
// Algorithmic bytes: sum over sequences of (unmasked tokens) * d * sizeof(T).
#include "mtr_device.h"

namespace mtr {

namespace {

template <typename T> __device__ __forceinline__ float load_as_f32(const T* p);
template <> __device__ __forceinline__ float load_as_f32<float>(const float* p) { return *p; }
template <> __device__ __forceinline__ float load_as_f32<bf16_t>(const bf16_t* p) { return bf16_to_f32(*p); }
template <> __device__ __forceinline__ float load_as_f32<_Float16>(const _Float16* p) { return (float)*p; }

constexpr int kPoolThreads = 256;
constexpr int kMaxCols = 32;  // columns per thread: d <= 256 * 32

template <typename T, int NC>
__global__ __launch_bounds__(kPoolThreads) void pool_normalize_kernel(const T* __restrict__ hidden,
                                                                       const long long* __restrict__ mask, int t, int d,
                                                                       float* __restrict__ out) {
    __shared__ float red[kPoolThreads / 64];
    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const T* h = hidden + (size_t)b * t * d;
    const long long* m = mask + (size_t)b * t;
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.f;
    float count = 0.f;
    for (int tok = 0; tok < t; ++tok) {
        const float w = (float)m[tok];  // wave-uniform; the reference multiplies by the mask value
        if (w == 0.f) continue;
        count += w;
        const T* row = h + (size_t)tok * d;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int col = c * kPoolThreads + tid;
            if (col < d) acc[c] = __builtin_fmaf(load_as_f32<T>(row + col), w, acc[c]);
        }
    }
    const float denom = fmaxf(count, 1e-9f);
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        acc[c] = acc[c] / denom;
        ss += acc[c] * acc[c];
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    float total = 0.f;
#pragma unroll
    for (int w = 0; w < kPoolThreads / 64; ++w) total += red[w];
    const float inv = 1.f / fmaxf(sqrtf(total), 1e-12f);
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int col = c * kPoolThreads + tid;
        if (col < d) out[(size_t)b * d + col] = acc[c] * inv;
    }
}

// ---- long sequences: partial sums per token slice, then a finishing pass -----

constexpr int kSliceThreads = 256;

template <typename T> struct Vec16;   // 16 bytes of T as floats
template <> struct Vec16<float> {
    static constexpr int N = 4;
    static __device__ __forceinline__ void load(const float* p, float (&f)[4]) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(p);
        f[0] = v[0]; f[1] = v[1]; f[2] = v[2]; f[3] = v[3];
    }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const bf16_t* p, float (&f)[8]) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(p);
        Chunk<bf16_t>::unpack(v, f);
    }
};
template <> struct Vec16<_Float16> {
    static constexpr int N = 8;
    static __device__ __forceinline__ void load(const _Float16* p, float (&f)[8]) {
        using h8 = __attribute__((ext_vector_type(8))) _Float16;
        const h8 v = *reinterpret_cast<const h8*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = (float)v[i];
    }
};

// grid (column blocks, token slices, sequences).  partial: [b][n_slices][d] fp32,
// counts: [b][n_slices] (sum of the slice's mask values).  VEC: 16-byte loads
// (row bytes a multiple of 16), else one element per load.
template <typename T, bool VEC>
__global__ __launch_bounds__(kSliceThreads) void pool_partial_kernel(const T* __restrict__ hidden,
                                                                     const long long* __restrict__ mask, int t, int d,
                                                                     int slice_tokens, float* __restrict__ partial,
                                                                     float* __restrict__ counts) {
    constexpr int N = VEC ? Vec16<T>::N : 1;
    const int b = blockIdx.z, slice = blockIdx.y, n_slices = gridDim.y;
    const int col = (blockIdx.x * kSliceThreads + threadIdx.x) * N;
    const int t0 = slice * slice_tokens;
    const int t1 = t0 + slice_tokens < t ? t0 + slice_tokens : t;
    const T* h = hidden + (size_t)b * t * d;
    const long long* m = mask + (size_t)b * t;
    float acc[N];
#pragma unroll
    for (int i = 0; i < N; ++i) acc[i] = 0.f;
    float count = 0.f;
    for (int tok = t0; tok < t1; ++tok) {
        const float w = (float)m[tok];   // workgroup-uniform
        if (w == 0.f) continue;
        count += w;
        if (col < d) {
            if constexpr (VEC) {
                float f[N];
                Vec16<T>::load(h + (size_t)tok * d + col, f);
#pragma unroll
                for (int i = 0; i < N; ++i) acc[i] = __builtin_fmaf(f[i], w, acc[i]);
            } else {
                acc[0] = __builtin_fmaf(load_as_f32<T>(h + (size_t)tok * d + col), w, acc[0]);
            }
        }
    }
    if (col < d) {
        float* dst = partial + ((size_t)b * n_slices + slice) * d + col;
#pragma unroll
        for (int i = 0; i < N; ++i) dst[i] = acc[i];   // d is a multiple of N in the vector form
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) counts[(size_t)b * n_slices + slice] = count;
}

// One workgroup per sequence: slices added in slice order, mean, L2 normalisation.
__global__ __launch_bounds__(kSliceThreads) void pool_finish_kernel(const float* __restrict__ partial,
                                                                    const float* __restrict__ counts, int n_slices, int d,
                                                                    float* __restrict__ out) {
    __shared__ float red[kSliceThreads / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    float count = 0.f;
    for (int s = 0; s < n_slices; ++s) count += counts[(size_t)b * n_slices + s];
    const float denom = fmaxf(count, 1e-9f);
    float ss = 0.f;
    for (int col = tid; col < d; col += kSliceThreads) {
        float v = 0.f;
        for (int s = 0; s < n_slices; ++s) v += partial[((size_t)b * n_slices + s) * d + col];
        v /= denom;
        out[(size_t)b * d + col] = v;
        ss += v * v;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if ((tid & 63) == 0) red[tid >> 6] = ss;
    __syncthreads();
    float total = 0.f;
#pragma unroll
    for (int w = 0; w < kSliceThreads / 64; ++w) total += red[w];
    const float inv = 1.f / fmaxf(sqrtf(total), 1e-12f);
    for (int col = tid; col < d; col += kSliceThreads) out[(size_t)b * d + col] *= inv;   // each thread rereads its own writes
}

// Token slices for the two-kernel form: enough workgroups to fill the chip
// (about four per CU), slices of at least 32 tokens.
int pool_slices(int b, int t, int col_blocks) {
    long long want = (1024 + (long long)b * col_blocks - 1) / ((long long)b * col_blocks);
    const int most = t / 32 > 1 ? t / 32 : 1;
    if (want > most) want = most;
    return want < 1 ? 1 : (int)want;
}

template <typename T>
hipError_t launch_sliced(const void* hidden, const long long* mask, int b, int t, int d, float* out, hipStream_t stream) {
    const T* h = static_cast<const T*>(hidden);
    const bool vec = ((size_t)d * sizeof(T)) % 16 == 0 && reinterpret_cast<uintptr_t>(hidden) % 16 == 0;
    const int per_block = kSliceThreads * (vec ? Vec16<T>::N : 1);
    const int col_blocks = (d + per_block - 1) / per_block;
    const int n_slices = pool_slices(b, t, col_blocks);
    const int slice_tokens = (t + n_slices - 1) / n_slices;
    // stream-ordered scratch: nothing is held between calls, and calls on different streams do not share it
    float* partial = nullptr;
    const size_t bytes = ((size_t)b * n_slices * d + (size_t)b * n_slices) * sizeof(float);
    hipError_t e = hipMallocAsync(reinterpret_cast<void**>(&partial), bytes, stream);
    if (e != hipSuccess) return e;
    float* counts = partial + (size_t)b * n_slices * d;
    const dim3 grid(col_blocks, n_slices, b);
    if (vec)
        hipLaunchKernelGGL((pool_partial_kernel<T, true>), grid, dim3(kSliceThreads), 0, stream, h, mask, t, d, slice_tokens,
                           partial, counts);
    else
        hipLaunchKernelGGL((pool_partial_kernel<T, false>), grid, dim3(kSliceThreads), 0, stream, h, mask, t, d, slice_tokens,
                           partial, counts);
    hipLaunchKernelGGL(pool_finish_kernel, dim3(b), dim3(kSliceThreads), 0, stream, partial, counts, n_slices, d, out);
    e = hipGetLastError();
    const hipError_t e2 = hipFreeAsync(partial, stream);
    return e != hipSuccess ? e : e2;
}

template <typename T>
void launch_typed(const void* hidden, const long long* mask, int b, int t, int d, float* out, hipStream_t stream) {
    const T* h = static_cast<const T*>(hidden);
    if (d <= kPoolThreads * 4)
        hipLaunchKernelGGL((pool_normalize_kernel<T, 4>), dim3(b), dim3(kPoolThreads), 0, stream, h, mask, t, d, out);
    else if (d <= kPoolThreads * 16)
        hipLaunchKernelGGL((pool_normalize_kernel<T, 16>), dim3(b), dim3(kPoolThreads), 0, stream, h, mask, t, d, out);
    else
        hipLaunchKernelGGL((pool_normalize_kernel<T, kMaxCols>), dim3(b), dim3(kPoolThreads), 0, stream, h, mask, t, d,
                           out);
}

}  // namespace

// Sequences from this length on (or rows wider than the one-kernel form's 8192
// columns) take the two-kernel form; batches that fill the chip by themselves
// with short sequences do not need it.
constexpr int kSlicedFromTokens = 256;

hipError_t launch_pool_normalize(const void* hidden, int dtype, const long long* mask, int b, int t, int d, float* out,
                                 hipStream_t stream) {
    if (b < 1 || t < 1 || d < 1) return hipErrorInvalidValue;
    int form = tuning().pool_sliced;   // -1: by shape
    if (form < 0) form = (t >= kSlicedFromTokens || d > kPoolThreads * kMaxCols) ? 1 : 0;
    if (d > kPoolThreads * kMaxCols) form = 1;
    if (b > 65535) form = 0;           // grid.z limit of the sliced form
    if (form == 0 && d > kPoolThreads * kMaxCols) return hipErrorInvalidValue;
    if (form == 1) {
        switch (dtype) {
            case 0: return launch_sliced<float>(hidden, mask, b, t, d, out, stream);
            case 2: return launch_sliced<bf16_t>(hidden, mask, b, t, d, out, stream);
            case 3: return launch_sliced<_Float16>(hidden, mask, b, t, d, out, stream);
            default: return hipErrorInvalidValue;
        }
    }
    switch (dtype) {
        case 0: launch_typed<float>(hidden, mask, b, t, d, out, stream); break;
        case 2: launch_typed<bf16_t>(hidden, mask, b, t, d, out, stream); break;
        case 3: launch_typed<_Float16>(hidden, mask, b, t, d, out, stream); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace mtr
