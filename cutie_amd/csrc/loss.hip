// K9-K11: per-cell softmax, sparse cross-entropy, hard negative mining, d(loss)/d(logits).
// Model.softmax (/root/reference/model_framework.py:318-320) and CUTIE.build_loss
// (/root/reference/model_cutie.py:107-130).  HBM-bound: (2C+2)*4 B per cell fwd+bwd.
#include "common.h"

#include <math.h>

namespace {
using namespace cutie;

constexpr int CE_CELLS = 256;  // cells per workgroup

// A tile of 256 cells x C logits goes through LDS so that global reads and writes are
// coalesced rows while each lane reduces one cell's C values (rows padded to C+1 words:
// conflict-free column walks).
template <typename TL>
__global__ __launch_bounds__(CE_CELLS) void softmax_ce_fwd_kernel(const TL* __restrict__ logits, int ld,
                                                                  const int32_t* __restrict__ labels,
                                                                  int64_t n_cells, int C, float* __restrict__ ce,
                                                                  float* __restrict__ probs,
                                                                  int32_t* __restrict__ bad_labels) {
    extern __shared__ float tile[];  // [CE_CELLS][C+1]
    const int64_t cell0 = (int64_t)blockIdx.x * CE_CELLS;
    const int n_here = (int)((n_cells - cell0) < CE_CELLS ? (n_cells - cell0) : CE_CELLS);
    const int stride = C + 1;
    for (int i = threadIdx.x; i < n_here * C; i += CE_CELLS) {
        const int r = i / C, c = i - r * C;
        tile[r * stride + c] = load_as_f32(logits + (cell0 + r) * ld + c);
    }
    __syncthreads();
    const int r = threadIdx.x;
    if (r < n_here) {
        float* row = tile + r * stride;
        float mx = row[0];
        for (int c = 1; c < C; ++c) mx = fmaxf(mx, row[c]);
        float sum = 0.f;
        for (int c = 0; c < C; ++c) {
            const float e = expf(row[c] - mx);
            sum += e;
        }
        const int lab = labels[cell0 + r];
        float loss;
        if (lab >= 0 && lab < C) {
            loss = logf(sum) - (row[lab] - mx);
        } else {
            loss = nanf("");
            atomicAdd(bad_labels, 1);
        }
        ce[cell0 + r] = loss;
        if (probs) {
            const float inv = 1.f / sum;
            for (int c = 0; c < C; ++c) row[c] = expf(row[c] - mx) * inv;
        }
    }
    if (probs) {
        __syncthreads();
        for (int i = threadIdx.x; i < n_here * C; i += CE_CELLS) {
            const int rr = i / C, c = i - rr * C;
            probs[(cell0 + rr) * C + c] = tile[rr * stride + c];
        }
    }
}

template <typename TD>
__global__ __launch_bounds__(256) void softmax_ce_bwd_kernel(const float* __restrict__ probs,
                                                             const int32_t* __restrict__ labels,
                                                             const float* __restrict__ weight, int64_t n_cells,
                                                             int C, TD* __restrict__ dlogits, int ld) {
    const int64_t total = n_cells * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / C;
        const int c = (int)(i - n * C);
        const float w = weight[n];
        const float g = w * (probs[i] - (labels[n] == c ? 1.f : 0.f));
        store_from_f32(dlogits + n * ld + c, w == 0.f ? 0.f : g);
    }
}

__device__ __forceinline__ unsigned sortable(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

constexpr int HNM_T = 1024;

__device__ __forceinline__ float block_sum_f(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < HNM_T / 64; ++w) t += sh[w];
    return t;
}
__device__ __forceinline__ int block_sum_i(int v, int* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
    for (int w = 0; w < HNM_T / 64; ++w) t += sh[w];
    return t;
}

// Single workgroup: counts, radix select of the num_bg-th largest background ce (4 x 8 bit),
// then an index-ordered pass that breaks ties toward the lower index (tf.nn.top_k).
__global__ __launch_bounds__(HNM_T) void hnm_kernel(const float* __restrict__ ce, const int32_t* __restrict__ labels,
                                                    int64_t n_cells, int ratio, float* __restrict__ weight,
                                                    float* __restrict__ loss_out) {
    __shared__ int hist[256];
    __shared__ float shf[HNM_T / 64];
    __shared__ int shi[HNM_T / 64];
    __shared__ unsigned s_prefix;
    __shared__ long long s_k;
    __shared__ int s_base;
    const int tid = threadIdx.x;

    int nfg = 0, nbg = 0;
    float sum_fg = 0.f;
    for (int64_t i = tid; i < n_cells; i += HNM_T) {
        if (labels[i] != 0) { ++nfg; sum_fg += ce[i]; } else { ++nbg; }
    }
    nfg = block_sum_i(nfg, shi);
    nbg = block_sum_i(nbg, shi);
    sum_fg = block_sum_f(sum_fg, shf);
    const long long want = (long long)ratio * (long long)nfg;
    const long long num_bg = want < (long long)nbg ? want : (long long)nbg;  // model_cutie.py:124-125

    unsigned thr = 0;           // sortable key of the num_bg-th largest background ce
    long long k_eq = 0;         // how many cells equal to thr are selected
    const bool all_bg = (num_bg == (long long)nbg);
    if (!all_bg && num_bg > 0) {
        if (tid == 0) { s_prefix = 0; s_k = num_bg; }
        __syncthreads();
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            for (int i = tid; i < 256; i += HNM_T) hist[i] = 0;
            __syncthreads();
            const unsigned prefix = s_prefix;
            const unsigned mask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
            for (int64_t i = tid; i < n_cells; i += HNM_T) {
                if (labels[i] == 0) {
                    const unsigned u = sortable(ce[i]);
                    if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255], 1);
                }
            }
            __syncthreads();
            if (tid == 0) {
                long long k = s_k;
                int b = 255;
                for (; b > 0; --b) {
                    if (k <= hist[b]) break;
                    k -= hist[b];
                }
                s_k = k;
                s_prefix = prefix | ((unsigned)b << shift);
            }
            __syncthreads();
        }
        thr = s_prefix;
        k_eq = s_k;
    }

    const float w_fg = 1.f / (float)nfg;          // inf when nfg == 0; unused then
    const float w_bg = 1.f / (float)num_bg;
    float sum_sel = 0.f;
    if (tid == 0) s_base = 0;
    __syncthreads();
    for (int64_t i0 = 0; i0 < n_cells; i0 += HNM_T) {
        const int64_t i = i0 + tid;
        const bool in = i < n_cells;
        const int lab = in ? labels[i] : 1;
        const float c = in ? ce[i] : 0.f;
        const bool bg = in && lab == 0;
        bool take = false, eq = false;
        if (bg && num_bg > 0) {
            if (all_bg) take = true;
            else {
                const unsigned u = sortable(c);
                take = u > thr;
                eq = u == thr;
            }
        }
        if (!all_bg && num_bg > 0) {
            // ordinal of this cell among the cells equal to thr, in index order
            const unsigned long long bal = __ballot(eq);
            const int lane = tid & 63, wv = tid >> 6;
            const int before = __popcll(bal & ((1ull << lane) - 1ull));
            __syncthreads();
            if (lane == 0) shi[wv] = __popcll(bal);
            __syncthreads();
            int wave_off = 0, total = 0;
            for (int w = 0; w < HNM_T / 64; ++w) {
                if (w < wv) wave_off += shi[w];
                total += shi[w];
            }
            const int base = s_base;
            if (eq && (long long)(base + wave_off + before) < k_eq) take = true;
            __syncthreads();
            if (tid == 0) s_base = base + total;
        }
        if (in) {
            float w = 0.f;
            if (lab != 0) w = w_fg;
            else if (take) { w = w_bg; sum_sel += c; }
            weight[i] = w;
        }
    }
    sum_sel = block_sum_f(sum_sel, shf);
    if (tid == 0) {
        // mean over an empty set is NaN (tf.reduce_mean of an empty tensor): 0/0
        loss_out[0] = sum_fg / (float)nfg + sum_sel / (float)num_bg;
        loss_out[1] = (float)nfg;
        loss_out[2] = (float)num_bg;
    }
}

}  // namespace

extern "C" int cutie_softmax_ce_fwd(const void* logits, int ld, int dtype, const int32_t* labels, int64_t n_cells,
                                    int C, float* ce, float* probs, int32_t* bad_labels, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(n_cells >= 0 && C >= 1 && C <= 127 && ld >= C, "softmax_ce_fwd: n=%lld C=%d ld=%d (C <= 127)",
                    (long long)n_cells, C, ld);
    if (n_cells == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(logits && labels && ce && bad_labels, "softmax_ce_fwd: null pointer");
    const int g = (int)cutie::cdiv(n_cells, CE_CELLS);
    const size_t lds = (size_t)CE_CELLS * (C + 1) * sizeof(float);
    CUTIE_DISPATCH_DTYPE(dtype, DT,
        hipLaunchKernelGGL(softmax_ce_fwd_kernel<typename elem<DT>::type>, dim3(g), dim3(CE_CELLS), lds, stream,
                           (const typename elem<DT>::type*)logits, ld, labels, n_cells, C, ce, probs, bad_labels));
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

extern "C" size_t cutie_hnm_workspace_bytes(int64_t) { return 16; }

extern "C" int cutie_hnm_weights(const float* ce, const int32_t* labels, int64_t n_cells, int ratio, float* weight,
                                 float* loss_out, void* /*workspace*/, size_t /*workspace_bytes*/,
                                 cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(n_cells >= 0 && ratio >= 0, "hnm_weights: n=%lld ratio=%d", (long long)n_cells, ratio);
    CUTIE_CHECK_ARG(ce && labels && weight && loss_out, "hnm_weights: null pointer");
    hipLaunchKernelGGL(hnm_kernel, dim3(1), dim3(HNM_T), 0, stream, ce, labels, n_cells, ratio, weight, loss_out);
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

extern "C" int cutie_softmax_ce_bwd(const float* probs, const int32_t* labels, const float* weight, int64_t n_cells,
                                    int C, void* dlogits, int ld, int dtype, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(n_cells >= 0 && C >= 1 && ld >= C, "softmax_ce_bwd: n=%lld C=%d ld=%d", (long long)n_cells, C, ld);
    if (n_cells == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(probs && labels && weight && dlogits, "softmax_ce_bwd: null pointer");
    int64_t g = cutie::cdiv(n_cells * C, 256);
    if (g > 2048) g = 2048;
    CUTIE_DISPATCH_DTYPE(dtype, DT,
        hipLaunchKernelGGL(softmax_ce_bwd_kernel<typename elem<DT>::type>, dim3((int)g), dim3(256), 0, stream, probs,
                           labels, weight, n_cells, C, (typename elem<DT>::type*)dlogits, ld));
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}
