// K9-K11: per-cell softmax, sparse cross-entropy, hard negative mining, d(loss)/d(logits).
// Model.softmax (/root/reference/model_framework.py:318-320) and CUTIE.build_loss
// (/root/reference/model_cutie.py:107-130).  HBM-bound: (2C+2)*4 B per cell fwd+bwd.
#include "common.h"

#include <mutex>

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
                                                             int C, TD* __restrict__ dlogits, int ld,
                                                             const float* __restrict__ loss_scale,
                                                             const float* __restrict__ elem_weight) {
    const float ls = loss_scale ? loss_scale[0] : 1.f;
    const int64_t total = n_cells * C;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = i / C;
        const int c = (int)(i - n * C);
        const float w = weight[n];
        float g = ls * w * (probs[i] - (labels[n] == c ? 1.f : 0.f));
        if (elem_weight) g *= elem_weight[i];      // GHM: the CE saw logits * elem_weight
        store_from_f32(dlogits + n * ld + c, w == 0.f ? 0.f : g);
    }
}

__device__ __forceinline__ unsigned sortable(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---------------------------------------------------------------------------------------------
// Hard negative mining (model_cutie.py:116-130) over many workgroups.
//   num_bg = min(#background, ratio * #foreground); the selected background cells are the num_bg
//   with the largest ce, ties broken toward the lower index (tf.nn.top_k).  The num_bg-th largest
//   key is found by a three-level radix select (11 + 11 + 10 bits of the order-preserving integer
//   image of the float) - one histogram kernel per level, every workgroup redoing the small bin
//   scan of the previous level - then the cells equal to the threshold are counted per 1024-cell
//   chunk so that the final kernel can take the first k_eq of them in index order.
//   All sums are fixed-order (per-chunk partials, then one thread): the loss is deterministic.
constexpr int HNM_T = 256;
constexpr int HNM_CHUNK = 1024;       // cells per workgroup (4 per lane)
constexpr int HNM_B0 = 2048, HNM_B1 = 2048, HNM_B2 = 1024;

struct HnmWs {
    int* counts;     // [0] nfg [1] nbg [2..7] sel: b0, k1, b1, k2, (unused)
    int* hist0;      // [2048]
    int* hist1;      // [2048]
    int* hist2;      // [1024]
    int* tie_cnt;    // [chunks]
    float* sum_fg;   // [chunks]
    float* sum_sel;  // [chunks]
};
constexpr int HNM_INT_HEAD = 8 + HNM_B0 + HNM_B1 + HNM_B2;

__device__ __forceinline__ HnmWs hnm_ws(void* ws, int chunks) {
    HnmWs w;
    w.counts = reinterpret_cast<int*>(ws);
    w.hist0 = w.counts + 8;
    w.hist1 = w.hist0 + HNM_B0;
    w.hist2 = w.hist1 + HNM_B1;
    w.tie_cnt = w.hist2 + HNM_B2;
    w.sum_fg = reinterpret_cast<float*>(w.tie_cnt + chunks);
    w.sum_sel = w.sum_fg + chunks;
    return w;
}

__device__ __forceinline__ int block_sum_i256(int v, int* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return sh[0] + sh[1] + sh[2] + sh[3];
}
__device__ __forceinline__ float block_sum_f256(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

__device__ __forceinline__ long long hnm_num_bg(const int* counts, int ratio, bool& all_bg) {
    const long long nfg = counts[0], nbg = counts[1];
    const long long want = (long long)ratio * nfg;
    const long long num_bg = want < nbg ? want : nbg;     // model_cutie.py:124-125
    all_bg = num_bg == nbg;
    return num_bg;
}

// Which bin holds the k-th largest key (k >= 1), and how many of that bin's keys are wanted.
// All 256 lanes take part; hist has `bins` (a multiple of 256) entries.  sh: 260 ints.
__device__ __forceinline__ void hnm_select(const int* __restrict__ hist, int bins, long long k, int* sh, int& bin,
                                           long long& k_in) {
    const int per = bins / HNM_T;
    const int tid = threadIdx.x;
    int mine = 0;
    for (int i = 0; i < per; ++i) mine += hist[tid * per + i];
    __syncthreads();
    sh[tid] = mine;
    __syncthreads();
    if (tid == 0) {
        // walk the lane sums from the top bin down, then the bins of the lane that holds rank k
        long long rem = k;
        int t = HNM_T - 1;
        for (; t > 0; --t) {
            if (rem <= sh[t]) break;
            rem -= sh[t];
        }
        int bsel = t * per + per - 1;
        for (; bsel > t * per; --bsel) {
            const int h = hist[bsel];
            if (rem <= h) break;
            rem -= h;
        }
        sh[256] = bsel;
        sh[257] = (int)rem;
    }
    __syncthreads();
    bin = sh[256];
    k_in = sh[257];
    __syncthreads();
}

// level 0: counts, foreground ce partial sums, histogram of the top 11 key bits of background cells
__global__ __launch_bounds__(HNM_T) void hnm_count_kernel(const float* __restrict__ ce, const int32_t* __restrict__ labels,
                                                          int64_t n_cells, int chunks, void* ws_) {
    __shared__ int hist[HNM_B0];
    __shared__ int shi[4];
    __shared__ float shf[4];
    const HnmWs ws = hnm_ws(ws_, chunks);
    const int tid = threadIdx.x;
    for (int i = tid; i < HNM_B0; i += HNM_T) hist[i] = 0;
    __syncthreads();
    int nfg = 0, nbg = 0;
    float sum_fg = 0.f;
    const int64_t base = (int64_t)blockIdx.x * HNM_CHUNK;
#pragma unroll
    for (int u = 0; u < HNM_CHUNK / HNM_T; ++u) {
        const int64_t i = base + u * HNM_T + tid;
        if (i < n_cells) {
            const float c = ce[i];
            if (labels[i] != 0) { ++nfg; sum_fg += c; }
            else { ++nbg; atomicAdd(&hist[sortable(c) >> 21], 1); }
        }
    }
    nfg = block_sum_i256(nfg, shi);
    nbg = block_sum_i256(nbg, shi);
    sum_fg = block_sum_f256(sum_fg, shf);
    if (tid == 0) {
        if (nfg) atomicAdd(&ws.counts[0], nfg);
        if (nbg) atomicAdd(&ws.counts[1], nbg);
        ws.sum_fg[blockIdx.x] = sum_fg;
    }
    for (int i = tid; i < HNM_B0; i += HNM_T)
        if (hist[i]) atomicAdd(&ws.hist0[i], hist[i]);
}

// level 1 / 2: redo the previous level's bin selection, histogram the next key bits of the
// background cells inside the selected prefix
template <int LEVEL>
__global__ __launch_bounds__(HNM_T) void hnm_hist_kernel(const float* __restrict__ ce, const int32_t* __restrict__ labels,
                                                         int64_t n_cells, int chunks, int ratio, void* ws_) {
    constexpr int BINS = LEVEL == 1 ? HNM_B1 : HNM_B2;
    __shared__ int hist[BINS];
    __shared__ int sh[260];
    const HnmWs ws = hnm_ws(ws_, chunks);
    const int tid = threadIdx.x;
    bool all_bg;
    const long long num_bg = hnm_num_bg(ws.counts, ratio, all_bg);
    if (all_bg || num_bg <= 0) return;
    unsigned prefix;     // the key bits fixed so far, right-aligned
    if (LEVEL == 1) {
        int b0; long long k1;
        hnm_select(ws.hist0, HNM_B0, num_bg, sh, b0, k1);
        if (blockIdx.x == 0 && tid == 0) { ws.counts[2] = b0; ws.counts[3] = (int)k1; }
        prefix = (unsigned)b0;
    } else {
        const int b0 = ws.counts[2];
        int b1; long long k2;
        hnm_select(ws.hist1, HNM_B1, (long long)ws.counts[3], sh, b1, k2);
        if (blockIdx.x == 0 && tid == 0) { ws.counts[4] = b1; ws.counts[5] = (int)k2; }
        prefix = ((unsigned)b0 << 11) | (unsigned)b1;
    }
    for (int i = tid; i < BINS; i += HNM_T) hist[i] = 0;
    __syncthreads();
    const int64_t base = (int64_t)blockIdx.x * HNM_CHUNK;
#pragma unroll
    for (int u = 0; u < HNM_CHUNK / HNM_T; ++u) {
        const int64_t i = base + u * HNM_T + tid;
        if (i < n_cells && labels[i] == 0) {
            const unsigned key = sortable(ce[i]);
            if (LEVEL == 1) {
                if ((key >> 21) == prefix) atomicAdd(&hist[(key >> 10) & 2047], 1);
            } else {
                if ((key >> 10) == prefix) atomicAdd(&hist[key & 1023], 1);
            }
        }
    }
    __syncthreads();
    int* out = LEVEL == 1 ? ws.hist1 : ws.hist2;
    for (int i = tid; i < BINS; i += HNM_T)
        if (hist[i]) atomicAdd(&out[i], hist[i]);
}

// threshold key and the number of cells equal to it that are selected; per-chunk tie counts
__global__ __launch_bounds__(HNM_T) void hnm_tie_kernel(const float* __restrict__ ce, const int32_t* __restrict__ labels,
                                                        int64_t n_cells, int chunks, int ratio, void* ws_) {
    __shared__ int sh[260];
    __shared__ int shi[4];
    const HnmWs ws = hnm_ws(ws_, chunks);
    const int tid = threadIdx.x;
    bool all_bg;
    const long long num_bg = hnm_num_bg(ws.counts, ratio, all_bg);
    if (all_bg || num_bg <= 0) return;
    int b2; long long k_eq;
    hnm_select(ws.hist2, HNM_B2, (long long)ws.counts[5], sh, b2, k_eq);
    const unsigned thr = ((unsigned)ws.counts[2] << 21) | ((unsigned)ws.counts[4] << 10) | (unsigned)b2;
    if (blockIdx.x == 0 && tid == 0) { ws.counts[6] = (int)thr; ws.counts[7] = (int)k_eq; }
    int ties = 0;
    const int64_t base = (int64_t)blockIdx.x * HNM_CHUNK;
#pragma unroll
    for (int u = 0; u < HNM_CHUNK / HNM_T; ++u) {
        const int64_t i = base + u * HNM_T + tid;
        if (i < n_cells && labels[i] == 0 && sortable(ce[i]) == thr) ++ties;
    }
    ties = block_sum_i256(ties, shi);
    if (tid == 0) ws.tie_cnt[blockIdx.x] = ties;
}

// per-cell weights (1/nfg, 1/num_bg or 0) and the per-chunk sums of the selected background ce
__global__ __launch_bounds__(HNM_T) void hnm_weight_kernel(const float* __restrict__ ce, const int32_t* __restrict__ labels,
                                                           int64_t n_cells, int chunks, int ratio, void* ws_,
                                                           float* __restrict__ weight) {
    __shared__ int shi[4];
    __shared__ float shf[4];
    __shared__ int wave_cnt[4];
    const HnmWs ws = hnm_ws(ws_, chunks);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    bool all_bg;
    const long long num_bg = hnm_num_bg(ws.counts, ratio, all_bg);
    const bool select = !all_bg && num_bg > 0;
    const unsigned thr = select ? (unsigned)ws.counts[6] : 0u;
    const long long k_eq = select ? (long long)ws.counts[7] : 0;
    // cells equal to thr in earlier chunks (index order)
    long long before_chunk = 0;
    if (select) {
        int part = 0;
        for (int c = tid; c < (int)blockIdx.x; c += HNM_T) part += ws.tie_cnt[c];
        before_chunk = block_sum_i256(part, shi);
    }
    const float w_fg = 1.f / (float)ws.counts[0];          // inf when nfg == 0; unused then
    const float w_bg = 1.f / (float)num_bg;
    float sum_sel = 0.f;
    const int64_t base = (int64_t)blockIdx.x * HNM_CHUNK;
    for (int u = 0; u < HNM_CHUNK / HNM_T; ++u) {
        const int64_t i = base + u * HNM_T + tid;
        const bool in = i < n_cells;
        const int lab = in ? labels[i] : 1;
        const float c = in ? ce[i] : 0.f;
        const bool bg = in && lab == 0;
        bool take = false, eq = false;
        if (bg && num_bg > 0) {
            if (all_bg) take = true;
            else {
                const unsigned key = sortable(c);
                take = key > thr;
                eq = key == thr;
            }
        }
        if (select) {      // ordinal of this cell among the cells equal to thr, in index order
            const unsigned long long bal = __ballot(eq);
            const int before = __popcll(bal & ((1ull << lane) - 1ull));
            __syncthreads();
            if (lane == 0) wave_cnt[wv] = __popcll(bal);
            __syncthreads();
            int wave_off = 0, total = 0;
            for (int w = 0; w < 4; ++w) {
                if (w < wv) wave_off += wave_cnt[w];
                total += wave_cnt[w];
            }
            if (eq && before_chunk + wave_off + before < k_eq) take = true;
            before_chunk += total;
        }
        if (in) {
            float w = 0.f;
            if (lab != 0) w = w_fg;
            else if (take) { w = w_bg; sum_sel += c; }
            weight[i] = w;
        }
    }
    sum_sel = block_sum_f256(sum_sel, shf);
    if (tid == 0) ws.sum_sel[blockIdx.x] = sum_sel;
}

__global__ __launch_bounds__(64) void hnm_final_kernel(int chunks, int ratio, void* ws_, float* __restrict__ loss_out) {
    const HnmWs ws = hnm_ws(ws_, chunks);
    if (threadIdx.x != 0) return;
    bool all_bg;
    const long long num_bg = hnm_num_bg(ws.counts, ratio, all_bg);
    float sum_fg = 0.f, sum_sel = 0.f;
    for (int c = 0; c < chunks; ++c) { sum_fg += ws.sum_fg[c]; sum_sel += ws.sum_sel[c]; }
    // mean over an empty set is NaN (tf.reduce_mean of an empty tensor): 0/0
    loss_out[0] = sum_fg / (float)ws.counts[0] + sum_sel / (float)num_bg;
    loss_out[1] = (float)ws.counts[0];
    loss_out[2] = (float)num_bg;
}

// ---------------------------------------------------------------------------------------------
// Gradient harmonising weights, calc_ghm_weights of main_train_json.py:86-119 (use_ghm = 1).
//   grad = |logit - onehot| per (cell, class) in double (numpy promotes float32 - int64 to float64);
//   bin i holds edges[i] <= grad < edges[i+1], edges[i] = i / bins, the last edge + eps;
//   acc_sum[i] = momentum * acc_sum[i] + (1 - momentum) * count[i] for the non-empty bins (M of them);
//   weight = N / acc_sum[bin] / M inside a bin, 1 / M outside every bin (grad >= 1 + eps), N = cells * C.
// The histogram runs on the whole grid, the M-bin update on one lane, the weights (and the weighted
// logits the loss is taken of, model_cutie.py:110) on the whole grid again.  State stays on the device.
constexpr int GHM_MAX_BINS = 256;

__device__ __forceinline__ int ghm_bin(float logit, bool hot, int bins, double eps) {
    const double g = fabs((double)logit - (hot ? 1.0 : 0.0));
    int i = (int)(g * bins);
    if (i > bins - 1) i = bins - 1;
    while (i > 0 && g < (double)i / bins) --i;
    while (i < bins - 1 && g >= (double)(i + 1) / bins) ++i;
    const double hi = (i == bins - 1) ? (double)bins / bins + eps : (double)(i + 1) / bins;
    return (g >= (double)i / bins && g < hi) ? i : bins;     // `bins` = outside every bin
}

template <typename TL>
__global__ __launch_bounds__(256) void ghm_hist_kernel(const TL* __restrict__ logits, int ld,
                                                       const int32_t* __restrict__ labels, int64_t n_cells, int C,
                                                       int bins, double eps, unsigned* __restrict__ counts) {
    __shared__ unsigned h[GHM_MAX_BINS + 1];
    for (int i = threadIdx.x; i <= bins; i += 256) h[i] = 0u;
    __syncthreads();
    const int64_t total = n_cells * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / C;
        const int c = (int)(i - n * C);
        atomicAdd(&h[ghm_bin(load_as_f32(logits + n * ld + c), labels[n] == c, bins, eps)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < bins; i += 256)
        if (h[i]) atomicAdd(counts + i, h[i]);
}

__global__ void ghm_update_kernel(const unsigned* __restrict__ counts, int bins, double momentum, double n_total,
                                  double* __restrict__ acc_sum, float* __restrict__ table) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int m = 0;
    for (int i = 0; i < bins; ++i) {
        if (counts[i] > 0u) {
            acc_sum[i] = momentum * acc_sum[i] + (1.0 - momentum) * (double)counts[i];
            ++m;
        }
    }
    const double div = m > 0 ? (double)m : 1.0;
    for (int i = 0; i < bins; ++i) table[i] = (float)((counts[i] > 0u ? n_total / acc_sum[i] : 1.0) / div);
    table[bins] = (float)(1.0 / div);
}

template <typename TL>
__global__ __launch_bounds__(256) void ghm_apply_kernel(const TL* __restrict__ logits, int ld,
                                                        const int32_t* __restrict__ labels, int64_t n_cells, int C,
                                                        int bins, double eps, const float* __restrict__ table,
                                                        float* __restrict__ weights, float* __restrict__ weighted) {
    const int64_t total = n_cells * C;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t n = i / C;
        const int c = (int)(i - n * C);
        const float x = load_as_f32(logits + n * ld + c);
        const float w = table[ghm_bin(x, labels[n] == c, bins, eps)];
        weights[i] = w;
        if (weighted) weighted[i] = x * w;
    }
}

}  // namespace

extern "C" size_t cutie_ghm_state_bytes(int bins) {
    if (bins < 1 || bins > GHM_MAX_BINS) return 0;
    return (size_t)bins * sizeof(double) + (size_t)(bins + 1) * sizeof(float) + (size_t)bins * sizeof(unsigned) + 16;
}

extern "C" int cutie_ghm_weights(const void* logits, int ld, int dtype, const int32_t* labels, int64_t n_cells, int C,
                                 int bins, float momentum, float eps, void* state, size_t state_bytes, float* weights,
                                 float* weighted_logits, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(n_cells >= 0 && C >= 1 && ld >= C, "ghm_weights: n=%lld C=%d ld=%d", (long long)n_cells, C, ld);
    CUTIE_CHECK_ARG(bins >= 1 && bins <= GHM_MAX_BINS, "ghm_weights: bins=%d (1..%d)", bins, GHM_MAX_BINS);
    CUTIE_CHECK_ARG(momentum >= 0.f && momentum < 1.f, "ghm_weights: momentum %g", (double)momentum);
    if (!state || state_bytes < cutie_ghm_state_bytes(bins) || (reinterpret_cast<uintptr_t>(state) & 7)) {
        cutie::set_error("ghm_weights: state %zu < %zu bytes (or not 8-byte aligned)", state_bytes, cutie_ghm_state_bytes(bins));
        return CUTIE_E_WORKSPACE;
    }
    if (n_cells == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(logits && labels && weights, "ghm_weights: null pointer");
    double* acc_sum = reinterpret_cast<double*>(state);
    float* table = reinterpret_cast<float*>(acc_sum + bins);
    unsigned* counts = reinterpret_cast<unsigned*>(table + bins + 1);
    CUTIE_CHECK_HIP(hipMemsetAsync(counts, 0, (size_t)bins * sizeof(unsigned), stream));
    int64_t g = cutie::cdiv(n_cells * C, 256 * 8);
    if (g > 1024) g = 1024;
    if (g < 1) g = 1;
    CUTIE_DISPATCH_DTYPE(dtype, DT,
        hipLaunchKernelGGL(ghm_hist_kernel<typename elem<DT>::type>, dim3((int)g), dim3(256), 0, stream,
                           (const typename elem<DT>::type*)logits, ld, labels, n_cells, C, bins, (double)eps, counts));
    CUTIE_CHECK_LAUNCH();
    hipLaunchKernelGGL(ghm_update_kernel, dim3(1), dim3(64), 0, stream, (const unsigned*)counts, bins, (double)momentum,
                       (double)n_cells * C, acc_sum, table);
    CUTIE_CHECK_LAUNCH();
    CUTIE_DISPATCH_DTYPE(dtype, DT,
        hipLaunchKernelGGL(ghm_apply_kernel<typename elem<DT>::type>, dim3((int)g), dim3(256), 0, stream,
                           (const typename elem<DT>::type*)logits, ld, labels, n_cells, C, bins, (double)eps,
                           (const float*)table, weights, weighted_logits));
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

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

static inline int64_t hnm_chunks(int64_t n_cells) { return n_cells > 0 ? cutie::cdiv(n_cells, HNM_CHUNK) : 1; }

extern "C" size_t cutie_hnm_workspace_bytes(int64_t n_cells) {
    return (size_t)(HNM_INT_HEAD + 3 * hnm_chunks(n_cells)) * 4 + 64;
}

// the six launches of the select over a zero-filled work area `ws` (16-byte aligned, hnm_ws layout)
static int hnm_launch(const float* ce, const int32_t* labels, int64_t n_cells, int ratio, float* weight, float* loss_out,
                      void* ws, hipStream_t stream) {
    const int chunks = (int)hnm_chunks(n_cells);
    CUTIE_CHECK_HIP(hipMemsetAsync(ws, 0, (size_t)(HNM_INT_HEAD + 3 * chunks) * 4, stream));
    const dim3 g(chunks), t(HNM_T);
    hipLaunchKernelGGL(hnm_count_kernel, g, t, 0, stream, ce, labels, n_cells, chunks, ws);
    hipLaunchKernelGGL(hnm_hist_kernel<1>, g, t, 0, stream, ce, labels, n_cells, chunks, ratio, ws);
    hipLaunchKernelGGL(hnm_hist_kernel<2>, g, t, 0, stream, ce, labels, n_cells, chunks, ratio, ws);
    hipLaunchKernelGGL(hnm_tie_kernel, g, t, 0, stream, ce, labels, n_cells, chunks, ratio, ws);
    hipLaunchKernelGGL(hnm_weight_kernel, g, t, 0, stream, ce, labels, n_cells, chunks, ratio, ws, weight);
    hipLaunchKernelGGL(hnm_final_kernel, dim3(1), dim3(64), 0, stream, chunks, ratio, ws, loss_out);
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

extern "C" int cutie_hnm_weights(const float* ce, const int32_t* labels, int64_t n_cells, int ratio, float* weight,
                                 float* loss_out, void* workspace, size_t workspace_bytes, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(n_cells >= 0 && ratio >= 0, "hnm_weights: n=%lld ratio=%d", (long long)n_cells, ratio);
    CUTIE_CHECK_ARG(ce && labels && weight && loss_out, "hnm_weights: null pointer");
    CUTIE_CHECK_ARG(n_cells < (1ll << 31) - HNM_CHUNK, "hnm_weights: too many cells");
    const size_t need = cutie_hnm_workspace_bytes(n_cells);
    if (!workspace || workspace_bytes < need) {
        cutie::set_error("hnm_weights: workspace %zu < %zu", workspace_bytes, need);
        return CUTIE_E_WORKSPACE;
    }
    void* ws = (void*)(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
    return hnm_launch(ce, labels, n_cells, ratio, weight, loss_out, ws, stream);
}

// ---------------------------------------------------------------------------------------------
// cutie_hnm_weights for callers that cannot allocate (HIP graph capture, serving loops): the work area of the
// select - three histograms, eight counters and three words per 1024-cell chunk - lives in a device array of the
// code object instead of a caller workspace.  Each GPU the library runs on holds its own copy at its own address
// until the process ends; the host only ever uses the address hipGetSymbolAddress returned for the CURRENT device
// (the array's name is a host-side stand-in, never a device pointer).
constexpr int64_t HNM_STATIC_MAX_CELLS = (int64_t)CUTIE_HNM_STATIC_MAX_CELLS;
constexpr size_t HNM_STATIC_BYTES = (size_t)(HNM_INT_HEAD + 3 * (HNM_STATIC_MAX_CELLS / HNM_CHUNK)) * 4;
static __device__ __align__(256) unsigned char g_hnm_scratch[HNM_STATIC_BYTES];

namespace {
constexpr int HNM_MAX_DEVICES = 64;
std::mutex g_hnm_scratch_mu;
void* g_hnm_scratch_ptr[HNM_MAX_DEVICES];      // per device, resolved on the entry point's first call there

int hnm_scratch_for_current_device(void** out) {
    int dev = -1;
    CUTIE_CHECK_HIP(hipGetDevice(&dev));
    CUTIE_CHECK_ARG(dev >= 0 && dev < HNM_MAX_DEVICES, "hnm_weights_static: device ordinal %d (max %d)", dev, HNM_MAX_DEVICES - 1);
    std::lock_guard<std::mutex> lock(g_hnm_scratch_mu);
    if (!g_hnm_scratch_ptr[dev]) {
        void* p = nullptr;
        CUTIE_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(g_hnm_scratch)));
        CUTIE_CHECK_ARG(p && ((uintptr_t)p & 15) == 0, "hnm_weights_static: symbol address %p", p);
        g_hnm_scratch_ptr[dev] = p;
    }
    *out = g_hnm_scratch_ptr[dev];
    return CUTIE_OK;
}
}  // namespace

extern "C" int cutie_hnm_weights_static(const float* ce, const int32_t* labels, int64_t n_cells, int ratio,
                                        float* weight, float* loss_out, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(n_cells >= 0 && ratio >= 0, "hnm_weights_static: n=%lld ratio=%d", (long long)n_cells, ratio);
    CUTIE_CHECK_ARG(ce && labels && weight && loss_out, "hnm_weights_static: null pointer");
    if (n_cells > HNM_STATIC_MAX_CELLS) {
        cutie::set_error("hnm_weights_static: %lld cells exceed the static work area's %lld (use cutie_hnm_weights with a "
                         "caller workspace)", (long long)n_cells, (long long)HNM_STATIC_MAX_CELLS);
        return CUTIE_E_ARG;
    }
    void* ws = nullptr;
    if (int e = hnm_scratch_for_current_device(&ws)) return e;
    // the array keeps the last call's histograms: hnm_launch clears what this call's size uses before anything reads it
    return hnm_launch(ce, labels, n_cells, ratio, weight, loss_out, ws, stream);
}

extern "C" int cutie_softmax_ce_bwd(const float* probs, const int32_t* labels, const float* weight, int64_t n_cells,
                                    int C, void* dlogits, int ld, int dtype, const float* loss_scale,
                                    const float* elem_weight, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(n_cells >= 0 && C >= 1 && ld >= C, "softmax_ce_bwd: n=%lld C=%d ld=%d", (long long)n_cells, C, ld);
    if (n_cells == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(probs && labels && weight && dlogits, "softmax_ce_bwd: null pointer");
    int64_t g = cutie::cdiv(n_cells * C, 256);
    if (g > 2048) g = 2048;
    CUTIE_DISPATCH_DTYPE(dtype, DT,
        hipLaunchKernelGGL(softmax_ce_bwd_kernel<typename elem<DT>::type>, dim3((int)g), dim3(256), 0, stream, probs,
                           labels, weight, n_cells, C, (typename elem<DT>::type*)dlogits, ld, loss_scale, elem_weight));
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}
