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
//   scan of the previous level - then the cells equal to the threshold are counted per 256-cell
//   chunk so that the final kernel can take the first k_eq of them in index order.
//   All sums are fixed-order (per-chunk partials, then one thread): the loss is deterministic.
constexpr int HNM_T = 256;
constexpr int HNM_CHUNK = 256;        // cells per workgroup (one per lane; 1024 in round 2: a quarter of the workgroups, each four times as long, on kernels that are pure latency)
constexpr int HNM_B0 = 2048, HNM_B1 = 2048, HNM_B2 = 1024;

struct HnmWs {
    int* counts;     // [0] nfg [1] nbg [2..7] sel: b0, k1, b1, k2, (unused)
    int* hist0;      // [2048]
    int* hist1;      // [2048]
    int* hist2;      // [1024]
    int* tie_cnt;    // [chunks]
    float* sum_fg;   // [chunks]
    float* sum_sel;  // [chunks]
    int* cnt_fg;     // [chunks]   per-chunk cell counts, STORED by level 0 (512 workgroups adding to two words by atomics
    int* cnt_bg;     // [chunks]   queue up behind each other); level 1 sums them and leaves the totals in counts[0..1]
};
constexpr int HNM_INT_HEAD = 8 + HNM_B0 + HNM_B1 + HNM_B2;
constexpr int HNM_PER_CHUNK = 5;      // words per chunk behind the head

__device__ __forceinline__ HnmWs hnm_ws(void* ws, int chunks) {
    HnmWs w;
    w.counts = reinterpret_cast<int*>(ws);
    w.hist0 = w.counts + 8;
    w.hist1 = w.hist0 + HNM_B0;
    w.hist2 = w.hist1 + HNM_B1;
    w.tie_cnt = w.hist2 + HNM_B2;
    w.sum_fg = reinterpret_cast<float*>(w.tie_cnt + chunks);
    w.sum_sel = w.sum_fg + chunks;
    w.cnt_fg = reinterpret_cast<int*>(w.sum_sel + chunks);
    w.cnt_bg = w.cnt_fg + chunks;
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

__device__ __forceinline__ long long hnm_num_bg(long long nfg, long long nbg, int ratio, bool& all_bg) {
    const long long want = (long long)ratio * nfg;
    const long long num_bg = want < nbg ? want : nbg;     // model_cutie.py:124-125
    all_bg = num_bg == nbg;
    return num_bg;
}
__device__ __forceinline__ long long hnm_num_bg(const int* counts, int ratio, bool& all_bg) {
    const long long nfg = counts[0], nbg = counts[1];
    const long long want = (long long)ratio * nfg;
    const long long num_bg = want < nbg ? want : nbg;     // model_cutie.py:124-125
    all_bg = num_bg == nbg;
    return num_bg;
}

// Which bin holds the k-th largest key (k >= 1), and how many of that bin's keys are wanted.
// All 256 lanes take part; hist has `bins` (a multiple of 256) entries.  sh: 264 ints.
// Lane t owns bins [t*per, (t+1)*per); a suffix scan over the 256 lane sums (shuffles inside a wave, the four wave
// totals through LDS) finds the one lane whose bins hold rank k from the top, and that lane walks its own `per` bins.
// (Round 2 walked the 256 lane sums on ONE lane - a chain of dependent LDS reads that was most of the 10 us each of
// these launch-floor kernels took.)
__device__ __forceinline__ void hnm_select(const int* __restrict__ hist, int bins, long long k, int* sh, int& bin,
                                           long long& k_in) {
    const int per = bins / HNM_T;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int mine = 0;
    for (int i = 0; i < per; ++i) mine += hist[tid * per + i];
    int incl = mine;                              // sum over lanes >= mine inside the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_down(incl, o, 64);
        if (lane + o < 64) incl += u;
    }
    __syncthreads();
    if (lane == 0) sh[258 + wv] = incl;
    if (tid == 0) { sh[256] = 0; sh[257] = 0; }
    __syncthreads();
    long long above = 0;                          // keys in the waves above this one
    for (int w = wv + 1; w < HNM_T / 64; ++w) above += sh[258 + w];
    const long long hi = above + incl, lo = hi - mine;      // keys in bins >= mine / > mine
    if (lo < k && k <= hi) {
        long long rem = k - lo;
        int bsel = tid * per + per - 1;
        for (; bsel > tid * per; --bsel) {
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

// Cells of the flattened batch as HNM_CHUNK-cell chunks that never straddle a sample: chunk (b, cs) covers cells
// b*HW + cs*HNM_CHUNK .. of sample b (the last chunk of a sample may be short).  Chunk ids grow with the cell index, so
// "index order" is chunk order, then position.  The flat entry points run with one sample of n_cells cells.
struct CellMap {
    long long HW;
    int cps;           // chunks per sample
};
__device__ __forceinline__ long long chunk_cell(const CellMap& cm, int chunk, int pos, bool& valid) {
    const int b = chunk / cm.cps, cs = chunk - b * cm.cps;
    const long long local = (long long)cs * HNM_CHUNK + pos;
    valid = local < cm.HW;
    return (long long)b * cm.HW + local;
}
// labels as the loader's int8 grid (data_loader_json.py:455) or as int32
__device__ __forceinline__ int label_at(const void* labels, int label_bytes, long long i) {
    return label_bytes == 1 ? (int)reinterpret_cast<const int8_t*>(labels)[i] : reinterpret_cast<const int32_t*>(labels)[i];
}

// level 0: counts, foreground ce partial sums, histogram of the top 11 key bits of background cells
__global__ __launch_bounds__(HNM_T) void hnm_count_kernel(const float* __restrict__ ce, const void* __restrict__ labels,
                                                          int label_bytes, CellMap cm, int chunks, void* ws_) {
    __shared__ int hist[HNM_B0];
    __shared__ int shi[4];
    __shared__ float shf[4];
    const HnmWs ws = hnm_ws(ws_, chunks);
    const int tid = threadIdx.x;
    for (int i = tid; i < HNM_B0; i += HNM_T) hist[i] = 0;
    __syncthreads();
    int nfg = 0, nbg = 0;
    float sum_fg = 0.f;
#pragma unroll
    for (int u = 0; u < HNM_CHUNK / HNM_T; ++u) {
        bool in;
        const long long i = chunk_cell(cm, blockIdx.x, u * HNM_T + tid, in);
        if (in) {
            const float c = ce[i];
            if (label_at(labels, label_bytes, i) != 0) { ++nfg; sum_fg += c; }
            else { ++nbg; atomicAdd(&hist[sortable(c) >> 21], 1); }
        }
    }
    nfg = block_sum_i256(nfg, shi);
    nbg = block_sum_i256(nbg, shi);
    sum_fg = block_sum_f256(sum_fg, shf);
    if (tid == 0) {
        ws.cnt_fg[blockIdx.x] = nfg;
        ws.cnt_bg[blockIdx.x] = nbg;
        ws.sum_fg[blockIdx.x] = sum_fg;
    }
    for (int i = tid; i < HNM_B0; i += HNM_T)
        if (hist[i]) atomicAdd(&ws.hist0[i], hist[i]);
}

// level 1 / 2: redo the previous level's bin selection, histogram the next key bits of the
// background cells inside the selected prefix
template <int LEVEL>
__global__ __launch_bounds__(HNM_T) void hnm_hist_kernel(const float* __restrict__ ce, const void* __restrict__ labels,
                                                         int label_bytes, CellMap cm, int chunks, int ratio, void* ws_) {
    constexpr int BINS = LEVEL == 1 ? HNM_B1 : HNM_B2;
    __shared__ int hist[BINS];
    __shared__ int sh[264];
    const HnmWs ws = hnm_ws(ws_, chunks);
    const int tid = threadIdx.x;
    bool all_bg;
    long long num_bg;
    if (LEVEL == 1) {
        // the totals of level 0's per-chunk counts: every workgroup adds them for itself, workgroup 0 leaves them in
        // counts[0..1] for the kernels that follow
        int pf = 0, pb = 0;
        for (int c = tid; c < chunks; c += HNM_T) { pf += ws.cnt_fg[c]; pb += ws.cnt_bg[c]; }
        const int nfg = block_sum_i256(pf, sh), nbg = block_sum_i256(pb, sh);
        if (blockIdx.x == 0 && tid == 0) { ws.counts[0] = nfg; ws.counts[1] = nbg; }
        num_bg = hnm_num_bg((long long)nfg, (long long)nbg, ratio, all_bg);
    } else {
        num_bg = hnm_num_bg(ws.counts, ratio, all_bg);
    }
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
#pragma unroll
    for (int u = 0; u < HNM_CHUNK / HNM_T; ++u) {
        bool in;
        const long long i = chunk_cell(cm, blockIdx.x, u * HNM_T + tid, in);
        if (in && label_at(labels, label_bytes, i) == 0) {
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
__global__ __launch_bounds__(HNM_T) void hnm_tie_kernel(const float* __restrict__ ce, const void* __restrict__ labels,
                                                        int label_bytes, CellMap cm, int chunks, int ratio, void* ws_) {
    __shared__ int sh[264];
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
#pragma unroll
    for (int u = 0; u < HNM_CHUNK / HNM_T; ++u) {
        bool in;
        const long long i = chunk_cell(cm, blockIdx.x, u * HNM_T + tid, in);
        if (in && label_at(labels, label_bytes, i) == 0 && sortable(ce[i]) == thr) ++ties;
    }
    ties = block_sum_i256(ties, shi);
    if (tid == 0) ws.tie_cnt[blockIdx.x] = ties;
}

// The logits head's backward reduction, optionally taken by hnm_weight_kernel while it has the cell weights at hand:
// dy[n][c] = S * w[n] * (probs[n][c] - [c == label]) is the gradient of the loss in the instance-normalised logits y
// (cutie_softmax_ce_bwd); the norm's backward needs (sum dy, sum dy * y) per (sample, class): chunk (b, cs) STORES its
// share into red[b][cs][c][2].  Eight lanes share a cell, four classes each (C <= 32).
struct HeadRed {
    const float* probs;      // [N][C]
    const float* logits;     // [N][ld_l], the normalised logits = xhat
    int ld_l, C;
    const float* loss_scale; // nullable
    float* red;              // [B][cps][C][2]; null = no reduction (the plain cutie_hnm_weights)
};

// per-cell weights (1/nfg, 1/num_bg or 0) and the per-chunk sums of the selected background ce
__global__ __launch_bounds__(HNM_T) void hnm_weight_kernel(const float* __restrict__ ce, const void* __restrict__ labels,
                                                           int label_bytes, CellMap cm, int chunks, int ratio, void* ws_,
                                                           float* __restrict__ weight, HeadRed hr) {
    __shared__ int shi[4];
    __shared__ float shf[4];
    __shared__ int wave_cnt[4];
    __shared__ float wsh[HNM_CHUNK];
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
    for (int u = 0; u < HNM_CHUNK / HNM_T; ++u) {
        bool in;
        const long long i = chunk_cell(cm, blockIdx.x, u * HNM_T + tid, in);
        const int lab = in ? label_at(labels, label_bytes, i) : 1;
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
        float w = 0.f;
        if (in) {
            if (lab != 0) w = w_fg;
            else if (take) { w = w_bg; sum_sel += c; }
            weight[i] = w;
        }
        wsh[u * HNM_T + tid] = w;
    }
    sum_sel = block_sum_f256(sum_sel, shf);
    if (tid == 0) ws.sum_sel[blockIdx.x] = sum_sel;
    if (!hr.red) return;
    // ---- (sum dy, sum dy * y) of this chunk: lane = (cell slot g, class quad q); cells without a weight have dy = 0
    __shared__ float racc[32][32][2];      // [cell slot][class]
    __syncthreads();                       // wsh is complete
    const int q = tid & 7, g = tid >> 3;
    const float S = hr.loss_scale ? hr.loss_scale[0] : 1.f;
    float r1[4] = {0.f, 0.f, 0.f, 0.f}, r2[4] = {0.f, 0.f, 0.f, 0.f};
    if (q * 4 < hr.C) {
        constexpr int NP = HNM_CHUNK / 32;
        float wq[NP], pq[NP][4];
        float4 yq[NP];
        int lq[NP];
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {      // the loads of every weighted cell in flight together
            wq[pass] = wsh[pass * 32 + g];
            yq[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
            lq[pass] = -1;
#pragma unroll
            for (int e = 0; e < 4; ++e) pq[pass][e] = 0.f;
            if (wq[pass] != 0.f) {
                bool in;
                const long long i = chunk_cell(cm, blockIdx.x, pass * 32 + g, in);
                lq[pass] = label_at(labels, label_bytes, i);
                yq[pass] = *reinterpret_cast<const float4*>(hr.logits + i * hr.ld_l + q * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) if (q * 4 + e < hr.C) pq[pass][e] = hr.probs[i * hr.C + q * 4 + e];
            }
        }
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            if (wq[pass] == 0.f) continue;
            const float ye[4] = {yq[pass].x, yq[pass].y, yq[pass].z, yq[pass].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = q * 4 + e;
                if (c < hr.C) {
                    const float dy = S * wq[pass] * (pq[pass][e] - (lq[pass] == c ? 1.f : 0.f));
                    r1[e] += dy;
                    r2[e] = __builtin_fmaf(dy, ye[e], r2[e]);
                }
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { racc[g][q * 4 + e][0] = r1[e]; racc[g][q * 4 + e][1] = r2[e]; }
    __syncthreads();
    if (tid < 2 * hr.C) {
        const int c = tid >> 1, k = tid & 1;
        float t = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) t += racc[r][c][k];      // fixed order
        hr.red[((long long)blockIdx.x * hr.C + c) * 2 + k] = t;
    }
}

// the loss from the per-chunk sums: all 256 lanes of a workgroup take part (lane t adds chunks t, t + 256, ...; the lane sums
// are then added in a fixed order), lane 0 writes.  sh: 4 floats.
__device__ __forceinline__ void hnm_final(const HnmWs& ws, int chunks, int ratio, float* __restrict__ loss_out, float* sh) {
    bool all_bg;
    const long long num_bg = hnm_num_bg(ws.counts, ratio, all_bg);
    float sum_fg = 0.f, sum_sel = 0.f;
    for (int c = threadIdx.x; c < chunks; c += HNM_T) { sum_fg += ws.sum_fg[c]; sum_sel += ws.sum_sel[c]; }
    sum_fg = block_sum_f256(sum_fg, sh);
    sum_sel = block_sum_f256(sum_sel, sh);
    if (threadIdx.x != 0) return;
    // mean over an empty set is NaN (tf.reduce_mean of an empty tensor): 0/0
    loss_out[0] = sum_fg / (float)ws.counts[0] + sum_sel / (float)num_bg;
    loss_out[1] = (float)ws.counts[0];
    loss_out[2] = (float)num_bg;
}

__global__ __launch_bounds__(HNM_T) void hnm_final_kernel(int chunks, int ratio, void* ws_, float* __restrict__ loss_out) {
    __shared__ float sh[4];
    const HnmWs ws = hnm_ws(ws_, chunks);
    hnm_final(ws, chunks, ratio, loss_out, sh);
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


// ---------------------------------------------------------------------------------------------
// The logits head as a handful of small launches (round 2 took fourteen, ~155 us for 0.3 GFLOP):
//   cls_logits = Model.conv(1, 1, C) + bias + ReLU + instance norm   (model_cutie_aspp.py:72-73, model_framework.py:414-424)
//   softmax (model_framework.py:318-320), sparse softmax CE + hard negative mining (model_cutie.py:107-130)
// forward:  head_conv_kernel     a = relu(x W + b) in f32 (MFMA, one 256-pixel tile per workgroup) + the tile's moments
//           head_norm_ce_kernel  y = instance_norm(a), softmax, CE by wavefront reduction (eight lanes share a cell),
//                                and level 0 of the hard-negative select (counts, foreground CE sums, first histogram)
// backward: hnm_hist<1>, hnm_hist<2>, hnm_tie, hnm_weight (+ the norm gradient's two sums), head_bwd_kernel
//           (dz of the logits layer, its bias-gradient partials, the loss)
typedef __attribute__((ext_vector_type(8))) short hd_short8_t;
typedef __attribute__((ext_vector_type(4))) float hd_float4_t;
typedef _Float16 hd_half8_t __attribute__((ext_vector_type(8)));
constexpr int HEAD_TILE = 256;        // pixels per head_conv workgroup
constexpr int HEAD_MAX_KK = 8;        // cin <= 256

template <int DT>
__device__ __forceinline__ hd_float4_t head_mma(const uint4& a, const uint4& b, hd_float4_t c) {
    if constexpr (DT == CUTIE_BF16) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(hd_short8_t, a), __builtin_bit_cast(hd_short8_t, b), c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(hd_half8_t, a), __builtin_bit_cast(hd_half8_t, b), c, 0, 0, 0);
    }
}
__device__ __forceinline__ float hd_row16_sum(float v) {      // sum over the 16 lanes of a DPP row
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// grid (tiles per sample, B); 256 lanes = 4 waves x 64 pixels.  The MFMAs take the weight fragment as the row operand:
// lane (li, lg) of accumulator [m][nb] holds pixel m*16 + li and the four classes nb*16 + lg*4 .. +3.
template <int DT>
__global__ __launch_bounds__(256) void head_conv_kernel(const typename elem<DT>::type* __restrict__ x, int ld_x, int HW, int cin,
                                                        const float* __restrict__ w, const float* __restrict__ bias, int C,
                                                        float* __restrict__ a, int ld_a, float* __restrict__ stats, int tiles,
                                                        int* __restrict__ ws_zero, int ws_words) {
    typedef typename elem<DT>::type T;
    // the select's counters and histograms start at zero (this launch precedes head_norm_ce_kernel, which adds to them)
    for (int i = (blockIdx.y * gridDim.x + blockIdx.x) * 256 + threadIdx.x; i < ws_words; i += gridDim.x * gridDim.y * 256) ws_zero[i] = 0;
    __shared__ uint4 wf[2][HEAD_MAX_KK][64];
    __shared__ float2 wsum[4][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lg = lane >> 4;
    const int KK = cin >> 5;
    for (int idx = tid; idx < 2 * KK * 64; idx += 256) {
        const int ln = idx & 63, kk = (idx >> 6) % KK, nb = idx / (64 * KK);
        const int n = nb * 16 + (ln & 15), k0 = kk * 32 + (ln >> 4) * 8;
        T h[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) store_from_f32(&h[j], n < C ? w[(long long)(k0 + j) * C + n] : 0.f);
        wf[nb][kk][ln] = *reinterpret_cast<const uint4*>(h);
    }
    __syncthreads();
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * HEAD_TILE + wave * 64;
    hd_float4_t acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m) { acc[m][0] = hd_float4_t{0.f, 0.f, 0.f, 0.f}; acc[m][1] = hd_float4_t{0.f, 0.f, 0.f, 0.f}; }
    for (int kk = 0; kk < KK; ++kk) {
        uint4 xf[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int px = p0 + m * 16 + li;
            xf[m] = make_uint4(0u, 0u, 0u, 0u);
            if (px < HW) xf[m] = *reinterpret_cast<const uint4*>(x + ((long long)b * HW + px) * ld_x + kk * 32 + lg * 8);
        }
        const uint4 w0 = wf[0][kk][lane], w1 = wf[1][kk][lane];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[m][0] = head_mma<DT>(w0, xf[m], acc[m][0]);
            acc[m][1] = head_mma<DT>(w1, xf[m], acc[m][1]);
        }
    }
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int cb = nb * 16 + lg * 4;
        float bv[4], sh[4], s1[4], s2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bv[r] = cb + r < C ? bias[cb + r] : 0.f;
            sh[r] = fmaxf(bv[r], 0.f);          // norm_shift(bias, ReLU, c): the moments are kept about it
            s1[r] = 0.f;
            s2[r] = 0.f;
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int px = p0 + m * 16 + li;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = fmaxf(acc[m][nb][r] + bv[r], 0.f);
                const float d = px < HW ? v[r] - sh[r] : 0.f;
                s1[r] += d;
                s2[r] = __builtin_fmaf(d, d, s2[r]);
            }
            if (px < HW && cb < ld_a)
                *reinterpret_cast<float4*>(a + ((long long)b * HW + px) * ld_a + cb) = make_float4(v[0], v[1], v[2], v[3]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            s1[r] = hd_row16_sum(s1[r]);
            s2[r] = hd_row16_sum(s2[r]);
            if (li == 0) wsum[wave][cb + r] = make_float2(s1[r], s2[r]);
        }
    }
    __syncthreads();
    if (tid < C) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int wv = 0; wv < 4; ++wv) { t1 += wsum[wv][tid].x; t2 += wsum[wv][tid].y; }
        *reinterpret_cast<float2*>(stats + (((long long)b * tiles + blockIdx.x) * C + tid) * 2) = make_float2(t1, t2);
    }
}

// Sum of the T slots of a [T][C][2] partial-sum array for channel lanes tid < 32 (C <= 32), in slot order (norm.hip's
// fold), with the loads spread over the workgroup: eight slots per round trip instead of one.  lds: [8][32] float2.
__device__ __forceinline__ void fold_slots_par(const float* __restrict__ base, int T, int C, float2* lds, float& s, float& ss) {
    const int tid = threadIdx.x, c = tid & 31, tr = tid >> 5;
    s = 0.f;
    ss = 0.f;
    for (int t0 = 0; t0 < T; t0 += 8) {
        float2 v = make_float2(0.f, 0.f);
        if (t0 + tr < T && c < C) v = *reinterpret_cast<const float2*>(base + ((long long)(t0 + tr) * C + c) * 2);
        __syncthreads();
        lds[tr * 32 + c] = v;
        __syncthreads();
        if (tid < 32) {
#pragma unroll
            for (int k = 0; k < 8; ++k) { s += lds[k * 32 + tid].x; ss += lds[k * 32 + tid].y; }
        }
    }
}

// reductions over the eight lanes that share a cell, on the vector ALU (DPP: quad xor 1, quad xor 2, half-row mirror);
// every lane of the group ends up with the result
__device__ __forceinline__ float dpp_mov(float v, int ctrl_b1, int ctrl_4e, int ctrl_141, int which) {
    const int x = __builtin_bit_cast(int, v);
    int r;
    if (which == 0) r = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, true);
    else if (which == 1) r = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, true);
    else r = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, true);
    return __builtin_bit_cast(float, r);
}
__device__ __forceinline__ float grp8_max(float v) {
    v = fmaxf(v, dpp_mov(v, 0, 0, 0, 0));
    v = fmaxf(v, dpp_mov(v, 0, 0, 0, 1));
    return fmaxf(v, dpp_mov(v, 0, 0, 0, 2));
}
__device__ __forceinline__ float grp8_sum(float v) {
    v += dpp_mov(v, 0, 0, 0, 0);
    v += dpp_mov(v, 0, 0, 0, 1);
    return v + dpp_mov(v, 0, 0, 0, 2);
}

// grid (chunks per sample, B), 256 lanes: a wave takes eight cells at a time, eight lanes to a cell, four classes to
// a lane (C <= 32); the max, the sum of exponentials and the label's logit are reductions over the cell's eight lanes.
// labels == nullptr: inference (no CE, no counts).
__global__ __launch_bounds__(HNM_T) void head_norm_ce_kernel(const float* __restrict__ a, int ld_a, const float* __restrict__ stats,
                                                             int tiles, CellMap cm, int C, float eps, const float* __restrict__ bias,
                                                             float* __restrict__ logits, int ld_l, float* __restrict__ probs,
                                                             const void* __restrict__ labels, int label_bytes,
                                                             float* __restrict__ ce, int32_t* __restrict__ bad_labels, void* ws_,
                                                             int chunks) {
    __shared__ float tab[2][32];
    __shared__ int hist[HNM_B0];
    __shared__ int shi[4];
    __shared__ float shf[4];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, chunk = b * gridDim.x + blockIdx.x;
    __shared__ float2 fold[8 * 32];
    float fs, fss;
    fold_slots_par(stats + (long long)b * tiles * C * 2, tiles, C, fold, fs, fss);
    if (tid < 32) {
        float mean = 0.f, inv = 0.f;
        if (tid < C) norm_mean_inv(fs, fss, 1.f / (float)cm.HW, norm_shift(bias, 1, tid), eps, mean, inv);
        tab[0][tid] = mean;
        tab[1][tid] = inv;
    }
    if (labels) for (int i = tid; i < HNM_B0; i += HNM_T) hist[i] = 0;
    __syncthreads();
    const int q = tid & 7, g = tid >> 3;
    float mean[4], inv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) { mean[e] = tab[0][q * 4 + e]; inv[e] = tab[1][q * 4 + e]; }
    int nfg = 0, nbg = 0;
    float sum_fg = 0.f;
    constexpr int NP = HNM_CHUNK / 32;      // passes: 32 cells at a time, all loads of the chunk in flight together
    float4 av[NP];
    int labv[NP];
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        bool in;
        const long long i = chunk_cell(cm, chunk, pass * 32 + g, in);
        av[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (in && q * 4 < ld_a) av[pass] = *reinterpret_cast<const float4*>(a + i * ld_a + q * 4);
        labv[pass] = (labels && in) ? label_at(labels, label_bytes, i) : 0;
    }
#pragma unroll
    for (int pass = 0; pass < NP; ++pass) {
        bool in;
        const long long i = chunk_cell(cm, chunk, pass * 32 + g, in);
        float y[4] = {0.f, 0.f, 0.f, 0.f};
        if (in && q * 4 < ld_a) {
            const float4 v = av[pass];
            y[0] = (v.x - mean[0]) * inv[0]; y[1] = (v.y - mean[1]) * inv[1];
            y[2] = (v.z - mean[2]) * inv[2]; y[3] = (v.w - mean[3]) * inv[3];
        }
        if (in && q * 4 < ld_l) *reinterpret_cast<float4*>(logits + i * ld_l + q * 4) = make_float4(y[0], y[1], y[2], y[3]);
        float mx = -3.0e38f;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (q * 4 + e < C) mx = fmaxf(mx, y[e]);
        mx = grp8_max(mx);
        float ex[4], sum = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) { ex[e] = q * 4 + e < C ? expf(y[e] - mx) : 0.f; sum += ex[e]; }
        sum = grp8_sum(sum);
        if (in && probs) {
            const float r = 1.f / sum;
#pragma unroll
            for (int e = 0; e < 4; ++e) if (q * 4 + e < C) probs[i * C + q * 4 + e] = ex[e] * r;
        }
        if (!labels) continue;
        const int lab = labv[pass];
        float yl = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) if (q * 4 + e == lab) yl = y[e];
        yl = grp8_sum(yl);
        if (in && q == 0) {
            float loss;
            if (lab >= 0 && lab < C) loss = logf(sum) - (yl - mx);
            else { loss = nanf(""); atomicAdd(bad_labels, 1); }
            ce[i] = loss;
            if (lab != 0) { ++nfg; sum_fg += loss; }
            else { ++nbg; atomicAdd(&hist[sortable(loss) >> 21], 1); }
        }
    }
    if (!labels) return;
    const HnmWs ws = hnm_ws(ws_, chunks);
    nfg = block_sum_i256(nfg, shi);
    nbg = block_sum_i256(nbg, shi);
    sum_fg = block_sum_f256(sum_fg, shf);
    if (tid == 0) {
        ws.cnt_fg[chunk] = nfg;
        ws.cnt_bg[chunk] = nbg;
        ws.sum_fg[chunk] = sum_fg;
    }
    for (int i = tid; i < HNM_B0; i += HNM_T)
        if (hist[i]) atomicAdd(&ws.hist0[i], hist[i]);
}

// dz of the logits layer: dy = S * w * (probs - onehot) (cutie_softmax_ce_bwd), then the instance norm's and the
// ReLU's gradient, dz = [a > 0] * inv * (dy - mean(dy) - y * mean(dy * y)) (cutie_in_bwd_apply's operations), in the
// activation dtype for the data gradient and the weight gradient that follow; chunk (b, cs) STORES row b*cps + cs of
// the bias-gradient partials.  Workgroup (0, 0) also writes the loss.  grid (chunks per sample, B).
template <typename TZ>
__global__ __launch_bounds__(HNM_T) void head_bwd_kernel(const float* __restrict__ a, int ld_a, const float* __restrict__ stats, int tiles,
                                                         const float* __restrict__ logits, int ld_l, const float* __restrict__ probs,
                                                         const void* __restrict__ labels, int label_bytes,
                                                         const float* __restrict__ weight, const float* __restrict__ red,
                                                         CellMap cm, int C, float eps, const float* __restrict__ bias,
                                                         const float* __restrict__ loss_scale, TZ* __restrict__ dz, int ld_dz,
                                                         float* __restrict__ dbias, void* ws_, int chunks, int ratio,
                                                         float* __restrict__ loss_out) {
    __shared__ float tab[3][32];          // inv, mean(dy), mean(dy * y)
    __shared__ float racc[32][32];
    const int tid = threadIdx.x;
    const int b = blockIdx.y, chunk = b * gridDim.x + blockIdx.x;
    if (chunk == 0) {
        __shared__ float shfin[4];
        hnm_final(hnm_ws(ws_, chunks), chunks, ratio, loss_out, shfin);
    }
    __shared__ float2 fold[8 * 32];
    float fs, fss, r1, r2;
    fold_slots_par(stats + (long long)b * tiles * C * 2, tiles, C, fold, fs, fss);
    fold_slots_par(red + (long long)b * cm.cps * C * 2, cm.cps, C, fold, r1, r2);
    if (tid < 32) {
        float inv = 0.f;
        if (tid < C) {
            float mean;
            const float inv_n = 1.f / (float)cm.HW;
            norm_mean_inv(fs, fss, inv_n, norm_shift(bias, 1, tid), eps, mean, inv);
            r1 *= inv_n;
            r2 *= inv_n;
        } else {
            r1 = 0.f;
            r2 = 0.f;
        }
        tab[0][tid] = inv;
        tab[1][tid] = r1;
        tab[2][tid] = r2;
    }
    __syncthreads();
    const int q = tid & 7, g = tid >> 3;
    const float S = loss_scale ? loss_scale[0] : 1.f;
    float inv[4], m1[4], m2[4], sb[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e) { inv[e] = tab[0][q * 4 + e]; m1[e] = tab[1][q * 4 + e]; m2[e] = tab[2][q * 4 + e]; }
    if (q * 4 < ld_a) {
        constexpr int NP = HNM_CHUNK / 32;
        float4 avv[NP], yvv[NP];
        float wv[NP], pv[NP][4];
        int lv[NP];
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {      // every load of the chunk in flight together (probs of unweighted cells too)
            bool in;
            const long long i = chunk_cell(cm, chunk, pass * 32 + g, in);
            avv[pass] = yvv[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
            wv[pass] = 0.f;
            lv[pass] = -1;
#pragma unroll
            for (int e = 0; e < 4; ++e) pv[pass][e] = 0.f;
            if (in) {
                avv[pass] = *reinterpret_cast<const float4*>(a + i * ld_a + q * 4);
                yvv[pass] = *reinterpret_cast<const float4*>(logits + i * ld_l + q * 4);
                wv[pass] = weight[i];
                lv[pass] = label_at(labels, label_bytes, i);
#pragma unroll
                for (int e = 0; e < 4; ++e) if (q * 4 + e < C) pv[pass][e] = probs[i * C + q * 4 + e];
            }
        }
#pragma unroll
        for (int pass = 0; pass < NP; ++pass) {
            bool in;
            const long long i = chunk_cell(cm, chunk, pass * 32 + g, in);
            if (!in) continue;
            const float4 av = avv[pass], yv = yvv[pass];
            const float ae[4] = {av.x, av.y, av.z, av.w}, ye[4] = {yv.x, yv.y, yv.z, yv.w};
            const float w = wv[pass];
            const int lab = lv[pass];
            TZ o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = q * 4 + e;
                float dy = 0.f;
                if (w != 0.f && c < C) dy = S * w * (pv[pass][e] - (lab == c ? 1.f : 0.f));
                float d = inv[e] * __builtin_fmaf(-ye[e], m2[e], dy - m1[e]);
                if (!(ae[e] > 0.f) || c >= C) d = 0.f;
                sb[e] += d;
                store_from_f32(&o[e], d);
            }
            if (q * 4 < ld_dz) {
                if constexpr (sizeof(TZ) == 2) *reinterpret_cast<uint2*>(dz + i * ld_dz + q * 4) = *reinterpret_cast<const uint2*>(o);
                else *reinterpret_cast<float4*>(dz + i * ld_dz + q * 4) = *reinterpret_cast<const float4*>(o);
            }
        }
    }
    if (!dbias) return;
#pragma unroll
    for (int e = 0; e < 4; ++e) racc[g][q * 4 + e] = sb[e];
    __syncthreads();
    if (tid < C) {
        float t = 0.f;
#pragma unroll 8
        for (int r = 0; r < 32; ++r) t += racc[r][tid];
        dbias[(long long)chunk * C + tid] = t;
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
    return (size_t)(HNM_INT_HEAD + HNM_PER_CHUNK * hnm_chunks(n_cells)) * 4 + 64;
}

// the six launches of the select over a zero-filled work area `ws` (16-byte aligned, hnm_ws layout)
static int hnm_launch(const float* ce, const int32_t* labels, int64_t n_cells, int ratio, float* weight, float* loss_out,
                      void* ws, hipStream_t stream) {
    const int chunks = (int)hnm_chunks(n_cells);
    CUTIE_CHECK_HIP(hipMemsetAsync(ws, 0, (size_t)(HNM_INT_HEAD + HNM_PER_CHUNK * chunks) * 4, stream));
    const dim3 g(chunks), t(HNM_T);
    const CellMap cm = {(long long)n_cells, chunks};      // one "sample" of n_cells cells
    const HeadRed none = {nullptr, nullptr, 0, 0, nullptr, nullptr};
    hipLaunchKernelGGL(hnm_count_kernel, g, t, 0, stream, ce, (const void*)labels, 4, cm, chunks, ws);
    hipLaunchKernelGGL(hnm_hist_kernel<1>, g, t, 0, stream, ce, (const void*)labels, 4, cm, chunks, ratio, ws);
    hipLaunchKernelGGL(hnm_hist_kernel<2>, g, t, 0, stream, ce, (const void*)labels, 4, cm, chunks, ratio, ws);
    hipLaunchKernelGGL(hnm_tie_kernel, g, t, 0, stream, ce, (const void*)labels, 4, cm, chunks, ratio, ws);
    hipLaunchKernelGGL(hnm_weight_kernel, g, t, 0, stream, ce, (const void*)labels, 4, cm, chunks, ratio, ws, weight, none);
    hipLaunchKernelGGL(hnm_final_kernel, dim3(1), dim3(HNM_T), 0, stream, chunks, ratio, ws, loss_out);
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
// select - three histograms, eight counters and three words per 256-cell chunk - lives in a device array of the
// code object instead of a caller workspace.  Each GPU the library runs on holds its own copy at its own address
// until the process ends; the host only ever uses the address hipGetSymbolAddress returned for the CURRENT device
// (the array's name is a host-side stand-in, never a device pointer).
constexpr int64_t HNM_STATIC_MAX_CELLS = (int64_t)CUTIE_HNM_STATIC_MAX_CELLS;
constexpr size_t HNM_STATIC_BYTES = (size_t)(HNM_INT_HEAD + HNM_PER_CHUNK * (HNM_STATIC_MAX_CELLS / HNM_CHUNK)) * 4;
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

// ---------------------------------------------------------------------------------------------
extern "C" int cutie_head_tiles(int HW) { return HW < 1 ? 1 : (HW + HEAD_TILE - 1) / HEAD_TILE; }
extern "C" int cutie_head_chunks(int HW) { return HW < 1 ? 1 : (HW + HNM_CHUNK - 1) / HNM_CHUNK; }
extern "C" size_t cutie_head_workspace_bytes(int B, int HW) {
    return (size_t)(HNM_INT_HEAD + HNM_PER_CHUNK * (int64_t)(B < 1 ? 1 : B) * cutie_head_chunks(HW)) * 4 + 64;
}

static int head_check(const char* name, int B, int HW, int C, int ld_a, int ld_l, const void* a, const void* logits) {
    CUTIE_CHECK_ARG(B >= 0 && HW >= 1 && C >= 1 && C <= 32, "%s: B=%d HW=%d C=%d (C <= 32)", name, B, HW, C);
    CUTIE_CHECK_ARG(ld_a >= C && ld_a % 4 == 0 && ld_l >= C && ld_l % 4 == 0 && cutie::aligned16(a) && cutie::aligned16(logits),
                    "%s: a / logits need 16-byte aligned rows (ld %d / %d)", name, ld_a, ld_l);
    CUTIE_CHECK_ARG((int64_t)B * HW < (1ll << 31) - HNM_CHUNK, "%s: too many cells", name);
    return CUTIE_OK;
}

extern "C" int cutie_head_fwd(const void* x, int ld_x, int dtype, int B, int HW, int cin, const float* w_hwio,
                              const float* bias, int C, float eps, float* a, int ld_a, float* stats, int stats_tiles,
                              float* logits, int ld_logits, float* probs, const void* labels, int label_bytes, float* ce,
                              int32_t* bad_labels, void* workspace, size_t workspace_bytes, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int e = head_check("head_fwd", B, HW, C, ld_a, ld_logits, a, logits)) return e;
    CUTIE_CHECK_ARG(dtype == CUTIE_BF16 || dtype == CUTIE_F16, "head_fwd: 16-bit activations only (dtype %d)", dtype);
    CUTIE_CHECK_ARG(cin >= 32 && cin % 32 == 0 && cin <= 32 * HEAD_MAX_KK && ld_x >= cin && (ld_x * 2) % 16 == 0 && cutie::aligned16(x),
                    "head_fwd: cin=%d (multiple of 32, <= %d), ld_x=%d", cin, 32 * HEAD_MAX_KK, ld_x);
    if (B == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(x && w_hwio && bias && a && stats && logits, "head_fwd: null pointer");
    CUTIE_CHECK_ARG(stats_tiles >= cutie_head_tiles(HW), "head_fwd: stats_tiles=%d < cutie_head_tiles(%d)", stats_tiles, HW);
    CUTIE_CHECK_ARG(!labels || ((label_bytes == 1 || label_bytes == 4) && ce && bad_labels), "head_fwd: labels need ce and bad_labels");
    const int tiles = cutie_head_tiles(HW), cps = cutie_head_chunks(HW);
    void* ws = nullptr;
    if (labels) {
        const size_t need = cutie_head_workspace_bytes(B, HW);
        if (!workspace || workspace_bytes < need) {
            cutie::set_error("head_fwd: workspace %zu < %zu", workspace_bytes, need);
            return CUTIE_E_WORKSPACE;
        }
        ws = (void*)(((uintptr_t)workspace + 15) & ~(uintptr_t)15);      // cleared by head_conv_kernel's workgroups
    }
    const int ws_words = labels ? HNM_INT_HEAD : 0;      // the per-chunk words are all STORED before they are read
    if (dtype == CUTIE_BF16) {
        hipLaunchKernelGGL(head_conv_kernel<CUTIE_BF16>, dim3(tiles, B), dim3(256), 0, stream, (const bf16_t*)x, ld_x, HW, cin,
                           w_hwio, bias, C, a, ld_a, stats, stats_tiles, (int*)ws, ws_words);
    } else {
        hipLaunchKernelGGL(head_conv_kernel<CUTIE_F16>, dim3(tiles, B), dim3(256), 0, stream, (const f16_t*)x, ld_x, HW, cin,
                           w_hwio, bias, C, a, ld_a, stats, stats_tiles, (int*)ws, ws_words);
    }
    CUTIE_CHECK_LAUNCH();
    const CellMap cm = {(long long)HW, cps};
    hipLaunchKernelGGL(head_norm_ce_kernel, dim3(cps, B), dim3(HNM_T), 0, stream, (const float*)a, ld_a, (const float*)stats,
                       stats_tiles, cm, C, eps, bias, logits, ld_logits, probs, labels, label_bytes, ce, bad_labels, ws, B * cps);
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

extern "C" int cutie_head_bwd(const float* a, int ld_a, const float* stats, int stats_tiles, const float* logits,
                              int ld_logits, const float* probs, const void* labels, int label_bytes, const float* ce,
                              int B, int HW, int C, float eps, const float* bias, int ratio, const float* loss_scale,
                              float* weight, float* red, float* loss_out, void* dz, int ld_dz, int dz_dtype,
                              float* dbias_part, void* workspace, size_t workspace_bytes, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    if (int e = head_check("head_bwd", B, HW, C, ld_a, ld_logits, a, logits)) return e;
    CUTIE_CHECK_ARG(ratio >= 0 && (label_bytes == 1 || label_bytes == 4), "head_bwd: ratio=%d label_bytes=%d", ratio, label_bytes);
    if (B == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(a && stats && logits && probs && labels && ce && bias && weight && red && loss_out && dz,
                    "head_bwd: null pointer");
    CUTIE_CHECK_ARG(stats_tiles >= cutie_head_tiles(HW), "head_bwd: stats_tiles=%d", stats_tiles);
    const int es = dz_dtype == CUTIE_F32 ? 4 : 2;
    CUTIE_CHECK_ARG(ld_dz >= C && (ld_dz * es) % (4 * es) == 0 && (reinterpret_cast<uintptr_t>(dz) % (4 * es)) == 0,
                    "head_bwd: dz rows must hold whole groups of four classes (ld_dz=%d)", ld_dz);
    const size_t need = cutie_head_workspace_bytes(B, HW);
    if (!workspace || workspace_bytes < need) {
        cutie::set_error("head_bwd: workspace %zu < %zu", workspace_bytes, need);
        return CUTIE_E_WORKSPACE;
    }
    void* ws = (void*)(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
    const int cps = cutie_head_chunks(HW), chunks = B * cps;
    const CellMap cm = {(long long)HW, cps};
    const dim3 g(chunks), t(HNM_T);
    const HeadRed hr = {probs, logits, ld_logits, C, loss_scale, red};
    hipLaunchKernelGGL(hnm_hist_kernel<1>, g, t, 0, stream, ce, labels, label_bytes, cm, chunks, ratio, ws);
    hipLaunchKernelGGL(hnm_hist_kernel<2>, g, t, 0, stream, ce, labels, label_bytes, cm, chunks, ratio, ws);
    hipLaunchKernelGGL(hnm_tie_kernel, g, t, 0, stream, ce, labels, label_bytes, cm, chunks, ratio, ws);
    hipLaunchKernelGGL(hnm_weight_kernel, g, t, 0, stream, ce, labels, label_bytes, cm, chunks, ratio, ws, weight, hr);
    CUTIE_CHECK_LAUNCH();
    CUTIE_DISPATCH_DTYPE(dz_dtype, DT,
        hipLaunchKernelGGL(head_bwd_kernel<typename elem<DT>::type>, dim3(cps, B), t, 0, stream, a, ld_a, stats, stats_tiles,
                           logits, ld_logits, probs, labels, label_bytes, (const float*)weight, (const float*)red, cm, C, eps,
                           bias, loss_scale, (typename elem<DT>::type*)dz, ld_dz, dbias_part, ws, chunks, ratio, loss_out));
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
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
