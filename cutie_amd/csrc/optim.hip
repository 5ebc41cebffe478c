// K12 / K14: L2 regulariser, global-norm pieces and Adam over flat f32 parameter buffers,
// plus the weight packer that feeds the conv kernels.
// l2_regularizer /root/reference/model_framework.py:427-432; clip_by_global_norm + Adam
// /root/reference/main_train_json.py:152-156.  Reductions are two-stage with a fixed grid and
// a fixed summation order, so every data-parallel replica computes bit-identical norms.
#include "common.h"

namespace {
using namespace cutie;

constexpr int RED_BLOCKS = CUTIE_REDUCE_SCRATCH;

__device__ __forceinline__ void block_sum_store(float v, float* dst) {
    __shared__ float sh[4];
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) *dst = sh[0] + sh[1] + sh[2] + sh[3];
}

template <typename T>
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const T* __restrict__ x, int64_t n,
                                                            float* __restrict__ partials) {
    float s = 0.f;
    const int64_t n4 = n / 4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)RED_BLOCKS * 256) {
        const float4 v = load4(x + i * 4);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    if (blockIdx.x == 0) {
        for (int64_t i = n4 * 4 + threadIdx.x; i < n; i += 256) {
            const float v = load_as_f32(x + i);
            s += v * v;
        }
    }
    block_sum_store(s, partials + blockIdx.x);
}

__global__ __launch_bounds__(256) void l2_partial_kernel(const float* __restrict__ w, float* __restrict__ g,
                                                         int64_t n, float wd, float* __restrict__ partials,
                                                         const float* __restrict__ loss_scale) {
    float s = 0.f;
    const float inv = loss_scale ? loss_scale[1] : 1.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)RED_BLOCKS * 256) {
        const float v = w[i];
        g[i] = loss_scale ? g[i] * inv + wd * v : g[i] + wd * v;
        s += v * v;
    }
    block_sum_store(s, partials + blockIdx.x);
}

__global__ __launch_bounds__(256) void final_sum_kernel(const float* __restrict__ partials, float scale,
                                                        float* __restrict__ out, const float* __restrict__ loss_scale) {
    float s = 0.f;
    for (int i = threadIdx.x; i < RED_BLOCKS; i += 256) s += partials[i];
    __shared__ float sh[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float inv = loss_scale ? loss_scale[1] : 1.f;    // the summed squares carry the scale twice
        *out += scale * inv * inv * (sh[0] + sh[1] + sh[2] + sh[3]);
    }
}

// dynamic loss scale, kept on the device so that no step waits for the host:
// state = {scale, 1/scale, good steps in a row, steps skipped so far, steps skipped as invalid, -, -, -}
__global__ void loss_scale_update_kernel(float* __restrict__ state, const float* __restrict__ gnorm_sq, float backoff,
                                         float growth, int interval, float min_scale, float max_scale,
                                         const int32_t* __restrict__ invalid) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float s = state[0], good = state[2];
    const float n = *gnorm_sq;
    if (invalid && *invalid != 0) {
        // the step's results were declared invalid (a conv tile gave up waiting for its sample's other tiles): the
        // optimizer skipped it; the scale has nothing to learn from it
        state[3] += 1.f;
        state[4] += 1.f;
        return;
    }
    if (!(fabsf(n) <= 3.0e38f)) {          // inf or NaN: the step was skipped by adam_kernel
        s = fmaxf(s * backoff, min_scale);
        good = 0.f;
        state[3] += 1.f;
    } else if (++good >= (float)interval) {
        s = fminf(s * growth, max_scale);
        good = 0.f;
    }
    state[0] = s;
    state[1] = 1.f / s;
    state[2] = good;
}

// steps the device skipped (f16 overflow, invalid results) never reached m and v: the bias correction belongs to the count of
// steps that did, t - skipped, not to the caller's t (lr_t = lr * c(t), c(t) = sqrt(1 - b2^t) / (1 - b1^t))
__device__ float rebased_lr(float lr_t, float b1, float b2, int t, float skipped) {
    if (!(skipped > 0.f)) return lr_t;
    const double te = fmax((double)t - (double)skipped, 1.0);
    const double c_t = sqrt(1.0 - pow((double)b2, (double)t)) / (1.0 - pow((double)b1, (double)t));
    const double c_e = sqrt(1.0 - pow((double)b2, te)) / (1.0 - pow((double)b1, te));
    return (float)((double)lr_t * (c_e / c_t));
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, float lr_t,
                                                   float b1, float b2, float eps, const float* __restrict__ gnorm_sq,
                                                   float clip, float grad_scale, int t,
                                                   const float* __restrict__ loss_scale,
                                                   const int32_t* __restrict__ invalid) {
    float scale = grad_scale;
    if (invalid && *invalid != 0) return;      // the gradients are not to be trusted: leave p, m, v untouched
    if (loss_scale && t > 0) {
        __shared__ float lr_sh;
        if (threadIdx.x == 0) lr_sh = rebased_lr(lr_t, b1, b2, t, loss_scale[3]);      // once per workgroup, not per lane
        __syncthreads();
        lr_t = lr_sh;
    }
    // an overflowed (f16, loss-scaled) backward shows as a non-finite norm: leave p, m, v untouched
    if (gnorm_sq && !(fabsf(*gnorm_sq) <= 3.0e38f)) return;
    if (gnorm_sq) {
        // tf.clip_by_global_norm: t * clip_norm / max(global_norm, clip_norm)
        const float norm = sqrtf(*gnorm_sq) * grad_scale;
        scale = grad_scale * (clip / fmaxf(norm, clip));
    }
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float gi = g[i] * scale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= lr_t * mi / (sqrtf(vi) + eps);
    }
}

// HWIO f32 [tap][cin][cout] -> [tap][cout][cin_pad] (fwd operand) / [tap][cin][cout_pad] (dgrad operand)
template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ w, int taps, int cin, int cout,
                                                   T* __restrict__ w_fwd, int cin_pad, T* __restrict__ w_dg,
                                                   int cout_pad) {
    const int64_t n_fwd = w_fwd ? (int64_t)taps * cout * cin_pad : 0;
    const int64_t n_dg = w_dg ? (int64_t)taps * cin * cout_pad : 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_fwd + n_dg;
         i += (int64_t)gridDim.x * blockDim.x) {
        if (i < n_fwd) {
            const int c = (int)(i % cin_pad);
            const int64_t r = i / cin_pad;
            const int n = (int)(r % cout);
            const int t = (int)(r / cout);
            const float v = c < cin ? w[((int64_t)t * cin + c) * cout + n] : 0.f;
            store_from_f32(w_fwd + i, v);
        } else {
            const int64_t j = i - n_fwd;
            const int n = (int)(j % cout_pad);
            const int64_t r = j / cout_pad;   // t*cin + c
            const float v = n < cout ? w[r * cout + n] : 0.f;
            store_from_f32(w_dg + j, v);
        }
    }
}

}  // namespace

static int launch_final(float* partials, float scale, float* out, const float* loss_scale, hipStream_t stream) {
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, stream, partials, scale, out, loss_scale);
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

extern "C" int cutie_sumsq(const void* x, int64_t n, int dtype, float* out, float* scratch, const float* loss_scale,
                           cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(n >= 0 && out && scratch, "sumsq: bad arguments");
    if (n == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(x && cutie::aligned16(x), "sumsq: x must be 16-byte aligned");
    CUTIE_DISPATCH_DTYPE(dtype, DT,
        hipLaunchKernelGGL(sumsq_partial_kernel<typename elem<DT>::type>, dim3(RED_BLOCKS), dim3(256), 0, stream,
                           (const typename elem<DT>::type*)x, n, scratch));
    CUTIE_CHECK_LAUNCH();
    return launch_final(scratch, 1.f, out, loss_scale, stream);
}

extern "C" int cutie_l2_add(const float* w, float* g, int64_t n, float wd, float* reg_loss, float* scratch,
                            const float* loss_scale, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(n >= 0 && reg_loss && scratch, "l2_add: bad arguments");
    if (n == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(w && g, "l2_add: null pointer");
    hipLaunchKernelGGL(l2_partial_kernel, dim3(RED_BLOCKS), dim3(256), 0, stream, w, g, n, wd, scratch, loss_scale);
    CUTIE_CHECK_LAUNCH();
    return launch_final(scratch, 0.5f * wd, reg_loss, nullptr, stream);
}

extern "C" int cutie_loss_scale_init(float* state, float scale, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(state && scale > 0.f && scale <= 3.0e38f, "loss_scale_init: scale %g", (double)scale);
    const float h[CUTIE_LOSS_SCALE_STATE] = {scale, 1.f / scale, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    CUTIE_CHECK_HIP(hipMemcpyAsync(state, h, sizeof(h), hipMemcpyHostToDevice, stream));
    CUTIE_CHECK_HIP(hipStreamSynchronize(stream));   // h lives on this stack frame
    return CUTIE_OK;
}

extern "C" int cutie_loss_scale_update(float* state, const float* gnorm_sq, float backoff, float growth, int interval,
                                       float min_scale, float max_scale, const int32_t* invalid, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(state && gnorm_sq, "loss_scale_update: null pointer");
    CUTIE_CHECK_ARG(backoff > 0.f && backoff <= 1.f && growth >= 1.f && interval >= 1 && min_scale > 0.f && max_scale >= min_scale,
                    "loss_scale_update: backoff %g growth %g interval %d range [%g, %g]", (double)backoff, (double)growth,
                    interval, (double)min_scale, (double)max_scale);
    hipLaunchKernelGGL(loss_scale_update_kernel, dim3(1), dim3(64), 0, stream, state, gnorm_sq, backoff, growth, interval,
                       min_scale, max_scale, invalid);
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

extern "C" int cutie_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr_t, float beta1,
                               float beta2, float eps, const float* gnorm_sq, float clip, float grad_scale,
                               int t, const float* loss_scale, const int32_t* invalid, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(n >= 0, "adam_step: n < 0");
    if (n == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(p && g && m && v, "adam_step: null pointer");
    int64_t grid = cutie::cdiv(n, 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(adam_kernel, dim3((int)grid), dim3(256), 0, stream, p, g, m, v, n, lr_t, beta1, beta2, eps,
                       gnorm_sq, clip, grad_scale, t, loss_scale, invalid);
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

namespace {
struct PackBatch {
    cutie_pack_desc d[CUTIE_MAX_BATCH];
};
// One workgroup per (tap, 64 cin x 64 cout) tile: the f32 tile is read once in rows of 64 consecutive cout
// (coalesced), kept in LDS, and written out twice - as it stands into w_dgrad [tap][cin][cout_pad] and
// transposed into w_fwd [tap][cout][cin_pad] (rows of 64 consecutive cin).  The plain per-element form read
// w_fwd's source at a stride of cout: 178 MB of fetches for 44 MB of weights (profiles/r01_f).
template <typename T>
__global__ __launch_bounds__(256) void pack_batched_kernel(PackBatch pb) {
    __shared__ float tile[64][65];
    const cutie_pack_desc d = pb.d[blockIdx.y];
    const float* __restrict__ w = d.w_hwio;
    T* __restrict__ w_fwd = (T*)d.w_fwd;
    T* __restrict__ w_dg = (T*)d.w_dgrad;
    const int cin = d.cin, cout = d.cout;
    const int c_ext = w_fwd ? d.cin_pad : cin;       // rows / columns that have to be produced (padding = zeros)
    const int n_ext = w_dg ? d.cout_pad : cout;
    const int CT = (c_ext + 63) / 64, NT = (n_ext + 63) / 64;
    const int tiles = d.taps * CT * NT;
    const int lo = threadIdx.x & 63, hi = threadIdx.x >> 6;      // hi = 0..3
    for (int tl = blockIdx.x; tl < tiles; tl += gridDim.x) {
        const int nt = tl % NT;
        const int ct = (tl / NT) % CT;
        const int t = tl / (NT * CT);
        const int c0 = ct * 64, n0 = nt * 64;
        __syncthreads();                 // the previous tile has been read out
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const int c = c0 + hi + 4 * k, n = n0 + lo;
            tile[hi + 4 * k][lo] = (c < cin && n < cout) ? w[((int64_t)t * cin + c) * cout + n] : 0.f;
        }
        __syncthreads();
        if (w_dg) {
#pragma unroll 4
            for (int k = 0; k < 16; ++k) {
                const int c = c0 + hi + 4 * k, n = n0 + lo;
                if (c < cin && n < d.cout_pad) store_from_f32(w_dg + ((int64_t)t * cin + c) * d.cout_pad + n, tile[hi + 4 * k][lo]);
            }
        }
        if (w_fwd) {
#pragma unroll 4
            for (int k = 0; k < 16; ++k) {
                const int n = n0 + hi + 4 * k, c = c0 + lo;
                if (n < cout && c < d.cin_pad) store_from_f32(w_fwd + ((int64_t)t * cout + n) * d.cin_pad + c, tile[lo][hi + 4 * k]);
            }
        }
    }
}
}  // namespace

// ---------------------------------------------------------------------------------------------
// The optimizer tail of a train step in two launches (cutie_opt_norm, cutie_opt_step).  Round 2 ran it as eleven:
// l2 (x2), their final sums, two sums of squares and their final sums, Adam, the weight packer, the loss-scale update -
// every one of them a pass over (part of) the same 54.6 MB.  Here pass 1 reads w and g once and leaves 2 x 1024 partial
// sums; pass 2 folds them (every workgroup for itself, in one fixed order: the same bits everywhere), applies
// g' = (g / S + wd * w) * clip scale on the fly, updates p, m, v, and - for the conv kernels, which it walks in
// 64 x 64 (cin x cout) tiles - writes the two 16-bit operand layouts the conv kernels read straight from the updated tile.
namespace {
constexpr int OPT_T = 256;

template <typename T>
__global__ __launch_bounds__(OPT_T) void opt_norm_kernel(const float* __restrict__ w, const float* __restrict__ g, int64_t n_l2,
                                                         int64_t n_dense, float wd, const T* __restrict__ rows,
                                                         int64_t n_rows, const float* __restrict__ state,
                                                         float* __restrict__ partials) {
    const float inv = state ? state[1] : 1.f;
    float s = 0.f, sw = 0.f;
    const int64_t stride = (int64_t)RED_BLOCKS * OPT_T;
    const int64_t t0 = (int64_t)blockIdx.x * OPT_T + threadIdx.x;
    for (int64_t i = t0; i < n_l2 / 4; i += stride) {
        const float4 wv = load4(w + i * 4), gv = load4(g + i * 4);
        const float a = gv.x * inv + wd * wv.x, b = gv.y * inv + wd * wv.y, c = gv.z * inv + wd * wv.z, d = gv.w * inv + wd * wv.w;
        s += a * a + b * b + c * c + d * d;
        sw += wv.x * wv.x + wv.y * wv.y + wv.z * wv.z + wv.w * wv.w;
    }
    for (int64_t i = n_l2 / 4 + t0; i < n_dense / 4; i += stride) {      // biases: no L2 term
        const float4 gv = load4(g + i * 4);
        const float a = gv.x * inv, b = gv.y * inv, c = gv.z * inv, d = gv.w * inv;
        s += a * a + b * b + c * c + d * d;
    }
    if (rows) {
        // the embedding's share of tf.clip_by_global_norm: its IndexedSlices values, one row per grid cell, still scaled
        float r = 0.f;
        for (int64_t i = t0; i < n_rows / 4; i += stride) {
            const float4 v = load4(rows + i * 4);
            r += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
        }
        s += r * inv * inv;
    }
    block_sum_store(s, partials + blockIdx.x);
    __syncthreads();
    block_sum_store(sw, partials + RED_BLOCKS + blockIdx.x);
}

struct OptLayer {
    long long off;       // of the layer's HWIO kernel in the flat buffers, elements
    void* w_fwd;         // [taps][cout][cin_pad], nullable
    void* w_dg;          // [taps][cin][cout_pad], nullable
    int taps, cin, cout, cin_pad, cout_pad;
    int tile0;           // first tile (workgroup) of the layer
};
struct OptArgs {
    float *p, *g, *m, *v;
    long long n, flat_from, zero_from;
    int n_layers, n_tile_blocks, n_flat_blocks;
    float wd, reg_scale, lr_t, b1, b2, eps, clip, grad_scale;
    int t;
    const float* partials;
    const float* extra_sq;
    float* state;
    const int32_t* invalid;
    float* gnorm_sq_out;
    float* reg_out;
    OptLayer layer[CUTIE_MAX_BATCH];
};

// sum of 1024 partials by 256 lanes in a fixed order; every lane returns it
__device__ __forceinline__ float fold_partials(const float* __restrict__ part, float* sh) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < RED_BLOCKS / OPT_T; ++k) s += part[k * OPT_T + threadIdx.x];
    s = wave_sum(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    return ((sh[0] + sh[1]) + sh[2]) + sh[3];
}

template <typename T>
__global__ __launch_bounds__(OPT_T) void opt_step_kernel(OptArgs a) {
    __shared__ float tile[64][65];
    __shared__ float sh[4];
    __shared__ float lr_sh;
    float gn2 = fold_partials(a.partials, sh);
    if (a.extra_sq) gn2 += *a.extra_sq;
    const bool bad = (a.invalid && *a.invalid != 0) || !(fabsf(gn2) <= 3.0e38f);
    if (blockIdx.x == 0) {
        const float sw = fold_partials(a.partials + RED_BLOCKS, sh);
        if (threadIdx.x == 0) {
            if (a.gnorm_sq_out) *a.gnorm_sq_out = gn2;
            if (a.reg_out) *a.reg_out = a.reg_scale * sw;
        }
    }
    if (bad) {
        // an overflowed or invalid backward: p, m, v and the packed operands stay as they are; the accumulated part of
        // the gradient buffer is cleared all the same (the next backward adds into it)
        if ((int)blockIdx.x >= a.n_tile_blocks) {
            const long long stride = (long long)a.n_flat_blocks * OPT_T;
            long long i = a.flat_from + (long long)(blockIdx.x - a.n_tile_blocks) * OPT_T + threadIdx.x;
            if (i < a.zero_from) i += (a.zero_from - i + stride - 1) / stride * stride;
            for (; i < a.n; i += stride) a.g[i] = 0.f;
        }
        return;
    }
    if (threadIdx.x == 0) lr_sh = (a.state && a.t > 0) ? rebased_lr(a.lr_t, a.b1, a.b2, a.t, a.state[3]) : a.lr_t;
    __syncthreads();
    const float lr_t = lr_sh;
    const float inv = a.state ? a.state[1] : 1.f;
    // tf.clip_by_global_norm: t * clip_norm / max(global_norm, clip_norm), on the mean gradient (grad_scale = 1 / world)
    const float scale = a.grad_scale * (a.clip / fmaxf(sqrtf(gn2) * a.grad_scale, a.clip));
    const float b1 = a.b1, b2 = a.b2, eps = a.eps;
    auto update = [&](long long i, float wdi) -> float {
        const float pi = a.p[i];
        const float gi = (a.g[i] * inv + wdi * pi) * scale;
        const float mi = b1 * a.m[i] + (1.f - b1) * gi;
        const float vi = b2 * a.v[i] + (1.f - b2) * gi * gi;
        a.m[i] = mi;
        a.v[i] = vi;
        const float pn = pi - lr_t * mi / (sqrtf(vi) + eps);
        a.p[i] = pn;
        return pn;
    };
    if ((int)blockIdx.x >= a.n_tile_blocks) {
        // biases, embedding (and whatever else lies behind the conv kernels): plain elements, no L2 term
        const long long stride = (long long)a.n_flat_blocks * OPT_T;
        for (long long i = a.flat_from + (long long)(blockIdx.x - a.n_tile_blocks) * OPT_T + threadIdx.x; i < a.n; i += stride) {
            update(i, 0.f);
            if (i >= a.zero_from) a.g[i] = 0.f;      // the scatter-added embedding gradient starts the next step at zero
        }
        return;
    }
    int li = 0;
    while (li + 1 < a.n_layers && (int)blockIdx.x >= a.layer[li + 1].tile0) ++li;
    const OptLayer L = a.layer[li];
    T* __restrict__ w_fwd = (T*)L.w_fwd;
    T* __restrict__ w_dg = (T*)L.w_dg;
    const int cin = L.cin, cout = L.cout;
    const int c_ext = w_fwd ? L.cin_pad : cin, n_ext = w_dg ? L.cout_pad : cout;
    const int CT = (c_ext + 63) / 64, NT = (n_ext + 63) / 64;
    const int tl = blockIdx.x - L.tile0;
    const int nt = tl % NT, ct = (tl / NT) % CT, t = tl / (NT * CT);
    const int c0 = ct * 64, n0 = nt * 64;
    const int lo = threadIdx.x & 63, hi = threadIdx.x >> 6;
#pragma unroll 4
    for (int k = 0; k < 16; ++k) {
        const int c = c0 + hi + 4 * k, n = n0 + lo;
        float pn = 0.f;
        if (c < cin && n < cout) pn = update(L.off + ((long long)t * cin + c) * cout + n, a.wd);
        tile[hi + 4 * k][lo] = pn;
    }
    __syncthreads();
    if (w_dg) {
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const int c = c0 + hi + 4 * k, n = n0 + lo;
            if (c < cin && n < L.cout_pad) store_from_f32(w_dg + ((long long)t * cin + c) * L.cout_pad + n, tile[hi + 4 * k][lo]);
        }
    }
    if (w_fwd) {
#pragma unroll 4
        for (int k = 0; k < 16; ++k) {
            const int n = n0 + hi + 4 * k, c = c0 + lo;
            if (n < cout && c < L.cin_pad) store_from_f32(w_fwd + ((long long)t * cout + n) * L.cin_pad + c, tile[lo][hi + 4 * k]);
        }
    }
}
}  // namespace

extern "C" int cutie_opt_norm(const float* w, const float* g, int64_t n_l2, int64_t n_dense, float wd, const void* rows,
                              int64_t n_rows, int rows_dtype, const float* loss_scale, float* partials,
                              cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(w && g && partials && n_l2 >= 0 && n_dense >= n_l2, "opt_norm: bad arguments");
    CUTIE_CHECK_ARG(n_l2 % 4 == 0 && n_dense % 4 == 0 && n_rows % 4 == 0 && n_rows >= 0, "opt_norm: ranges must be whole float4s");
    CUTIE_CHECK_ARG(cutie::aligned16(w) && cutie::aligned16(g) && cutie::aligned16(rows), "opt_norm: 16-byte alignment");
    if (!rows) { n_rows = 0; rows_dtype = CUTIE_F32; }
    CUTIE_DISPATCH_DTYPE(rows_dtype, DT,
        hipLaunchKernelGGL(opt_norm_kernel<typename elem<DT>::type>, dim3(RED_BLOCKS), dim3(OPT_T), 0, stream, w, g, n_l2,
                           n_dense, wd, (const typename elem<DT>::type*)rows, n_rows, loss_scale, partials));
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

extern "C" int cutie_opt_step(float* p, float* g, float* m, float* v, int64_t n, int64_t n_l2, float wd, float reg_scale,
                              const cutie_pack_desc* layers, int n_layers, int dtype, float lr_t, float beta1, float beta2,
                              float eps, float clip, float grad_scale, int t, const float* partials, const float* extra_sq,
                              float* loss_scale, const int32_t* invalid, float* gnorm_sq_out, float* reg_out,
                              int64_t zero_from, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(p && g && m && v && partials && n >= 0 && n_l2 >= 0 && n_l2 <= n, "opt_step: bad arguments");
    CUTIE_CHECK_ARG(n_layers >= 0 && n_layers <= CUTIE_MAX_BATCH && (n_layers == 0 || layers), "opt_step: n_layers=%d (0..%d)",
                    n_layers, CUTIE_MAX_BATCH);
    if (n == 0) return CUTIE_OK;
    OptArgs a;
    a.p = p; a.g = g; a.m = m; a.v = v; a.n = n; a.zero_from = zero_from < 0 ? n : zero_from;
    a.wd = wd; a.reg_scale = reg_scale; a.lr_t = lr_t; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.clip = clip;
    a.grad_scale = grad_scale; a.t = t; a.partials = partials; a.extra_sq = extra_sq; a.state = loss_scale;
    a.invalid = invalid; a.gnorm_sq_out = gnorm_sq_out; a.reg_out = reg_out; a.n_layers = n_layers;
    // the layers must tile [0, n_l2) of the flat buffers front to back (padding between them holds zeros and stays zero)
    long long tiles = 0, end = 0;
    for (int i = 0; i < n_layers; ++i) {
        const cutie_pack_desc& d = layers[i];
        CUTIE_CHECK_ARG(d.taps >= 1 && d.cin >= 1 && d.cout >= 1 && d.w_hwio, "opt_step: bad layer %d", i);
        CUTIE_CHECK_ARG(!d.w_fwd || d.cin_pad >= d.cin, "opt_step: layer %d cin_pad < cin", i);
        CUTIE_CHECK_ARG(!d.w_dgrad || d.cout_pad >= d.cout, "opt_step: layer %d cout_pad < cout", i);
        const long long off = d.w_hwio - p;
        const long long cnt = (long long)d.taps * d.cin * d.cout;
        CUTIE_CHECK_ARG(off >= end && off + cnt <= n_l2, "opt_step: layer %d does not lie in the L2 range behind layer %d", i, i - 1);
        end = off + cnt;
        OptLayer& L = a.layer[i];
        L.off = off; L.w_fwd = d.w_fwd; L.w_dg = d.w_dgrad; L.taps = d.taps; L.cin = d.cin; L.cout = d.cout;
        L.cin_pad = d.cin_pad; L.cout_pad = d.cout_pad; L.tile0 = (int)tiles;
        tiles += (long long)d.taps * cutie::cdiv(d.w_fwd ? d.cin_pad : d.cin, 64) * cutie::cdiv(d.w_dgrad ? d.cout_pad : d.cout, 64);
    }
    CUTIE_CHECK_ARG(tiles < (1ll << 30), "opt_step: too many tiles");
    // what the layers do not cover of [0, n_l2) is padding; without layers the L2 range is walked as plain elements too
    a.flat_from = n_layers ? n_l2 : 0;
    CUTIE_CHECK_ARG(n_layers || n_l2 == 0, "opt_step: an L2 range needs its layer table");
    CUTIE_CHECK_ARG(a.zero_from >= a.flat_from, "opt_step: zero_from lies inside the conv kernels");
    a.n_tile_blocks = (int)tiles;
    long long fb = cutie::cdiv(n - a.flat_from, (long long)OPT_T * 4);
    if (fb > 1024) fb = 1024;
    if (fb < 1) fb = 1;
    a.n_flat_blocks = (int)fb;
    CUTIE_DISPATCH_DTYPE(dtype, DT,
        hipLaunchKernelGGL(opt_step_kernel<typename elem<DT>::type>, dim3((unsigned)(tiles + fb)), dim3(OPT_T), 0, stream, a));
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

extern "C" int cutie_pack_weights_batched(const cutie_pack_desc* descs, int n, int dtype, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(n >= 0 && n <= CUTIE_MAX_BATCH && (n == 0 || descs), "pack_weights_batched: n=%d (0..%d)", n, CUTIE_MAX_BATCH);
    if (n == 0) return CUTIE_OK;
    PackBatch pb;
    int64_t most = 0;
    for (int i = 0; i < n; ++i) {
        const cutie_pack_desc& d = descs[i];
        CUTIE_CHECK_ARG(d.taps >= 1 && d.cin >= 1 && d.cout >= 1 && d.w_hwio, "pack_weights_batched: bad layer %d", i);
        CUTIE_CHECK_ARG(!d.w_fwd || d.cin_pad >= d.cin, "pack_weights_batched: layer %d cin_pad < cin", i);
        CUTIE_CHECK_ARG(!d.w_dgrad || d.cout_pad >= d.cout, "pack_weights_batched: layer %d cout_pad < cout", i);
        pb.d[i] = d;
        if (!d.w_fwd && !d.w_dgrad) continue;
        const int64_t tiles = (int64_t)d.taps * cutie::cdiv(d.w_fwd ? d.cin_pad : d.cin, 64) * cutie::cdiv(d.w_dgrad ? d.cout_pad : d.cout, 64);
        if (tiles > most) most = tiles;
    }
    if (most == 0) return CUTIE_OK;
    int64_t gx = most;
    if (gx > 1024) gx = 1024;
    CUTIE_DISPATCH_DTYPE(dtype, DT,
        hipLaunchKernelGGL(pack_batched_kernel<typename elem<DT>::type>, dim3((int)gx, n), dim3(256), 0, stream, pb));
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

extern "C" int cutie_pack_weights(const float* w_hwio, int taps, int cin, int cout, void* w_fwd, int cin_pad,
                                  void* w_dgrad, int cout_pad, int dtype, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(taps >= 1 && cin >= 1 && cout >= 1 && w_hwio, "pack_weights: bad arguments");
    CUTIE_CHECK_ARG(!w_fwd || cin_pad >= cin, "pack_weights: cin_pad < cin");
    CUTIE_CHECK_ARG(!w_dgrad || cout_pad >= cout, "pack_weights: cout_pad < cout");
    const int64_t total = (w_fwd ? (int64_t)taps * cout * cin_pad : 0) + (w_dgrad ? (int64_t)taps * cin * cout_pad : 0);
    if (total == 0) return CUTIE_OK;
    int64_t grid = cutie::cdiv(total, 256);
    if (grid > 4096) grid = 4096;
    CUTIE_DISPATCH_DTYPE(dtype, DT,
        hipLaunchKernelGGL(pack_kernel<typename elem<DT>::type>, dim3((int)grid), dim3(256), 0, stream, w_hwio, taps,
                           cin, cout, (typename elem<DT>::type*)w_fwd, cin_pad, (typename elem<DT>::type*)w_dgrad,
                           cout_pad));
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}
