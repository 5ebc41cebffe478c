// K6 / K8: instance norm (center=False, scale=False, eps 1e-6, biased variance) forward and
// backward, and the global-pool broadcast.  tf.contrib.layers.instance_norm at
// /root/reference/model_framework.py:423; Model.global_pool at :309-315.
//
// All are HBM-bound passes over [B][HW][C] views.  One lane owns VEC consecutive channels;
// a workgroup covers `rows_par` pixels side by side and walks a slice of HW, so every
// per-(b,c) reduction is: registers -> LDS across the side-by-side rows -> one f32 atomic.
#include "common.h"

namespace {
using namespace cutie;

template <int VEC> struct vecf { float v[VEC]; };

template <bool NT = false>
__device__ __forceinline__ void load8(const float* p, float* v) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
typedef unsigned nt_u32x4 __attribute__((ext_vector_type(4)));
// NT: non-temporal access for the streaming passes (each byte is touched once per pass; measured on the
// MI355X: in_apply 30.2 -> 27.5 us, in_bwd_apply 47.0 -> 43.2 us on cache-cold tensors, +1.1 % docs/s)
template <bool NT = false>
__device__ __forceinline__ void load8(const bf16_t* p, float* v) {
    nt_u32x4 u;
    if constexpr (NT) u = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4*>(p));
    else u = *reinterpret_cast<const nt_u32x4*>(p);
    v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
    v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
    v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
    v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
}
template <bool NT = false>
__device__ __forceinline__ void load8(const f16_t* p, float* v) {
    typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
    half8_t h;
    if constexpr (NT) h = __builtin_nontemporal_load(reinterpret_cast<const half8_t*>(p));
    else h = *reinterpret_cast<const half8_t*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = static_cast<float>(h[i]);
}
template <bool NT = false>
__device__ __forceinline__ void store8(f16_t* p, const float* v) {
    typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
    half8_t h;
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = static_cast<f16_t>(v[i]);
    if constexpr (NT) __builtin_nontemporal_store(h, reinterpret_cast<half8_t*>(p));
    else *reinterpret_cast<half8_t*>(p) = h;
}
template <bool NT = false>
__device__ __forceinline__ void store8(float* p, const float* v) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}
template <bool NT = false>
__device__ __forceinline__ void store8(bf16_t* p, const float* v) {
    uint4 u;
    u.x = static_cast<unsigned>(f2bf(v[0])) | (static_cast<unsigned>(f2bf(v[1])) << 16);
    u.y = static_cast<unsigned>(f2bf(v[2])) | (static_cast<unsigned>(f2bf(v[3])) << 16);
    u.z = static_cast<unsigned>(f2bf(v[4])) | (static_cast<unsigned>(f2bf(v[5])) << 16);
    u.w = static_cast<unsigned>(f2bf(v[6])) | (static_cast<unsigned>(f2bf(v[7])) << 16);
    if constexpr (NT) __builtin_nontemporal_store(nt_u32x4{u.x, u.y, u.z, u.w}, reinterpret_cast<nt_u32x4*>(p));
    else *reinterpret_cast<uint4*>(p) = u;
}

template <int VEC, bool NT = false, typename T> __device__ __forceinline__ vecf<VEC> loadv(const T* p) {
    vecf<VEC> r;
    if constexpr (VEC == 8) {
        load8<NT>(p, r.v);
    } else if constexpr (VEC == 4) {
        const float4 f = load4(p);
        r.v[0] = f.x; r.v[1] = f.y; r.v[2] = f.z; r.v[3] = f.w;
    } else {
        r.v[0] = load_as_f32(p);
    }
    return r;
}
template <int VEC, bool NT = false, typename T> __device__ __forceinline__ void storev(T* p, const vecf<VEC>& r) {
    if constexpr (VEC == 8) {
        store8<NT>(p, r.v);
    } else if constexpr (VEC == 4) {
        store4(p, make_float4(r.v[0], r.v[1], r.v[2], r.v[3]));
    } else {
        store_from_f32(p, r.v[0]);
    }
}

struct Map {      // workgroup -> (b, hw slice); thread -> (channel lane, side-by-side row)
    int b, hw0, hw1, cl, sub, rows_par;
    bool active;
};
__device__ __forceinline__ Map make_map(int HW, int CV) {
    Map m;
    m.b = blockIdx.y;
    const int per = (HW + gridDim.x - 1) / gridDim.x;
    m.hw0 = blockIdx.x * per;
    m.hw1 = min(HW, m.hw0 + per);
    m.rows_par = 256 / CV;
    m.cl = threadIdx.x % CV;
    m.sub = threadIdx.x / CV;
    m.active = m.sub < m.rows_par;
    return m;
}

// Partial-sum layout shared by stats, red and dbias_part: [B][T][C](x2) - every producer workgroup (a
// conv tile, a slice of H*W) owns slot t and STORES its sum there; consumers add the T slots in slot
// order.  No float atomics anywhere on this path: reruns are bit-identical.
__device__ __forceinline__ void sum_tiles(const float* p, int b, int T, int C, int c, float& s0, float& s1) {
    s0 = 0.f;
    s1 = 0.f;
#pragma unroll 4
    for (int t = 0; t < T; ++t) {
        const float2 v = *reinterpret_cast<const float2*>(p + (((int64_t)b * T + t) * C + c) * 2);
        s0 += v.x;
        s1 += v.y;
    }
}

// Workgroup-wide fold of the T slots of sample b into LDS tables (one channel per lane, coalesced float2 reads):
// every lane of a kernel then picks its VEC channels out of LDS instead of walking T slots per channel itself.
// MODE 0: t0 = sum0, t1 = sum1;  MODE 1: t0 = mean, t1 = 1/sigma (the arithmetic of mean_inv).
// The caller synchronises before reading the tables.
template <int MODE>
__device__ __forceinline__ void fold_slots(const float* p, int b, int T, int C, float inv_n, float eps, float* t0, float* t1,
                                           const float* bias = nullptr, int act = 0) {
    for (int c = threadIdx.x; c < C; c += 256) {
        float s, ss;
        sum_tiles(p, b, T, C, c, s, ss);
        if (MODE == 1) {
            norm_mean_inv(s, ss, inv_n, norm_shift(bias, act, c), eps, s, ss);
        }
        t0[c] = s;
        t1[c] = ss;
    }
}

// red[NRED][256][VEC] in LDS -> sum over the side-by-side rows (fixed order) -> *dst(c, k) = sum
template <int VEC, int NRED, typename F>
__device__ __forceinline__ void block_reduce_store(float (*red)[256][VEC], const Map& m, int CV, F&& dst) {
    __syncthreads();
    const int C = CV * VEC;
    for (int idx = threadIdx.x; idx < NRED * C; idx += 256) {      // all lanes share the tail
        const int k = idx / C, c = idx - k * C;
        const int cl = c / VEC, e = c - cl * VEC;
        float s = 0.f;
        for (int r = 0; r < m.rows_par; ++r) s += red[k][r * CV + cl][e];
        *dst(c, k) = s;
    }
}

template <int VEC, typename TA>
__global__ __launch_bounds__(256) void in_stats_kernel(const TA* __restrict__ a, int ld_a, int HW, int C,
                                                       float* __restrict__ stats, const float* __restrict__ bias, int act) {
    __shared__ float red[2][256][VEC];
    const int CV = C / VEC;
    const Map m = make_map(HW, CV);
    float s[VEC] = {}, ss[VEC] = {};
    if (m.active) {
        float sh[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) sh[e] = norm_shift(bias, act, m.cl * VEC + e);
        for (int hw = m.hw0 + m.sub; hw < m.hw1; hw += m.rows_par) {
            const vecf<VEC> x = loadv<VEC>(a + ((int64_t)m.b * HW + hw) * ld_a + m.cl * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { const float d = x.v[e] - sh[e]; s[e] += d; ss[e] += d * d; }
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) { red[0][threadIdx.x][e] = s[e]; red[1][threadIdx.x][e] = ss[e]; }
    const int64_t slot = (int64_t)m.b * gridDim.x + blockIdx.x;
    block_reduce_store<VEC, 2>(red, m, CV, [&](int c, int k) { return stats + (slot * C + c) * 2 + k; });
}

template <int VEC, typename TA, typename TY>
__global__ __launch_bounds__(256) void in_apply_kernel(const TA* __restrict__ a, int ld_a,
                                                       const float* __restrict__ stats, int st, float eps,
                                                       TY* __restrict__ y, int ld_y, int HW, int C,
                                                       const float* __restrict__ bias, int act) {
    extern __shared__ float tab[];      // [2][C]
    const int CV = C / VEC;
    const Map m = make_map(HW, CV);
    const float inv_n = 1.f / (float)HW;
    fold_slots<1>(stats, m.b, st, C, inv_n, eps, tab, tab + C, bias, act);
    __syncthreads();
    if (!m.active) return;
    float mean[VEC], inv[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { mean[e] = tab[m.cl * VEC + e]; inv[e] = tab[C + m.cl * VEC + e]; }
#pragma unroll 4
    for (int hw = m.hw0 + m.sub; hw < m.hw1; hw += m.rows_par) {
        const int64_t row = (int64_t)m.b * HW + hw;
        vecf<VEC> x = loadv<VEC, true>(a + row * ld_a + m.cl * VEC);
#pragma unroll
        for (int e = 0; e < VEC; ++e) x.v[e] = (x.v[e] - mean[e]) * inv[e];
        storev<VEC, true>(y + row * ld_y + m.cl * VEC, x);
    }
}

template <int VEC, typename TD, typename TA>
__global__ __launch_bounds__(256) void in_bwd_stats_kernel(const TD* __restrict__ dy, int ld_dy,
                                                           const TA* __restrict__ a, int ld_a,
                                                           const float* __restrict__ stats, int st, float eps,
                                                           float* __restrict__ redout, int HW, int C,
                                                           const float* __restrict__ bias, int act) {
    __shared__ float red[2][256][VEC];
    extern __shared__ float tab[];      // [2][C]
    const int CV = C / VEC;
    const Map m = make_map(HW, CV);
    fold_slots<1>(stats, m.b, st, C, 1.f / (float)HW, eps, tab, tab + C, bias, act & 15);
    const bool from_y = act & CUTIE_ACT_FROM_Y;      // `a` is the normalised output: xhat = y
    __syncthreads();
    float s1[VEC] = {}, s2[VEC] = {};
    if (m.active) {
        float mean[VEC], inv[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) { mean[e] = tab[m.cl * VEC + e]; inv[e] = tab[C + m.cl * VEC + e]; }
#pragma unroll 4
        for (int hw = m.hw0 + m.sub; hw < m.hw1; hw += m.rows_par) {
            const int64_t row = (int64_t)m.b * HW + hw;
            const vecf<VEC> g = loadv<VEC, true>(dy + row * ld_dy + m.cl * VEC);
            const vecf<VEC> x = loadv<VEC, true>(a + row * ld_a + m.cl * VEC);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                s1[e] += g.v[e];
                s2[e] = __builtin_fmaf(g.v[e], from_y ? x.v[e] : (x.v[e] - mean[e]) * inv[e], s2[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) { red[0][threadIdx.x][e] = s1[e]; red[1][threadIdx.x][e] = s2[e]; }
    const int64_t slot = (int64_t)m.b * gridDim.x + blockIdx.x;
    block_reduce_store<VEC, 2>(red, m, CV, [&](int c, int k) { return redout + (slot * C + c) * 2 + k; });
}

template <int VEC, typename TD, typename TA, typename TZ>
__global__ __launch_bounds__(256) void in_bwd_apply_kernel(const TD* __restrict__ dy, int ld_dy,
                                                           const TA* __restrict__ a, int ld_a,
                                                           const float* __restrict__ stats, int st, float eps,
                                                           const float* __restrict__ redin, int rt, int relu,
                                                           TZ* __restrict__ dz, int ld_dz,
                                                           float* __restrict__ dbias, int HW, int C,
                                                           const float* __restrict__ bias) {
    __shared__ float red[1][256][VEC];
    extern __shared__ float tab[];      // [4][C]
    const int CV = C / VEC;
    const Map m = make_map(HW, CV);
    const float inv_n = 1.f / (float)HW;
    const bool from_y = relu & CUTIE_ACT_FROM_Y;      // `a` is the normalised output with the ReLU pattern inside (common.h)
    relu &= 15;
    fold_slots<1>(stats, m.b, st, C, inv_n, eps, tab, tab + C, bias, relu);
    fold_slots<0>(redin, m.b, rt, C, inv_n, eps, tab + 2 * C, tab + 3 * C);
    __syncthreads();
    float sb[VEC] = {};
    if (m.active) {
        float mean[VEC], inv[VEC], m1[VEC], m2[VEC];
        unsigned short y0b[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const int c = m.cl * VEC + e;
            mean[e] = tab[c];
            inv[e] = tab[C + c];
            m1[e] = tab[2 * C + c] * inv_n;
            m2[e] = tab[3 * C + c] * inv_n;
            y0b[e] = y0_bits<TA>(mean[e], inv[e]);
        }
#pragma unroll 4
        for (int hw = m.hw0 + m.sub; hw < m.hw1; hw += m.rows_par) {
            const int64_t row = (int64_t)m.b * HW + hw;
            const vecf<VEC> g = loadv<VEC, true>(dy + row * ld_dy + m.cl * VEC);
            const vecf<VEC> x = loadv<VEC, true>(a + row * ld_a + m.cl * VEC);
            vecf<VEC> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xh = from_y ? x.v[e] : (x.v[e] - mean[e]) * inv[e];
                float d = inv[e] * __builtin_fmaf(-xh, m2[e], g.v[e] - m1[e]);      // conv_wide's meeting tiles: the same operations
                if (relu == 1) {
                    bool active = x.v[e] > 0.f;
                    if (from_y) {
                        if constexpr (sizeof(TA) == 2) {      // x came out of a 16-bit value: rounding it back is exact
                            TA t;
                            store_from_f32(&t, x.v[e]);
                            active = bits16(t) != y0b[e];
                        }
                    }
                    if (!active) d = 0.f;
                } else if (relu == 2) {
                    d *= x.v[e] * (1.f - x.v[e]);      // sigmoid': a is the stored sigmoid output
                }
                o.v[e] = d;
                sb[e] += d;
            }
            storev<VEC>(dz + row * ld_dz + m.cl * VEC, o);     // plain store: dz is read again at once by dgrad and wgrad
        }
    }
    if (dbias) {   // one row of partials per (sample, H*W slice): [B * gridDim.x][C], folded by sum_rows
#pragma unroll
        for (int e = 0; e < VEC; ++e) red[0][threadIdx.x][e] = sb[e];
        const int64_t slot = (int64_t)m.b * gridDim.x + blockIdx.x;
        block_reduce_store<VEC, 1>(red, m, CV, [&](int c, int) { return dbias + slot * C + c; });
    }
}


// one workgroup per (b, chunk of CV_BLK channel lanes): mean over HW, then broadcast
template <int VEC, typename TX, typename TO>
__global__ __launch_bounds__(256) void mean_bcast_kernel(const TX* __restrict__ x, int ld_x, TO* __restrict__ out,
                                                         int ld_out, int HW, int C, int cv_blk) {
    __shared__ float red[256][VEC];
    __shared__ float mean_s[256][VEC];
    const int b = blockIdx.y;
    const int rows_par = 256 / cv_blk;
    const int cl = threadIdx.x % cv_blk, sub = threadIdx.x / cv_blk;
    const int c0 = (blockIdx.x * cv_blk + cl) * VEC;
    const bool active = sub < rows_par && c0 < C;
    float s[VEC] = {};
    if (active) {
        for (int hw = sub; hw < HW; hw += rows_par) {
            const vecf<VEC> v = loadv<VEC>(x + ((int64_t)b * HW + hw) * ld_x + c0);
#pragma unroll
            for (int e = 0; e < VEC; ++e) s[e] += v.v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) red[threadIdx.x][e] = s[e];
    __syncthreads();
    if (threadIdx.x < cv_blk) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float t = 0.f;
            for (int r = 0; r < rows_par; ++r) t += red[r * cv_blk + threadIdx.x][e];
            mean_s[threadIdx.x][e] = t / (float)HW;
        }
    }
    __syncthreads();
    if (active) {
        vecf<VEC> mv;
#pragma unroll
        for (int e = 0; e < VEC; ++e) mv.v[e] = mean_s[cl][e];
        for (int hw = sub; hw < HW; hw += rows_par) storev<VEC>(out + ((int64_t)b * HW + hw) * ld_out + c0, mv);
    }
}

// second half of the wide-grid global pool: out[b, hw, c] = sums[b][c][0] / HW
template <int VEC, typename TO>
__global__ __launch_bounds__(256) void bcast_mean_kernel(const float* __restrict__ sums, int st, TO* __restrict__ out,
                                                         int ld_out, int HW, int C, int rows_per_wg) {
    extern __shared__ float tab[];      // [2][C]
    const int CV = C / VEC;
    const int b = blockIdx.y;
    const int rows_par = 256 / CV;
    const int cl = threadIdx.x % CV, sub = threadIdx.x / CV;
    fold_slots<0>(sums, b, st, C, 0.f, 0.f, tab, tab + C);
    __syncthreads();
    if (sub >= rows_par) return;
    vecf<VEC> mv;
    const float inv = 1.f / (float)HW;
#pragma unroll
    for (int e = 0; e < VEC; ++e) mv.v[e] = tab[cl * VEC + e] * inv;
    const int r0 = blockIdx.x * rows_per_wg;
    const int r1 = min(HW, r0 + rows_per_wg);
    for (int hw = r0 + sub; hw < r1; hw += rows_par) storev<VEC>(out + ((int64_t)b * HW + hw) * ld_out + cl * VEC, mv);
}

struct Launch { dim3 grid; int vec; };
inline int pick_vec(int C, std::initializer_list<int> lds, std::initializer_list<const void*> ptrs) {
    if (C % 4) return 1;
    for (int l : lds) if (l % 4) return 1;
    for (const void* p : ptrs) if (p && ((uintptr_t)p % 16)) return 1;
    if (C % 8 || C < 64) return 4;
    for (int l : lds) if (l % 8) return 4;
    return 8;      // 16-byte (bf16) / 32-byte (f32) accesses per lane
}
inline int check_common(const char* name, int B, int HW, int C, int vec) {
    if (B < 0 || HW < 1 || C < 1) { cutie::set_error("%s: bad sizes B=%d HW=%d C=%d", name, B, HW, C); return CUTIE_E_ARG; }
    if (C / vec > 256) { cutie::set_error("%s: C=%d too wide for this layout (max %d)", name, C, 256 * vec); return CUTIE_E_ARG; }
    return CUTIE_OK;
}
inline dim3 reduce_grid(int B, int HW, int C, int vec) {
    const int rows_par = 256 / (C / vec);
    // 8 row groups a workgroup and at most ~1024 workgroups: measured best on the MI355X (tools/norm_bench.py) -
    // more, shorter workgroups pay the slot fold more often (2x as many: the same; 4x: 1.2x slower; 8x: 1.7x), half as
    // many lose bandwidth (1.2x slower)
    int s = HW / (rows_par * 8);
    const int cap = (1024 + B - 1) / (B > 0 ? B : 1);
    if (s > cap) s = cap;
    if (s < 1) s = 1;
    return dim3(s, B);
}

inline int hint_tiles(int B, int HW, int C) {
    const int vec = (C % 8 == 0 && C >= 64) ? 8 : (C % 4 == 0 ? 4 : 1);
    if (C / vec > 256) return 1;
    return (int)reduce_grid(B, HW, C, vec).x;
}

}  // namespace

#define NORM_DISPATCH_VEC(vec, ...)                                   \
    if ((vec) == 8) { constexpr int VEC = 8; __VA_ARGS__; }           \
    else if ((vec) == 4) { constexpr int VEC = 4; __VA_ARGS__; }      \
    else { constexpr int VEC = 1; __VA_ARGS__; }

extern "C" int cutie_in_tiles(int B, int HW, int C) {
    if (B < 1 || HW < 1 || C < 1) return 1;
    return hint_tiles(B, HW, C);
}

extern "C" int cutie_in_stats(const void* a, int ld_a, int a_dtype, int B, int HW, int C, float* stats, int stats_tiles,
                              const float* shift_bias, int shift_act, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int vec = pick_vec(C, {ld_a}, {a});
    if (int e = check_common("in_stats", B, HW, C, vec)) return e;
    if (B == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(a && stats && ld_a >= C, "in_stats: null pointer or ld < C");
    CUTIE_CHECK_ARG(stats_tiles >= 1 && stats_tiles <= 65535, "in_stats: stats_tiles=%d", stats_tiles);
    const dim3 g(stats_tiles, B);
    CUTIE_DISPATCH_DTYPE(a_dtype, ADT, NORM_DISPATCH_VEC(vec,
        hipLaunchKernelGGL((in_stats_kernel<VEC, typename elem<ADT>::type>), g, dim3(256), 0, stream,
                           (const typename elem<ADT>::type*)a, ld_a, HW, C, stats, shift_bias, shift_act)));
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

extern "C" int cutie_in_apply(const void* a, int ld_a, int a_dtype, const float* stats, int stats_tiles, float eps,
                              void* y, int ld_y, int y_dtype, int B, int HW, int C, const float* shift_bias,
                              int shift_act, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int vec = pick_vec(C, {ld_a, ld_y}, {a, y});
    if (int e = check_common("in_apply", B, HW, C, vec)) return e;
    if (B == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(a && stats && y && ld_a >= C && ld_y >= C, "in_apply: null pointer or ld < C");
    CUTIE_CHECK_ARG(stats_tiles >= 1, "in_apply: stats_tiles=%d", stats_tiles);
    const dim3 g = reduce_grid(B, HW, C, vec);
    CUTIE_DISPATCH_DTYPE(a_dtype, ADT, CUTIE_DISPATCH_DTYPE(y_dtype, YDT, NORM_DISPATCH_VEC(vec,
        hipLaunchKernelGGL((in_apply_kernel<VEC, typename elem<ADT>::type, typename elem<YDT>::type>), g, dim3(256),
                           2 * C * sizeof(float), stream, (const typename elem<ADT>::type*)a, ld_a, stats, stats_tiles, eps,
                           (typename elem<YDT>::type*)y, ld_y, HW, C, shift_bias, shift_act))));
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

extern "C" int cutie_in_bwd_stats(const void* dy, int ld_dy, int dy_dtype, const void* a, int ld_a, int a_dtype,
                                  const float* stats, int stats_tiles, float eps, float* red, int red_tiles, int B,
                                  int HW, int C, const float* shift_bias, int shift_act, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int vec = pick_vec(C, {ld_dy, ld_a}, {dy, a});
    if (int e = check_common("in_bwd_stats", B, HW, C, vec)) return e;
    if (B == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(dy && a && stats && red && ld_dy >= C && ld_a >= C, "in_bwd_stats: null pointer or ld < C");
    CUTIE_CHECK_ARG(stats_tiles >= 1 && red_tiles >= 1 && red_tiles <= 65535, "in_bwd_stats: tiles %d / %d", stats_tiles, red_tiles);
    const dim3 g(red_tiles, B);
    CUTIE_DISPATCH_DTYPE(dy_dtype, DDT, CUTIE_DISPATCH_DTYPE(a_dtype, ADT, NORM_DISPATCH_VEC(vec,
        hipLaunchKernelGGL((in_bwd_stats_kernel<VEC, typename elem<DDT>::type, typename elem<ADT>::type>), g,
                           dim3(256), 2 * C * sizeof(float), stream, (const typename elem<DDT>::type*)dy, ld_dy,
                           (const typename elem<ADT>::type*)a, ld_a, stats, stats_tiles, eps, red, HW, C, shift_bias,
                           shift_act))));
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

extern "C" int cutie_in_bwd_apply(const void* dy, int ld_dy, int dy_dtype, const void* a, int ld_a, int a_dtype,
                                  const float* stats, int stats_tiles, float eps, const float* red, int red_tiles,
                                  int relu, void* dz, int ld_dz, int dz_dtype, float* dbias, int dbias_tiles, int B,
                                  int HW, int C, const float* shift_bias, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int vec = pick_vec(C, {ld_dy, ld_a, ld_dz}, {dy, a, dz});
    if (int e = check_common("in_bwd_apply", B, HW, C, vec)) return e;
    if (B == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(dy && a && stats && red && dz && ld_dy >= C && ld_a >= C && ld_dz >= C,
                    "in_bwd_apply: null pointer or ld < C");
    CUTIE_CHECK_ARG(a_dtype == dz_dtype, "in_bwd_apply: dz dtype must equal a dtype");
    CUTIE_CHECK_ARG(!(relu & CUTIE_ACT_FROM_Y) || (a_dtype != CUTIE_F32 && (relu & 15) != 2),
                    "in_bwd_apply: CUTIE_ACT_FROM_Y goes with a 16-bit y and no sigmoid");
    CUTIE_CHECK_ARG(stats_tiles >= 1 && red_tiles >= 1 && (!dbias || (dbias_tiles >= 1 && dbias_tiles <= 65535)),
                    "in_bwd_apply: tiles %d / %d / %d", stats_tiles, red_tiles, dbias_tiles);
    const dim3 g = dbias ? dim3(dbias_tiles, B) : reduce_grid(B, HW, C, vec);
    CUTIE_DISPATCH_DTYPE(dy_dtype, DDT, CUTIE_DISPATCH_DTYPE(a_dtype, ADT, NORM_DISPATCH_VEC(vec,
        hipLaunchKernelGGL((in_bwd_apply_kernel<VEC, typename elem<DDT>::type, typename elem<ADT>::type,
                                                typename elem<ADT>::type>), g, dim3(256), 4 * C * sizeof(float), stream,
                           (const typename elem<DDT>::type*)dy, ld_dy, (const typename elem<ADT>::type*)a, ld_a,
                           stats, stats_tiles, eps, red, red_tiles, relu, (typename elem<ADT>::type*)dz, ld_dz, dbias,
                           HW, C, shift_bias))));
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

namespace {
struct SumRowsBatch {
    cutie_sum_rows_desc d[CUTIE_MAX_BATCH];
};
// out[c] (+)= sum over rows of part[row][c]: 16 channels x 16 row lanes per workgroup, each lane adds its
// rows (row, row + 16, ...) in order, the 16 lane sums are added in lane order - a fixed order throughout
__global__ __launch_bounds__(256) void sum_rows_batched_kernel(SumRowsBatch sb, int B, int accumulate) {
    __shared__ float red[16][17];
    const cutie_sum_rows_desc d = sb.d[blockIdx.y];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    if (blockIdx.x * 16 >= d.C) return;
    float s = 0.f;
    if (c < d.C) {
#pragma unroll 4
        for (int b = rl; b < B; b += 16) s += d.part[(int64_t)b * d.C + c];
    }
    red[rl][cl] = s;
    __syncthreads();
    if (rl == 0 && c < d.C) {
        float t = accumulate ? d.out[c] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][cl];
        d.out[c] = t;
    }
}
}  // namespace

extern "C" int cutie_sum_rows_batched(const cutie_sum_rows_desc* descs, int n, int B, int accumulate,
                                      cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(n >= 0 && n <= CUTIE_MAX_BATCH && B >= 0 && (n == 0 || descs), "sum_rows_batched: n=%d B=%d", n, B);
    if (n == 0) return CUTIE_OK;
    SumRowsBatch sb;
    int cmax = 0;
    for (int i = 0; i < n; ++i) {
        CUTIE_CHECK_ARG(descs[i].part && descs[i].out && descs[i].C >= 1, "sum_rows_batched: bad entry %d", i);
        sb.d[i] = descs[i];
        if (descs[i].C > cmax) cmax = descs[i].C;
    }
    hipLaunchKernelGGL(sum_rows_batched_kernel, dim3((cmax + 15) / 16, n), dim3(256), 0, stream, sb, B, accumulate);
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

extern "C" int cutie_sum_rows(const float* part, int B, int C, float* out, int accumulate, cutie_stream_t stream_) {
    CUTIE_CHECK_ARG(B >= 0 && C >= 1 && part && out, "sum_rows: bad arguments");
    const cutie_sum_rows_desc d = {part, out, C};
    return cutie_sum_rows_batched(&d, 1, B, accumulate, stream_);      // one order of summation for both entry points
}

extern "C" int cutie_spatial_mean_bcast(const void* x, int ld_x, int x_dtype, void* out, int ld_out, int out_dtype,
                                        int B, int HW, int C, float* scratch, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    const int vec = pick_vec(C, {ld_x, ld_out}, {x, out});
    CUTIE_CHECK_ARG(B >= 0 && HW >= 1 && C >= 1, "spatial_mean_bcast: bad sizes");
    if (B == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(x && out && ld_x >= C && ld_out >= C, "spatial_mean_bcast: null pointer or ld < C");
    const int CV = C / vec;
    if (scratch && CV <= 256) {
        // wide grid: per-(b, c) partial sums by the instance-norm statistics pass (CUTIE_POOL_TILES
        // slices of H*W per sample, one slot each), then a store pass that adds the slots in order
        const int st = CUTIE_POOL_TILES;
        const int rc = cutie_in_stats(x, ld_x, x_dtype, B, HW, C, scratch, st, nullptr, 0, stream_);
        if (rc != CUTIE_OK) return rc;
        const int rows_par = 256 / CV;
        int rows_per_wg = rows_par * 8;
        const dim3 g((HW + rows_per_wg - 1) / rows_per_wg, B);
        CUTIE_DISPATCH_DTYPE(out_dtype, ODT, NORM_DISPATCH_VEC(vec,
            hipLaunchKernelGGL((bcast_mean_kernel<VEC, typename elem<ODT>::type>), g, dim3(256), 2 * C * sizeof(float), stream,
                               (const float*)scratch, st, (typename elem<ODT>::type*)out, ld_out, HW, C, rows_per_wg)));
        CUTIE_CHECK_LAUNCH();
        return CUTIE_OK;
    }
    const int cv_blk = CV < 16 ? CV : 16;
    const dim3 g((CV + cv_blk - 1) / cv_blk, B);
    CUTIE_DISPATCH_DTYPE(x_dtype, XDT, CUTIE_DISPATCH_DTYPE(out_dtype, ODT, NORM_DISPATCH_VEC(vec,
        hipLaunchKernelGGL((mean_bcast_kernel<VEC, typename elem<XDT>::type, typename elem<ODT>::type>), g, dim3(256),
                           0, stream, (const typename elem<XDT>::type*)x, ld_x, (typename elem<ODT>::type*)out,
                           ld_out, HW, C, cv_blk))));
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}
