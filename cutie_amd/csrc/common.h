// Shared helpers for the gfx950 kernels (error reporting, dtype load/store, wave reductions).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/cutie_hip.h"

namespace cutie {

void set_error(const char* fmt, ...);

// environment switches of the A/B scripts, read once at library load (core.hip)
struct Config {
    bool conv_wide;       // CUTIE_CONV_WIDE=0: every conv on the general kernel
    bool fuse_norm_bwd;   // CUTIE_FUSE_NORM_BWD=0: data gradient and norm-backward sums as two passes
    bool conv_lpt;        // CUTIE_CONV_LPT=0: conv_wide tiles in row-major order
    bool rendezvous;      // CUTIE_RENDEZVOUS=0: cutie_conv_dgrad_in_bwd always as data gradient + cutie_in_bwd_apply
    bool conv_ring;       // CUTIE_CONV_RING=1 in a -DCW_WITH_RING variant build: conv_wide's four-stage ring (measured slower)
    int conv_patch;       // CUTIE_CONV_PATCH=1 in a -DCW_WITH_PATCH variant build: conv_patch_kernel (A operand staged as a patch once
                          // per chunk) for multi-tap dilation-1 launches on the 64 / 128 tiles (measured, not shipped)
    int conv_duo;         // CUTIE_CONV_DUO: conv_duo_kernel (4-wave workgroups, two a CU, 128-channel tiles) - 0 never, 1 (default)
                          // where the channel tile is 128 anyway (n_out % 256 != 0), 2 for the 256-channel layers as well (measured slower)
};
const Config& config();

// grants `bytes` of dynamic LDS to a kernel on the current device, once (core.hip); returns CUTIE_OK / CUTIE_E_HIP
int ensure_dyn_lds(const void* fn, int bytes);
int device_cus();      // compute units of the current device, 0 when unknown

#define CUTIE_CHECK_ARG(cond, ...)                     \
    do {                                               \
        if (!(cond)) {                                 \
            cutie::set_error(__VA_ARGS__);             \
            return CUTIE_E_ARG;                        \
        }                                              \
    } while (0)

#define CUTIE_CHECK_HIP(expr)                                                          \
    do {                                                                               \
        hipError_t e__ = (expr);                                                       \
        if (e__ != hipSuccess) {                                                       \
            cutie::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__),   \
                             __FILE__, __LINE__);                                      \
            return CUTIE_E_HIP;                                                        \
        }                                                                              \
    } while (0)

#define CUTIE_CHECK_LAUNCH() CUTIE_CHECK_HIP(hipGetLastError())

#define CUTIE_ENSURE_LDS(kernel, bytes)                                                   \
    do {                                                                                  \
        if (int e__ = cutie::ensure_dyn_lds((const void*)(kernel), (int)(bytes))) return e__; \
    } while (0)

static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

typedef unsigned short bf16_t;  // raw bits

// round-to-nearest-even f32 -> bf16 (NaN stays NaN): lowers to v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ bf16_t f2bf(float f) {
    return __builtin_bit_cast(bf16_t, static_cast<__bf16>(f));
}
__device__ __forceinline__ float bf2f(bf16_t b) {
    return __uint_as_float(static_cast<unsigned>(b) << 16);
}

typedef _Float16 f16_t;         // IEEE half: 11 significant bits against bf16's 8, range 6e-8 .. 65504
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

template <int DT> struct elem;
template <> struct elem<CUTIE_F32> { typedef float type; };
template <> struct elem<CUTIE_BF16> { typedef bf16_t type; };
template <> struct elem<CUTIE_F16> { typedef f16_t type; };

__device__ __forceinline__ float load_as_f32(const float* p) { return *p; }
__device__ __forceinline__ float load_as_f32(const bf16_t* p) { return bf2f(*p); }
__device__ __forceinline__ float load_as_f32(const f16_t* p) { return static_cast<float>(*p); }
__device__ __forceinline__ void store_from_f32(float* p, float v) { *p = v; }
__device__ __forceinline__ void store_from_f32(bf16_t* p, float v) { *p = f2bf(v); }
__device__ __forceinline__ void store_from_f32(f16_t* p, float v) { *p = static_cast<f16_t>(v); }   // RNE, overflow -> inf

// 4 consecutive elements <-> float4 (16 B for f32, 8 B for bf16); pointers must be aligned
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const bf16_t* p) {
    uint2 u = *reinterpret_cast<const uint2*>(p);
    float4 r;
    r.x = __uint_as_float(u.x << 16);
    r.y = __uint_as_float(u.x & 0xffff0000u);
    r.z = __uint_as_float(u.y << 16);
    r.w = __uint_as_float(u.y & 0xffff0000u);
    return r;
}
__device__ __forceinline__ float4 load4(const f16_t* p) {
    const half4_t h = *reinterpret_cast<const half4_t*>(p);
    return make_float4((float)h.x, (float)h.y, (float)h.z, (float)h.w);
}
__device__ __forceinline__ void store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void store4(f16_t* p, float4 v) {
    *reinterpret_cast<half4_t*>(p) = half4_t{(f16_t)v.x, (f16_t)v.y, (f16_t)v.z, (f16_t)v.w};
}
__device__ __forceinline__ void store4(bf16_t* p, float4 v) {
    uint2 u;
    u.x = static_cast<unsigned>(f2bf(v.x)) | (static_cast<unsigned>(f2bf(v.y)) << 16);
    u.y = static_cast<unsigned>(f2bf(v.z)) | (static_cast<unsigned>(f2bf(v.w)) << 16);
    *reinterpret_cast<uint2*>(p) = u;
}

// Instance-norm moments are kept about a per-channel shift, (sum (x - sh), sum (x - sh)^2), with sh = act(bias[c]): a
// channel that its bias dominates (mean >> std) would lose its whole variance in f32 E[x^2] - E[x]^2.  The shift is a
// function of the layer's bias and activation code only, so producers and consumers compute the same value; bias
// NULL means sh = 0 (plain sums).  mean = sh + S1/N, var = S2/N - (S1/N)^2.
__device__ __forceinline__ float norm_shift(const float* bias, int act, int c) {
    if (!bias) return 0.f;
    const float b = bias[c];
    return act == 1 ? fmaxf(b, 0.f) : (act == 2 ? 1.f / (1.f + expf(-b)) : b);
}

// mean and 1/sigma of a (sample, channel) plane from its folded sums about shift.  Every kernel that needs them calls this
// one function with contraction off: whether the compiler fuses a multiply into the next add would otherwise be its own
// choice per call site, and the backward kernels must reproduce the forward launch's mean and 1/sigma BIT FOR BIT to
// recognise y0 below.
__device__ __forceinline__ void norm_mean_inv(float s1, float s2, float inv_n, float shift, float eps, float& mean, float& inv) {
#pragma clang fp contract(off)
    const float md = s1 * inv_n;                     // mean of (x - shift)
    const float var = fmaxf(s2 * inv_n - md * md, 0.f);
    mean = md + shift;
    inv = rsqrtf(var + eps);
}

// The one-launch forward (cutie_conv_in_fwd) stores only y = (a - mean) * inv.  The ReLU pattern the backward pass needs
// travels INSIDE y: every a == 0 element of a (sample, channel) plane holds exactly y0 = round16((0 - mean) * inv), and an
// a > 0 element whose y rounds onto y0 is stored one unit in the last place above it (the forward stores y0_bits itself
// into the a == 0 elements, it does not recompute them).  So mask = (bits of y != bits of y0),
// exactly, with no second tensor.  y >= y0 always (rounding is monotone), so the bump moves y by at most one ulp.
__device__ __forceinline__ unsigned short bits16(bf16_t v) { return v; }
__device__ __forceinline__ unsigned short bits16(f16_t v) { return __builtin_bit_cast(unsigned short, v); }
__device__ __forceinline__ unsigned short next_up16(unsigned short u) {      // the next 16-bit float above u (same for f16 / bf16)
    if (u == 0x8000u) return 0x0001u;              // -0 -> smallest positive
    return (u & 0x8000u) ? (unsigned short)(u - 1u) : (unsigned short)(u + 1u);
}
template <typename T> __device__ __forceinline__ unsigned short y0_bits(float mean, float inv) {
#pragma clang fp contract(off)
    float y0 = (0.f - mean) * inv;
    // the product must exist as an f32 value before it is narrowed: left alone the compiler narrows straight out of the
    // multiplier in some kernels (v_fma_mixlo_f16, one rounding) and not in others (two roundings) - one unit apart when
    // the product sits next to a tie
    asm volatile("" : "+v"(y0));
    T t;
    store_from_f32(&t, y0);
    const unsigned short u = bits16(t);
    return u == 0x8000u ? (unsigned short)0 : u;      // one zero: the compiler may turn 0 - mean into a negation
}
template <> __device__ __forceinline__ unsigned short y0_bits<float>(float, float) { return 0; }      // f32 path never stores y only

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// runtime dtype dispatch over one or two activation dtypes
#define CUTIE_DISPATCH_DTYPE(dt, NAME, ...)                              \
    switch (dt) {                                                        \
        case CUTIE_F32: { constexpr int NAME = CUTIE_F32; __VA_ARGS__; break; }   \
        case CUTIE_BF16: { constexpr int NAME = CUTIE_BF16; __VA_ARGS__; break; } \
        case CUTIE_F16: { constexpr int NAME = CUTIE_F16; __VA_ARGS__; break; }   \
        default: cutie::set_error("unknown dtype %d", (int)(dt)); return CUTIE_E_ARG; \
    }

}  // namespace cutie
