// Shared by conv_mfma.hip (general kernel) and conv_wide.hip (256-channel-tile kernel).
#pragma once
#include "common.h"

namespace cutie {
namespace convk {

typedef __attribute__((ext_vector_type(8))) short short8_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(4))) float float4_t;

struct Seg {
    const char* x;
    const char* w;
    long long w_tap_stride;  // elements
    int ld_x, cin, kh, kw, dil, sgn;
};

constexpr int CONV_MAX_PERM = 128;

struct ConvParams {
    Seg seg[CUTIE_MAX_SEGS];
    int nseg;
    int B, H, W, n_out;
    char* out;
    int ld_out;
    const float* bias;
    int relu;
    float* stats;        // [B][stats_tiles][n_out][2]: this launch stores slot (tile row * tiles_w + tile col)
    int stats_tiles;
    int TH, TWB, tiles_h, tiles_w, n_tiles_n;
    int a_bytes;  // LDS bytes reserved for the A patch
    // Instance-norm backward reduction fused into a data-gradient launch (conv_wide only; nullable):
    // `out` is dy of a layer whose forward pre-norm activation is bw_a with moments bw_stats; tile t
    // of a sample STORES (sum dy, sum dy * xhat) over its pixels into slot t of bw_red.
    const char* bw_a;
    int bw_ld_a;
    const float* bw_stats;   // [B][bw_stats_tiles][n_out][2]
    int bw_stats_tiles;
    float bw_eps;
    const float* bw_bias;    // the forward layer's bias and activation code: the shift its moments are kept about (nullable)
    int bw_act;
    float* bw_red;           // [B][bw_red_tiles][n_out][2]
    int bw_red_tiles;
    // Rendezvous of a sample's tiles inside such a data-gradient launch (conv_wide only, with bw_*; nullable): after
    // publishing its slot of bw_red every tile waits until the sample's other tiles have published theirs, folds all
    // slots and stores dz = act'(a) * inv * (dy - mean(dy) - xhat * mean(dy * xhat)) - what cutie_in_bwd_apply computes -
    // instead of dy, plus its share of the bias gradient.  `out` is not written then.
    unsigned long long* rv_slots;   // [B][bw_red_tiles][n_out] 16-byte {s1, tag, s2, tag} granule pairs (zeroed once by the caller)
    char* rv_dz;
    int rv_ld_dz;
    float* rv_dbias;         // [B][rv_dbias_tiles][n_out], tile t of a sample stores row t
    int rv_dbias_tiles;
    int rv_relu;             // 0 none, 1 ReLU mask a > 0
    int rv_plain_order;      // 1: block ids in launch order (a sample's tiles spread over the XCDs)
    int* rv_err;             // [1]: waves whose bounded wait ran out (the launch's results are then invalid); word 0 of the sync area
    int* rv_late;            // [1], nullable: the caller's own counter of the same event (one word for a whole step: the
                             // optimizer entry points skip the update when it is non-zero)
    // Forward rendezvous (conv_wide only, with stats, nullable): the sample's tiles meet after publishing their moments
    // and `out` receives the NORMALISED activation y = (a - mean) * rsqrt(var + fw_eps); a itself is not stored.
    // Shares rv_plain_order / rv_err / rv_late with the backward form.
    unsigned long long* fw_slots;   // [B][stats_tiles][n_out] 16-byte {s1, tag, s2, tag} granule pairs (zeroed once by the caller)
    float fw_eps;
    // conv_wide: order in which an XCD takes the pixel tiles of its samples - heaviest (most live taps)
    // first, so that the light corner tiles of a dilated layer fill the end of the launch.  0 = row-major.
    int perm_n;
    unsigned char perm[CONV_MAX_PERM];
    // cutie_conv_clock_probe: nullable; workgroup 0 of a conv_wide launch adds {shader-clock cycles, 100 MHz ticks, 1} of its
    // own lifetime to these three words - the clock the chip holds under the launch is cycles / ticks x 100 MHz
    unsigned long long* clk_probe;
};
unsigned long long* conv_clock_probe();      // the registered probe buffer of the current process (conv_mfma.hip), or nullptr

template <int DT> struct Traits;
template <> struct Traits<CUTIE_BF16> { static constexpr int ES = 2; static constexpr int BKE = 64; };
template <> struct Traits<CUTIE_F16> { static constexpr int ES = 2; static constexpr int BKE = 64; };
template <> struct Traits<CUTIE_F32> { static constexpr int ES = 4; static constexpr int BKE = 32; };

__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));

// one 16x16x32 MFMA on 16-bit fragments (8 K values per lane), bf16 or f16 by the element type
template <int DT>
__device__ __forceinline__ float4_t mma16(short8_t a, short8_t b, float4_t c) {
    static_assert(DT == CUTIE_BF16 || DT == CUTIE_F16, "16-bit element types only");
    if constexpr (DT == CUTIE_BF16) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    } else {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8_t, a), __builtin_bit_cast(half8_t, b), c, 0, 0, 0);
    }
}

template <int DT>
__device__ __forceinline__ void mma_frag(const uint4& a, const uint4& b, float4_t& c) {
    if constexpr (DT != CUTIE_F32) {
        c = mma16<DT>(__builtin_bit_cast(short8_t, a), __builtin_bit_cast(short8_t, b), c);
    } else {
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    }
}

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
constexpr unsigned BUF_OOB = 0x80000000u;   // beyond any num_records: the buffer load returns zeros

__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff) {
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, 0, 0);
    return make_uint4(v.x, v.y, v.z, v.w);
}


typedef __attribute__((address_space(3))) void lds_void_t;
typedef unsigned u32x4s_t __attribute__((ext_vector_type(4)));

// raw buffer descriptor (stride 0, 32-bit records): base, base_hi, num_records, flags
__device__ __forceinline__ u32x4s_t make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    u32x4s_t r;
    r.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    r.z = __builtin_amdgcn_readfirstlane(bytes);
    r.w = 0x00020000u;
    return r;
}

// `buffer_load_dwordx4 ... lds`: 64 lanes x 16 B land at LDS byte address lds_dst + 16*lane; each
// lane's source offset is its own.  Written as inline asm so that hipcc does not put a
// `s_waitcnt vmcnt(0)` in front of every later ds_read (it cannot tell the two LDS buffers
// apart); the pipeline's own counted wait + barrier order the data.  M0 is saved and restored.
__device__ __forceinline__ void glds16(u32x4s_t rsrc, unsigned lds_dst, unsigned voff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(rsrc), "s"(lds_dst)
                 : "memory");
}

}  // namespace convk

// conv_wide.hip: 16-bit in / out of one type, n_out a multiple of 64 (channel tile 256 / 128 / 64), every
// segment cin a multiple of 64
int launch_conv_wide(const convk::ConvParams& p, int dtype, int bn, hipStream_t stream);
int conv_wide_tile(const convk::ConvParams& p);
bool conv_patch_ok(const convk::ConvParams& p);      // p.TH / p.TWB set: conv_patch_kernel can take the launch (bn code 1000 + tile)
bool conv_wide_rendezvous_ok(const convk::ConvParams& p, int bn, int* plain_order);
bool conv_wide_fw_rendezvous_ok(const convk::ConvParams& p, int bn, int* plain_order);

}  // namespace cutie
