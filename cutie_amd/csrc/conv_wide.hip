// K2w: forward / data-gradient convolution for the wide layers (bf16 or f16, n_out a multiple of 256),
// built like a 256 x 256 GEMM tile with an asynchronous two-stage LDS ring.
//
// Same contract as conv_kernel in conv_mfma.hip (tf.nn.conv2d / atrous_conv2d + bias + ReLU of
// /root/reference/model_framework.py:168-204,414-422 and their input gradients); what differs is
// how the operands reach the matrix cores:
//   * workgroup = 512 lanes = 8 waves as 2 (pixel halves) x 4 (channel quarters); tile = 256
//     pixels (TH rows x TWB*16 columns of one sample; 16 x 16 for the 64 x 64 grid) x 256 output
//     channels; a wave owns 128 x 64 = 8 x 4 accumulators of v_mfma_f32_16x16x32_{bf16,f16};
//   * one K step = (tap, 64 input channels).  Its A tile (256 shifted pixels x 128 B) and B tile
//     (256 weight rows x 128 B) go global -> LDS by `buffer_load_dwordx4 ... lds` (no registers,
//     no ds_write): each lane's SOURCE address carries the tap shift, the XOR swizzle and the
//     zero padding (an out-of-range offset makes the hardware write zeros), the LDS image is
//     lane-linear.  Step s+1 is in flight while step s computes (2 x 64 KB of LDS);
//   * one barrier per K step: wait for my own loads, barrier (everybody's loads have landed and
//     everybody is done reading the other buffer), issue the next step's loads, compute;
//   * a tap whose shifted tile lies wholly outside the image is not a step at all (uniform for
//     the workgroup), which keeps the in-bounds FLOP accounting for 16-aligned dilations; since that makes
//     the tiles of a dilated layer unequal, an XCD takes its samples' tiles heaviest first (plan_tile_order);
//   * epilogue: bias + activation, the tile's instance-norm moments STORED into the tile's slot, and - for a
//     data gradient into an instance-normalised layer - that layer's backward sums (sum dy, sum dy * xhat)
//     taken in the same store pass (ConvParams::bw_*), so no separate reduction pass reads dy and a again.
#include "conv_common.h"
#include <type_traits>

namespace cutie {
namespace {
using namespace cutie::convk;

constexpr int CW_NT = 512;
constexpr int CW_A0 = 0, CW_B0 = 65536, CW_ABUF = 32768;    // LDS map: A0 A1 (32 KB each), B0 B1 (BN * 128 B each)
template <int BN, int NW = 8> struct Shape {
    static constexpr int NT = NW * 64;          // lanes of the workgroup: 8 waves, or 4 in the two-workgroups-a-CU form (conv_duo_kernel)
    static constexpr int WN = BN / 64;          // waves across the channel tile (each owns 64 channels)
    static constexpr int WM = NW / WN;          // waves across the 256 pixels
    static constexpr int MB = 16 / WM;          // 16-pixel blocks per wave: 8 / 4 / 2
    static constexpr int NB = BN / 64;          // 1-KB pieces of the B tile each wave stages per step
    static constexpr int NP = 4 + NB;           // pieces per wave and step (A: always 4)
    static constexpr int BBUF = BN * 128;
    static constexpr int WSUM = 256 * BN * 2;   // per-wave instance-norm partials sit behind the staged output tile
    static constexpr int TAB = WSUM + WM * BN * 8;   // [2][BN] f32 mean, inv of the fused norm-backward reduction
    static constexpr int LDS = (CW_B0 + 2 * BBUF > TAB + 2 * BN * 4) ? CW_B0 + 2 * BBUF : TAB + 2 * BN * 4;
};

// The four-stage ring of 32-channel K steps (conv_wide_kernel<..., RING = true>): a stage holds the A tile of a step
// (256 pixel rows x 64 B) and its B tile (BN weight rows x 64 B); 4 x 32 KB at BN = 256.  Rows are 64 B = four 16-byte
// chunks; chunk c of row r sits at chunk position c ^ rswz(r), which keeps the 16 lanes of every ds_read_b128 lane group
// (MI355X_MICROARCH.md, LDS) on 16 different 16-byte bank groups.
constexpr int RG_ABUF = 16384;
template <int BN> struct Ring {
    static constexpr int BBUF = BN * 64;
    static constexpr int STAGE = RG_ABUF + BBUF;
    static constexpr int NBP = BN / 128;        // 1-KB pieces of the B tile per wave and step
    static constexpr int NP = 2 + NBP;          // pieces per wave and step (A: always 2)
    static_assert(BN % 128 == 0, "the ring wants whole pieces per wave");
    static_assert(4 * STAGE <= Shape<BN>::LDS, "ring does not fit the kernel's LDS grant");
};
__device__ __forceinline__ int rswz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }      // {0, 2, 3, 1}[(row >> 2) & 3]

__device__ __forceinline__ unsigned short u16_of(const uint4& u, int e) {      // e-th 16-bit element of a 16-byte piece
    const unsigned c = e < 2 ? u.x : (e < 4 ? u.y : (e < 6 ? u.z : u.w));
    return (unsigned short)((e & 1) ? (c >> 16) : (c & 0xffffu));
}

template <int DT> __device__ __forceinline__ void unpack8(const uint4& u, float* v) {
    if constexpr (DT == CUTIE_BF16) {
        v[0] = __uint_as_float(u.x << 16); v[1] = __uint_as_float(u.x & 0xffff0000u);
        v[2] = __uint_as_float(u.y << 16); v[3] = __uint_as_float(u.y & 0xffff0000u);
        v[4] = __uint_as_float(u.z << 16); v[5] = __uint_as_float(u.z & 0xffff0000u);
        v[6] = __uint_as_float(u.w << 16); v[7] = __uint_as_float(u.w & 0xffff0000u);
    } else {
        const half8_t h = __builtin_bit_cast(half8_t, u);
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = static_cast<float>(h[i]);
    }
}

// v_fma_mix_f32: an f32 fused multiply-add whose f16 sources are taken straight from the halves of a packed register and
// widened exactly - the same value as v_cvt_f32_f16 followed by v_fma_f32 / v_add_f32, in one instruction instead of two or
// three.  The store passes of the backward epilogues are VALU-bound; hipcc finds this form for about a
// quarter of their elements only.
#ifndef CW_MIX
#define CW_MIX 1      // -DCW_MIX=0 (tools/build_variant.sh): the widened form everywhere, for A/B runs
#endif
__device__ __forceinline__ float mix_add(unsigned h2, bool hi, float c) {                    // (float)h + c
    float d;
    if (hi) asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(c));
    else asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(c));
    return d;
}
__device__ __forceinline__ float mix_fma_hh(unsigned a2, unsigned b2, bool hi, float c) {    // fma((float)a, (float)b, c)
    float d;
    if (hi) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a2), "v"(b2), "v"(c));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(d) : "v"(a2), "v"(b2), "v"(c));
    return d;
}
__device__ __forceinline__ float mix_fma_hf(unsigned a2, bool hi, float b, float c) {        // fma((float)a, b, c)
    float d;
    if (hi) asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a2), "v"(b), "v"(c));
    else asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(a2), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ unsigned pair_of(const uint4& u, int e) { return e < 2 ? u.x : (e < 4 ? u.y : (e < 6 ? u.z : u.w)); }

// Rendezvous of a sample's tiles on tagged granules (MI355X_MICROARCH.md, granule hand-off).  slots:
// [B][slot_tiles][n_out] 16-byte pairs of 8-byte granules {s1, tag, s2, tag}; a tile publishes its two sums for channel c
// with the launch's tag in ONE 16-byte write-through store, then every channel lane sweeps the sample's T tiles with ONE
// 16-byte load a tile until both halves of all of them carry that tag (bounded) and folds them in slot order.  Each 8-byte
// half validates itself, so nothing depends on the 16 bytes arriving as one (round 3: the two sums were two 8-byte
// stores and two 8-byte loads a tile before - `sc1` accesses of 8 B cost 2.7x their bytes, and the sweep was 15 us of a
// dense launch).  The tag of a launch is the slot's previous tag + 1: every launch rewrites all slots of its area.
constexpr int GR_SC1 = 16;                        // cache policy bit of the buffer builtins: sc1 (served by / written through to L2's memory side)
constexpr int GR_VOLATILE = (int)0x80000000u;     // ... and "volatile": the polling loads are not to be hoisted or merged
__device__ __forceinline__ __amdgpu_buffer_rsrc_t granule_rsrc(const unsigned long long* slots, int B, int slot_tiles, int n_out) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)slots, 0, (int)((long long)B * slot_tiles * n_out * 16), 0x00020000);
}
__device__ __forceinline__ unsigned granule_off(int b, int slot_tiles, int tile, int n_out, int c) {
    return (unsigned)((((long long)b * slot_tiles + tile) * n_out + c) * 16);
}
__device__ __forceinline__ unsigned granule_tag(const unsigned long long* slots, int b, int slot_tiles, int tile, int n_out, int c) {
    return (unsigned)(__hip_atomic_load(slots + (((long long)b * slot_tiles + tile) * n_out + c) * 2, __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_AGENT) >> 32) + 1u;
}
__device__ __forceinline__ void granule_publish(unsigned long long* slots, int B, int b, int slot_tiles, int tile, int n_out, int c,
                                                unsigned tag, float s1, float s2) {
    const u32x4_t v = {__float_as_uint(s1), tag, __float_as_uint(s2), tag};
    __builtin_amdgcn_raw_buffer_store_b128(v, granule_rsrc(slots, B, slot_tiles, n_out), (int)granule_off(b, slot_tiles, tile, n_out, c), 0, GR_SC1);
}
__device__ __forceinline__ void granule_sweep(const unsigned long long* slots, int B, int b, int slot_tiles, int T, int n_out, int c,
                                              unsigned tag, int* err, int* err2, float& r1, float& r2) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();      // 100 MHz
    const __amdgpu_buffer_rsrc_t rs = granule_rsrc(slots, B, slot_tiles, n_out);
    bool late = false;
    while (true) {
        bool all = true;
        r1 = 0.f;
        r2 = 0.f;
        for (int tb = 0; tb < T; tb += 16) {       // sixteen tiles' granules in flight together
            u32x4_t v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int t = tb + u < T ? tb + u : T - 1;
                v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)granule_off(b, slot_tiles, t, n_out, c), 0, GR_SC1 | GR_VOLATILE);
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (tb + u < T) {
                    all = all && v[u].y == tag && v[u].w == tag;
                    r1 += __uint_as_float(v[u].x);
                    r2 += __uint_as_float(v[u].z);
                }
            }
        }
#ifdef CW_DIAG_NOWAIT      // diagnostic build only (tools/build_variant.sh): prices the protocol without the waiting - WRONG results
        break;
#endif
        if (all) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 20000000ull) {      // 0.2 s: the other tiles never came
            late = true;
            break;
        }
        __builtin_amdgcn_s_sleep(4);
    }
    // a wait that ran out: the sums folded above are incomplete and what this tile stores is wrong.  Counted in the
    // sync area's word 0 and in the caller's step-wide word, which gates the optimizer (cutie_opt_step / cutie_adam_step)
    if (late && (threadIdx.x & 63) == 0) {
        atomicAdd(err, 1);
        if (err2) atomicAdd(err2, 1);
    }
}

// element offset of (pixel row, channel) in the staged output tile: [256][BN] with the 16-byte chunk index XORed by the
// row, so that the sixteen lanes of an 8-byte store group (sixteen rows, one chunk) land on different banks
template <int BN>
__device__ __forceinline__ int cs_off(int row, int ch) {
    constexpr int PPR = BN / 8;
    return row * BN + ((((ch >> 3) ^ (row & (PPR < 16 ? PPR - 1 : 15))) << 3) | (ch & 7));
}
// sum over the 16 lanes of a DPP row (all of them end up with it): xor 1, xor 2, half-row mirror, row mirror
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}

// bias, ReLU, instance-norm moments, LDS-staged coalesced store of a 256-pixel x BN-channel tile.
// With p.bw_a (data gradient of an instance-normalised layer) the store pass also takes this tile's share of
// (sum dy, sum dy * xhat) from the ROUNDED dy it stores and the layer's forward activation, so that
// cutie_in_bwd_apply can follow without a separate reduction pass over dy and a.
// MODE: 0 plain (forward conv / plain data gradient), 1 data gradient + the norm gradient's two sums, 2 the same with the
// sample's tiles meeting (dz out), 3 forward with the tiles meeting (y out).  A compile-time mode so that each path
// allocates registers for itself (the backward modes hold 64 registers of forward activation across the staging).
// FY (backward modes): bw_a is the layer's normalised output y (CUTIE_ACT_FROM_Y) - xhat = y, ReLU pattern out of y's bits;
// RELU (mode 2): the layer has a ReLU.  Compile-time, so that the per-element code holds one formula, not a select of two.
// The forward moments of the layer a backward launch differentiates, folded in slot order (exactly as norm.hip's fold takes
// them) by the channel lanes.  Called BEFORE the K loop: sixteen dependent-free loads whose latency then passes under the
// main loop instead of standing in front of the epilogue (two registers a lane carried across it).
template <int BN>
__device__ __forceinline__ void fold_bw_stats(const ConvParams& p, int b, int n0, float& fs, float& fss) {
    fs = 0.f;
    fss = 0.f;
    const int tid = threadIdx.x;
    if (tid < BN && n0 + tid < p.n_out) {
        const float* q = p.bw_stats + ((long long)b * p.bw_stats_tiles * p.n_out + n0 + tid) * 2;
        for (int t0 = 0; t0 < p.bw_stats_tiles; t0 += 16) {      // sixteen slots in flight together, added in slot order
            float2 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                v[u] = t0 + u < p.bw_stats_tiles ? *reinterpret_cast<const float2*>(q + (long long)(t0 + u) * p.n_out * 2) : make_float2(0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (t0 + u < p.bw_stats_tiles) {
                    fs += v[u].x;
                    fss += v[u].y;
                }
            }
        }
    }
}

template <int DT, int BN, int MODE, bool FY, bool RELU, int NW = 8>
__device__ __forceinline__ void conv_wide_epilogue(const ConvParams& p, float4_t (&acc)[Shape<BN, NW>::MB][4], char* smem, int b,
                                                   int tile, int h0, int w0, int n0, float fs, float fss) {
    typedef Shape<BN, NW> S;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / S::WN, wn = wave % S::WN;
    const int li = lane & 15, lg = lane >> 4;
    const int H = p.H, W = p.W, TCB = p.TWB, n_out = p.n_out;
    typedef typename elem<DT>::type TO;
    constexpr int PPR = BN / 8;          // 16-byte pieces per pixel row
    constexpr int RG = S::NT / PPR;      // pixel rows side by side in the store pass: 16 / 32 / 64
    constexpr int IT = 256 / RG;         // store iterations per lane: 16 / 8 / 4
    const int pc = tid % PPR, rg = tid / PPR;
    constexpr bool bw = MODE == 1 || MODE == 2;
    constexpr bool rv = MODE == 2;
    unsigned rv_tag = 0;
    if (rv) rv_tag = granule_tag(p.rv_slots, b, p.bw_red_tiles, tile, n_out, n0);
    constexpr bool fw = MODE == 3;                                           // forward: the tiles meet, y is stored, a is not
#ifdef CW_DIAG_NOEPI      // diagnostic build (tools/build_variant.sh): the K loop alone, results discarded
    {
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < S::MB; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
        if (s == 1.2345e-33f) reinterpret_cast<float*>(p.out)[0] = s;
        return;
    }
#endif
    unsigned fw_tag = 0;
    if (fw) fw_tag = granule_tag(p.fw_slots, b, p.stats_tiles, tile, n_out, n0);

    // the forward activation pieces and the forward moments are requested first: their latency passes
    // under the staging of the accumulators (mode 2 holds the tile's dy in registers as well and has no room for them
    // beside the accumulators: there they are requested right after the staging stores)
    uint4 av[IT];
    auto load_av = [&]() {
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int ml = rg + k * RG;
            const int mb = ml >> 4;
            const int h = h0 + mb / TCB, w = w0 + (mb % TCB) * 16 + (ml & 15);
            av[k] = make_uint4(0u, 0u, 0u, 0u);
            if (h < H && w < W && n0 + pc * 8 < n_out)
                av[k] = *reinterpret_cast<const uint4*>(reinterpret_cast<const TO*>(p.bw_a) +
                                                        ((long long)(b * H + h) * W + w) * p.bw_ld_a + n0 + pc * 8);
        }
    };
    if constexpr (bw && !rv) load_av();      // (fs, fss: the forward moments, folded before the K loop - fold_bw_stats)
    __syncthreads();
    TO* Cs = reinterpret_cast<TO*>(smem);  // [256 pixels][BN channels]
    float2* wsum = reinterpret_cast<float2*>(smem + S::WSUM);   // [WM][BN]
    float* tab = reinterpret_cast<float*>(smem + S::TAB);       // [2][BN]
    if (bw && tid < BN) {
        const float inv_n = 1.f / (float)(H * W);
        // the moments are kept about act(bias): norm.hip's fold_slots<1>
        norm_mean_inv(fs, fss, inv_n, n0 + tid < n_out ? norm_shift(p.bw_bias, p.bw_act & 15, n0 + tid) : 0.f, p.bw_eps,
                      tab[tid], tab[BN + tid]);
    }
    const bool full_tile = h0 + p.TH <= H && w0 + TCB * 16 <= W;
    // The MFMAs run with the weight fragment as the row operand: lane (li, lg) of accumulator [m][n] holds pixel
    // mb * 16 + li and the FOUR CONSECUTIVE channels n * 16 + lg * 4 + r.  So a lane rounds and packs them into one 8-byte
    // LDS store (32 ds_write_b64 per lane where the pixel-row layout took 128 two-byte stores, four lanes to a bank);
    // the staged image is [256 pixels][BN channels] with the 16-byte chunk index XORed by the pixel row (cs_chunk).
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int cb = wn * 64 + n * 16 + lg * 4;
        if constexpr (bw) {      // a data gradient: no bias, no activation, no moments of its own (checked by the launcher)
#pragma unroll
            for (int m = 0; m < S::MB; ++m) {
                const int mb = wm * S::MB + m;
                TO o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) store_from_f32(&o[r], acc[m][n][r]);
                *reinterpret_cast<uint2*>(Cs + cs_off<BN>(mb * 16 + li, cb)) = *reinterpret_cast<const uint2*>(o);
            }
        } else {
            float bv[4], sh[4], s1[4], s2[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                bv[r] = (p.bias && n0 + cb + r < n_out) ? p.bias[n0 + cb + r] : 0.f;
                sh[r] = p.relu ? fmaxf(bv[r], 0.f) : bv[r];      // norm_shift(p.bias, p.relu, col): the moments are kept about it
                s1[r] = 0.f;
                s2[r] = 0.f;
            }
            // a tile that lies wholly inside the image (uniform per workgroup) needs no per-pixel masking
            auto stage = [&](auto ragged, auto relu) {
#pragma unroll
                for (int m = 0; m < S::MB; ++m) {
                    const int mb = wm * S::MB + m;
                    bool inside = true;
                    if constexpr (decltype(ragged)::value) {
                        const int h = h0 + mb / TCB, w = w0 + (mb % TCB) * 16 + li;
                        inside = h < H && w < W;
                    }
                    TO o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float v = acc[m][n][r] + bv[r];
                        if constexpr (decltype(relu)::value) v = fmaxf(v, 0.f);
                        float d = v - sh[r];
                        if constexpr (decltype(ragged)::value) {
                            v = inside ? v : 0.f;
                            d = inside ? d : 0.f;
                        }
                        s1[r] += d;
                        s2[r] = __builtin_fmaf(d, d, s2[r]);
                        if constexpr (fw) acc[m][n][r] = v;        // stays in registers until the sample's moments are known
                        else store_from_f32(&o[r], v);
                    }
                    if constexpr (!fw) *reinterpret_cast<uint2*>(Cs + cs_off<BN>(mb * 16 + li, cb)) = *reinterpret_cast<const uint2*>(o);
                }
            };
            if (full_tile) {
                if (p.relu) stage(std::false_type{}, std::true_type{});
                else stage(std::false_type{}, std::false_type{});
            } else {
                if (p.relu) stage(std::true_type{}, std::true_type{});
                else stage(std::true_type{}, std::false_type{});
            }
            if (p.stats) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    s1[r] = row16_sum(s1[r]);
                    s2[r] = row16_sum(s2[r]);
                    if (li == 0) wsum[wm * BN + cb + r] = make_float2(s1[r], s2[r]);
                }
            }
        }
    }
    if constexpr (rv) load_av();
    __syncthreads();
    if (p.stats && tid < BN && n0 + tid < n_out) {
        // the tile's moments: the waves' partials added in wave order, STORED into this tile's slot -
        // no float atomics, so reruns (and the ReLU patterns downstream) are bit-identical
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int w = 0; w < S::WM; ++w) {
            const float2 v = wsum[w * BN + tid];
            s1 += v.x;
            s2 += v.y;
        }
        *reinterpret_cast<float2*>(p.stats + (((long long)b * p.stats_tiles + tile) * n_out + n0 + tid) * 2) = make_float2(s1, s2);
        if (fw) {
            // forward rendezvous: publish, wait for the sample's other tiles, fold all of them in slot order (the
            // arithmetic of norm.hip's fold_slots<1>) - this lane then knows its channel's mean and 1/sigma
            granule_publish(p.fw_slots, p.B, b, p.stats_tiles, tile, n_out, n0 + tid, fw_tag, s1, s2);
            float r1, r2;
            granule_sweep(p.fw_slots, p.B, b, p.stats_tiles, p.tiles_h * p.tiles_w, n_out, n0 + tid, fw_tag, p.rv_err, p.rv_late, r1, r2);
            const float inv_n = 1.f / (float)(H * W);
            norm_mean_inv(r1, r2, inv_n, norm_shift(p.bias, p.relu, n0 + tid), p.fw_eps, tab[tid], tab[BN + tid]);
        }
    }
    if (fw) {
        __syncthreads();
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const int cb = wn * 64 + n * 16 + lg * 4;
            float mean[4], inv[4];
            unsigned short y0[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mean[r] = tab[cb + r];
                inv[r] = tab[BN + cb + r];
                y0[r] = y0_bits<TO>(mean[r], inv[r]);
            }
#pragma unroll
            for (int m = 0; m < S::MB; ++m) {
                const int mb = wm * S::MB + m;
                TO o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {    // y = (a - mean) * inv from the UNROUNDED activation: one rounding
                    const float v = acc[m][n][r];
                    TO yh;
                    store_from_f32(&yh, (v - mean[r]) * inv[r]);
                    // the ReLU pattern travels inside y (common.h): an active element never holds y0's bits
                    if (p.relu) {
                        if (!(v > 0.f)) yh = __builtin_bit_cast(TO, y0[r]);
                        else if (bits16(yh) == y0[r]) yh = __builtin_bit_cast(TO, next_up16(y0[r]));
                    }
                    o[r] = yh;
                }
                *reinterpret_cast<uint2*>(Cs + cs_off<BN>(mb * 16 + li, cb)) = *reinterpret_cast<const uint2*>(o);
            }
        }
        __syncthreads();
    }
    const int col = n0 + pc * 8;
    float mean8[8], inv8[8], g1[8], g2[8];
    constexpr bool from_y = bw && FY;
    unsigned short y0b[8];
    if (bw) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            mean8[e] = tab[pc * 8 + e];
            inv8[e] = tab[BN + pc * 8 + e];
            y0b[e] = y0_bits<TO>(mean8[e], inv8[e]);
            g1[e] = 0.f;
            g2[e] = 0.f;
        }
    }
    uint4 gvk[IT];                       // rendezvous: the tile's rounded dy stays in registers for the second pass
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int ml = rg + k * RG;
        const int mb = ml >> 4;
        const int h = h0 + mb / TCB, w = w0 + (mb % TCB) * 16 + (ml & 15);
        if (h < H && w < W && col < n_out) {
            const uint4 gv = *reinterpret_cast<const uint4*>(Cs + cs_off<BN>(ml, pc * 8));
            if (rv) {
                gvk[k] = gv;
            } else {
                TO* dst = reinterpret_cast<TO*>(p.out) + ((long long)(b * H + h) * W + w) * p.ld_out + col;
#ifdef CW_DIAG_NOSTORE    // diagnostic build: everything but the output store
                if (gv.x == 0x12345678u)
#endif
                *reinterpret_cast<uint4*>(dst) = gv;
            }
            if constexpr (bw && from_y && DT == CUTIE_F16 && CW_MIX) {
                // g1 += dy, g2 = fma(dy, y, g2) on the packed halves: the same bits as the widened form below
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    g1[e] = mix_add(pair_of(gv, e), e & 1, g1[e]);
                    g2[e] = mix_fma_hh(pair_of(gv, e), pair_of(av[k], e), e & 1, g2[e]);
                }
            } else if (bw) {
                float g[8], x[8];
                unpack8<DT>(gv, g);
                unpack8<DT>(av[k], x);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    g1[e] += g[e];
                    g2[e] = __builtin_fmaf(g[e], from_y ? x[e] : (x[e] - mean8[e]) * inv8[e], g2[e]);
                }
            }
        }
    }
    if (!bw) return;
    __syncthreads();                     // the staged tile has been read: its LDS becomes the reduction buffer
    float2* rb = reinterpret_cast<float2*>(smem);      // [RG][BN]
#pragma unroll
    for (int e = 0; e < 8; ++e) rb[rg * BN + pc * 8 + e] = make_float2(g1[e], g2[e]);
    __syncthreads();
    if (tid < BN && n0 + tid < n_out) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll 4
        for (int r = 0; r < RG; ++r) {   // fixed order
            const float2 v = rb[r * BN + tid];
            s1 += v.x;
            s2 += v.y;
        }
        if (rv) {
            granule_publish(p.rv_slots, p.B, b, p.bw_red_tiles, tile, n_out, n0 + tid, rv_tag, s1, s2);
        } else {
            *reinterpret_cast<float2*>(p.bw_red + (((long long)b * p.bw_red_tiles + tile) * n_out + n0 + tid) * 2) = make_float2(s1, s2);
        }
    }
    if (!rv) return;

    // ---- rendezvous: every channel lane sweeps the granules of the sample's tiles until all carry this launch's tag
    // (bounded), then folds them in slot order - the arithmetic of norm.hip's fold.  No flag, no counter, no fence.
    const int T = p.tiles_h * p.tiles_w;
    float* tab2 = reinterpret_cast<float*>(smem + S::WSUM);      // [2][BN]: mean(dy), mean(dy * xhat)
    if (tid < BN && n0 + tid < n_out) {
        const float inv_n = 1.f / (float)(H * W);
        float r1, r2;
        granule_sweep(p.rv_slots, p.B, b, p.bw_red_tiles, T, n_out, n0 + tid, rv_tag, p.rv_err, p.rv_late, r1, r2);
        tab2[tid] = r1 * inv_n;
        tab2[BN + tid] = r2 * inv_n;
    }
    __syncthreads();
    float sb[8];
    {
        float m1[8], m2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            m1[e] = tab2[pc * 8 + e];
            m2[e] = tab2[BN + pc * 8 + e];
            sb[e] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < IT; ++k) {
            const int ml = rg + k * RG;
            const int mb = ml >> 4;
            const int h = h0 + mb / TCB, w = w0 + (mb % TCB) * 16 + (ml & 15);
            if (h < H && w < W && col < n_out) {
                float g[8], x[8];
                constexpr bool mix = from_y && DT == CUTIE_F16 && CW_MIX;
                if constexpr (!mix) {
                    unpack8<DT>(gvk[k], g);
                    unpack8<DT>(av[k], x);
                }
                TO o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {     // the arithmetic of in_bwd_apply_kernel, operation for operation
                    float d;
                    if constexpr (mix) {
                        // fma(-y, m2, dy - m1) = fma(y, -m2, dy + (-m1)), both steps straight off the packed halves
                        const float t = mix_add(pair_of(gvk[k], e), e & 1, -m1[e]);
                        d = inv8[e] * mix_fma_hf(pair_of(av[k], e), e & 1, -m2[e], t);
                    } else {
                        const float xh = from_y ? x[e] : (x[e] - mean8[e]) * inv8[e];
                        d = inv8[e] * __builtin_fmaf(-xh, m2[e], g[e] - m1[e]);
                    }
                    if constexpr (RELU) {
                        const bool active = from_y ? (u16_of(av[k], e) != y0b[e]) : (x[e] > 0.f);
                        if (!active) d = 0.f;
                    }
                    sb[e] += d;
                    store_from_f32(&o[e], d);
                }
                TO* dst = reinterpret_cast<TO*>(p.rv_dz) + ((long long)(b * H + h) * W + w) * p.rv_ld_dz + col;
                *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(o);
            }
        }
    }
    if (!p.rv_dbias) return;
    __syncthreads();
    float* rbf = reinterpret_cast<float*>(smem);        // [RG][BN]
#pragma unroll
    for (int e = 0; e < 8; ++e) rbf[rg * BN + pc * 8 + e] = sb[e];
    __syncthreads();
    if (tid < BN && n0 + tid < n_out) {
        float s = 0.f;
#pragma unroll 4
        for (int r = 0; r < RG; ++r) s += rbf[r * BN + tid];
        p.rv_dbias[((long long)b * p.rv_dbias_tiles + tile) * n_out + n0 + tid] = s;
    }
}

// MODE: the epilogue (conv_wide_epilogue) - one kernel instance per mode, so that each allocates registers for its own path
template <int DT, int BN, int MODE, bool FY, bool RELU, bool RING = false>
__global__ __launch_bounds__(CW_NT) void conv_wide_kernel(ConvParams p) {
    typedef Shape<BN> S;
    constexpr int MB = S::MB, NB = S::NB, NP = S::NP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / S::WN, wn = wave % S::WN;
    const int li = lane & 15, lg = lane >> 4;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_void_t*)smem);
    const int H = p.H, W = p.W, TCB = p.TWB, n_out = p.n_out;
    const int TR = p.TH, TC = TCB * 16;

    // consecutive block ids go round-robin over the 8 XCDs: give each XCD a contiguous run of
    // tiles so that neighbouring tiles (shared halos) meet in one L2
    int bid = blockIdx.x;
    const int nwg = gridDim.x;
    int nt, tw, th, b;
    if (p.perm_n) {
        // B % 8 == 0 (checked on the host): XCD x owns samples [x*S, (x+1)*S) and takes their tiles position by
        // position in the order of p.perm - all samples' heaviest tile first
        const int S = p.B >> 3;
        int l = bid >> 3;
        nt = l % p.n_tiles_n; l /= p.n_tiles_n;
        b = (bid & 7) * S + l % S;
        const int tile = p.perm[l / S];
        th = tile / p.tiles_w;
        tw = tile - th * p.tiles_w;
    } else {
        if ((nwg & 7) == 0 && !p.rv_plain_order) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
        nt = bid % p.n_tiles_n; bid /= p.n_tiles_n;
        tw = bid % p.tiles_w; bid /= p.tiles_w;
        th = bid % p.tiles_h;
        b = bid / p.tiles_h;
    }
    const int h0 = th * TR, w0 = tw * TC, n0 = nt * BN;
    // clock probe (cutie_conv_clock_probe): the first workgroup's lifetime in shader-clock cycles and in 100 MHz ticks
    const bool probe = p.clk_probe != nullptr && blockIdx.x == 0;
    unsigned long long ck0 = 0, rt0 = 0;
    if (probe) {
        ck0 = __builtin_amdgcn_s_memtime();
        rt0 = __builtin_amdgcn_s_memrealtime();
    }
    float bw_fs = 0.f, bw_fss = 0.f;
    if constexpr (MODE == 1 || MODE == 2) fold_bw_stats<BN>(p, b, n0, bw_fs, bw_fss);

    float4_t acc[MB][4];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = float4_t{0.f, 0.f, 0.f, 0.f};

    // staging plan: wave w moves 1-KB pieces 4w..4w+3 of the A tile and NB*w.. of the B tile; lane l
    // of a piece is (row = l >> 3, 16-byte slot = l & 7) and fetches chunk slot ^ swz(row)
    int a_h[4], a_w[4], a_c16[4], b_row[NB], b_c16[NB];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = wave * 32 + k * 8 + (lane >> 3);     // tile pixel 0..255
        const int mb = r >> 4;
        a_h[k] = h0 + mb / TCB;
        a_w[k] = w0 + (mb % TCB) * 16 + (r & 15);
        a_c16[k] = ((lane & 7) ^ swz(r)) * 16;
    }
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int r = wave * (NB * 8) + k * 8 + (lane >> 3);   // weight row 0..BN-1
        b_row[k] = n0 + r;
        b_c16[k] = ((lane & 7) ^ swz(r)) * 16;
    }
    // fragment read offsets (bytes from the A / B tile base); + m*2048 / n*2048 per 16-row block
    int a_rd[2], b_rd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int c16 = ((kk * 4 + lg) ^ ((li >> 1) & 7)) * 16;
        a_rd[kk] = (wm * (MB * 16) + li) * 128 + c16;
        b_rd[kk] = (wn * 64 + li) * 128 + c16;
    }
    if constexpr (RING) {
        // ---- Four-stage ring, one K step = (tap, 32 input channels) = MB x 4 MFMAs a wave.  While step s computes from
        // registers, the fragments of step s+1 are read out of their (already complete) stage and the pieces of step
        // s+3 are sent on their way, so the barrier that closes a step only orders buffer reuse: nothing waits for LDS
        // right behind it (the two-stage loop below reads a step's first fragments AFTER its barrier).  Round-3 experiment:
        // bit-identical results, no spills, and 5 % SLOWER per launch (194 -> 204 us, whole step 7.66 -> 7.93 ms, A/B on one
        // box, tools/gpu_ring_ab.sh) - the round trip behind the barrier is not what the loop waits for.  Kept for
        // variant builds (-DCW_WITH_RING, CUTIE_CONV_RING=1); never instantiated in the shipped library.
        typedef Ring<BN> R;
        constexpr int NBP = R::NBP, NPR = R::NP;
        int ra_h[2], ra_w[2], ra_c16[2], rb_row[NBP], rb_c16[NBP];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = (wave * 2 + k) * 16 + (lane >> 2);        // tile pixel 0..255
            const int mb = r >> 4;
            ra_h[k] = h0 + mb / TCB;
            ra_w[k] = w0 + (mb % TCB) * 16 + (r & 15);
            ra_c16[k] = ((lane & 3) ^ rswz(r)) * 16;
        }
#pragma unroll
        for (int k = 0; k < NBP; ++k) {
            const int r = (wave * NBP + k) * 16 + (lane >> 2);      // weight row 0..BN-1
            rb_row[k] = n0 + r;
            rb_c16[k] = ((lane & 3) ^ rswz(r)) * 16;
        }
        const int rc16 = (lg ^ rswz(li)) * 16;
        const int ra_rd = (wm * (MB * 16) + li) * 64 + rc16;          // + m * 1024 per 16-pixel block
        const int rb_rd = RG_ABUF + (wn * 64 + li) * 64 + rc16;       // + n * 1024 per 16-channel block
        uint4 fb0[4], fa0[MB], fb1[4], fa1[MB];                       // the fragments of the step in hand / of the next one

        for (int s = 0; s < p.nseg; ++s) {
            const Seg g = p.seg[s];
            const int ncc = g.cin >> 5;
            unsigned live = 0;
            for (int i = 0; i < g.kh; ++i) {
                const int sy = g.sgn * (i - g.kh / 2) * g.dil;
                if (h0 + sy + TR - 1 < 0 || h0 + sy >= H) continue;
                for (int j = 0; j < g.kw; ++j) {
                    const int sx = g.sgn * (j - g.kw / 2) * g.dil;
                    if (w0 + sx + TC - 1 < 0 || w0 + sx >= W) continue;
                    live |= 1u << (i * g.kw + j);
                }
            }
            live = __builtin_amdgcn_readfirstlane(live);
            if (live == 0u) continue;
            const int nsteps = __builtin_popcount(live) * ncc;

            const long long xbytes = ((long long)(H * W - 1) * g.ld_x + g.cin) * 2;
            const u32x4s_t rsrc_x = make_rsrc(g.x + (long long)b * H * W * g.ld_x * 2, (unsigned)xbytes);
            const long long wbytes = ((long long)(g.kh * g.kw - 1) * g.w_tap_stride + (long long)n_out * g.cin) * 2;
            const u32x4s_t rsrc_w = make_rsrc(g.w, (unsigned)wbytes);
            int a_base[2];
            unsigned b_base[NBP];
#pragma unroll
            for (int k = 0; k < 2; ++k) a_base[k] = (ra_h[k] * W + ra_w[k]) * g.ld_x * 2 + ra_c16[k];
#pragma unroll
            for (int k = 0; k < NBP; ++k) b_base[k] = rb_row[k] < n_out ? (unsigned)(rb_row[k] * g.cin * 2 + rb_c16[k]) : BUF_OOB;

            // the step whose pieces go out next: taps innermost (consecutive steps read the same 64-byte channel chunk of
            // overlapping pixels: L2), then the next 32 channels; beyond the last step the pieces are zero fills
            unsigned irest = live;
            int icc = 0, issued = 0, da = 0, sy = 0, sx = 0;
            unsigned db = 0;
            bool ivalid = false;
            auto advance = [&]() {
                ivalid = issued < nsteps;
                ++issued;
                if (!ivalid) return;
                if (irest == 0u) { irest = live; ++icc; }
                const int tap = __builtin_ctz(irest);
                irest &= irest - 1;
                const int i = tap / g.kw, j = tap - i * g.kw;
                sy = g.sgn * (i - g.kh / 2) * g.dil;
                sx = g.sgn * (j - g.kw / 2) * g.dil;
                da = ((sy * W + sx) * g.ld_x + icc * 32) * 2;
                db = (unsigned)(((long long)tap * g.w_tap_stride + icc * 32) * 2);
            };
            auto issue_piece = [&](int piece, int stage) {
                const unsigned dst = lds0 + stage * R::STAGE;
                if (piece < 2) {
                    const int k = piece;
                    const bool ok = ivalid && (unsigned)(ra_h[k] + sy) < (unsigned)H && (unsigned)(ra_w[k] + sx) < (unsigned)W;
                    glds16(rsrc_x, dst + (wave * 2 + k) * 1024, ok ? (unsigned)(a_base[k] + da) : BUF_OOB);
                } else {
                    const int k = piece - 2;
                    glds16(rsrc_w, dst + RG_ABUF + (wave * NBP + k) * 1024, ivalid ? b_base[k] + db : BUF_OOB);
                }
            };
            auto load_frags = [&](int stage, uint4 (&fb)[4], uint4 (&fa)[MB]) {
                const char* St = smem + stage * R::STAGE;
#pragma unroll
                for (int n = 0; n < 4; ++n) fb[n] = *reinterpret_cast<const uint4*>(St + rb_rd + n * 1024);
#pragma unroll
                for (int m = 0; m < MB; ++m) fa[m] = *reinterpret_cast<const uint4*>(St + ra_rd + m * 1024);
            };
            // one step: MB groups of 4 MFMAs on (cb, ca); between the groups the next step's fragments are read into
            // (nb, na) and the pieces of the step three ahead are issued
            auto do_step = [&](int st, uint4 (&cb)[4], uint4 (&ca)[MB], uint4 (&nb)[4], uint4 (&na)[MB]) {
                const char* Sn = smem + ((st + 1) & 3) * R::STAGE;
                const int sti = (st + 3) & 3;
                constexpr int G0 = MB >= 8 ? 2 : 1;      // first group that carries a piece
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    if (m < 2) {
                        nb[2 * m] = *reinterpret_cast<const uint4*>(Sn + rb_rd + (2 * m) * 1024);
                        nb[2 * m + 1] = *reinterpret_cast<const uint4*>(Sn + rb_rd + (2 * m + 1) * 1024);
                    }
                    na[m] = *reinterpret_cast<const uint4*>(Sn + ra_rd + m * 1024);
#pragma unroll
                    for (int q = 0; q < NPR; ++q)
                        if (G0 + q == m || (G0 + q >= MB && m == MB - 1)) issue_piece(q, sti);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int n = 0; n < 4; ++n) mma_frag<DT>(cb[n], ca[m], acc[m][n]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };

            // ---- prologue: steps 0..2 on their way, steps 0 and 1 complete, the fragments of step 0 in registers
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the previous segment's trailing zero fills
            __builtin_amdgcn_s_barrier();                         // ... and its last fragment reads
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                advance();
#pragma unroll
                for (int q = 0; q < NPR; ++q) issue_piece(q, j);
            }
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPR) : "memory");
            __builtin_amdgcn_s_barrier();
            load_frags(0, fb0, fa0);
            int st = 0;
            while (true) {
                advance();                                        // step st + 3
                do_step(st, fb0, fa0, fb1, fa1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPR) : "memory");      // my pieces of step st + 2 have landed ...
                __builtin_amdgcn_s_barrier();                                    // ... everybody's have; stage st is free
                if (++st == nsteps) break;
                advance();
                do_step(st, fb1, fa1, fb0, fa0);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPR) : "memory");
                __builtin_amdgcn_s_barrier();
                if (++st == nsteps) break;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // trailing zero fills must not land in the epilogue's LDS
        conv_wide_epilogue<DT, BN, MODE, FY, RELU>(p, acc, smem, b, th * p.tiles_w + tw, h0, w0, n0, bw_fs, bw_fss);
        return;
    }
    int buf = 0;
#ifdef CW_DIAG_NOLDSRD      // diagnostic build: fragments are read from LDS in the first two steps only - WRONG results
#define CW_RD(dst, src) do { if (diag_steps < 2) dst = src; } while (0)
#else
#define CW_RD(dst, src) dst = src
#endif
#if defined(CW_DIAG_NODMA) || defined(CW_DIAG_NOLDSRD) || defined(CW_DIAG_OOBDMA)
    int diag_steps = 0;      // steps issued so far (diagnostic builds only)
#endif
    uint4 bf0[4], af0[MB], bf1[4], af1[MB];      // K-half 0 / 1 fragments; a staggered wave keeps half 1 across the barrier
    // the two orders are two separate instances of the whole K loop (one branch per kernel, none per step)
    auto run = [&](auto stag_tag) {
    constexpr bool stag = decltype(stag_tag)::value;
    bool pend = false;                           // a staggered wave owes the second K half of its last step

    for (int s = 0; s < p.nseg; ++s) {
        const Seg g = p.seg[s];
        const int ncc = g.cin >> 6;
        // taps whose shifted tile touches the image
        unsigned live = 0;
        for (int i = 0; i < g.kh; ++i) {
            const int sy = g.sgn * (i - g.kh / 2) * g.dil;
            if (h0 + sy + TR - 1 < 0 || h0 + sy >= H) continue;
            for (int j = 0; j < g.kw; ++j) {
                const int sx = g.sgn * (j - g.kw / 2) * g.dil;
                if (w0 + sx + TC - 1 < 0 || w0 + sx >= W) continue;
                live |= 1u << (i * g.kw + j);
            }
        }
        live = __builtin_amdgcn_readfirstlane(live);
        if (live == 0u) continue;

        const long long xbytes = ((long long)(H * W - 1) * g.ld_x + g.cin) * 2;
        const u32x4s_t rsrc_x = make_rsrc(g.x + (long long)b * H * W * g.ld_x * 2, (unsigned)xbytes);
        const long long wbytes = ((long long)(g.kh * g.kw - 1) * g.w_tap_stride + (long long)n_out * g.cin) * 2;
        const u32x4s_t rsrc_w = make_rsrc(g.w, (unsigned)wbytes);
        int a_base[4];
        unsigned b_base[NB];
#pragma unroll
        for (int k = 0; k < 4; ++k) a_base[k] = (a_h[k] * W + a_w[k]) * g.ld_x * 2 + a_c16[k];
#pragma unroll
        for (int k = 0; k < NB; ++k) b_base[k] = b_row[k] < n_out ? (unsigned)(b_row[k] * g.cin * 2 + b_c16[k]) : BUF_OOB;

        // one step's loads: 4 A pieces + NB B pieces per wave.  `piece` 0..NP-1 so that the pipelined
        // loop can drop the pieces between MFMA groups instead of a burst after the barrier.
        int da = 0, sy = 0, sx = 0;
        unsigned db = 0;
#ifdef CW_DIAG_SKIPA      // diagnostic build: the A tile is fetched for the first tap of a kw row only - an upper bound of what
        int diag_j = 0;       // sharing one staged patch between the kw taps could save - WRONG results
#endif
        auto step_offsets = [&](int tap, int cc) {
            const int i = tap / g.kw, j = tap - i * g.kw;
#ifdef CW_DIAG_SKIPA
            diag_j = j;
#endif
            sy = g.sgn * (i - g.kh / 2) * g.dil;
            sx = g.sgn * (j - g.kw / 2) * g.dil;
            da = ((sy * W + sx) * g.ld_x + cc * 64) * 2;
            db = (unsigned)(((long long)tap * g.w_tap_stride + cc * 64) * 2);
        };
        auto issue_piece = [&](int piece, int bsel) {
#ifdef CW_DIAG_NODMA      // diagnostic build (tools/build_variant.sh): only the first two steps fetch their tiles - WRONG results
            if (diag_steps >= 2) return;
#endif
#ifdef CW_DIAG_OOBDMA      // diagnostic build: after the first two steps every piece is issued out of range - the LDS-DMA's issue and
            const bool oob = diag_steps >= 2;      // its LDS write (zeros) without the memory side; use with CW_DIAG_NOLDSRD - WRONG results
#else
            constexpr bool oob = false;
#endif
            if (piece < 4) {
#ifdef CW_DIAG_SKIPA
                if (diag_j != 0) return;
#endif
                const int k = piece;
                const bool ok = !oob && (unsigned)(a_h[k] + sy) < (unsigned)H && (unsigned)(a_w[k] + sx) < (unsigned)W;
                glds16(rsrc_x, lds0 + CW_A0 + bsel * CW_ABUF + wave * 4096 + k * 1024,
                       ok ? (unsigned)(a_base[k] + da) : BUF_OOB);
            } else {
                const int k = piece - 4;    // an out-of-range weight row stays out of range after + db
                glds16(rsrc_w, lds0 + CW_B0 + bsel * S::BBUF + wave * (NB * 1024) + k * 1024, oob ? BUF_OOB : b_base[k] + db);
            }
        };
        // first-half fragments; a staggered wave takes the upper half of its A fragments later (first_half, m == 0),
        // so that they are not live beside the previous step's second-half fragments (register budget)
        auto load_f0 = [&](int bsel, int m_end) {
            const char* At = smem + CW_A0 + bsel * CW_ABUF;
            const char* Bt = smem + CW_B0 + bsel * S::BBUF;
#pragma unroll
            for (int n = 0; n < 4; ++n) CW_RD(bf0[n], *reinterpret_cast<const uint4*>(Bt + b_rd[0] + n * 2048));
#pragma unroll
            for (int m = 0; m < MB; ++m)
                if (m < m_end) CW_RD(af0[m], *reinterpret_cast<const uint4*>(At + a_rd[0] + m * 2048));
        };
        // first K half: between the groups of 4 MFMAs the pieces of the next step's tiles are
        // sent on their way and the second half's fragments are fetched; the scheduling
        // barriers keep hipcc from gathering the loads into one burst
        auto first_half = [&](int bsel, auto prefetch) {
            const char* At = smem + CW_A0 + bsel * CW_ABUF;
            const char* Bt = smem + CW_B0 + bsel * S::BBUF;
#pragma unroll
            for (int m = 0; m < MB; ++m) {
#ifndef CW_ISSUE_MODE
#define CW_ISSUE_MODE 1      // A/B on the MI355X (tools/gpu_variants.sh): mode 1 is 0.5 % of a step ahead of 0 and 2
#endif
#pragma unroll
                for (int q = 0; q < NP; ++q) {
                    if (CW_ISSUE_MODE == 0 && (q * MB) / NP == m) prefetch(q);            // spread over the first K half
                    if (CW_ISSUE_MODE == 1 && (q * MB) / (2 * NP) == m) prefetch(q);      // first quarter, two per group
                    if (CW_ISSUE_MODE == 2 && m == 0) prefetch(q);                        // burst before the first MFMA
                }
                if (stag && m == 0 && MB > 1) {
#pragma unroll
                    for (int k = (MB + 1) / 2; k < MB; ++k) CW_RD(af0[k], *reinterpret_cast<const uint4*>(At + a_rd[0] + k * 2048));
                }
                if (!stag && MB > 2 && m + 2 < MB) CW_RD(af0[m + 2], *reinterpret_cast<const uint4*>(At + a_rd[0] + (m + 2) * 2048));
                if (m == MB / 2 - 1) {
#pragma unroll
                    for (int n = 0; n < 4; ++n) CW_RD(bf1[n], *reinterpret_cast<const uint4*>(Bt + b_rd[1] + n * 2048));
                }
                if (m >= MB / 2) {
                    CW_RD(af1[2 * (m - MB / 2)], *reinterpret_cast<const uint4*>(At + a_rd[1] + (2 * (m - MB / 2)) * 2048));
                    CW_RD(af1[2 * (m - MB / 2) + 1], *reinterpret_cast<const uint4*>(At + a_rd[1] + (2 * (m - MB / 2) + 1) * 2048));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < 4; ++n) mma_frag<DT>(bf0[n], af0[m], acc[m][n]);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        // second K half, from registers; the staggered waves issue the next step's pieces between its groups
        auto second_half = [&](auto prefetch) {
#pragma unroll
            for (int m = 0; m < MB; ++m) {
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    if ((q * MB) / (2 * NP) == m) prefetch(q);
#pragma unroll
                for (int n = 0; n < 4; ++n) mma_frag<DT>(bf1[n], af1[m], acc[m][n]);
            }
        };
        auto compute = [&](int bsel, auto prefetch) {
            // not staggered: only the first two A fragments up front, the others two groups ahead of their MFMAs
            // (first_half) - the burst of twelve ds_read_b128 a wave right behind the barrier becomes six (round 3:
            // SQ_LDS_CMD_FIFO_FULL is non-zero in this kernel and zero in wgrad5; +0.2 % of a step, A/B on one box)
            load_f0(bsel, stag ? (MB > 1 ? (MB + 1) / 2 : MB) : (MB > 2 ? 2 : MB));
            if constexpr (stag) {
                if (pend) {
                    __builtin_amdgcn_sched_barrier(0);
                    second_half(prefetch);
                    __builtin_amdgcn_sched_barrier(0);
                } else {
#pragma unroll
                    for (int q = 0; q < NP; ++q) prefetch(q);
                }
                first_half(bsel, [&](int) {});
                pend = true;
                // the reads of this step's second-half fragments must have left LDS before the barrier that lets
                // the next pieces overwrite the buffer (the MFMAs that wait for them come after it)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            } else {
                first_half(bsel, prefetch);
                second_half([&](int) {});
            }
        };

        // ---- software pipeline over the live (tap, channel chunk) steps of this segment
        unsigned rest = live;
        int tap = __builtin_ctz(rest);
        rest &= rest - 1;
        int cc = 0;
        __builtin_amdgcn_s_barrier();        // the previous segment's last reads of `buf` are done
        step_offsets(tap, cc);
#pragma unroll
        for (int q = 0; q < NP; ++q) issue_piece(q, buf);
        while (true) {
            // taps innermost: consecutive steps read the same 128-byte channel chunk of
            // overlapping pixels, so the A lines are still in L2 (with channels innermost the
            // reuse distance is 4 steps x ~1 MB per XCD - the whole L2)
            int ntap, nc = cc;
            bool more = true;
            if (rest != 0u) {
                ntap = __builtin_ctz(rest);
                rest &= rest - 1;
            } else {
                nc = cc + 1;
                more = nc < ncc;
                rest = live;
                ntap = __builtin_ctz(rest);
                rest &= rest - 1;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef CW_DIAG_NOBAR      // diagnostic build: the waves of a workgroup drift apart - WRONG results
            if (!more)
#endif
            __builtin_amdgcn_s_barrier();
            if (!more) break;
            step_offsets(ntap, nc);
            const int nb = buf ^ 1;
#if defined(CW_DIAG_NODMA) || defined(CW_DIAG_NOLDSRD) || defined(CW_DIAG_OOBDMA)
            ++diag_steps;
#endif
            compute(buf, [&](int q) { issue_piece(q, nb); });
            tap = ntap;
            cc = nc;
            buf = nb;
        }
        compute(buf, [&](int) {});
        buf ^= 1;
    }
    if (stag && pend) {
#pragma unroll
        for (int m = 0; m < MB; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) mma_frag<DT>(bf1[n], af1[m], acc[m][n]);
    }
    };
    run(std::false_type{});      // (std::true_type: the staggered schedule of round 2, measured 0.4 % slower, never instantiated)

    if (probe && tid == 0) {      // the K loop is over: cycles and ticks since the workgroup started (its epilogue is not in it)
        atomicAdd(p.clk_probe, __builtin_amdgcn_s_memtime() - ck0);
        atomicAdd(p.clk_probe + 1, __builtin_amdgcn_s_memrealtime() - rt0);
        atomicAdd(p.clk_probe + 2, 1ull);
    }
    conv_wide_epilogue<DT, BN, MODE, FY, RELU>(p, acc, smem, b, th * p.tiles_w + tw, h0, w0, n0, bw_fs, bw_fss);
}


#ifdef CW_WITH_PATCH      // variant builds only (tools/build_variant.sh patch -DCW_WITH_PATCH, CUTIE_CONV_PATCH=1): measured, not shipped
// ---- The A operand as a PATCH (round 3): conv_wide_kernel stages the 256 shifted pixels of every tap anew, although the
// taps of a 3 x 5 kernel at dilation 1 read the same 18 x 20 pixels fifteen times over; tools/gpu_diag_loop.sh prices the
// LDS-DMA at 13 % of a dense launch, 60 % of it on the issue / LDS-write side - fewer pieces is the one lever.  Here the
// patch of a 64-channel chunk (tile + halo, at most 360 pixels x 128 B = 45 KB) is staged ONCE per chunk, two buffers, and
// only the B tile (weights of one tap) still moves every step.  The patch's LDS image is pixel-major with a PADDED pixel
// stride of 144 B (nine 16-byte units, the ninth never read): sixteen consecutive pixels at one chunk then sit 36 banks
// apart - sixteen different 4-bank groups whatever the tap's shift - so there is no swizzle and a tap is a constant added
// to the read address, while eight neighbouring DMA lanes still fetch one pixel's 128 contiguous bytes (the ninth lane of
// each group fetches nothing).  A first version with a chunk-major image (no padding, the DMA gathering 64 cache lines an
// instruction) ran the decoder's 512 -> 64 layer in 214 us against 157.  Same wave layout, same step order (chunk, tap,
// K half), same epilogue as conv_wide_kernel: the same bits (the kernel, model and full-shape tests pass with it).
// Measured per launch inside real steps (bench.py --launch-table, one box): chunk-major form, 256 -> 256 3x5 232.7-237 us
// against 236 (no gain: 45 % fewer pieces, but a gathering piece costs several coalesced ones), whole step -0.75 %; padded
// form (does not fit the 256-channel tile's B ring beside two patches): 512 -> 64 3x5 200 us against 159 - the 8-wave
// 64-channel kernel needs 80 KB of LDS and runs two workgroups a CU, this one 118 KB and one -, 256 -> 128 3x5 130 us against
// 145 on conv_wide's 128 tile and 129 on conv_duo.  Nothing to ship: what the staging costs is not the number of A bytes.
constexpr int PT_NPX = 360;                     // patch pixels the image is laid out for
constexpr int PT_STRIDE = 144;                  // bytes a pixel: eight 16-byte chunks + one unit of padding
constexpr int PT_PIECES = (PT_NPX * PT_STRIDE + 1023) / 1024;      // 51 one-KB pieces
constexpr int PT_ABUF = PT_PIECES * 1024;       // 52224 B
constexpr int PT_ROUNDS = (PT_PIECES + 7) / 8;  // pieces per wave and chunk: 7
template <int BN> struct Patch {
    static constexpr int BBUF = BN * 128;
    static constexpr int B0 = 2 * PT_ABUF;
    static constexpr int RING = B0 + 2 * BBUF;
    static constexpr int EPI = Shape<BN>::TAB + 2 * BN * 4;
    static constexpr int LDS = RING > EPI ? RING : EPI;
    static constexpr bool FITS = LDS <= 160 * 1024;      // the 256-channel tile's B ring leaves no room for two patches
};

template <int DT, int BN, int MODE, bool FY, bool RELU>
__global__ __launch_bounds__(CW_NT) void conv_patch_kernel(ConvParams p) {
    typedef Shape<BN> S;
    typedef Patch<BN> PT;
    constexpr int MB = S::MB, NB = S::NB;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / S::WN, wn = wave % S::WN;
    const int li = lane & 15, lg = lane >> 4;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_void_t*)smem);
    const int H = p.H, W = p.W, TCB = p.TWB, n_out = p.n_out;
    const int TR = p.TH, TC = TCB * 16;

    int bid = blockIdx.x;
    const int nwg = gridDim.x;
    int nt, tw, th, b;
    if (p.perm_n) {
        const int SX = p.B >> 3;
        int l = bid >> 3;
        nt = l % p.n_tiles_n; l /= p.n_tiles_n;
        b = (bid & 7) * SX + l % SX;
        const int tile = p.perm[l / SX];
        th = tile / p.tiles_w;
        tw = tile - th * p.tiles_w;
    } else {
        if ((nwg & 7) == 0 && !p.rv_plain_order) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
        nt = bid % p.n_tiles_n; bid /= p.n_tiles_n;
        tw = bid % p.tiles_w; bid /= p.tiles_w;
        th = bid % p.tiles_h;
        b = bid / p.tiles_h;
    }
    const int h0 = th * TR, w0 = tw * TC, n0 = nt * BN;
    float bw_fs = 0.f, bw_fss = 0.f;
    if constexpr (MODE == 1 || MODE == 2) fold_bw_stats<BN>(p, b, n0, bw_fs, bw_fss);

    float4_t acc[MB][4];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = float4_t{0.f, 0.f, 0.f, 0.f};

    // B tile: staged and read exactly as in conv_wide_kernel (row-major, XOR swizzle)
    int b_row[NB], b_c16[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int r = wave * (NB * 8) + k * 8 + (lane >> 3);
        b_row[k] = n0 + r;
        b_c16[k] = ((lane & 7) ^ swz(r)) * 16;
    }
    int b_rd[2];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) b_rd[kk] = (wn * 64 + li) * 128 + (((kk * 4 + lg) ^ ((li >> 1) & 7)) * 16);

    int buf = 0, abuf = 0;
    uint4 bf0[4], af0[MB], bf1[4], af1[MB];
    for (int s = 0; s < p.nseg; ++s) {
        const Seg g = p.seg[s];
        const int ncc = g.cin >> 6;
        const int ntaps = g.kh * g.kw;
        // the patch: tile + halo (dilation 1, checked by the launcher), rows of PW pixels
        const int PW = TC + g.kw - 1, PH = TR + g.kh - 1;
        const int npx = PH * PW;
        const int ph0 = h0 - g.kh / 2, pw0 = w0 - g.kw / 2;

        const long long xbytes = ((long long)(H * W - 1) * g.ld_x + g.cin) * 2;
        const u32x4s_t rsrc_x = make_rsrc(g.x + (long long)b * H * W * g.ld_x * 2, (unsigned)xbytes);
        const long long wbytes = ((long long)(ntaps - 1) * g.w_tap_stride + (long long)n_out * g.cin) * 2;
        const u32x4s_t rsrc_w = make_rsrc(g.w, (unsigned)wbytes);
        // wave w moves pieces w, w + 8, ...: unit u = piece * 64 + lane is (patch pixel u / 9, chunk u % 9; chunk 8 = padding)
        unsigned a_off[PT_ROUNDS];
#pragma unroll
        for (int r = 0; r < PT_ROUNDS; ++r) {
            const int piece = wave + 8 * r;
            const int u = piece * 64 + lane;
            const int pp = u / 9, c = u - pp * 9;
            const int py = pp / PW, px = pp - py * PW;
            const int h = ph0 + py, w = pw0 + px;
            const bool ok = piece < PT_PIECES && c < 8 && pp < npx && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
            a_off[r] = ok ? (unsigned)((h * W + w) * g.ld_x * 2 + c * 16) : BUF_OOB;
        }
        unsigned b_base[NB];
#pragma unroll
        for (int k = 0; k < NB; ++k) b_base[k] = b_row[k] < n_out ? (unsigned)(b_row[k] * g.cin * 2 + b_c16[k]) : BUF_OOB;
        // fragment read address of (pixel block m, lane): the block's patch pixel + li at tap offset 0, chunk lg
        int a_m[MB];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int mb = wm * MB + m;
            a_m[m] = ((mb / TCB) * PW + (mb % TCB) * 16 + li) * PT_STRIDE + lg * 16;
        }

        unsigned db = 0;
        auto issue_b = [&](int k, int bsel) {
            glds16(rsrc_w, lds0 + PT::B0 + bsel * PT::BBUF + wave * (NB * 1024) + k * 1024, b_base[k] + db);
        };
        auto issue_a = [&](int r, int asel, int cc) {
            if (wave + 8 * r < PT_PIECES)      // uniform per wave
                glds16(rsrc_x, lds0 + asel * PT_ABUF + (wave + 8 * r) * 1024, a_off[r] + (unsigned)(cc * 128));
        };
        // one sub-step = one tap of one chunk: K halves 0 and 1 from the patch at the tap's offset and the tap's B tile
        auto compute = [&](int bsel, int asel, int tapoff, auto prefetch) {
            const char* At = smem + asel * PT_ABUF + tapoff * PT_STRIDE;
            const char* Bt = smem + PT::B0 + bsel * PT::BBUF;
            constexpr int KK1 = 64;      // K half 1: chunks 4..7
#pragma unroll
            for (int n = 0; n < 4; ++n) bf0[n] = *reinterpret_cast<const uint4*>(Bt + b_rd[0] + n * 2048);
#pragma unroll
            for (int m = 0; m < MB; ++m)
                if (m < 2) af0[m] = *reinterpret_cast<const uint4*>(At + a_m[m]);
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                prefetch(m);
                if (MB > 2 && m + 2 < MB) af0[m + 2] = *reinterpret_cast<const uint4*>(At + a_m[m + 2]);
                if (m == MB / 2 - 1) {
#pragma unroll
                    for (int n = 0; n < 4; ++n) bf1[n] = *reinterpret_cast<const uint4*>(Bt + b_rd[1] + n * 2048);
                }
                if (m >= MB / 2) {
                    af1[2 * (m - MB / 2)] = *reinterpret_cast<const uint4*>(At + a_m[2 * (m - MB / 2)] + KK1);
                    af1[2 * (m - MB / 2) + 1] = *reinterpret_cast<const uint4*>(At + a_m[2 * (m - MB / 2) + 1] + KK1);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < 4; ++n) mma_frag<DT>(bf0[n], af0[m], acc[m][n]);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) mma_frag<DT>(bf1[n], af1[m], acc[m][n]);
        };
        auto tap_offset = [&](int t) {      // patch pixels between the tile's origin and the tap's shifted origin
            const int i = t / g.kw, j = t - i * g.kw;
            return g.sgn > 0 ? i * PW + j : (g.kh - 1 - i) * PW + (g.kw - 1 - j);
        };

        __builtin_amdgcn_s_barrier();        // the previous segment's last reads of both rings are done
#pragma unroll
        for (int r = 0; r < PT_ROUNDS; ++r) issue_a(r, abuf, 0);
        db = 0;
#pragma unroll
        for (int k = 0; k < NB; ++k) issue_b(k, buf);
        int t = 0, cc = 0;
        while (true) {
            int ntp = t + 1, nc = cc;
            bool more = true;
            if (ntp == ntaps) {
                ntp = 0;
                nc = cc + 1;
                more = nc < ncc;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (!more) break;
            db = (unsigned)(((long long)ntp * g.w_tap_stride + nc * 64) * 2);
            const int nb = buf ^ 1, na = abuf ^ 1;
            const bool pa = cc + 1 < ncc;      // the next chunk's patch goes out under this chunk's taps, a piece every other tap or so
            const int tnow = t, ccn = cc + 1;
            compute(buf, abuf, tap_offset(t), [&](int m) {
#pragma unroll
                for (int q = 0; q < NB; ++q)
                    if ((q * MB) / (2 * NB) == m) issue_b(q, nb);
                if (m == (MB > 1 ? 1 : 0) && pa) {
#pragma unroll
                    for (int r = 0; r < PT_ROUNDS; ++r)
                        if ((r * ntaps) / PT_ROUNDS == tnow) issue_a(r, na, ccn);
                }
            });
            t = ntp;
            if (nc != cc) {
                cc = nc;
                abuf = na;
            }
            buf = nb;
        }
        compute(buf, abuf, tap_offset(t), [&](int) {});
        buf ^= 1;
        abuf ^= 1;
    }
    conv_wide_epilogue<DT, BN, MODE, FY, RELU>(p, acc, smem, b, th * p.tiles_w + tw, h0, w0, n0, bw_fs, bw_fss);
}

#endif      // CW_WITH_PATCH

// ---- Two workgroups a CU (round 3, the review's "overlap epilogues with main loops", form (i)): 256 lanes = 4 waves as
// 2 (pixel halves) x 2 (channel halves), tile = 256 pixels x 128 output channels, a wave owns the same 128 x 64 = 8 x 4
// accumulators as in conv_wide_kernel.  One K step = (tap, 32 input channels): A tile 256 rows x 64 B, B tile 128 rows x 64 B,
// 24 KB a stage, THREE stages (72 KB), so that two workgroups share a CU's 160 KB of LDS, its four SIMDs (one wave of each
// workgroup per SIMD, 256 registers each) and its matrix pipes: one workgroup's barrier, its waits and its epilogue run
// under the other one's MFMAs.  Stage s+2 is filled while stage s computes; one barrier per step orders both the landing
// of step s+1 and the reuse of the stage read at step s-1.  The steps run (64-channel chunk, tap, half) - the order in
// which conv_wide_kernel adds its K halves - and the epilogue is the same function with the same wave / lane partition
// of the tile, so both kernels produce the same bits.
constexpr int DUO_NW = 4;
constexpr int DUO_ABUF = 16384;
template <int BN> struct Duo {
    static_assert(BN == 128 || BN == 64, "channel tiles of the 4-wave form");
    static constexpr int BBUF = BN * 64;
    static constexpr int STAGE = DUO_ABUF + BBUF;
    static constexpr int RING = 3 * STAGE;
    static constexpr int EPI = Shape<BN, DUO_NW>::TAB + 2 * BN * 4;      // the epilogue's staged tile + its tables
    static constexpr int LDS = RING > EPI ? RING : EPI;
    static_assert(2 * LDS <= 160 * 1024, "two workgroups a CU");
};

// BN = 128: waves 2 x 2, a wave owns 128 pixels x 64 channels; BN = 64 (the decoder's 64-channel layers): waves 4 x 1, 64 x 64
template <int DT, int BN, int MODE, bool FY, bool RELU>
__global__ __launch_bounds__(DUO_NW * 64, 2) void conv_duo_kernel(ConvParams p) {
    typedef Shape<BN, DUO_NW> S;
    typedef Duo<BN> D;
    constexpr int MB = S::MB;                  // 8 / 4
    constexpr int NPA = 4, NPB = BN / 64, NP = NPA + NPB;      // 1-KB pieces per wave and step
    constexpr int DUO_STAGE = D::STAGE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / S::WN, wn = wave % S::WN;
    const int li = lane & 15, lg = lane >> 4;
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_void_t*)smem);
    const int H = p.H, W = p.W, TCB = p.TWB, n_out = p.n_out;
    const int TR = p.TH, TC = TCB * 16;

    int bid = blockIdx.x;
    const int nwg = gridDim.x;
    int nt, tw, th, b;
    if (p.perm_n) {
        const int SX = p.B >> 3;
        int l = bid >> 3;
        nt = l % p.n_tiles_n; l /= p.n_tiles_n;
        b = (bid & 7) * SX + l % SX;
        const int tile = p.perm[l / SX];
        th = tile / p.tiles_w;
        tw = tile - th * p.tiles_w;
    } else {
        if ((nwg & 7) == 0 && !p.rv_plain_order) bid = (bid & 7) * (nwg >> 3) + (bid >> 3);
        nt = bid % p.n_tiles_n; bid /= p.n_tiles_n;
        tw = bid % p.tiles_w; bid /= p.tiles_w;
        th = bid % p.tiles_h;
        b = bid / p.tiles_h;
    }
    const int h0 = th * TR, w0 = tw * TC, n0 = nt * BN;
    float bw_fs = 0.f, bw_fss = 0.f;
    if constexpr (MODE == 1 || MODE == 2) fold_bw_stats<BN>(p, b, n0, bw_fs, bw_fss);

    float4_t acc[MB][4];
#pragma unroll
    for (int m = 0; m < MB; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = float4_t{0.f, 0.f, 0.f, 0.f};

    // staging plan: a 1-KB piece is 16 rows x 64 B, lane l = (row l >> 2, 16-byte slot l & 3) fetching chunk slot ^ rswz(row);
    // wave w moves A pieces 4w..4w+3 and B pieces 2w, 2w+1
    int ra_h[NPA], ra_w[NPA], ra_c16[NPA], rb_row[NPB], rb_c16[NPB];
#pragma unroll
    for (int k = 0; k < NPA; ++k) {
        const int r = (wave * NPA + k) * 16 + (lane >> 2);      // tile pixel 0..255
        const int mb = r >> 4;
        ra_h[k] = h0 + mb / TCB;
        ra_w[k] = w0 + (mb % TCB) * 16 + (r & 15);
        ra_c16[k] = ((lane & 3) ^ rswz(r)) * 16;
    }
#pragma unroll
    for (int k = 0; k < NPB; ++k) {
        const int r = (wave * NPB + k) * 16 + (lane >> 2);      // weight row 0..127 of the channel tile
        rb_row[k] = n0 + r;
        rb_c16[k] = ((lane & 3) ^ rswz(r)) * 16;
    }
    const int rc16 = (lg ^ rswz(li)) * 16;
    const int ra_rd = (wm * (MB * 16) + li) * 64 + rc16;              // + m * 1024 per 16-pixel block
    const int rb_rd = DUO_ABUF + (wn * 64 + li) * 64 + rc16;          // + n * 1024 per 16-channel block

    int stage = 0;       // stage of the step in hand; the ring position survives from segment to segment
    for (int s = 0; s < p.nseg; ++s) {
        const Seg g = p.seg[s];
        const int ncc = g.cin >> 6;
        unsigned live = 0;
        for (int i = 0; i < g.kh; ++i) {
            const int sy = g.sgn * (i - g.kh / 2) * g.dil;
            if (h0 + sy + TR - 1 < 0 || h0 + sy >= H) continue;
            for (int j = 0; j < g.kw; ++j) {
                const int sx = g.sgn * (j - g.kw / 2) * g.dil;
                if (w0 + sx + TC - 1 < 0 || w0 + sx >= W) continue;
                live |= 1u << (i * g.kw + j);
            }
        }
        live = __builtin_amdgcn_readfirstlane(live);
        if (live == 0u) continue;
        const int nsteps = __builtin_popcount(live) * ncc * 2;

        const long long xbytes = ((long long)(H * W - 1) * g.ld_x + g.cin) * 2;
        const u32x4s_t rsrc_x = make_rsrc(g.x + (long long)b * H * W * g.ld_x * 2, (unsigned)xbytes);
        const long long wbytes = ((long long)(g.kh * g.kw - 1) * g.w_tap_stride + (long long)n_out * g.cin) * 2;
        const u32x4s_t rsrc_w = make_rsrc(g.w, (unsigned)wbytes);
        int a_base[NPA];
        unsigned b_base[NPB];
#pragma unroll
        for (int k = 0; k < NPA; ++k) a_base[k] = (ra_h[k] * W + ra_w[k]) * g.ld_x * 2 + ra_c16[k];
#pragma unroll
        for (int k = 0; k < NPB; ++k) b_base[k] = rb_row[k] < n_out ? (unsigned)(rb_row[k] * g.cin * 2 + rb_c16[k]) : BUF_OOB;

        // the step whose pieces go out next: K half innermost (the two halves of a 128-byte channel chunk), then the taps
        // (overlapping pixels of one chunk: L2), then the next 64 channels; beyond the last step the pieces are zero fills
        unsigned irest = live;
        int icc = 0, ihalf = 1, issued = 0, da = 0, sy = 0, sx = 0;
        unsigned db = 0;
        bool ivalid = false;
        auto advance = [&]() {
            ivalid = issued < nsteps;
            ++issued;
            if (!ivalid) return;
            if (ihalf == 0) {
                ihalf = 1;
                da += 64;
                db += 64u;
                return;
            }
            ihalf = 0;
            if (irest == 0u) { irest = live; ++icc; }
            const int tap = __builtin_ctz(irest);
            irest &= irest - 1;
            const int i = tap / g.kw, j = tap - i * g.kw;
            sy = g.sgn * (i - g.kh / 2) * g.dil;
            sx = g.sgn * (j - g.kw / 2) * g.dil;
            da = ((sy * W + sx) * g.ld_x + icc * 64) * 2;
            db = (unsigned)(((long long)tap * g.w_tap_stride + icc * 64) * 2);
        };
        auto issue_piece = [&](int piece, int stg) {
            const unsigned dst = lds0 + stg * DUO_STAGE;
            if (piece < NPA) {
                const int k = piece;
                const bool ok = ivalid && (unsigned)(ra_h[k] + sy) < (unsigned)H && (unsigned)(ra_w[k] + sx) < (unsigned)W;
                glds16(rsrc_x, dst + (wave * NPA + k) * 1024, ok ? (unsigned)(a_base[k] + da) : BUF_OOB);
            } else {
                const int k = piece - NPA;
                glds16(rsrc_w, dst + DUO_ABUF + (wave * NPB + k) * 1024, ivalid ? b_base[k] + db : BUF_OOB);
            }
        };

        // ---- prologue: the previous segment's trailing zero fills and last fragment reads are over, steps 0 and 1 go out
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int s1 = stage == 2 ? 0 : stage + 1;
        advance();
#pragma unroll
        for (int q = 0; q < NP; ++q) issue_piece(q, stage);
        advance();
#pragma unroll
        for (int q = 0; q < NP; ++q) issue_piece(q, s1);
        for (int st = 0; st < nsteps; ++st) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP) : "memory");      // my pieces of this step have landed ...
            __builtin_amdgcn_s_barrier();                                   // ... everybody's have; the stage of step st - 1 is free
            const int s2 = s1 == 2 ? 0 : s1 + 1;
            advance();                                                      // step st + 2 goes into it
            const char* St = smem + stage * DUO_STAGE;
            uint4 fb[4], fa[MB];
#pragma unroll
            for (int n = 0; n < 4; ++n) fb[n] = *reinterpret_cast<const uint4*>(St + rb_rd + n * 1024);
            fa[0] = *reinterpret_cast<const uint4*>(St + ra_rd);
            fa[1] = *reinterpret_cast<const uint4*>(St + ra_rd + 1024);
#pragma unroll
            for (int m = 0; m < MB; ++m) {
                if (m + 2 < MB) fa[m + 2] = *reinterpret_cast<const uint4*>(St + ra_rd + (m + 2) * 1024);
#pragma unroll
                for (int q = 0; q < NP; ++q)
                    if ((q * MB) / NP == m) issue_piece(q, s2);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int n = 0; n < 4; ++n) mma_frag<DT>(fb[n], fa[m], acc[m][n]);
                __builtin_amdgcn_sched_barrier(0);
            }
            stage = s1;
            s1 = s2;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // trailing zero fills must not land in the epilogue's LDS
    conv_wide_epilogue<DT, BN, MODE, FY, RELU, DUO_NW>(p, acc, smem, b, th * p.tiles_w + tw, h0, w0, n0, bw_fs, bw_fss);
}


}  // namespace

// conv_patch_kernel takes a launch when every segment is a multi-tap kernel at dilation 1 whose patch (tile + halo) fits the
// 360-pixel image: 3 x 5 on 16 x 16 or 8 x 32 tiles
bool conv_patch_ok(const ConvParams& p) {
#ifndef CW_WITH_PATCH
    return false;
#else
    const int TC = p.TWB * 16;
    for (int s = 0; s < p.nseg; ++s) {
        const Seg& g = p.seg[s];
        if (g.dil != 1 || g.kh * g.kw < 2) return false;
        if ((p.TH + g.kh - 1) * (TC + g.kw - 1) > PT_NPX) return false;
    }
    return true;
#endif
}

// channel tile for n_out: 256 where it divides, else 128, else 64 (also for n_out < 64 in whole 16-byte pieces);
// 0 = not a wide-kernel shape
int conv_wide_tile(const ConvParams& p) {
    if ((p.ld_out * 2) % 16 != 0 || !aligned16(p.out) || p.relu == 2) return 0;    // sigmoid: generic kernel only
    for (int s = 0; s < p.nseg; ++s)
        if (p.seg[s].cin % 64 != 0 || p.seg[s].kh * p.seg[s].kw > 32) return 0;
    if (p.n_out % 256 == 0) return 256;
    if (p.n_out % 128 == 0) return 128;
    if (p.n_out % 64 == 0) return 64;
    // a narrow layer (CUTIE2's 32-channel image branch): one 64-channel tile, the rows and columns beyond n_out are
    // zero weight rows and masked stores - half the tile idles, where the general kernel pads to 128 channels
    if (p.n_out < 64 && p.n_out % 8 == 0) return 64;
    return 0;
}

template <int DT, int BN, int MODE, bool FY = false, bool RELU = false>
static int launch_mode(const ConvParams& p, hipStream_t stream) {
    const long long grid = (long long)p.B * p.tiles_h * p.tiles_w * p.n_tiles_n;
#ifndef CW_WITH_PATCH
    if constexpr (BN > 1000) {
        set_error("conv_patch: not in this build (-DCW_WITH_PATCH)");
        return CUTIE_E_ARG;
    } else
#else
    if constexpr (BN > 1000 && !Patch<(BN > 1000 ? BN - 1000 : 64)>::FITS) {
        set_error("conv_patch: no patch form of the %d-channel tile", BN - 1000);
        return CUTIE_E_ARG;
    } else if constexpr (BN > 1000) {      // the A operand staged as a patch, once per 64-channel chunk (conv_patch_kernel)
        constexpr int PBN = BN - 1000;
        CUTIE_ENSURE_LDS((conv_patch_kernel<DT, PBN, MODE, FY, RELU>), Patch<PBN>::LDS);
        hipLaunchKernelGGL((conv_patch_kernel<DT, PBN, MODE, FY, RELU>), dim3((unsigned)grid), dim3(CW_NT), Patch<PBN>::LDS, stream, p);
        CUTIE_CHECK_LAUNCH();
        return CUTIE_OK;
    } else
#endif
    if constexpr (BN < 0) {      // the two-workgroups-a-CU form of the 128- and 64-channel tiles
        CUTIE_ENSURE_LDS((conv_duo_kernel<DT, -BN, MODE, FY, RELU>), Duo<-BN>::LDS);
        hipLaunchKernelGGL((conv_duo_kernel<DT, -BN, MODE, FY, RELU>), dim3((unsigned)grid), dim3(DUO_NW * 64), Duo<-BN>::LDS, stream, p);
        CUTIE_CHECK_LAUNCH();
        return CUTIE_OK;
    } else {
#ifdef CW_WITH_RING      // variant builds only (tools/build_variant.sh ring -DCW_WITH_RING): measured 3.5 % SLOWER
    if constexpr (BN >= 128) {
        if (config().conv_ring) {      // CUTIE_CONV_RING=1 selects it
            CUTIE_ENSURE_LDS((conv_wide_kernel<DT, BN, MODE, FY, RELU, true>), Shape<BN>::LDS);
            hipLaunchKernelGGL((conv_wide_kernel<DT, BN, MODE, FY, RELU, true>), dim3((unsigned)grid), dim3(CW_NT), Shape<BN>::LDS, stream, p);
            CUTIE_CHECK_LAUNCH();
            return CUTIE_OK;
        }
    }
#endif
    CUTIE_ENSURE_LDS((conv_wide_kernel<DT, BN, MODE, FY, RELU>), Shape<BN>::LDS);
    hipLaunchKernelGGL((conv_wide_kernel<DT, BN, MODE, FY, RELU>), dim3((unsigned)grid), dim3(CW_NT), Shape<BN>::LDS, stream, p);
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
    }
}

template <int DT, int BN>
static int launch_one(const ConvParams& p, hipStream_t stream) {
    if (p.bw_a != nullptr) {
        if (p.bias || p.relu || p.stats) {
            set_error("conv_wide: a data gradient with fused norm sums takes no bias, activation or moments of its own");
            return CUTIE_E_ARG;
        }
        const bool fy = (p.bw_act & CUTIE_ACT_FROM_Y) != 0;
        if (p.rv_slots == nullptr) return fy ? launch_mode<DT, BN, 1, true>(p, stream) : launch_mode<DT, BN, 1, false>(p, stream);
        if (p.rv_relu == 1)
            return fy ? launch_mode<DT, BN, 2, true, true>(p, stream) : launch_mode<DT, BN, 2, false, true>(p, stream);
        return fy ? launch_mode<DT, BN, 2, true, false>(p, stream) : launch_mode<DT, BN, 2, false, false>(p, stream);
    }
    if (p.fw_slots != nullptr && p.stats != nullptr) return launch_mode<DT, BN, 3>(p, stream);
    return launch_mode<DT, BN, 0>(p, stream);
}

// K steps of every pixel tile (live taps x channel chunks); returns false when there are more tiles than CONV_MAX_PERM
static bool tile_weights(const ConvParams& p, int* weight, int& wmin, int& wmax) {
    const int nt = p.tiles_h * p.tiles_w;
    if (nt > CONV_MAX_PERM) return false;
    const int TR = p.TH, TC = p.TWB * 16;
    wmin = 1 << 30;
    wmax = 0;
    for (int t = 0; t < nt; ++t) {
        const int h0 = (t / p.tiles_w) * TR, w0 = (t % p.tiles_w) * TC;
        int w = 0;
        for (int s = 0; s < p.nseg; ++s) {
            const Seg& g = p.seg[s];
            int live = 0;
            for (int i = 0; i < g.kh; ++i) {
                const int sy = g.sgn * (i - g.kh / 2) * g.dil;
                if (h0 + sy + TR - 1 < 0 || h0 + sy >= p.H) continue;
                for (int j = 0; j < g.kw; ++j) {
                    const int sx = g.sgn * (j - g.kw / 2) * g.dil;
                    if (w0 + sx + TC - 1 < 0 || w0 + sx >= p.W) continue;
                    ++live;
                }
            }
            w += live * (g.cin >> 6);
        }
        weight[t] = w;
        if (w < wmin) wmin = w;
        if (w > wmax) wmax = w;
    }
    return true;
}

// A rendezvous of a sample's tiles is deadlock-free when all of a sample's workgroups can be resident together: one
// 8-wave workgroup per CU, dispatched in block order, so at most 32 a sample where the block order keeps a sample on
// one XCD and at most 128 where it spreads them (plain_order); a bounded wait in the kernel backs this up.  It pays
// when what the tiles wait for each other (every round of workgroups lasts as long as its heaviest tile, and the
// heaviest-first order is given up) costs less than the pass over dy, a and dz it replaces (6 bytes per element at
// ~5 TB/s); a K step of the 256 tile takes ~1.9 us of a tile's wall time (tools/dgrad_bench.py: 230 us for two rounds of 60 steps).
bool conv_wide_rendezvous_ok(const ConvParams& p, int bn, int* plain_order) {
    if (bn > 1000) bn -= 1000;      // conv_patch_kernel: the same tiles, one workgroup a CU
    int weight[CONV_MAX_PERM];
    int wmin, wmax;
    if (!tile_weights(p, weight, wmin, wmax)) return false;
    const int nt = p.tiles_h * p.tiles_w;
    const int per_sample = nt * p.n_tiles_n;
    // a sample's workgroups must all be resident at once, one per CU (two in the 4-wave form, bn < 0), with room for a
    // second sample beside it
    const int per_cu = bn < 0 ? 2 : 1;
    if (per_sample > 128 || 2 * per_sample > per_cu * device_cus()) return false;
    double wavg = 0.0;
    for (int t = 0; t < nt; ++t) wavg += weight[t];
    wavg /= nt;
    const int cus = per_cu * (device_cus() > 0 ? device_cus() : 256);
    const double rounds = (double)((long long)p.B * per_sample + cus - 1) / (double)cus;
    const double step_us = 1.9 * (bn < 0 ? -2 * bn : bn) / 256.0;      // (4-wave form: two tiles share a CU's matrix pipes)
    // every channel lane sweeps the granules of all nt tiles, sixteen tiles in flight at a time (~2 us a batch): the first
    // batch is part of the protocol's measured cost below, the others are not (CUTIE2-dilate 256x128, 128 tiles a sample:
    // 363 docs/s with the tiles meeting against 381 without, tools/cutie2_bench.py)
    const double sweep_us = (rounds < 1.0 ? 1.0 : rounds) * ((nt + 15) / 16 - 1) * 2.0;
    const double cost_us = (rounds < 1.0 ? 1.0 : rounds) * (wmax - wavg) * step_us + sweep_us;
    const double saved_us = 6.0 * p.B * p.H * p.W * p.n_out / 5e6;
    if (cost_us > 0.5 * saved_us) return false;      // the launch's own protocol takes the other half (tools/dgrad_bench.py)
    *plain_order = per_sample > 32 ? 1 : 0;
    return true;
}

// The forward form (y stored instead of a, cutie_conv_in_fwd): the same residency rule; taken only where the tiles
// are balanced to 6 % - a rule that does NOT depend on the batch size, because the fused and the two-call form round y
// differently (from the f32 activation / from the stored 16-bit one) and a sample's output must not depend on how
// many other samples share its batch.
bool conv_wide_fw_rendezvous_ok(const ConvParams& p, int bn, int* plain_order) {
    if (bn > 1000) bn -= 1000;
    int weight[CONV_MAX_PERM];
    int wmin, wmax;
    if (!tile_weights(p, weight, wmin, wmax)) return false;
    const int nt = p.tiles_h * p.tiles_w;
    const int per_sample = nt * p.n_tiles_n;
    if (per_sample > 128 || nt > 32 || 2 * per_sample > (bn < 0 ? 2 : 1) * device_cus()) return false;      // (nt: more than two batches of granules to sweep cost more than the pass saves)
    double wavg = 0.0;
    for (int t = 0; t < nt; ++t) wavg += weight[t];
    wavg /= nt;
    if (wmax - wavg > 0.06 * wavg) return false;
    *plain_order = per_sample > 32 ? 1 : 0;
    return true;
}

// Heaviest-first tile order (ConvParams::perm): the K steps of a tile are its live taps x channel chunks, which on
// a dilated layer differ up to 2x between the centre and the corners of the image.  Dispatch is in block order, so
// putting the heavy tiles first lets the light ones even out the end of the launch.
static void plan_tile_order(ConvParams& p) {
    const bool on = config().conv_lpt;
    p.perm_n = 0;
    const int nt = p.tiles_h * p.tiles_w;
    if (p.rv_slots || p.fw_slots) return;      // a rendezvous wants a sample's tiles side by side in dispatch order
    if (!on || nt < 2 || nt > CONV_MAX_PERM || (p.B & 7) != 0) return;
    int weight[CONV_MAX_PERM];
    int wmin, wmax;
    tile_weights(p, weight, wmin, wmax);
    if (wmin == wmax) return;
    for (int t = 0; t < nt; ++t) p.perm[t] = (unsigned char)t;
    for (int i = 1; i < nt; ++i) {          // stable insertion sort, heaviest first
        const unsigned char v = p.perm[i];
        int j = i - 1;
        while (j >= 0 && weight[p.perm[j]] < weight[v]) { p.perm[j + 1] = p.perm[j]; --j; }
        p.perm[j + 1] = v;
    }
    p.perm_n = nt;
}

// p.TH / p.TWB / tiles_* must describe 256-pixel tiles, p.n_tiles_n = n_out / bn
int launch_conv_wide(const ConvParams& p0, int dtype, int bn, hipStream_t stream) {
    if (dtype != CUTIE_BF16 && dtype != CUTIE_F16) {
        set_error("conv_wide: dtype %d", dtype);
        return CUTIE_E_ARG;
    }
    ConvParams p = p0;
    p.clk_probe = conv_clock_probe();
    plan_tile_order(p);
    const bool bf = dtype == CUTIE_BF16;
    switch (bn) {
        case 1256: return bf ? launch_one<CUTIE_BF16, 1256>(p, stream) : launch_one<CUTIE_F16, 1256>(p, stream);
        case 1128: return bf ? launch_one<CUTIE_BF16, 1128>(p, stream) : launch_one<CUTIE_F16, 1128>(p, stream);
        case 1064: return bf ? launch_one<CUTIE_BF16, 1064>(p, stream) : launch_one<CUTIE_F16, 1064>(p, stream);
        case -128: return bf ? launch_one<CUTIE_BF16, -128>(p, stream) : launch_one<CUTIE_F16, -128>(p, stream);
        case -64: return bf ? launch_one<CUTIE_BF16, -64>(p, stream) : launch_one<CUTIE_F16, -64>(p, stream);
        case 256: return bf ? launch_one<CUTIE_BF16, 256>(p, stream) : launch_one<CUTIE_F16, 256>(p, stream);
        case 128: return bf ? launch_one<CUTIE_BF16, 128>(p, stream) : launch_one<CUTIE_F16, 128>(p, stream);
        case 64: return bf ? launch_one<CUTIE_BF16, 64>(p, stream) : launch_one<CUTIE_F16, 64>(p, stream);
    }
    set_error("conv_wide: channel tile %d", bn);
    return CUTIE_E_ARG;
}

}  // namespace cutie
