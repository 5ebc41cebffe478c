// K2-K5: stride-1 SAME (dilated) convolution as an implicit GEMM on the gfx950 matrix cores.
//
// Replaces tf.nn.conv2d / tf.nn.atrous_conv2d (+ bias_add + relu) of
// /root/reference/model_framework.py:168-204,414-422 and the two gradients tf.gradients
// derives for them (data gradient = the same kernel with negated tap shifts and [tap][cin][cout]
// weights; weight gradient = conv_wgrad below).
//
// Forward / data-gradient kernel (conv_kernel)
//   GEMM view: M = output pixels, N = output channels, K = taps x input channels.
//   Workgroup = 256 lanes = 4 waves (2 x 2); tile = 128 pixels x 128 channels; a wave owns
//   64 x 64 = 4 x 4 MFMA 16x16 accumulators.  The 128 pixels are TH rows x (TWB*16) columns
//   of ONE sample, so a whole input row patch (with its +-2*dil column halo, zero filled
//   outside the image) is staged in LDS once per (tap row, 128-byte channel chunk) and reused
//   by all kw column taps: the A operand costs 1/kw of a plain GEMM's staging traffic.
//   16-pixel MFMA row blocks whose shifted input lies wholly outside the image are skipped
//   (wave-uniform), which is where the in-bounds FLOP count of SURVEY.md 8(d) comes from.
//   LDS images hold 128 B per pixel / per weight row; the 16-byte chunk index is XOR-swizzled
//   with (row >> 1) & 7 so that ds_read_b128 fragment reads spread over the 64 banks.
//   bf16: v_mfma_f32_16x16x32_bf16;  f32 (parity mode): 4 x v_mfma_f32_16x16x4_f32 on the same
//   16-byte fragments (exact f32 FMA chains).
//   Epilogue: + bias, ReLU, per-(sample, channel) sum / sum-of-squares for the instance norm
//   (wave shuffle reduce + one f32 atomic per wave and channel), tile staged through LDS and
//   written as whole 16-byte row pieces into the (possibly wider) destination buffer.
//
// Weight-gradient kernel (wgrad_kernel)
//   GEMM view per tap: M = cin, N = cout, K = pixels.  Both operands are pixel-major in HBM,
//   i.e. K-major, so the bf16 fragments come from ds_read_b64_tr_b16 (hardware transposed
//   read) out of row-major [pixel][channel] LDS images; f32 uses plain ds_read_b32.
//   K is split over row groups into f32 slabs, summed by wgrad_reduce_kernel in a fixed order
//   (deterministic, no float atomics).
#include "conv_common.h"
#include <stdlib.h>

#include <mutex>
#include <vector>

namespace {
using namespace cutie;
using namespace cutie::convk;

constexpr int CONV_NT = 256;     // lanes per workgroup: 4 waves as 2 (pixel halves) x 2 (channel halves)
constexpr int CONV_MB = 8;       // 16-pixel blocks per wave: a wave owns 128 pixels x 64 channels
constexpr int CONV_NA = 6;       // 16-byte chunks of the A patch each lane stages per batch

// columns of the staged patch: image columns [w0 - padl, w0 + TW + padr); a live 16-pixel block
// never reads left of column -15 or right of W + 14, so the halo is clipped to that
__device__ __forceinline__ void conv_geom(int kw, int dil, int w0, int TW, int W, int& padl, int& pw) {
    const int pad = (kw / 2) * dil;
    padl = min(pad, w0 + 15);
    const int padr = max(0, min(pad, W + 15 - (w0 + TW)));
    pw = padl + TW + padr;
}

// Workgroup tile = 256 pixels (TH rows x TWB*16 columns of one sample) x 128 output channels.
// Per (segment, tap row, 128-byte channel chunk) the input row patch is staged once and reused by
// the kw column taps; the weight tile of tap j+1 is fetched into registers while tap j computes
// and lands in the other half of a double-buffered LDS tile (one barrier per tap).
template <int DT, int ODT>
__global__ __launch_bounds__(CONV_NT, 2) void conv_kernel(ConvParams p) {
    constexpr int ES = Traits<DT>::ES;
    constexpr int BKE = Traits<DT>::BKE;
    typedef typename elem<ODT>::type TO;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                    // [TH][PW][8 chunks] x 16 B, chunk index swizzled
    char* Bs = smem + p.a_bytes;        // 2 x [128 rows][8 chunks] x 16 B

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    const int H = p.H, W = p.W, TH = p.TH, TWB = p.TWB, n_out = p.n_out;

    // workgroup -> (sample, tile row, tile col, channel tile); channel tile fastest
    int bid = blockIdx.x;
    const int nt = bid % p.n_tiles_n; bid /= p.n_tiles_n;
    const int tw = bid % p.tiles_w; bid /= p.tiles_w;
    const int th = bid % p.tiles_h;
    const int b = bid / p.tiles_h;
    const int TW = TWB * 16;
    const int h0 = th * TH, w0 = tw * TW, n0 = nt * 128;

    float4_t acc[CONV_MB][4];
#pragma unroll
    for (int m = 0; m < CONV_MB; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = float4_t{0.f, 0.f, 0.f, 0.f};

    // weight-tile staging plan: constant over the whole kernel
    int b_lds[4], b_row[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int idx = u * CONV_NT + tid;
        const int c = idx & 7, n = idx >> 3;
        b_lds[u] = (n * 8 + (c ^ swz(n))) * 16;
        b_row[u] = (n0 + n < n_out) ? (n0 + n) : -1;
    }
    int buf = 0;

    for (int s = 0; s < p.nseg; ++s) {
        const Seg g = p.seg[s];
        int padl, pw;
        conv_geom(g.kw, g.dil, w0, TW, W, padl, pw);
        const int n_chunks_a = TH * pw * 8;
        const int ncc = g.cin / BKE;
        // tap rows whose shifted tile touches the image form a contiguous range [i_lo, i_hi]
        int i_lo = g.kh, i_hi = -1;
        for (int i = 0; i < g.kh; ++i) {
            const int sy = g.sgn * (i - g.kh / 2) * g.dil;
            if (!(h0 + sy + TH - 1 < 0 || h0 + sy >= H)) { i_lo = min(i_lo, i); i_hi = max(i_hi, i); }
        }
        if (i_hi < i_lo) continue;
        // Loads are buffer loads: a chunk outside the image (or a weight row beyond n_out) gets an
        // offset beyond num_records and comes back as zeros - no branch around any load.
        const long long xbytes = ((long long)(H * W - 1) * g.ld_x + g.cin) * ES;
        const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
            (void*)(g.x + (long long)b * H * W * g.ld_x * ES), 0, (int)xbytes, 0x00020000);
        const long long wbytes = ((long long)(g.kh * g.kw - 1) * g.w_tap_stride + (long long)n_out * g.cin) * ES;
        const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)g.w, 0, (int)wbytes, 0x00020000);
        int b_voff[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) b_voff[u] = b_row[u] >= 0 ? (b_row[u] * g.cin * ES + ((u * CONV_NT + tid) & 7) * 16) : -1;

        uint4 regB[4];
        auto load_b = [&](int i, int cc, int j) {
            const int step_off = (int)(((long long)(i * g.kw + j) * g.w_tap_stride + cc * BKE) * ES);
#pragma unroll
            for (int u = 0; u < 4; ++u)
                regB[u] = buf_load16(rsrc_w, b_voff[u] >= 0 ? (unsigned)(b_voff[u] + step_off) : BUF_OOB);
        };
        auto store_b = [&](int bsel) {
#pragma unroll
            for (int u = 0; u < 4; ++u) *reinterpret_cast<uint4*>(Bs + bsel * (128 * 128) + b_lds[u]) = regB[u];
        };
        auto stage_a = [&](int i, int cc) {     // TH rows x PW columns x 128 B, loads batched CONV_NA deep
            const int sy = g.sgn * (i - g.kh / 2) * g.dil;
            for (int base = 0; base < n_chunks_a; base += CONV_NA * CONV_NT) {
                uint4 v[CONV_NA];
                int dst[CONV_NA];
#pragma unroll
                for (int u = 0; u < CONV_NA; ++u) {
                    const int idx = base + u * CONV_NT + tid;
                    const int c = idx & 7, pix = idx >> 3;
                    const int r = pix / pw, pc = pix - r * pw;
                    const int hin = h0 + r + sy, win = w0 + pc - padl;
                    const bool ok = idx < n_chunks_a && hin >= 0 && hin < H && win >= 0 && win < W;
                    dst[u] = idx < n_chunks_a ? (pix * 8 + (c ^ swz(pc))) * 16 : -1;
                    v[u] = buf_load16(rsrc_x, ok ? (unsigned)(((hin * W + win) * g.ld_x + cc * BKE) * ES + c * 16) : BUF_OOB);
                }
#pragma unroll
                for (int u = 0; u < CONV_NA; ++u)
                    if (dst[u] >= 0) *reinterpret_cast<uint4*>(As + dst[u]) = v[u];
            }
        };

        for (int i = i_lo; i <= i_hi; ++i) {
            const int sy = g.sgn * (i - g.kh / 2) * g.dil;
            for (int cc = 0; cc < ncc; ++cc) {
                // ---- group start: patch + first weight tile, synchronously
                __syncthreads();                   // previous group is done with As and Bs[buf]
                load_b(i, cc, 0);
                stage_a(i, cc);
                store_b(buf);
                __syncthreads();
                for (int j = 0; j < g.kw; ++j) {
                    const int sx = g.sgn * (j - g.kw / 2) * g.dil;
                    const bool more = j + 1 < g.kw;
                    if (more) load_b(i, cc, j + 1);          // in flight while this tap computes
                    // ---- 2 x (4 + 8 fragment reads, up to 32 MFMA)
                    unsigned live = 0;
#pragma unroll
                    for (int m = 0; m < CONV_MB; ++m) {
                        const int mb = wm * CONV_MB + m;
                        const int h = h0 + mb / TWB, wc = w0 + (mb % TWB) * 16;
                        const bool l = h < H && wc < W && h + sy >= 0 && h + sy < H && wc + sx + 15 >= 0 && wc + sx < W;
                        live |= (l ? 1u : 0u) << m;
                    }
                    live = __builtin_amdgcn_readfirstlane(live);
                    const char* Bc = Bs + buf * (128 * 128);
                    if (live != 0u) {
                        // ONE code path (two paths made hipcc keep two accumulator sets): fragment
                        // reads are unconditional (a dead block reads pixel 0 of the patch), only
                        // the MFMAs of a dead 16-pixel block are skipped, by a scalar branch.
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
                            uint4 bf[4];
#pragma unroll
                            for (int n = 0; n < 4; ++n) {
                                const int nl = wn * 64 + n * 16 + li;
                                bf[n] = *reinterpret_cast<const uint4*>(Bc + (nl * 8 + ((kk * 4 + lg) ^ swz(nl))) * 16);
                            }
#pragma unroll
                            for (int mh = 0; mh < 2; ++mh) {
                                uint4 af[4];
#pragma unroll
                                for (int mq = 0; mq < 4; ++mq) {
                                    const int m = mh * 4 + mq;
                                    const int mb = wm * CONV_MB + m;
                                    const bool l = (live >> m) & 1u;
                                    const int pc = l ? padl + (mb % TWB) * 16 + sx + li : 0;
                                    const int pix = l ? (mb / TWB) * pw + pc : 0;
                                    af[mq] = *reinterpret_cast<const uint4*>(As + (pix * 8 + ((kk * 4 + lg) ^ swz(pc))) * 16);
                                }
#pragma unroll
                                for (int mq = 0; mq < 4; ++mq) {
                                    const int m = mh * 4 + mq;
                                    if ((live >> m) & 1u) {
#pragma unroll
                                        for (int n = 0; n < 4; ++n) mma_frag<DT>(af[mq], bf[n], acc[m][n]);
                                    }
                                }
                            }
                        }
                    }
                    if (more) {
                        store_b(buf ^ 1);          // that half was last read one barrier ago
                        __syncthreads();
                        buf ^= 1;
                    }
                }
            }
        }
    }

    // ---- epilogue: bias, ReLU, instance-norm moments, LDS-staged coalesced store
    __syncthreads();
    TO* Cs = reinterpret_cast<TO*>(smem);  // [256 pixels][128 channels]
    float ssum[4], ssq[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int col = n0 + wn * 64 + n * 16 + li;
        const float bv = (p.bias && col < n_out) ? p.bias[col] : 0.f;
        const float sh = (p.bias && col < n_out) ? cutie::norm_shift(p.bias, p.relu, col) : 0.f;      // the moments' shift
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int m = 0; m < CONV_MB; ++m) {
            const int mb = wm * CONV_MB + m;
            const int h = h0 + mb / TWB;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int w = w0 + (mb % TWB) * 16 + lg * 4 + r;
                float v = acc[m][n][r] + bv;
                if (p.relu == 1) v = fmaxf(v, 0.f);
                else if (p.relu == 2) v = 1.f / (1.f + expf(-v));
                const bool inside = h < H && w < W;
                if (!inside) v = 0.f;
                const float d = inside ? v - sh : 0.f;
                s1 += d;
                s2 += d * d;
                const int ml = mb * 16 + lg * 4 + r;
                store_from_f32(Cs + ml * 128 + wn * 64 + n * 16 + li, v);
            }
        }
        ssum[n] = s1;
        ssq[n] = s2;
    }
    float2* wsum = reinterpret_cast<float2*>(smem + 256 * 128 * sizeof(TO));   // [2 pixel halves][128 channels]
    if (p.stats) {
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            float s1 = ssum[n], s2 = ssq[n];
            s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
            if (lg == 0) wsum[wm * 128 + wn * 64 + n * 16 + li] = make_float2(s1, s2);
        }
    }
    __syncthreads();
    if (p.stats && tid < 128 && n0 + tid < n_out) {     // the tile's slot: stored, never added to (no atomics)
        const float2 u = wsum[tid], v = wsum[128 + tid];
        *reinterpret_cast<float2*>(p.stats + (((long long)b * p.stats_tiles + (th * p.tiles_w + tw)) * n_out + n0 + tid) * 2) =
            make_float2(u.x + v.x, u.y + v.y);
    }
    constexpr int EPC = 16 / sizeof(TO);         // elements per 16-byte piece
    constexpr int PIECES = 128 / EPC;            // pieces per pixel row of the tile
    for (int idx = tid; idx < 256 * PIECES; idx += CONV_NT) {
        const int ml = idx / PIECES, pc = idx - ml * PIECES;
        const int mb = ml >> 4;
        const int h = h0 + mb / TWB, w = w0 + (mb % TWB) * 16 + (ml & 15);
        const int col = n0 + pc * EPC;
        if (h < H && w < W && col < n_out) {
            TO* dst = reinterpret_cast<TO*>(p.out) + ((long long)(b * H + h) * W + w) * p.ld_out + col;
            const TO* src = Cs + ml * 128 + pc * EPC;
            if (col + EPC <= n_out && ((reinterpret_cast<uintptr_t>(dst) & 15) == 0)) {
                *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
            } else {
                for (int e = 0; e < EPC && col + e < n_out; ++e) dst[e] = src[e];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
struct WgradParams {
    const char* x;
    const char* dz;
    int ld_x, cin, ld_dz, cout;
    int B, H, W, kh, kw, dil;
    float* slabs;    // [splits][taps][cin][cout]
    int splits, mt, ntile;
};

__device__ __forceinline__ int tr_func(int r) { return (r & 3) | (((r >> 3) & 1) << 2); }

template <int DT>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradParams p) {
    constexpr int ES = Traits<DT>::ES;
    constexpr int ROWB = 128 * ES;  // bytes per staged pixel row (128 channels)
    constexpr int KPIX = 64;        // pixels per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* Xs = smem;
    char* Ds = smem + KPIX * ROWB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, lg = lane >> 4;

    int bid = blockIdx.x;
    const int nt = bid % p.ntile; bid /= p.ntile;
    const int mt = bid % p.mt;
    const int tap = bid / p.mt;
    const int ti = tap / p.kw, tj = tap - ti * p.kw;
    const int sy = (ti - p.kh / 2) * p.dil, sx = (tj - p.kw / 2) * p.dil;
    const int m0 = mt * 128, n0 = nt * 128;

    const int rows_total = p.B * p.H;
    const int per = (rows_total + p.splits - 1) / p.splits;
    const int row_begin = blockIdx.y * per;
    const int row_end = min(rows_total, row_begin + per);

    float4_t acc[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = float4_t{0.f, 0.f, 0.f, 0.f};

    constexpr int CPR = ROWB / 16;  // 16-byte chunks per staged row: 16 (bf16) / 32 (f32)
    constexpr int EPC = 16 / ES;    // elements per chunk
    for (int row = row_begin; row < row_end; ++row) {
        const int b = row / p.H, h = row - b * p.H;
        const int hh = h + sy;
        if (hh < 0 || hh >= p.H) continue;
        for (int ws = 0; ws < p.W; ws += KPIX) {
            if (ws + sx + KPIX <= 0 || ws + sx >= p.W) continue;  // whole x window outside: zeros
            __syncthreads();
#pragma unroll
            for (int u = 0; u < (KPIX * CPR) / 256; ++u) {
                const int idx = u * 256 + tid;
                const int c = idx % CPR, r = idx / CPR;
                const int w = ws + r;
                uint4 vx = make_uint4(0u, 0u, 0u, 0u), vd = vx;
                const int wsrc = w + sx;
                if (w < p.W && wsrc >= 0 && wsrc < p.W && m0 + c * EPC < p.cin) {
                    const long long off = ((long long)(b * p.H + hh) * p.W + wsrc) * p.ld_x + m0 + c * EPC;
                    vx = *reinterpret_cast<const uint4*>(p.x + off * ES);
                }
                if (w < p.W && n0 + c * EPC < p.cout) {
                    const long long off = ((long long)(b * p.H + h) * p.W + w) * p.ld_dz + n0 + c * EPC;
                    vd = *reinterpret_cast<const uint4*>(p.dz + off * ES);
                }
                int dst;
                if constexpr (ES == 2) {
                    dst = r * ROWB + (((c >> 1) ^ tr_func(r)) * 32) + (c & 1) * 16;
                } else {
                    dst = r * ROWB + c * 16;
                }
                *reinterpret_cast<uint4*>(Xs + dst) = vx;
                *reinterpret_cast<uint4*>(Ds + dst) = vd;
            }
            __syncthreads();
            if constexpr (ES == 2) {
#pragma unroll
                for (int ks = 0; ks < KPIX / 32; ++ks) {
                    const int q = li >> 2, pp = li & 3;
                    const int r1 = ks * 32 + lg * 8 + q, r2 = r1 + 4;
                    short8_t af[4], bf[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        const int blk = wm * 4 + m;
                        const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) short4_t*)(Xs + r1 * ROWB + ((blk ^ tr_func(r1)) * 32) + pp * 8));
                        const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) short4_t*)(Xs + r2 * ROWB + ((blk ^ tr_func(r2)) * 32) + pp * 8));
                        af[m] = short8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        const int blk = wn * 4 + n;
                        const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) short4_t*)(Ds + r1 * ROWB + ((blk ^ tr_func(r1)) * 32) + pp * 8));
                        const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) short4_t*)(Ds + r2 * ROWB + ((blk ^ tr_func(r2)) * 32) + pp * 8));
                        bf[n] = short8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    }
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            acc[m][n] = mma16<DT>(af[m], bf[n], acc[m][n]);
                }
            } else {
                for (int ks = 0; ks < KPIX / 4; ++ks) {
                    const int r = ks * 4 + lg;
                    float af[4], bf[4];
#pragma unroll
                    for (int m = 0; m < 4; ++m)
                        af[m] = *reinterpret_cast<const float*>(Xs + r * ROWB + ((wm * 4 + m) * 16 + li) * 4);
#pragma unroll
                    for (int n = 0; n < 4; ++n)
                        bf[n] = *reinterpret_cast<const float*>(Ds + r * ROWB + ((wn * 4 + n) * 16 + li) * 4);
#pragma unroll
                    for (int m = 0; m < 4; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[m], bf[n], acc[m][n], 0, 0, 0);
                }
            }
        }
    }

    float* slab = p.slabs + ((long long)blockIdx.y * p.kh * p.kw + tap) * p.cin * p.cout;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ci = m0 + (wm * 4 + m) * 16 + lg * 4 + r;
                const int co = n0 + (wn * 4 + n) * 16 + li;
                if (ci < p.cin && co < p.cout) slab[(long long)ci * p.cout + co] = acc[m][n][r];
            }
}

// ------------------------------------------------------------------------------------------
// 16-bit (bf16 / f16) weight gradient, all kw column taps of one tap row per workgroup.
//   workgroup = 512 lanes = 8 waves as 4 (cin) x 2 (cout); tile = 128 cin x 128 cout x kw taps;
//   a wave owns 32 cin x 64 cout for every tap: kw x 2 x 4 accumulators (160 registers at kw=5).
//   Per image row (64 pixels of K) one input row patch [halo + 64 + halo][128 cin] and one
//   output-gradient row [64][128 cout] are staged; the kw taps read the SAME patch at shifted
//   pixel rows through ds_read_b64_tr_b16, so the x operand costs 1/kw of a per-tap kernel.
//   Staging is LDS-DMA (`buffer_load_dwordx4 ... lds`, 1-KB pieces of 4 pixel rows, the swizzle
//   applied on the source address): the pieces of row r+1 are sent off between the taps of row
//   r's first K block and land in the other half of the LDS image; one barrier per row.
//   KW = 1 (1x1 layers) has no taps to share a patch between, so there the "tap" index walks TWO cin tiles
//   instead: a workgroup owns 256 cin x 128 cout, stages both 128-channel patches of a row (2 x 16 KB, the
//   space one haloed patch takes) and gets 32 MFMAs out of a staged row instead of 16 (1024 -> 256: 141 -> see
//   DESIGN.md).
//   MASK >= 0: bit ks*KW+t says whether K block ks of tap t overlaps the image - a compile-time
//   pattern (all ten for W = 64 and dilation <= 8; eight for dilation 16) that leaves the row loop
//   branch-free so the compiler can pipeline it; MASK < 0 is the general instance (uniform tests).
constexpr int WG5_NT = 512;
constexpr int WG5_KW = 5;      // at most 5 column taps
constexpr int WG5_PIX = 64;    // pixels of K per stage
constexpr int WG5_XROWS = 96;  // staged patch rows (24 pieces): 64 + 2 * pad <= 96 ...
constexpr int WG5_XROWS_WIDE = 128;   // ... or 32 pieces = the whole patch buffer (WIDEPATCH instances)
// LDS map: x patch 0 at 0, x patch 1 at 32 KB, dz rows at 64 KB and 80 KB (the halves toggle by XOR)
constexpr int WG5_XSTRIDE = 32768, WG5_D0 = 65536, WG5_DBUF = WG5_PIX * 256;
// LDS per workgroup: 96 KB; the wide 1x1 instance (256 cin x 256 cout a workgroup) stages two cout tiles of dz per half: 128 KB
constexpr int wg5_lds(int kw, bool narrow) { return WG5_D0 + 2 * WG5_DBUF * ((kw == 1 && !narrow) ? 2 : 1); }

struct Wgrad5Params {
    const char* x;
    const char* dz;
    int ld_x, cin, ld_dz, cout;
    int B, H, W, kh, kw, dil;
    float* slabs;    // [splits][taps][cin][cout]
    int splits, mt, ntile, nwseg, pad;   // pad = halo columns staged on each side (clipped)
};

//   NARROW (cout <= 64, one cout tile): the eight waves are 8 (cin) x 1 (cout) instead of 4 x 2 - a wave owns
//   16 cin x 64 cout - so that no wave multiplies the padding half of a 128-wide cout tile.
//   WIDEPATCH: a patch of up to 128 rows (64 + 2 x 32 halo: dilation 16 on an image wider than one 64-pixel
//   segment) - a fourth x piece per wave fills the 32 KB a patch buffer has.
// several layers in ONE launch (cutie_conv_wgrad_batched): workgroups [start[i], start[i+1]) belong to layer i.  With
// G = sum of the layers' (tap row, cin tile, cout tile) groups in one launch, 256 CUs are filled by 256 / G K splits
// instead of 256 / 12: a third of the slab bytes, of the reduction reads and of the per-workgroup prologues.
constexpr int WG5_MAXB = 16;
struct Wgrad5Batch {
    Wgrad5Params p[WG5_MAXB];
    int pair_start[WG5_MAXB + 1];   // (layer, K split) pairs are numbered layer by layer: layer i owns [pair_start[i], pair_start[i+1])
    int n, gmax;                    // gmax = the most (tap row, cin tile, cout tile) groups any layer of the launch has
};

template <int DT, int KW, int MASK, bool NARROW, bool WIDEPATCH>
__device__ __forceinline__ void wgrad5_body(const Wgrad5Params& p, const int split, int bid) {
    constexpr int ES = 2;
    constexpr int MBLK = NARROW ? 1 : 2;     // 16-channel cin blocks per wave
    constexpr int ROWB = 256;       // bytes per staged pixel row: 128 channels
    constexpr bool TWO = KW == 1;   // two cin tiles per workgroup in place of taps
    // ... and, on the wide 1x1 layers, two cout tiles as well: 256 cin x 256 cout a workgroup, 64 MFMAs a wave out of a
    // staged row (64 KB) where 256 x 128 gave 32 out of 48 KB - a 1x1 weight gradient is a plain GEMM with K = pixels and
    // lives on operand reuse alone (round 3: 111 -> see DESIGN.md)
    constexpr bool QUAD = TWO && !NARROW;
    constexpr int NCT = QUAD ? 2 : 1;        // cout tiles per workgroup
    constexpr int NT_ = TWO ? 2 : KW;        // accumulator sets: taps, or cin tiles
    constexpr int NX = (TWO || WIDEPATCH) ? 4 : 3;   // x pieces per wave and row
    constexpr int ND = 2 * NCT;              // dz pieces per wave and row
    constexpr int NPC = NX + ND;
    constexpr int DBUF = WG5_DBUF * NCT;     // one half of the dz image
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int pwx = WG5_PIX + 2 * p.pad;                    // patch rows that hold data

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = NARROW ? wave : wave >> 1, wn = NARROW ? 0 : wave & 1;
    const int li = lane & 15, lg = lane >> 4;
    const int H = p.H, W = p.W, dil = p.dil, pad = p.pad;

    const int nt = bid % p.ntile; bid /= p.ntile;
    const int mt = bid % p.mt;
    const int ti = bid / p.mt;                 // tap row
    const int sy = (ti - p.kh / 2) * dil;
    const int m0 = mt * (TWO ? 256 : 128), n0 = nt * (128 * NCT);

    const int units_total = p.B * H * p.nwseg;  // a unit = (sample, image row, 64-pixel column segment)
    const int per = (units_total + p.splits - 1) / p.splits;
    const int unit_begin = split * per;
    const int unit_end = min(units_total, unit_begin + per);

    float4_t acc[NT_][NCT][MBLK][4];
#pragma unroll
    for (int t = 0; t < NT_; ++t)
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int m = 0; m < MBLK; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) acc[t][c][m][n] = float4_t{0.f, 0.f, 0.f, 0.f};

    const long long xbytes = ((long long)(p.B * H * W - 1) * p.ld_x + p.cin) * ES;
    const long long dbytes = ((long long)(p.B * H * W - 1) * p.ld_dz + p.cout) * ES;
    const u32x4s_t rsrc_x = make_rsrc(p.x, (unsigned)xbytes);
    const u32x4s_t rsrc_d = make_rsrc(p.dz, (unsigned)dbytes);
    const unsigned lds0 = (unsigned)(uintptr_t)((lds_void_t*)smem);

    // staging plan: wave w moves x-patch pieces w, w+8, w+16 and dz pieces w, w+8 (1 KB = 4 pixel
    // rows x 256 B).  Lane l of a piece is (row = l >> 4, 16-byte slot s = l & 15) and fetches
    // channel chunk ((s >> 1) ^ tr_func(row)) * 2 + (s & 1): the swizzle the transposed reads
    // below expect, applied on the SOURCE address.  Rows past the patch get zeros.
    const int prow = lane >> 4, pslot = lane & 15;
    int x_r[NX], x_off[NX], d_r[ND], d_off[ND];
    bool x_ok[NX], d_ok[ND];
#pragma unroll
    for (int u = 0; u < NX; ++u) {
        // TWO: pieces 0, 1 are the 64 rows of cin tile 0, pieces 2, 3 those of cin tile 1
        x_r[u] = (wave + 8 * (TWO ? (u & 1) : u)) * 4 + prow;
        const int chn = m0 + (TWO ? (u >> 1) * 128 : 0) + ((((pslot >> 1) ^ tr_func(x_r[u])) << 1) | (pslot & 1)) * 8;
        x_ok[u] = x_r[u] < pwx && chn < p.cin;
        x_off[u] = ((x_r[u] - pad) * p.ld_x + chn) * ES;
    }
#pragma unroll
    for (int u = 0; u < ND; ++u) {      // pieces 0, 1: the 64 rows of cout tile 0; 2, 3 (QUAD): those of cout tile 1
        d_r[u] = (wave + 8 * (u & 1)) * 4 + prow;
        const int chn = n0 + (u >> 1) * 128 + ((((pslot >> 1) ^ tr_func(d_r[u])) << 1) | (pslot & 1)) * 8;
        d_ok[u] = chn < p.cout;
        d_off[u] = (d_r[u] * p.ld_dz + chn) * ES;
    }
    // unit -> (b, h, ws); returns false when the shifted input row is outside the image
    auto decode = [&](int unit, int& b, int& h, int& ws) {
        const int seg = unit % p.nwseg;
        const int row = unit / p.nwseg;
        b = row / H;
        h = row - b * H;
        ws = seg * WG5_PIX;
        const int hh = h + sy;
        return hh >= 0 && hh < H;
    };
    auto issue_piece = [&](int k, int b, int h, int ws, int bsel, bool live) {    // k = 0..NPC-1, compile-time at every call
        if (k < NX) {
            const int w = ws - pad + x_r[k];
            const bool ok = live && x_ok[k] && w >= 0 && w < W;
            const int piece = TWO ? (k >> 1) * 16 + wave + 8 * (k & 1) : wave + 8 * k;      // 1-KB slot in the x image
            glds16(rsrc_x, lds0 + bsel * WG5_XSTRIDE + piece * 1024,
                   ok ? (unsigned)(((b * H + h + sy) * W + ws) * p.ld_x * ES + x_off[k]) : BUF_OOB);
        } else {
            const int u = k - NX;
            const bool ok = live && d_ok[u] && ws + d_r[u] < W;
            glds16(rsrc_d, lds0 + WG5_D0 + bsel * DBUF + (u >> 1) * WG5_DBUF + (wave + 8 * (u & 1)) * 1024,
                   ok ? (unsigned)(((b * H + h) * W + ws) * p.ld_dz * ES + d_off[u]) : BUF_OOB);
        }
    };
    // next unit at or after `u` whose input row is inside the image
    auto seek = [&](int u, int& b, int& h, int& ws) {
        while (u < unit_end && !decode(u, b, h, ws)) ++u;
        return u;
    };

    // transposed-read addresses (LDS bytes) for buffer 0 / K block 0 / first 16-channel block of the
    // wave; the other blocks are an XOR of the chunk field (blk ^ f = (blk0 ^ f) ^ i for i below
    // blk0's alignment), K block 1 is + 32 rows (tr_func ignores bit 5) and the two halves are +- a fixed stride
    const int q = li >> 2, pp = li & 3;
    const int r1 = lg * 8 + q;
    unsigned da = lds0 + WG5_D0 + r1 * ROWB + (((wn * 4) ^ tr_func(r1)) * 32) + pp * 8;   // hi row = + 4 rows
    unsigned xa[NT_][2];
#pragma unroll
    for (int t = 0; t < NT_; ++t) {
        const int x1 = pad + (TWO ? 0 : (t - KW / 2) * dil) + r1, x2 = x1 + 4;
        const unsigned tile = TWO ? t * 16384u : 0u;        // the second cin tile's patch sits 16 KB further on
        xa[t][0] = lds0 + tile + x1 * ROWB + (((wm * MBLK) ^ tr_func(x1)) * 32) + pp * 8;
        xa[t][1] = lds0 + tile + x2 * ROWB + (((wm * MBLK) ^ tr_func(x2)) * 32) + pp * 8;
    }
    auto tr2 = [&](unsigned lo, unsigned hi) {
        const short4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(uintptr_t)(lo));
        const short4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(uintptr_t)(hi));
        return short8_t{a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
    };

    int cb, ch, cws;
    int cur = seek(unit_begin, cb, ch, cws);
    int buf = 0;
    if (cur < unit_end) {
#pragma unroll
        for (int k = 0; k < NPC; ++k) issue_piece(k, cb, ch, cws, 0, true);
    }
    while (cur < unit_end) {
        int nb, nh, nws;
        const int nxt = seek(cur + 1, nb, nh, nws);
        const bool more = nxt < unit_end;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my pieces of this row have landed ...
        __builtin_amdgcn_s_barrier();                        // ... everybody's have, and the other half is free
        // one image row: 2 K blocks of 32 pixels x KW taps x (2 x 4) MFMAs; the pieces of the next
        // row are sent on their way between the taps of the first K block (zeros when there is none)
#pragma unroll
        for (int ks = 0; ks < WG5_PIX / 32; ++ks) {
            const int px0 = cws + ks * 32;                   // first output pixel of this K block
            if (MASK < 0 && px0 >= W) {
                if (ks == 0) {
#pragma unroll
                    for (int k = 0; k < NPC; ++k) issue_piece(k, nb, nh, nws, buf ^ 1, more);
                }
                continue;
            }
            short8_t bf[NCT][4];
#pragma unroll
            for (int c = 0; c < NCT; ++c)
#pragma unroll
                for (int n = 0; n < 4; ++n)
                    bf[c][n] = tr2((da ^ (n * 32)) + c * WG5_DBUF + ks * 32 * ROWB, (da ^ (n * 32)) + c * WG5_DBUF + ks * 32 * ROWB + 4 * ROWB);
#pragma unroll
            for (int t = 0; t < NT_; ++t) {
                if (ks == 0) {
#pragma unroll
                    for (int k = 0; k < NPC; ++k)
                        if (k * NT_ / NPC == t) issue_piece(k, nb, nh, nws, buf ^ 1, more);
                }
                const int sx = TWO ? 0 : (t - KW / 2) * dil;
                // input pixels px0+sx .. px0+sx+31 all outside the image: nothing to add (uniform)
                if (MASK >= 0 && !((MASK >> (ks * KW + t)) & 1)) continue;     // compile-time pattern
                if (MASK < 0 && (px0 + sx + 31 < 0 || px0 + sx >= W)) continue;
                short8_t af[MBLK];
#pragma unroll
                for (int m = 0; m < MBLK; ++m)
                    af[m] = tr2((xa[t][0] ^ (m * 32)) + ks * 32 * ROWB, (xa[t][1] ^ (m * 32)) + ks * 32 * ROWB);
#pragma unroll
                for (int c = 0; c < NCT; ++c)
#pragma unroll
                    for (int m = 0; m < MBLK; ++m)
#pragma unroll
                        for (int n = 0; n < 4; ++n)
                            acc[t][c][m][n] = mma16<DT>(af[m], bf[c][n], acc[t][c][m][n]);
            }
        }
        // switch halves by +- the stride (not XOR: a dead tap's base row may be negative)
        const int dxa = buf ? -WG5_XSTRIDE : WG5_XSTRIDE, dda = buf ? -DBUF : DBUF;
        buf ^= 1;
        da += dda;
#pragma unroll
        for (int t = 0; t < NT_; ++t) { xa[t][0] += dxa; xa[t][1] += dxa; }
        cur = nxt; cb = nb; ch = nh; cws = nws;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last row's (all-zero) prefetch

#pragma unroll
    for (int t = 0; t < NT_; ++t) {
        float* slab = p.slabs + ((long long)split * p.kh * KW + (TWO ? ti : ti * KW + t)) * p.cin * p.cout;
#pragma unroll
        for (int c = 0; c < NCT; ++c)
#pragma unroll
            for (int m = 0; m < MBLK; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int ci = m0 + (TWO ? t * 128 : 0) + (wm * MBLK + m) * 16 + lg * 4 + r;
                        const int co = n0 + c * 128 + (wn * 4 + n) * 16 + li;
                        if (ci < p.cin && co < p.cout) slab[(long long)ci * p.cout + co] = acc[t][c][m][n][r];
                    }
    }
}

template <int DT, int KW, int MASK, bool NARROW = false, bool WIDEPATCH = false>
__global__ __launch_bounds__(WG5_NT, 2) void wgrad5_kernel(Wgrad5Params p) {
    // Workgroups that multiply the SAME pixels (one K split; different tap row / channel tiles)
    // should meet in one L2: hardware deals consecutive ids round-robin over the 8 XCDs, so give
    // each XCD a contiguous run of logical ids (bijective for any grid size) and make the K split
    // the slow index of the logical id.
    const int groups = p.kh * p.mt * p.ntile;
    const int nwg = groups * p.splits;
    const int hw_id = blockIdx.x;
    const int qx = nwg >> 3, rx = nwg & 7, xcd = hw_id & 7;
    const int logical = (xcd < rx ? xcd * (qx + 1) : rx * (qx + 1) + (xcd - rx) * qx) + (hw_id >> 3);
    const int split = logical / groups;
    wgrad5_body<DT, KW, MASK, NARROW, WIDEPATCH>(p, split, logical - split * groups);
}

template <int DT, int KW, int MASK, bool NARROW = false, bool WIDEPATCH = false>
__global__ __launch_bounds__(WG5_NT, 2) void wgrad5_batched_kernel(Wgrad5Batch bt) {
    // XCD x (= blockIdx & 7) takes the pairs x, x + 8, ...; the gmax consecutive workgroups it is dealt belong to ONE pair,
    // so that the workgroups that read the same pixels run side by side behind one L2
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int pair = (j / bt.gmax) * 8 + xcd, group = j % bt.gmax;
    if (pair >= bt.pair_start[bt.n]) return;       // (uniform)
    int i = 0;
#pragma unroll
    for (int k = 1; k < WG5_MAXB; ++k)
        if (k < bt.n && pair >= bt.pair_start[k]) i = k;
    const Wgrad5Params p = bt.p[i];
    if (group >= p.kh * p.mt * p.ntile) return;    // a layer with fewer groups than gmax
    wgrad5_body<DT, KW, MASK, NARROW, WIDEPATCH>(p, pair - bt.pair_start[i], group);
}

// slabs -> gradient for several layers in one launch: blocks [start[i], start[i+1]) walk layer i
struct ReduceBatch {
    const float* slabs[WG5_MAXB];
    float* dw[WG5_MAXB];
    long long n[WG5_MAXB];
    int splits[WG5_MAXB], accumulate[WG5_MAXB], start[WG5_MAXB + 1];
    int cnt;
};
__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ slabs, int splits, long long n, float* __restrict__ dw,
                                                  int accumulate, long long first, long long stride);

__global__ __launch_bounds__(256) void wgrad_reduce_batched_kernel(ReduceBatch rb) {
    int i = 0;
#pragma unroll
    for (int j = 1; j < WG5_MAXB; ++j)
        if (j < rb.cnt && (int)blockIdx.x >= rb.start[j]) i = j;
    const int nb = rb.start[i + 1] - rb.start[i];
    wgrad_reduce_body(rb.slabs[i], rb.splits[i], rb.n[i], rb.dw[i], rb.accumulate[i],
                      (long long)((int)blockIdx.x - rb.start[i]) * blockDim.x + threadIdx.x, (long long)nb * blockDim.x);
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ slabs, int splits, long long n,
                                                           float* __restrict__ dw, int accumulate) {
    wgrad_reduce_body(slabs, splits, n, dw, accumulate, (long long)blockIdx.x * blockDim.x + threadIdx.x,
                      (long long)gridDim.x * blockDim.x);
}

__device__ __forceinline__ void wgrad_reduce_body(const float* __restrict__ slabs, int splits, long long n, float* __restrict__ dw,
                                                  int accumulate, long long first, long long stride) {
    // n is a multiple of 4 whenever cout is (every layer but the 18-class logits): 16-byte lanes, four
    // slabs in flight per lane; the summation order over the slabs is fixed either way
    if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(slabs) | reinterpret_cast<uintptr_t>(dw)) & 15) == 0) {
        const long long n4 = n >> 2;
        const float4* __restrict__ s4 = reinterpret_cast<const float4*>(slabs);
        float4* __restrict__ d4 = reinterpret_cast<float4*>(dw);
        for (long long i = first; i < n4; i += stride) {
            float4 s = accumulate ? d4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            int k = 0;
            for (; k + 4 <= splits; k += 4) {
                const float4 a = s4[(long long)k * n4 + i], b = s4[(long long)(k + 1) * n4 + i];
                const float4 c = s4[(long long)(k + 2) * n4 + i], d = s4[(long long)(k + 3) * n4 + i];
                s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
                s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
                s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
                s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
            }
            for (; k < splits; ++k) {
                const float4 a = s4[(long long)k * n4 + i];
                s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            }
            d4[i] = s;
        }
        return;
    }
    for (long long i = first; i < n; i += stride) {
        float s = accumulate ? dw[i] : 0.f;
        for (int k = 0; k < splits; ++k) s += slabs[(long long)k * n + i];
        dw[i] = s;
    }
}

inline int wgrad5_splits(int B, int H, int W, int cin, int cout, int kh, int kw) {
    const long long groups = (long long)kh * cutie::cdiv(cin, kw == 1 ? 256 : 128) * cutie::cdiv(cout, (kw == 1 && cout > 64) ? 256 : 128);
    long long s = 256 / groups;                 // about one 512-lane workgroup per CU
    const long long units = (long long)B * H * cutie::cdiv(W, WG5_PIX);
    if (s > units) s = units;
    if (s < 1) s = 1;
    return (int)s;
}

inline int wgrad_splits(int B, int H, int cin, int cout, int taps) {
    const long long tiles = (long long)taps * cutie::cdiv(cin, 128) * cutie::cdiv(cout, 128);
    long long s = 1024 / tiles;
    if (s > 64) s = 64;
    if (s > (long long)B * H) s = (long long)B * H;
    if (s < 1) s = 1;
    return (int)s;
}

template <int DT, int ODT>
int launch_conv(const ConvParams& p, size_t lds, hipStream_t stream) {
    const void* fn = (const void*)conv_kernel<DT, ODT>;
    if (lds > 64 * 1024) CUTIE_ENSURE_LDS(fn, lds);
    const long long grid = (long long)p.B * p.tiles_h * p.tiles_w * p.n_tiles_n;
    hipLaunchKernelGGL((conv_kernel<DT, ODT>), dim3((unsigned)grid), dim3(CONV_NT), lds, stream, p);
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

}  // namespace

namespace {
// 256-pixel tilings of the two kernels: (rows, 16-pixel column blocks) per tile
inline void generic_tiling(int W, int& th, int& twb) {
    twb = 1;
    while (twb < 8 && twb * 16 < W) twb *= 2;
    th = 16 / twb;
}
inline void wide_tiling(int H, int W, int& th, int& tcb) {
    th = 16;
    tcb = 1;
    while (th > 1 && th / 2 >= H && tcb * 16 < W) { th /= 2; tcb *= 2; }
}
}  // namespace

extern "C" int cutie_conv_stats_tiles(int H, int W) {
    if (H < 1 || W < 1) return 1;
    int th, tb;
    generic_tiling(W, th, tb);
    const long long g = cutie::cdiv(H, th) * cutie::cdiv(W, tb * 16);
    wide_tiling(H, W, th, tb);
    const long long w = cutie::cdiv(H, th) * cutie::cdiv(W, tb * 16);
    return (int)(g > w ? g : w);
}

namespace {
struct NormBwd {      // the reduction of cutie_in_bwd_stats, asked of a data-gradient launch
    const void* a;
    int ld_a, a_dtype;
    const float* stats;
    int stats_tiles;
    float eps;
    float* red;
    int red_tiles;
    const float* bias;   // the forward layer's bias and activation code: the shift of its moments
    int act;
    // cutie_conv_dgrad_in_bwd: the second half of the norm's gradient as well (dz, bias-gradient partials)
    void* dz;
    int ld_dz, relu;
    float* dbias;
    int dbias_tiles;
    unsigned* sync;      // word 0: tiles whose wait ran out; from byte 64: [B][red_tiles][n_out] 16-byte granule pairs
    int* late;           // nullable: the caller's step-wide counter of the same event
};
}  // namespace

namespace {
struct NormFwd {      // cutie_conv_in_fwd: the normalised output, asked of a forward launch
    void* y;
    int ld_y;
    float eps;
    unsigned* sync;      // word 0: waits that ran out; from byte 64: [B][stats_tiles][n_out] 16-byte granule pairs
    int* late;           // nullable: the caller's step-wide counter of the same event
};
}  // namespace

// A sync area belongs to ONE (B, slot tiles, n_out): the tiles of a launch derive its tag from their own slots, which only
// agree when every earlier launch on the area had the same geometry.  The library cannot read device memory here, so it
// keeps the geometry each area was last used with on the host and clears the area (on the launch stream) when a call
// brings another one - a caller that reuses a buffer for a second layer or shape gets a fresh area instead of tiles that
// wait for tags nobody writes.
static int sync_area_claim(void* sync, size_t bytes, int B, int tiles, int n_out, hipStream_t stream) {
    struct Entry { const void* p; int B, tiles, n_out; };
    static std::mutex mu;
    static std::vector<Entry> seen;
    std::lock_guard<std::mutex> lock(mu);
    for (Entry& e : seen) {
        if (e.p != sync) continue;
        if (e.B == B && e.tiles == tiles && e.n_out == n_out) return CUTIE_OK;
        CUTIE_CHECK_HIP(hipMemsetAsync(sync, 0, bytes, stream));
        e.B = B; e.tiles = tiles; e.n_out = n_out;
        return CUTIE_OK;
    }
    seen.push_back({sync, B, tiles, n_out});      // first sight: zeroed by the caller (the documented contract)
    return CUTIE_OK;
}

static int conv_fwd_impl(const cutie_conv_seg* segs, int n_segs, int B, int H, int W, int n_out, void* out,
                         int ld_out, int out_dtype, const float* bias, int relu, float* stats, int stats_tiles,
                         int dtype, const NormBwd* nb, cutie_stream_t stream_, const NormFwd* nf = nullptr) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(segs && n_segs >= 1 && n_segs <= CUTIE_MAX_SEGS, "conv_fwd: n_segs=%d (1..%d)", n_segs, CUTIE_MAX_SEGS);
    CUTIE_CHECK_ARG(B >= 0 && H >= 1 && W >= 1 && n_out >= 1 && ld_out >= n_out, "conv_fwd: bad sizes B=%d H=%d W=%d n_out=%d ld_out=%d",
                    B, H, W, n_out, ld_out);
    CUTIE_CHECK_ARG(dtype == CUTIE_F32 || dtype == CUTIE_BF16 || dtype == CUTIE_F16, "conv_fwd: dtype %d", dtype);
    CUTIE_CHECK_ARG(out_dtype == CUTIE_F32 || out_dtype == dtype || (dtype == CUTIE_F32 && (out_dtype == CUTIE_BF16 || out_dtype == CUTIE_F16)),
                    "conv_fwd: out_dtype %d must be f32 or the input dtype %d", out_dtype, dtype);
    CUTIE_CHECK_ARG(relu >= 0 && relu <= 2, "conv_fwd: activation code %d (0 none, 1 ReLU, 2 sigmoid)", relu);
    if (B == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(out, "conv_fwd: null output");
    const int es = dtype == CUTIE_F32 ? 4 : 2;
    const int bke = dtype == CUTIE_F32 ? 32 : 64;

    ConvParams p;
    p.nseg = n_segs;
    p.bw_a = nullptr; p.bw_ld_a = 0; p.bw_stats = nullptr; p.bw_stats_tiles = 0; p.bw_eps = 0.f; p.bw_bias = nullptr; p.bw_act = 0; p.bw_red = nullptr; p.bw_red_tiles = 0; p.perm_n = 0;
    p.clk_probe = nullptr;
    p.rv_slots = nullptr; p.rv_dz = nullptr; p.rv_ld_dz = 0; p.rv_dbias = nullptr; p.rv_dbias_tiles = 0; p.rv_relu = 0; p.rv_plain_order = 0; p.rv_err = nullptr; p.rv_late = nullptr; p.fw_slots = nullptr; p.fw_eps = 0.f;
    int twb, th_g;
    generic_tiling(W, th_g, twb);
    p.TWB = twb;
    p.TH = th_g;          // 16 pixel blocks = 256 pixels per tile
    p.tiles_h = (int)cutie::cdiv(H, p.TH);
    p.tiles_w = (int)cutie::cdiv(W, twb * 16);
    p.n_tiles_n = (int)cutie::cdiv(n_out, 128);
    int a_bytes = 0;
    for (int s = 0; s < n_segs; ++s) {
        const cutie_conv_seg& g = segs[s];
        CUTIE_CHECK_ARG(g.x && g.w, "conv_fwd: segment %d has a null pointer", s);
        CUTIE_CHECK_ARG(g.kh >= 1 && g.kw >= 1 && (g.kh & 1) && (g.kw & 1) && g.dil >= 1,
                        "conv_fwd: segment %d kernel %dx%d dil %d (odd sizes, dil >= 1)", s, g.kh, g.kw, g.dil);
        CUTIE_CHECK_ARG(g.cin >= bke && g.cin % bke == 0, "conv_fwd: segment %d cin=%d must be a multiple of %d (zero pad)", s,
                        g.cin, bke);
        CUTIE_CHECK_ARG(g.ld_x >= g.cin && (g.ld_x * es) % 16 == 0 && cutie::aligned16(g.x) && cutie::aligned16(g.w),
                        "conv_fwd: segment %d needs 16-byte aligned rows (ld_x=%d)", s, g.ld_x);
        CUTIE_CHECK_ARG(g.w_tap_stride >= (int64_t)n_out * g.cin, "conv_fwd: segment %d w_tap_stride too small", s);
        CUTIE_CHECK_ARG((int64_t)H * W * g.ld_x * es < (1ll << 31) && (int64_t)g.kh * g.kw * g.w_tap_stride * es < (1ll << 31),
                        "conv_fwd: segment %d: one sample / one weight tensor must stay below 2 GiB (buffer addressing)", s);
        Seg& d = p.seg[s];
        d.x = (const char*)g.x; d.w = (const char*)g.w; d.w_tap_stride = g.w_tap_stride;
        d.ld_x = g.ld_x; d.cin = g.cin; d.kh = g.kh; d.kw = g.kw; d.dil = g.dil; d.sgn = g.flip ? -1 : 1;
        const int tw_px = twb * 16;
        int pad = (g.kw / 2) * g.dil;
        const int pad_cap = (W > tw_px ? W - tw_px : 0) + 15;   // see conv_geom: halo clipped to the image + 15
        if (pad > pad_cap) pad = pad_cap;
        const int pw = tw_px + 2 * pad;
        const int bytes = p.TH * pw * 128;
        if (bytes > a_bytes) a_bytes = bytes;
    }
    p.a_bytes = a_bytes;
    p.B = B; p.H = H; p.W = W; p.n_out = n_out;
    p.out = (char*)out; p.ld_out = ld_out; p.bias = bias; p.relu = relu; p.stats = stats; p.stats_tiles = stats_tiles;
    CUTIE_CHECK_ARG(!stats || stats_tiles >= cutie_conv_stats_tiles(H, W),
                    "conv_fwd: stats_tiles=%d < cutie_conv_stats_tiles(%d, %d) = %d", stats_tiles, H, W, cutie_conv_stats_tiles(H, W));
    CUTIE_CHECK_ARG(!stats || (reinterpret_cast<uintptr_t>(stats) & 7) == 0, "conv_fwd: stats must be 8-byte aligned");
    const size_t oes = out_dtype == CUTIE_F32 ? 4 : 2;
    size_t lds = (size_t)a_bytes + 2 * 128 * 128;
    if (lds < 256 * 128 * oes + 2 * 128 * 8) lds = 256 * 128 * oes + 2 * 128 * 8;    // staged tile + per-wave moments
    CUTIE_CHECK_ARG(lds <= 160 * 1024, "conv_fwd: tile needs %zu B of LDS (dilation %d too large for W=%d)", lds,
                    segs[0].dil, W);
    CUTIE_CHECK_ARG((long long)B * p.tiles_h * p.tiles_w * p.n_tiles_n < (1ll << 31), "conv_fwd: grid too large");
    // wide layers: 256-pixel x 256-channel tiles fed by LDS-DMA (conv_wide.hip)
    const bool wide_on = cutie::config().conv_wide;
    int bn = (wide_on && dtype != CUTIE_F32 && out_dtype == dtype) ? cutie::conv_wide_tile(p) : 0;
    // 128-channel tiles on 4-wave workgroups, two a CU (conv_duo_kernel): +10 % on the 128-channel layers, -13 % on the
    // 256-channel ones, so by default only where the tile is 128 channels anyway
    // (measured per launch, tools/gpu_launch_table.sh: 256 -> 128 3x5 140 -> 128 us, 64 -> 64 1x1 31 -> 22 us, but the
    // decoder's 512 -> 64 3x5 156 -> 194 us: a 64-channel tile moves 80 B of A tile per clock either way and the 8-wave
    // kernel's 64-channel steps suit that better; CUTIE2's 3x3 image-branch layers lost 0.7 % of CUTIE2-ASPP inference - so
    // 64-channel tiles only for 1x1 layers of up to 256 input channels)
    long long ksteps = 0;
    for (int s = 0; s < n_segs; ++s) ksteps += (long long)segs[s].kh * segs[s].kw * (segs[s].cin / 64);
    if ((bn == 128 || (bn == 64 && ksteps <= 4)) && cutie::config().conv_duo >= 1) bn = -bn;
    else if (bn == 256 && cutie::config().conv_duo >= 2) bn = -128;
    if (bn) {
        ConvParams q = p;
        int th, tcb;
        wide_tiling(H, W, th, tcb);
        q.TH = th; q.TWB = tcb;
        q.tiles_h = (int)cutie::cdiv(H, th);
        q.tiles_w = (int)cutie::cdiv(W, tcb * 16);
        q.n_tiles_n = (int)cutie::cdiv(n_out, bn < 0 ? -bn : bn);
        // multi-tap dilation-1 launches: the A operand as a patch staged once per chunk (conv_patch_kernel)
        if (bn > 0 && bn <= 128 && cutie::config().conv_patch >= 1 && cutie::conv_patch_ok(q)) bn += 1000;
        if ((long long)B * q.tiles_h * q.tiles_w * q.n_tiles_n < (1ll << 31)) {
            const bool fuse_on = cutie::config().fuse_norm_bwd;
            const bool fuse = nb && fuse_on && !bias && !relu && !stats && nb->a_dtype == dtype && (nb->ld_a * 2) % 16 == 0 && cutie::aligned16(nb->a) &&
                              q.tiles_h * q.tiles_w <= nb->red_tiles;
            bool meet = false;
            if (fuse) {
                q.bw_a = (const char*)nb->a; q.bw_ld_a = nb->ld_a; q.bw_stats = nb->stats; q.bw_stats_tiles = nb->stats_tiles;
                q.bw_eps = nb->eps; q.bw_red = nb->red; q.bw_red_tiles = nb->red_tiles; q.bw_bias = nb->bias; q.bw_act = nb->act;
                // the whole norm gradient inside this launch: a sample's tiles meet after publishing their sums
                int plain = 0;
                meet = nb->dz && nb->sync && cutie::config().rendezvous && (nb->relu & 15) != 2 && (nb->ld_dz * 2) % 16 == 0 &&
                       cutie::aligned16(nb->dz) && (!nb->dbias || q.tiles_h * q.tiles_w <= nb->dbias_tiles) &&
                       cutie::conv_wide_rendezvous_ok(q, bn, &plain);
                if (meet) {
                    q.rv_err = reinterpret_cast<int*>(nb->sync);      // word 0: timeouts; the granules start 64 bytes in
                    q.rv_slots = reinterpret_cast<unsigned long long*>(nb->sync + 16);
                    q.rv_dz = (char*)nb->dz; q.rv_ld_dz = nb->ld_dz; q.rv_dbias = nb->dbias; q.rv_dbias_tiles = nb->dbias_tiles;
                    q.rv_relu = nb->relu & 15; q.rv_plain_order = plain; q.rv_late = nb->late;
                }
            }
            if (nf && !nb && stats && cutie::config().rendezvous && relu != 2 && (nf->ld_y * 2) % 16 == 0 && cutie::aligned16(nf->y)) {
                int plain = 0;
                if (cutie::conv_wide_fw_rendezvous_ok(q, bn, &plain)) {      // the tiles meet: y goes out, a does not
                    q.out = (char*)nf->y; q.ld_out = nf->ld_y;
                    q.fw_slots = reinterpret_cast<unsigned long long*>(nf->sync + 16); q.fw_eps = nf->eps;
                    q.rv_err = reinterpret_cast<int*>(nf->sync); q.rv_plain_order = plain; q.rv_late = nf->late;
                    meet = true;
                }
            }
            const int rc = cutie::launch_conv_wide(q, dtype, bn, stream);
            if (rc != CUTIE_OK) return rc;
            if (meet) return 1;      // positive: the launch did the norm too (callers inside this file only)
            if (fuse || !nb) return rc;
            return cutie_in_bwd_stats(out, ld_out, out_dtype, nb->a, nb->ld_a, nb->a_dtype, nb->stats, nb->stats_tiles, nb->eps,
                                      nb->red, nb->red_tiles, B, H * W, n_out, nb->bias, nb->act, stream_);
        }
    }
    int rc;
    if (dtype == CUTIE_BF16) {
        rc = out_dtype == CUTIE_BF16 ? launch_conv<CUTIE_BF16, CUTIE_BF16>(p, lds, stream) : launch_conv<CUTIE_BF16, CUTIE_F32>(p, lds, stream);
    } else if (dtype == CUTIE_F16) {
        rc = out_dtype == CUTIE_F16 ? launch_conv<CUTIE_F16, CUTIE_F16>(p, lds, stream) : launch_conv<CUTIE_F16, CUTIE_F32>(p, lds, stream);
    } else if (out_dtype == CUTIE_BF16) {
        rc = launch_conv<CUTIE_F32, CUTIE_BF16>(p, lds, stream);
    } else if (out_dtype == CUTIE_F16) {
        rc = launch_conv<CUTIE_F32, CUTIE_F16>(p, lds, stream);
    } else {
        rc = launch_conv<CUTIE_F32, CUTIE_F32>(p, lds, stream);
    }
    if (rc != CUTIE_OK || !nb) return rc;
    // kernels without the fused reduction: the separate pass, into the same slots' array
    return cutie_in_bwd_stats(out, ld_out, out_dtype, nb->a, nb->ld_a, nb->a_dtype, nb->stats, nb->stats_tiles, nb->eps, nb->red,
                              nb->red_tiles, B, H * W, n_out, nb->bias, nb->act, stream_);
}

extern "C" int cutie_conv_fwd(const cutie_conv_seg* segs, int n_segs, int B, int H, int W, int n_out, void* out,
                              int ld_out, int out_dtype, const float* bias, int relu, float* stats, int stats_tiles,
                              int dtype, cutie_stream_t stream_) {
    return conv_fwd_impl(segs, n_segs, B, H, W, n_out, out, ld_out, out_dtype, bias, relu, stats, stats_tiles, dtype, nullptr,
                         stream_);
}

extern "C" int cutie_conv_in_fwd(const cutie_conv_seg* segs, int n_segs, int B, int H, int W, int n_out, void* a, int ld_a,
                                 void* y, int ld_y, const float* bias, int relu, float* stats, int stats_tiles, float eps,
                                 void* sync, size_t sync_bytes, int32_t* late_count, int* fused, int dtype,
                                 cutie_stream_t stream_) {
    CUTIE_CHECK_ARG(B >= 0 && H >= 1 && W >= 1 && n_out >= 1, "conv_in_fwd: bad sizes");
    if (fused) *fused = 0;
    if (B == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(a && y && stats && ld_a >= n_out && ld_y >= n_out, "conv_in_fwd: null pointer or leading dimension < n_out");
    if (sync && (sync_bytes < cutie_conv_sync_bytes(B, stats_tiles, n_out) || (reinterpret_cast<uintptr_t>(sync) & 63))) {
        cutie::set_error("conv_in_fwd: sync area %zu < %zu bytes (or not 64-byte aligned)", sync_bytes,
                         cutie_conv_sync_bytes(B, stats_tiles, n_out));
        return CUTIE_E_WORKSPACE;
    }
    if (sync) {
        if (int e = sync_area_claim(sync, cutie_conv_sync_bytes(B, stats_tiles, n_out), B, stats_tiles, n_out, (hipStream_t)stream_)) return e;
    }
    const NormFwd nf = {y, ld_y, eps, (unsigned*)sync, late_count};
    const int rc = conv_fwd_impl(segs, n_segs, B, H, W, n_out, a, ld_a, dtype, bias, relu, stats, stats_tiles, dtype, nullptr,
                                 stream_, sync ? &nf : nullptr);
    if (rc == 1) {      // one launch did all of it; `a` was not written
        if (fused) *fused = 1;
        return CUTIE_OK;
    }
    if (rc != CUTIE_OK) return rc;
    return cutie_in_apply(a, ld_a, dtype, stats, stats_tiles, eps, y, ld_y, dtype, B, H * W, n_out, bias, relu, stream_);
}

extern "C" int cutie_conv_dgrad_in_stats(const cutie_conv_seg* segs, int n_segs, int B, int H, int W, int n_out, void* out,
                                         int ld_out, int out_dtype, const void* a, int ld_a, int a_dtype, const float* stats,
                                         int stats_tiles, float eps, float* red, int red_tiles,
                                         const float* shift_bias, int shift_act, int dtype, cutie_stream_t stream_) {
    CUTIE_CHECK_ARG(B >= 0 && H >= 1 && W >= 1 && n_out >= 1, "conv_dgrad_in_stats: bad sizes");
    if (B == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(a && stats && red && ld_a >= n_out, "conv_dgrad_in_stats: null pointer or ld_a < n_out");
    CUTIE_CHECK_ARG(stats_tiles >= 1 && red_tiles >= cutie_conv_stats_tiles(H, W) && red_tiles <= 65535,
                    "conv_dgrad_in_stats: stats_tiles=%d, red_tiles=%d (>= cutie_conv_stats_tiles(%d, %d) = %d)", stats_tiles,
                    red_tiles, H, W, cutie_conv_stats_tiles(H, W));
    CUTIE_CHECK_ARG((reinterpret_cast<uintptr_t>(stats) & 7) == 0 && (reinterpret_cast<uintptr_t>(red) & 7) == 0,
                    "conv_dgrad_in_stats: stats and red must be 8-byte aligned");
    const NormBwd nb = {a, ld_a, a_dtype, stats, stats_tiles, eps, red, red_tiles, shift_bias, shift_act,
                        nullptr, 0, 0, nullptr, 0, nullptr, nullptr};
    return conv_fwd_impl(segs, n_segs, B, H, W, n_out, out, ld_out, out_dtype, nullptr, 0, nullptr, 0, dtype, &nb, stream_);
}

// measurement hook: the probe buffer the conv_wide launches of this process report into (nullptr: none)
static unsigned long long* g_clock_probe = nullptr;
namespace cutie { namespace convk { unsigned long long* conv_clock_probe() { return g_clock_probe; } } }
extern "C" int cutie_conv_clock_probe(void* probe) {
    CUTIE_CHECK_ARG((reinterpret_cast<uintptr_t>(probe) & 7) == 0, "conv_clock_probe: 8-byte aligned device words");
    g_clock_probe = static_cast<unsigned long long*>(probe);
    return CUTIE_OK;
}

extern "C" size_t cutie_conv_sync_bytes(int B, int red_tiles, int n_out) {
    if (B < 0 || red_tiles < 1 || n_out < 1) return 0;
    return 64 + (size_t)B * red_tiles * 2 * n_out * sizeof(unsigned long long);
}

extern "C" int cutie_conv_dgrad_in_bwd(const cutie_conv_seg* segs, int n_segs, int B, int H, int W, int n_out, void* dy,
                                       int ld_dy, const void* a, int ld_a, const float* stats, int stats_tiles, float eps,
                                       float* red, int red_tiles, int relu, const float* shift_bias, void* dz, int ld_dz,
                                       float* dbias_part, int dbias_tiles, void* sync, size_t sync_bytes,
                                       int32_t* late_count, int dtype, cutie_stream_t stream_) {
    CUTIE_CHECK_ARG(B >= 0 && H >= 1 && W >= 1 && n_out >= 1, "conv_dgrad_in_bwd: bad sizes");
    if (B == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(dy && a && stats && red && dz && ld_dy >= n_out && ld_a >= n_out && ld_dz >= n_out,
                    "conv_dgrad_in_bwd: null pointer or leading dimension < n_out");
    CUTIE_CHECK_ARG((relu & 15) <= 2 && (relu & ~(15 | CUTIE_ACT_FROM_Y)) == 0 && relu >= 0, "conv_dgrad_in_bwd: activation code %d", relu);
    CUTIE_CHECK_ARG(stats_tiles >= 1 && red_tiles >= cutie_conv_stats_tiles(H, W) && red_tiles <= 65535,
                    "conv_dgrad_in_bwd: stats_tiles=%d, red_tiles=%d (>= cutie_conv_stats_tiles(%d, %d) = %d)", stats_tiles,
                    red_tiles, H, W, cutie_conv_stats_tiles(H, W));
    CUTIE_CHECK_ARG(!dbias_part || (dbias_tiles >= 1 && dbias_tiles <= 65535), "conv_dgrad_in_bwd: dbias_tiles=%d", dbias_tiles);
    CUTIE_CHECK_ARG((reinterpret_cast<uintptr_t>(stats) & 7) == 0 && (reinterpret_cast<uintptr_t>(red) & 7) == 0,
                    "conv_dgrad_in_bwd: stats and red must be 8-byte aligned");
    if (sync && (sync_bytes < cutie_conv_sync_bytes(B, red_tiles, n_out) || (reinterpret_cast<uintptr_t>(sync) & 63))) {
        cutie::set_error("conv_dgrad_in_bwd: sync area %zu < %zu bytes (or not 64-byte aligned)", sync_bytes,
                         cutie_conv_sync_bytes(B, red_tiles, n_out));
        return CUTIE_E_WORKSPACE;
    }
    if (sync) {
        if (int e = sync_area_claim(sync, cutie_conv_sync_bytes(B, red_tiles, n_out), B, red_tiles, n_out, (hipStream_t)stream_)) return e;
    }
    const NormBwd nb = {a, ld_a, dtype, stats, stats_tiles, eps, red, red_tiles, shift_bias, relu, dz, ld_dz, relu, dbias_part,
                        dbias_tiles, (unsigned*)sync, late_count};
    const int rc = conv_fwd_impl(segs, n_segs, B, H, W, n_out, dy, ld_dy, dtype, nullptr, 0, nullptr, 0, dtype, &nb, stream_);
    if (rc == 1) return CUTIE_OK;      // one launch did all of it
    if (rc != CUTIE_OK) return rc;
    return cutie_in_bwd_apply(dy, ld_dy, dtype, a, ld_a, dtype, stats, stats_tiles, eps, red, red_tiles, relu, dz, ld_dz,
                              dtype, dbias_part, dbias_tiles, B, H * W, n_out, shift_bias, stream_);
}

namespace {
// which wgrad5 instance a layer runs on (canonical: the template arguments of the kernel)
struct Wg5Key {
    int kw, mask, narrow, wide;
    bool operator==(const Wg5Key& o) const { return kw == o.kw && mask == o.mask && narrow == o.narrow && wide == o.wide; }
};

// fills q (all but slabs / splits) and key; false when the layer is not a wgrad5 shape (f32, kw > 5, 2 GB views)
bool plan_wgrad5(const void* x, int ld_x, int cin, const void* dz, int ld_dz, int cout, int B, int H, int W, int kh, int kw,
                 int dil, int dtype, Wgrad5Params& q, Wg5Key& key) {
    const int es = 2;
    if (dtype == CUTIE_F32 || kw > WG5_KW || !(kw == 1 || kw == 3 || kw == 5)) return false;
    if ((long long)B * H * W * (ld_x > ld_dz ? ld_x : ld_dz) * es >= (1ll << 31)) return false;
    q.x = (const char*)x; q.dz = (const char*)dz; q.ld_x = ld_x; q.cin = cin; q.ld_dz = ld_dz; q.cout = cout;
    q.B = B; q.H = H; q.W = W; q.kh = kh; q.kw = kw; q.dil = dil;
    q.slabs = nullptr; q.splits = 1;
    q.mt = (int)cutie::cdiv(cin, kw == 1 ? 256 : 128);
    q.ntile = (int)cutie::cdiv(cout, (kw == 1 && cout > 64) ? 256 : 128);      // 1x1, not narrow: two cout tiles a workgroup
    q.nwseg = (int)cutie::cdiv(W, WG5_PIX);
    // halo columns: with one 64-pixel segment per row everything beyond the image is zero
    // padding and a live 32-pixel K block reaches at most |shift| mod 32 columns outside
    int pad = 0;
    for (int t = 0; t < kw; ++t) {
        int sx = (t - kw / 2) * dil;
        if (sx < 0) sx = -sx;
        const int need = q.nwseg == 1 ? sx % 32 : sx;
        if (need > pad) pad = need;
    }
    q.pad = pad;
    const int pwx = WG5_PIX + 2 * pad;
    const bool widepatch = pwx > WG5_XROWS;
    if (!(pwx <= WG5_XROWS || (pwx <= WG5_XROWS_WIDE && kw == 5 && cout > 64))) return false;
    // liveness pattern of the (K block, tap) pairs when a row is one 64-pixel segment
    int mask = -1;
    if (q.nwseg == 1 && kw == 5 && W > 32) {
        mask = 0;
        for (int ks = 0; ks < 2; ++ks)
            for (int t = 0; t < kw; ++t) {
                const int px0 = ks * 32, sx = (t - kw / 2) * dil;
                if (!(px0 + sx + 31 < 0 || px0 + sx >= W)) mask |= 1 << (ks * kw + t);
            }
    } else if (q.nwseg > 1 && kw == 5 && 2 * dil <= 31) {
        // several segments per row, every shift shorter than a K block: no (K block, tap) pair is ever wholly
        // outside on the left, and what lies beyond the right edge is staged as zeros - the all-live pattern
        // (compile-time, so the row loop pipelines) is exact
        mask = 0x3FF;
    }
    const bool narrow = cout <= 64;      // one cout tile, half of it (or more) padding
    key.kw = kw; key.wide = widepatch ? 1 : 0; key.narrow = (!widepatch && narrow) ? 1 : 0;
    if (widepatch) key.mask = -1;
    else if (narrow) key.mask = (kw == 5 && mask == 0x3FF) ? 0x3FF : -1;
    else key.mask = (kw == 5 && (mask == 0x3FF || mask == 0x1FE)) ? mask : -1;
    return true;
}

// calls f with the instance's template arguments as integral constants
template <typename F>
void wg5_dispatch(int dtype, const Wg5Key& k, F&& f) {
    using std::integral_constant;
    using std::bool_constant;
    auto with_dt = [&](auto kwc, auto maskc, auto narrowc, auto widec) {
        if (dtype == CUTIE_BF16) f(integral_constant<int, CUTIE_BF16>{}, kwc, maskc, narrowc, widec);
        else f(integral_constant<int, CUTIE_F16>{}, kwc, maskc, narrowc, widec);
    };
    typedef integral_constant<int, 5> K5;
    typedef integral_constant<int, 3> K3;
    typedef integral_constant<int, 1> K1;
    typedef integral_constant<int, -1> MG;
    typedef integral_constant<int, 0x3FF> MA;
    typedef integral_constant<int, 0x1FE> MD;
    typedef bool_constant<true> T;
    typedef bool_constant<false> N;
    if (k.wide) with_dt(K5{}, MG{}, N{}, T{});
    else if (k.narrow) {
        if (k.kw == 5 && k.mask == 0x3FF) with_dt(K5{}, MA{}, T{}, N{});
        else if (k.kw == 5) with_dt(K5{}, MG{}, T{}, N{});
        else if (k.kw == 3) with_dt(K3{}, MG{}, T{}, N{});
        else with_dt(K1{}, MG{}, T{}, N{});
    } else if (k.kw == 5 && k.mask == 0x3FF) with_dt(K5{}, MA{}, N{}, N{});
    else if (k.kw == 5 && k.mask == 0x1FE) with_dt(K5{}, MD{}, N{}, N{});
    else if (k.kw == 5) with_dt(K5{}, MG{}, N{}, N{});
    else if (k.kw == 3) with_dt(K3{}, MG{}, N{}, N{});
    else with_dt(K1{}, MG{}, N{}, N{});
}

// K splits for a batched launch: the same units per workgroup for every layer of the launch (s for the layer with the
// most units, proportionally fewer for the others); s minimises a time model - rounds of 256 workgroups x (rows per
// workgroup x 1.65 us + 8 us of prologue and slab store) + the slab bytes written and read back at 5 TB/s (constants
// from the single-layer kernel on the MI355X: 97 rows a workgroup in 160 us, tools/conv_bench.py)
int batched_splits(const Wgrad5Params* q, int n, int* splits) {
    long long umax = 1;
    int gmax = 1;
    for (int i = 0; i < n; ++i) {
        const long long u = (long long)q[i].B * q[i].H * q[i].nwseg;
        if (u > umax) umax = u;
        const int g = q[i].kh * q[i].mt * q[i].ntile;
        if (g > gmax) gmax = g;
    }
    auto split_of = [&](int i, int s) {
        const long long u = (long long)q[i].B * q[i].H * q[i].nwseg;
        long long si = (s * u + umax / 2) / umax;
        if (si > u) si = u;
        return (int)(si < 1 ? 1 : si);
    };
    double best = 1e30;
    int best_s = 1;
    for (int s = 1; s <= 256 && s <= umax; ++s) {
        long long pairs = 0;
        double slab_bytes = 0.0;
        for (int i = 0; i < n; ++i) {
            const int si = split_of(i, s);
            pairs += si;
            slab_bytes += (double)si * q[i].kh * q[i].kw * q[i].cin * q[i].cout * 4.0;
        }
        const long long wgs = ((pairs + 7) / 8) * 8 * gmax;       // what the launch dispatches (idle ones included)
        const long long rounds = (wgs + 255) / 256;
        const double t = rounds * ((double)umax / s * 1.65 + 8.0) + 2.0 * slab_bytes / 5e6;
        if (t < best) { best = t; best_s = s; }
    }
    for (int i = 0; i < n; ++i) splits[i] = split_of(i, best_s);
    return gmax;
}
}  // namespace

extern "C" size_t cutie_conv_wgrad_workspace_bytes(int B, int H, int W, int cin, int cout, int kh, int kw) {
    if (B < 1 || H < 1 || W < 1 || cin < 1 || cout < 1 || kh < 1 || kw < 1) return 0;
    int s = wgrad_splits(B, H, cin, cout, kh * kw);
    const int s5 = wgrad5_splits(B, H, W, cin, cout, kh, kw);
    if (s5 > s) s = s5;
    return (size_t)s * kh * kw * cin * cout * sizeof(float);
}

extern "C" int cutie_conv_wgrad(const void* x, int ld_x, int cin, const void* dz, int ld_dz, int cout, int B, int H,
                                int W, int kh, int kw, int dil, float* dw, int accumulate, int dtype, void* workspace,
                                size_t workspace_bytes, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(B >= 0 && H >= 1 && W >= 1 && cin >= 1 && cout >= 1, "conv_wgrad: bad sizes");
    CUTIE_CHECK_ARG(kh >= 1 && kw >= 1 && (kh & 1) && (kw & 1) && dil >= 1, "conv_wgrad: kernel %dx%d dil %d", kh, kw, dil);
    CUTIE_CHECK_ARG(dtype == CUTIE_F32 || dtype == CUTIE_BF16 || dtype == CUTIE_F16, "conv_wgrad: dtype %d", dtype);
    const int es = dtype == CUTIE_F32 ? 4 : 2;
    const int epc = 16 / es;
    CUTIE_CHECK_ARG(ld_x >= cin && ld_dz >= cout && (ld_x * es) % 16 == 0 && (ld_dz * es) % 16 == 0,
                    "conv_wgrad: rows must be 16-byte multiples (ld_x=%d ld_dz=%d)", ld_x, ld_dz);
    // a 16-byte chunk is loaded whenever its first channel is valid: the view must be readable
    // up to the chunk end (ld covers it) and hold zeros there
    CUTIE_CHECK_ARG(cutie::cdiv(cin, epc) * epc <= ld_x && cutie::cdiv(cout, epc) * epc <= ld_dz,
                    "conv_wgrad: ld must cover cin/cout rounded up to %d channels (zero padded)", epc);
    if (B == 0) {
        if (!accumulate) CUTIE_CHECK_HIP(hipMemsetAsync(dw, 0, (size_t)kh * kw * cin * cout * 4, stream));
        return CUTIE_OK;
    }
    CUTIE_CHECK_ARG(x && dz && dw && cutie::aligned16(x) && cutie::aligned16(dz), "conv_wgrad: null or misaligned pointer");
    const size_t need = cutie_conv_wgrad_workspace_bytes(B, H, W, cin, cout, kh, kw);
    if (!workspace || workspace_bytes < need) {
        cutie::set_error("conv_wgrad: workspace %zu < %zu", workspace_bytes, need);
        return CUTIE_E_WORKSPACE;
    }
    const long long n_w = (long long)kh * kw * cin * cout;
    long long g_red = cutie::cdiv((n_w & 3) == 0 ? n_w / 4 : n_w, 256);
    if (g_red > 4096) g_red = 4096;
    Wgrad5Params q;
    Wg5Key key;
    if (plan_wgrad5(x, ld_x, cin, dz, ld_dz, cout, B, H, W, kh, kw, dil, dtype, q, key)) {
        q.slabs = (float*)workspace;
        q.splits = wgrad5_splits(B, H, W, cin, cout, kh, kw);
        const dim3 grid5(kh * q.mt * q.ntile * q.splits);
        wg5_dispatch(dtype, key, [&](auto dt, auto kwc, auto maskc, auto narrowc, auto widec) {
            constexpr auto fn = wgrad5_kernel<decltype(dt)::value, decltype(kwc)::value, decltype(maskc)::value,
                                              decltype(narrowc)::value, decltype(widec)::value>;
            constexpr int lds5 = wg5_lds(decltype(kwc)::value, decltype(narrowc)::value);
            cutie::ensure_dyn_lds((const void*)fn, lds5);
            hipLaunchKernelGGL(fn, grid5, dim3(WG5_NT), lds5, stream, q);
        });
        CUTIE_CHECK_LAUNCH();
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)g_red), dim3(256), 0, stream, (const float*)workspace,
                           q.splits, n_w, dw, accumulate);
        CUTIE_CHECK_LAUNCH();
        return CUTIE_OK;
    }
    WgradParams p;
    p.x = (const char*)x; p.dz = (const char*)dz; p.ld_x = ld_x; p.cin = cin; p.ld_dz = ld_dz; p.cout = cout;
    p.B = B; p.H = H; p.W = W; p.kh = kh; p.kw = kw; p.dil = dil;
    p.slabs = (float*)workspace;
    p.splits = wgrad_splits(B, H, cin, cout, kh * kw);
    p.mt = (int)cutie::cdiv(cin, 128);
    p.ntile = (int)cutie::cdiv(cout, 128);
    const dim3 grid(kh * kw * p.mt * p.ntile, p.splits);
    const size_t lds = (size_t)2 * 64 * 128 * es;
    if (dtype == CUTIE_BF16) {
        hipLaunchKernelGGL(wgrad_kernel<CUTIE_BF16>, grid, dim3(256), lds, stream, p);
    } else if (dtype == CUTIE_F16) {
        hipLaunchKernelGGL(wgrad_kernel<CUTIE_F16>, grid, dim3(256), lds, stream, p);
    } else {
        hipLaunchKernelGGL(wgrad_kernel<CUTIE_F32>, grid, dim3(256), lds, stream, p);
    }
    CUTIE_CHECK_LAUNCH();
    const long long n = (long long)kh * kw * cin * cout;
    long long g = cutie::cdiv((n & 3) == 0 ? n / 4 : n, 256);
    if (g > 4096) g = 4096;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)g), dim3(256), 0, stream, (const float*)workspace, p.splits, n,
                       dw, accumulate);
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

namespace {
struct WgradPlan {      // how cutie_conv_wgrad_batched lays a call out: shared by the size query and the launch
    int n5;                              // layers on the wgrad5 path
    int idx[CUTIE_MAX_BATCH];            // their positions in descs, grouped by kernel instance
    int group_of[CUTIE_MAX_BATCH];       // group number of idx[k]
    Wgrad5Params q[CUTIE_MAX_BATCH];
    Wg5Key key[CUTIE_MAX_BATCH];
    size_t slab_off[CUTIE_MAX_BATCH];    // byte offset of the layer's slabs in the workspace
    size_t bytes5;                       // workspace of the wgrad5 layers
    size_t bytes_single;                 // largest workspace among the layers that take cutie_conv_wgrad by themselves
};

void plan_batched(const cutie_wgrad_desc* d, int n, int dtype, WgradPlan& pl) {
    pl.n5 = 0; pl.bytes5 = 0; pl.bytes_single = 0;
    Wgrad5Params q[CUTIE_MAX_BATCH];
    Wg5Key key[CUTIE_MAX_BATCH];
    bool ok[CUTIE_MAX_BATCH], taken[CUTIE_MAX_BATCH];
    for (int i = 0; i < n; ++i) {
        ok[i] = d[i].B > 0 && plan_wgrad5(d[i].x, d[i].ld_x, d[i].cin, d[i].dz, d[i].ld_dz, d[i].cout, d[i].B, d[i].H, d[i].W,
                                          d[i].kh, d[i].kw, d[i].dil, dtype, q[i], key[i]);
        taken[i] = false;
        if (!ok[i] && d[i].B > 0) {
            const size_t b = cutie_conv_wgrad_workspace_bytes(d[i].B, d[i].H, d[i].W, d[i].cin, d[i].cout, d[i].kh, d[i].kw);
            if (b > pl.bytes_single) pl.bytes_single = b;
        }
    }
    int group = 0;
    for (int i = 0; i < n; ++i) {
        if (!ok[i] || taken[i]) continue;
        const int first = pl.n5;
        int members = 0;
        for (int j = i; j < n && members < WG5_MAXB; ++j)
            if (ok[j] && !taken[j] && key[j] == key[i]) {
                taken[j] = true;
                pl.idx[pl.n5] = j; pl.group_of[pl.n5] = group; pl.q[pl.n5] = q[j]; pl.key[pl.n5] = key[j];
                ++pl.n5; ++members;
            }
        int splits[WG5_MAXB];
        batched_splits(pl.q + first, members, splits);
        for (int k = 0; k < members; ++k) {
            Wgrad5Params& qq = pl.q[first + k];
            qq.splits = splits[k];
            pl.slab_off[first + k] = pl.bytes5;
            pl.bytes5 += ((size_t)qq.splits * qq.kh * qq.kw * qq.cin * qq.cout * sizeof(float) + 255) & ~(size_t)255;
        }
        ++group;
    }
}
}  // namespace

extern "C" size_t cutie_conv_wgrad_batched_workspace_bytes(const cutie_wgrad_desc* descs, int n, int dtype) {
    if (!descs || n < 1 || n > CUTIE_MAX_BATCH) return 0;
    WgradPlan pl;
    plan_batched(descs, n, dtype, pl);
    return pl.bytes5 + pl.bytes_single;
}

extern "C" int cutie_conv_wgrad_batched(const cutie_wgrad_desc* descs, int n, int dtype, void* workspace, size_t workspace_bytes,
                                        cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(n >= 0 && n <= CUTIE_MAX_BATCH && (n == 0 || descs), "conv_wgrad_batched: n=%d (0..%d)", n, CUTIE_MAX_BATCH);
    CUTIE_CHECK_ARG(dtype == CUTIE_F32 || dtype == CUTIE_BF16 || dtype == CUTIE_F16, "conv_wgrad_batched: dtype %d", dtype);
    if (n == 0) return CUTIE_OK;
    const int es = dtype == CUTIE_F32 ? 4 : 2, epc = 16 / es;
    for (int i = 0; i < n; ++i) {
        const cutie_wgrad_desc& d = descs[i];
        CUTIE_CHECK_ARG(d.B >= 0 && d.H >= 1 && d.W >= 1 && d.cin >= 1 && d.cout >= 1, "conv_wgrad_batched[%d]: bad sizes", i);
        CUTIE_CHECK_ARG(d.kh >= 1 && d.kw >= 1 && (d.kh & 1) && (d.kw & 1) && d.dil >= 1, "conv_wgrad_batched[%d]: kernel %dx%d dil %d",
                        i, d.kh, d.kw, d.dil);
        CUTIE_CHECK_ARG(d.ld_x >= d.cin && d.ld_dz >= d.cout && (d.ld_x * es) % 16 == 0 && (d.ld_dz * es) % 16 == 0,
                        "conv_wgrad_batched[%d]: rows must be 16-byte multiples (ld_x=%d ld_dz=%d)", i, d.ld_x, d.ld_dz);
        CUTIE_CHECK_ARG(cutie::cdiv(d.cin, epc) * epc <= d.ld_x && cutie::cdiv(d.cout, epc) * epc <= d.ld_dz,
                        "conv_wgrad_batched[%d]: ld must cover cin/cout rounded up to %d channels (zero padded)", i, epc);
        CUTIE_CHECK_ARG(d.dw && (d.B == 0 || (d.x && d.dz && cutie::aligned16(d.x) && cutie::aligned16(d.dz))),
                        "conv_wgrad_batched[%d]: null or misaligned pointer", i);
    }
    WgradPlan pl;
    plan_batched(descs, n, dtype, pl);
    if (pl.bytes5 + pl.bytes_single > 0 && (!workspace || workspace_bytes < pl.bytes5 + pl.bytes_single ||
                                            (reinterpret_cast<uintptr_t>(workspace) & 255))) {
        cutie::set_error("conv_wgrad_batched: workspace %zu < %zu (or not 256-byte aligned)", workspace_bytes,
                         pl.bytes5 + pl.bytes_single);
        return CUTIE_E_WORKSPACE;
    }
    char* ws = (char*)workspace;
    // one launch per kernel instance
    for (int first = 0; first < pl.n5;) {
        int last = first;
        while (last < pl.n5 && pl.group_of[last] == pl.group_of[first]) ++last;
        Wgrad5Batch bt;
        bt.n = last - first;
        bt.gmax = 1;
        int at = 0;
        for (int k = first; k < last; ++k) {
            Wgrad5Params& q = pl.q[k];
            q.slabs = (float*)(ws + pl.slab_off[k]);
            bt.p[k - first] = q;
            bt.pair_start[k - first] = at;
            at += q.splits;
            if (q.kh * q.mt * q.ntile > bt.gmax) bt.gmax = q.kh * q.mt * q.ntile;
        }
        for (int k = bt.n; k <= WG5_MAXB; ++k) bt.pair_start[k] = at;
        for (int k = bt.n; k < WG5_MAXB; ++k) bt.p[k] = bt.p[0];
        const dim3 grid((unsigned)(((at + 7) / 8) * 8 * bt.gmax));
        wg5_dispatch(dtype, pl.key[first], [&](auto dt, auto kwc, auto maskc, auto narrowc, auto widec) {
            constexpr auto fn = wgrad5_batched_kernel<decltype(dt)::value, decltype(kwc)::value, decltype(maskc)::value,
                                                      decltype(narrowc)::value, decltype(widec)::value>;
            constexpr int lds5 = wg5_lds(decltype(kwc)::value, decltype(narrowc)::value);
            cutie::ensure_dyn_lds((const void*)fn, lds5);
            hipLaunchKernelGGL(fn, grid, dim3(WG5_NT), lds5, stream, bt);
        });
        CUTIE_CHECK_LAUNCH();
        first = last;
    }
    // one reduction launch over all of them (WG5_MAXB layers a launch)
    for (int first = 0; first < pl.n5; first += WG5_MAXB) {
        const int cnt = pl.n5 - first < WG5_MAXB ? pl.n5 - first : WG5_MAXB;
        ReduceBatch rb;
        rb.cnt = cnt;
        int at = 0;
        for (int k = 0; k < cnt; ++k) {
            const Wgrad5Params& q = pl.q[first + k];
            const cutie_wgrad_desc& d = descs[pl.idx[first + k]];
            const long long n_w = (long long)q.kh * q.kw * q.cin * q.cout;
            long long g = cutie::cdiv((n_w & 3) == 0 ? n_w / 4 : n_w, 256);
            if (g > 2048) g = 2048;
            rb.slabs[k] = q.slabs; rb.dw[k] = d.dw; rb.n[k] = n_w; rb.splits[k] = q.splits; rb.accumulate[k] = d.accumulate;
            rb.start[k] = at;
            at += (int)g;
        }
        for (int k = cnt; k <= WG5_MAXB; ++k) rb.start[k] = at;
        for (int k = cnt; k < WG5_MAXB; ++k) { rb.slabs[k] = nullptr; rb.dw[k] = nullptr; rb.n[k] = 0; rb.splits[k] = 0; rb.accumulate[k] = 0; }
        hipLaunchKernelGGL(wgrad_reduce_batched_kernel, dim3((unsigned)at), dim3(256), 0, stream, rb);
        CUTIE_CHECK_LAUNCH();
    }
    // what is not a wgrad5 shape (f32, wide kernels), and empty batches: one by one
    for (int i = 0; i < n; ++i) {
        bool in5 = false;
        for (int k = 0; k < pl.n5; ++k) in5 = in5 || pl.idx[k] == i;
        if (in5) continue;
        const cutie_wgrad_desc& d = descs[i];
        const int rc = cutie_conv_wgrad(d.x, d.ld_x, d.cin, d.dz, d.ld_dz, d.cout, d.B, d.H, d.W, d.kh, d.kw, d.dil, d.dw,
                                        d.accumulate, dtype, ws + pl.bytes5, workspace_bytes - pl.bytes5, stream_);
        if (rc != CUTIE_OK) return rc;
    }
    return CUTIE_OK;
}
