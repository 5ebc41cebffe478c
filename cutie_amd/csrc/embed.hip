// K1 / K13: embedding gather (+ its scatter-add gradient), batch gather, dtype convert.
// All HBM-bound row copies: one lane moves 4 consecutive elements, a row is read by
// consecutive lanes (model_framework.py:43-54, :73-86).
#include "common.h"

namespace {
using namespace cutie;

template <int ODT>
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int32_t* __restrict__ ids, int64_t n_cells,
                                                        const float* __restrict__ table, int V, int E4,
                                                        typename elem<ODT>::type* __restrict__ out,
                                                        int ld_out, int32_t* oob) {
    const int64_t total = n_cells * E4;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t cell = idx / E4;
        const int q = (int)(idx - cell * E4);
        const int id = ids[cell];
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (id >= 0 && id < V) {
            v = *reinterpret_cast<const float4*>(table + (int64_t)id * E4 * 4 + q * 4);
        } else if (q == 0) {
            atomicAdd(oob, 1);
        }
        store4(out + cell * ld_out + q * 4, v);
    }
}

// dtable[id] += dout[cell] (tf.nn.embedding_lookup's gradient, model_framework.py:51).
// What bounds a scatter-add into an embedding table is same-address float atomics (measured, tools/embed_bench.py:
// 8226 cells onto ONE row take 790 us, onto distinct rows 5 us), and a document grid is exactly that input: most
// cells hold id 0, a third of the tokens id 1 ([UNK]) and the rest follow a Zipf law.  So:
//   * ids 0 and 1 are summed in registers, then over the workgroup in LDS, and the workgroup's two partial rows are
//     STORED into `partials` [G][2][E] (folded in workgroup order by embed_bwd_fold_kernel: no atomics, same bits
//     every run).  Without a workspace they go out as one atomic per workgroup and element;
//   * every other id is first summed per workgroup in an LDS table of `slots` rows (open addressing on the id), so
//     that a frequent id costs one global atomic per workgroup and element instead of one per cell; a cell that
//     finds the table full adds straight into the global row.
// Bench batch (131072 cells, 8226 of them with an id >= 2): 96 -> 36 us; all of those cells on ONE id: 790 -> 42 us.
template <int DT>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const int32_t* __restrict__ ids, int64_t n_cells,
                                                        const typename elem<DT>::type* __restrict__ dout,
                                                        int ld, float* __restrict__ dtable, int V, int E4,
                                                        int cells_per_wg, float* __restrict__ partials, int slots) {
    __shared__ float4 red[2][256];
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    int* keys = reinterpret_cast<int*>(dyn);                               // [slots]
    float* rows = reinterpret_cast<float*>(dyn + ((slots * 4 + 15) & ~15)); // [slots][E]
    const int E = E4 * 4;
    for (int i = threadIdx.x; i < slots; i += 256) keys[i] = -1;
    for (int i = threadIdx.x; i < slots * E; i += 256) rows[i] = 0.f;
    __syncthreads();
    const int rows_par = 256 / E4;        // rows handled side by side
    const int q = threadIdx.x % E4;
    const int sub = threadIdx.x / E4;
    const bool active = sub < rows_par;
    float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f), acc1 = acc0;
    const int64_t c0 = (int64_t)blockIdx.x * cells_per_wg;
    const int64_t c1 = (c0 + cells_per_wg < n_cells) ? c0 + cells_per_wg : n_cells;
    if (active) {
        // four cells in flight per lane: the loads of a group are issued before any of them is used
        constexpr int U = 4;
        for (int64_t base = c0 + sub; base < c1; base += (int64_t)U * rows_par) {
            int id[U];
            float4 g[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t cell = base + (int64_t)u * rows_par;
                id[u] = cell < c1 ? ids[cell] : -1;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t cell = base + (int64_t)u * rows_par;
                g[u] = (id[u] >= 0 && id[u] < V) ? load4(dout + cell * ld + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (id[u] < 0 || id[u] >= V) continue;
                if (id[u] == 0) {
                    acc0.x += g[u].x; acc0.y += g[u].y; acc0.z += g[u].z; acc0.w += g[u].w;
                } else if (id[u] == 1) {
                    acc1.x += g[u].x; acc1.y += g[u].y; acc1.z += g[u].z; acc1.w += g[u].w;
                } else {
                    // every lane of the cell walks the same probe sequence, so all of them end on the same slot
                    int slot = -1;
                    if (slots > 0) {
                        unsigned h = ((unsigned)id[u] * 2654435761u) >> 8;
                        for (int pr = 0; pr < 8; ++pr) {
                            const int sidx = (int)((h + (unsigned)pr) % (unsigned)slots);
                            const int old = atomicCAS(&keys[sidx], -1, id[u]);
                            if (old == -1 || old == id[u]) { slot = sidx; break; }
                        }
                    }
                    float* dst = slot >= 0 ? rows + slot * E + q * 4 : dtable + (int64_t)id[u] * E + q * 4;
                    atomicAdd(dst + 0, g[u].x); atomicAdd(dst + 1, g[u].y);
                    atomicAdd(dst + 2, g[u].z); atomicAdd(dst + 3, g[u].w);
                }
            }
        }
    }
    red[0][threadIdx.x] = acc0;
    red[1][threadIdx.x] = acc1;
    __syncthreads();
    if (threadIdx.x < E4) {
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
        for (int s = 0; s < rows_par; ++s) {
            const float4 a = red[0][s * E4 + q], b = red[1][s * E4 + q];
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
        }
        if (partials) {
            float* p0 = partials + ((int64_t)blockIdx.x * 2) * E + q * 4;
            *reinterpret_cast<float4*>(p0) = s0;
            *reinterpret_cast<float4*>(p0 + E) = s1;
        } else {
            float* d0 = dtable + q * 4;
            atomicAdd(d0 + 0, s0.x); atomicAdd(d0 + 1, s0.y); atomicAdd(d0 + 2, s0.z); atomicAdd(d0 + 3, s0.w);
            if (V > 1) {
                float* d1 = dtable + (int64_t)E + q * 4;
                atomicAdd(d1 + 0, s1.x); atomicAdd(d1 + 1, s1.y); atomicAdd(d1 + 2, s1.z); atomicAdd(d1 + 3, s1.w);
            }
        }
    }
    // the workgroup's table: one global atomic per occupied slot and element
    for (int i = threadIdx.x; i < slots * E; i += 256) {
        const int key = keys[i / E];
        if (key >= 0) atomicAdd(dtable + (int64_t)key * E + (i % E), rows[i]);
    }
}

// rows 0 and 1 of dtable += the workgroups' partial rows, in a fixed order: a workgroup owns 8 of the 2E elements,
// 32 lanes per element each add every 32nd partial, the 32 lane sums are added in lane order
__global__ __launch_bounds__(256) void embed_bwd_fold_kernel(const float* __restrict__ partials, int G, int E, int V,
                                                             float* __restrict__ dtable) {
    __shared__ float sh[32][8];
    const int el = threadIdx.x & 7, part = threadIdx.x >> 3;
    const int i = blockIdx.x * 8 + el;                  // 0 .. 2E-1: (row, element)
    float s = 0.f;
    if (i < 2 * E) {
#pragma unroll 4
        for (int g = part; g < G; g += 32) s += partials[(int64_t)g * 2 * E + i];
    }
    sh[part][el] = s;
    __syncthreads();
    if (part == 0 && i < 2 * E && !(i >= E && V < 2)) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) t += sh[k][el];
        dtable[i] += t;
    }
}

template <int DT>
__global__ __launch_bounds__(256) void batch_gather_kernel(const typename elem<DT>::type* __restrict__ feat,
                                                           int ld_feat, int n_src, const int32_t* __restrict__ idx,
                                                           int n_cells, int B, int C4,
                                                           typename elem<DT>::type* __restrict__ out, int ld_out,
                                                           int32_t* __restrict__ oob) {
    const int64_t total = (int64_t)B * n_cells * C4;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t cell = t / C4;  // b*n_cells + cell
        const int q = (int)(t - cell * C4);
        const int b = (int)(cell / n_cells);
        int s = idx[cell];
        if (oob && q == 0 && (s < 0 || s >= n_src)) atomicAdd(oob, 1);      // tf.batch_gather raises; here it is counted
        s = s < 0 ? 0 : (s >= n_src ? n_src - 1 : s);
        const float4 v = load4(feat + ((int64_t)b * n_src + s) * ld_feat + q * 4);
        store4(out + cell * ld_out + q * 4, v);
    }
}

// backward of batch_gather: dfeat[b, idx[b,cell], :] += dout[b, cell, :] (f32 accumulate).  Grid cells
// without a token all point at pixel 0 (data_loader_json.py:554-555 leaves ps_x = ps_y = 0), so
// that row is summed in registers, then LDS, then one atomic per workgroup and channel.
template <int DT>
__global__ __launch_bounds__(256) void batch_scatter_add_kernel(const typename elem<DT>::type* __restrict__ dout,
                                                                int ld_dout, const int32_t* __restrict__ idx,
                                                                int n_cells, int n_src, int C4,
                                                                float* __restrict__ dfeat, int cells_per_wg) {
    __shared__ float4 red[256];
    const int b = blockIdx.y;
    const int rows_par = 256 / C4;
    const int q = threadIdx.x % C4;
    const int sub = threadIdx.x / C4;
    float4 acc0 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c0 = blockIdx.x * cells_per_wg;
    const int c1 = min(n_cells, c0 + cells_per_wg);
    if (sub < rows_par) {
        for (int cell = c0 + sub; cell < c1; cell += rows_par) {
            int s = idx[(int64_t)b * n_cells + cell];
            s = s < 0 ? 0 : (s >= n_src ? n_src - 1 : s);
            const float4 g = load4(dout + ((int64_t)b * n_cells + cell) * ld_dout + q * 4);
            if (s == 0) {
                acc0.x += g.x; acc0.y += g.y; acc0.z += g.z; acc0.w += g.w;
            } else {
                float* dst = dfeat + (((int64_t)b * n_src + s) * C4 + q) * 4;
                atomicAdd(dst + 0, g.x); atomicAdd(dst + 1, g.y);
                atomicAdd(dst + 2, g.z); atomicAdd(dst + 3, g.w);
            }
        }
    }
    red[threadIdx.x] = acc0;
    __syncthreads();
    if (threadIdx.x < C4) {
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int s = 0; s < rows_par; ++s) {
            const float4 a = red[s * C4 + q];
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
        float* d0 = dfeat + ((int64_t)b * n_src * C4 + q) * 4;
        atomicAdd(d0 + 0, s0.x); atomicAdd(d0 + 1, s0.y); atomicAdd(d0 + 2, s0.z); atomicAdd(d0 + 3, s0.w);
    }
}

template <int SDT, int DDT>
__global__ __launch_bounds__(256) void convert_kernel(const typename elem<SDT>::type* __restrict__ src, int ld_src,
                                                      typename elem<DDT>::type* __restrict__ dst, int ld_dst,
                                                      int64_t n_rows, int C4) {
    const int64_t total = n_rows * C4;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / C4;
        const int q = (int)(t - r * C4);
        store4(dst + r * ld_dst + q * 4, load4(src + r * ld_src + q * 4));
    }
}

inline int grid_for(int64_t work_items) {
    int64_t g = cutie::cdiv(work_items, 256);
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

extern "C" int cutie_embed_fwd(const int32_t* ids, int64_t n_cells, const float* table, int V, int E,
                               void* out, int ld_out, int out_dtype, int32_t* oob_count,
                               cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(n_cells >= 0 && V >= 1 && E >= 4 && (E % 4) == 0, "embed_fwd: n_cells=%lld V=%d E=%d (E %% 4 == 0 needed)",
                    (long long)n_cells, V, E);
    CUTIE_CHECK_ARG(ld_out >= E && (ld_out % 4) == 0, "embed_fwd: ld_out=%d must be >= E and a multiple of 4", ld_out);
    if (n_cells == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(ids && table && out && oob_count, "embed_fwd: null pointer");
    CUTIE_CHECK_ARG(cutie::aligned16(table) && ((uintptr_t)out % 8) == 0, "embed_fwd: misaligned pointers");
    const int g = grid_for(n_cells * (E / 4));
    CUTIE_DISPATCH_DTYPE(out_dtype, ODT,
        hipLaunchKernelGGL(embed_fwd_kernel<ODT>, dim3(g), dim3(256), 0, stream, ids, n_cells, table, V, E / 4,
                           (typename elem<ODT>::type*)out, ld_out, oob_count));
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

namespace {
struct EmbedBwdPlan { int groups, cells_per_wg, slots; size_t lds; };
inline EmbedBwdPlan embed_bwd_plan(int64_t n_cells, int E) {
    EmbedBwdPlan p;
    // ~128 cells a workgroup, at most 1024 workgroups: short per-workgroup loops (the pass is latency-bound) now that
    // the hot rows no longer cost one same-address atomic per workgroup
    constexpr int cells = 128, max_slots = 16;
    int64_t g = cutie::cdiv(n_cells, cells);
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    p.groups = (int)g;
    p.cells_per_wg = (int)cutie::cdiv(n_cells, g);
    // 16 slots (8 KB of LDS rows at E = 128): a workgroup's 128 cells of a document grid hold about ten tokens, and
    // clearing / flushing a larger table costs more than it saves (64 slots: +14 us on the bench batch)
    p.slots = 32768 / (E * 4);
    if (p.slots > max_slots) p.slots = max_slots;
    p.lds = ((size_t)p.slots * 4 + 15) / 16 * 16 + (size_t)p.slots * E * 4;
    return p;
}
}  // namespace

extern "C" size_t cutie_embed_bwd_workspace_bytes(int64_t n_cells, int E) {
    if (n_cells < 1 || E < 4) return 0;
    return (size_t)embed_bwd_plan(n_cells, E).groups * 2 * E * sizeof(float);
}

extern "C" int cutie_embed_bwd(const int32_t* ids, int64_t n_cells, const void* dout, int ld_dout,
                               int dout_dtype, float* dtable, int V, int E, void* workspace, size_t workspace_bytes,
                               cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(n_cells >= 0 && V >= 1 && E >= 4 && (E % 4) == 0 && E <= 1024, "embed_bwd: n_cells=%lld V=%d E=%d",
                    (long long)n_cells, V, E);
    CUTIE_CHECK_ARG(ld_dout >= E && (ld_dout % 4) == 0, "embed_bwd: ld_dout=%d", ld_dout);
    if (n_cells == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(ids && dout && dtable, "embed_bwd: null pointer");
    const EmbedBwdPlan pl = embed_bwd_plan(n_cells, E);
    float* partials = nullptr;
    if (workspace) {
        if (workspace_bytes < cutie_embed_bwd_workspace_bytes(n_cells, E) || !cutie::aligned16(workspace)) {
            cutie::set_error("embed_bwd: workspace %zu < %zu (or not 16-byte aligned)", workspace_bytes,
                             cutie_embed_bwd_workspace_bytes(n_cells, E));
            return CUTIE_E_WORKSPACE;
        }
        partials = (float*)workspace;
    }
    CUTIE_DISPATCH_DTYPE(dout_dtype, DT,
        hipLaunchKernelGGL(embed_bwd_kernel<DT>, dim3(pl.groups), dim3(256), pl.lds, stream, ids, n_cells,
                           (const typename elem<DT>::type*)dout, ld_dout, dtable, V, E / 4, pl.cells_per_wg, partials,
                           pl.slots));
    CUTIE_CHECK_LAUNCH();
    if (partials) {
        hipLaunchKernelGGL(embed_bwd_fold_kernel, dim3((2 * E + 7) / 8), dim3(256), 0, stream, partials, pl.groups, E, V,
                           dtable);
        CUTIE_CHECK_LAUNCH();
    }
    return CUTIE_OK;
}

extern "C" int cutie_batch_gather(const void* feat, int ld_feat, int n_src, const int32_t* idx, int n_cells,
                                  int B, int C, void* out, int ld_out, int dtype, int32_t* oob_count,
                                  cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(B >= 0 && n_cells >= 0 && n_src >= 1 && C >= 4 && (C % 4) == 0, "batch_gather: bad sizes");
    CUTIE_CHECK_ARG((ld_feat % 4) == 0 && (ld_out % 4) == 0 && ld_feat >= C && ld_out >= C, "batch_gather: bad ld");
    if (B == 0 || n_cells == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(feat && idx && out, "batch_gather: null pointer");
    const int g = grid_for((int64_t)B * n_cells * (C / 4));
    CUTIE_DISPATCH_DTYPE(dtype, DT,
        hipLaunchKernelGGL(batch_gather_kernel<DT>, dim3(g), dim3(256), 0, stream,
                           (const typename elem<DT>::type*)feat, ld_feat, n_src, idx, n_cells, B, C / 4,
                           (typename elem<DT>::type*)out, ld_out, oob_count));
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

extern "C" int cutie_batch_scatter_add(const void* dout, int ld_dout, int dtype, const int32_t* idx, int n_cells,
                                       int n_src, int B, int C, float* dfeat, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(B >= 0 && n_cells >= 0 && n_src >= 1 && C >= 4 && (C % 4) == 0 && C <= 1024, "batch_scatter_add: bad sizes");
    CUTIE_CHECK_ARG((ld_dout % 4) == 0 && ld_dout >= C, "batch_scatter_add: ld_dout=%d", ld_dout);
    if (B == 0 || n_cells == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(dout && idx && dfeat && cutie::aligned16(dfeat), "batch_scatter_add: null or misaligned pointer");
    CUTIE_CHECK_ARG(B <= 65535, "batch_scatter_add: B=%d", B);
    const int cells_per_wg = 512;
    const dim3 grid((unsigned)cutie::cdiv(n_cells, cells_per_wg), (unsigned)B);
    CUTIE_DISPATCH_DTYPE(dtype, DT,
        hipLaunchKernelGGL(batch_scatter_add_kernel<DT>, grid, dim3(256), 0, stream,
                           (const typename elem<DT>::type*)dout, ld_dout, idx, n_cells, n_src, C / 4, dfeat,
                           cells_per_wg));
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}

extern "C" int cutie_convert(const void* src, int ld_src, int src_dtype, void* dst, int ld_dst, int dst_dtype,
                             int64_t n_rows, int C, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(n_rows >= 0 && C >= 4 && (C % 4) == 0, "convert: C=%d must be a positive multiple of 4", C);
    CUTIE_CHECK_ARG((ld_src % 4) == 0 && (ld_dst % 4) == 0 && ld_src >= C && ld_dst >= C, "convert: bad ld");
    if (n_rows == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(src && dst, "convert: null pointer");
    const int g = grid_for(n_rows * (C / 4));
    CUTIE_DISPATCH_DTYPE(src_dtype, SDT, CUTIE_DISPATCH_DTYPE(dst_dtype, DDT,
        hipLaunchKernelGGL((convert_kernel<SDT, DDT>), dim3(g), dim3(256), 0, stream,
                           (const typename elem<SDT>::type*)src, ld_src, (typename elem<DDT>::type*)dst, ld_dst,
                           n_rows, C / 4)));
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}
