// K0: OCR tokens -> rows x cols index grid.  One workgroup per document.
//
// Follows DataLoader._positional_mapping (/root/reference/data_loader_json.py:437-577):
//   keys in IEEE double with the reference's operation order (:499-504), the stable sort of
//   :507 as a rank-by-counting over the total order (row, v_c, h_c, dict_id, input index),
//   then one lane per grid row replays the sequential slot search / left shift (:512-557).
// Rows are independent once sorted (strategies 0 and 1 never touch another row), so the
// sequential part is at most one row's tokens long.  The working grid lives in LDS.
#include "common.h"

#pragma clang fp contract(off)

namespace {

struct GridArgs {
    const int32_t* tok_box;
    const int32_t* tok_id;
    const int32_t* tok_ecid;
    const int32_t* doc_extent;
    const int32_t* doc_off;
    int rows, cols, strategy;
    int32_t* grid;
    int8_t* labels;
    int32_t* ps_x;
    int32_t* ps_y;
    int32_t* tok_order;
    int32_t* tok_place;
    int32_t* status;
    double* key_v;   // workspace [T]
    double* key_h;   // workspace [T]
    int32_t* key_row;  // workspace [T]
    int32_t* key_col;  // workspace [T]
    int cap;           // tokens per document whose scratch fits in LDS
};

__device__ __forceinline__ bool key_less(int rj, double vj, double hj, int dj, int j,
                                         int ri, double vi, double hi, int di, int i) {
    if (rj != ri) return rj < ri;
    if (vj != vi) return vj < vi;
    if (hj != hi) return hj < hi;
    if (dj != di) return dj < di;
    return j < i;
}

constexpr int GB_NT = 1024;

__global__ __launch_bounds__(GB_NT) void grid_build_kernel(GridArgs a) {
    extern __shared__ __attribute__((aligned(16))) int32_t lds[];
    const int rows = a.rows, cols = a.cols;
    int32_t* g = lds;                       // [rows*cols]
    int32_t* row_start = lds + rows * cols; // [rows]
    int32_t* row_cnt = row_start + rows;    // [rows]
    int32_t* flag = row_cnt + rows;         // [1] (+ pad to 8 bytes)

    const int doc = blockIdx.x;
    const int t0 = a.doc_off[doc];
    const int n = a.doc_off[doc + 1] - t0;
    const int tid = threadIdx.x;

    // per-token scratch (keys, ids, sorted order): in LDS when the document fits the reserved
    // capacity (every lane re-reads all keys in the rank loop), else in the caller's workspace
    const bool in_lds = n <= a.cap;
    char* kbase = reinterpret_cast<char*>(lds) + (((size_t)(rows * cols + 2 * rows + 2) * 4 + 7) & ~(size_t)7);
    double* key_v = in_lds ? reinterpret_cast<double*>(kbase) : a.key_v + t0;
    double* key_h = in_lds ? key_v + a.cap : a.key_h + t0;
    int32_t* key_row = in_lds ? reinterpret_cast<int32_t*>(key_h + a.cap) : a.key_row + t0;
    int32_t* key_col = in_lds ? key_row + a.cap : a.key_col + t0;
    int32_t* key_id = in_lds ? key_col + a.cap : const_cast<int32_t*>(a.tok_id) + t0;
    int32_t* order = in_lds ? key_id + a.cap : a.tok_order + t0;
    // what the placement lanes need per token besides the keys: in LDS too, so that the sequential walk
    // of a row never waits for a global load
    int32_t* t_ecid = in_lds ? order + a.cap : nullptr;
    int32_t* t_px = in_lds ? t_ecid + a.cap : nullptr;
    int32_t* t_py = in_lds ? t_px + a.cap : nullptr;
    int32_t* members = in_lds ? t_py + a.cap : nullptr;     // token indices grouped by grid row

    // the document's output cells start at zero (the barrier below orders these stores before the
    // placement lanes' stores to the same cells)
    {
        const size_t c0 = (size_t)doc * rows * cols;
        for (int i = tid; i < rows * cols; i += GB_NT) {
            g[i] = 0;
            a.labels[c0 + i] = 0;
            a.ps_x[c0 + i] = 0;
            a.ps_y[c0 + i] = 0;
        }
    }
    for (int i = tid; i < rows; i += GB_NT) { row_start[i] = -1; row_cnt[i] = 0; }
    if (tid == 0) *flag = 0;
    __syncthreads();

    const int left = a.doc_extent[doc * 4 + 0], top = a.doc_extent[doc * 4 + 1];
    const int right = a.doc_extent[doc * 4 + 2], bottom = a.doc_extent[doc * 4 + 3];

    // ---- keys (:499-504)
    for (int i = tid; i < n; i += GB_NT) {
        const int4 box = *reinterpret_cast<const int4*>(a.tok_box + (size_t)(t0 + i) * 4);
        const int x_left = box.x, y_top = box.y, x_right = box.z, y_bottom = box.w;
        bool bad = (bottom == top) || (right == left);
        double v_c = 0.0, h_c = 0.0;
        int row = 0, col = 0;
        if (!bad) {
            v_c = ((double)(y_top - top) + (double)(y_bottom - y_top) / 2.0) / (double)(bottom - top);
            h_c = ((double)(x_left - left) + (double)(x_right - x_left) / 2.0) / (double)(right - left);
            const double fr = (double)rows * v_c, fc = (double)cols * h_c;
            // int() truncates toward zero; anything that lands outside the grid is an error
            bad = !(fr > -1.0 && fr < (double)rows && fc > -1.0 && fc < (double)cols);
            if (!bad) {  // (-1,0) truncates to 0 and stays inside, as Python's int() does
                row = (int)fr;
                col = (int)fc;
            }
        }
        if (bad) atomicOr(flag, 1);
        key_v[i] = v_c;
        key_h[i] = h_c;
        key_row[i] = row;
        key_col[i] = col;
        if (in_lds) {
            key_id[i] = a.tok_id[t0 + i];
            t_ecid[i] = a.tok_ecid[t0 + i];
            t_px[i] = x_left;
            t_py[i] = (int)((double)y_top + (double)(y_bottom - y_top) / 2.0);
        }
    }
    __syncthreads();  // workgroup-scope fence: the keys are re-read below by other lanes
    if (*flag) {
        if (tid == 0) a.status[doc] = 1;
        for (int i = tid; i < rows * cols; i += GB_NT) a.grid[(size_t)doc * rows * cols + i] = 0;
        for (int i = tid; i < n; i += GB_NT) {      // a rejected document places nothing
            a.tok_order[t0 + i] = -1;
            a.tok_place[(t0 + i) * 2 + 0] = -1;
            a.tok_place[(t0 + i) * 2 + 1] = -1;
        }
        return;
    }

    // ---- stable sort (:507) as a rank by counting over the total order (row, v_c, h_c, dict_id, input index)
    if (in_lds) {
        // The row is the leading key, so a token's rank is (tokens in earlier rows) + (its rank inside its own row):
        // group the tokens by row first and compare each only with its row mates - about n * (tokens per row)
        // comparisons instead of n * n, which on ONE compute unit was 70 of this kernel's 113 us.
        __shared__ int wave_tot[GB_NT / 64];
        for (int i = tid; i < n; i += GB_NT) order[i] = atomicAdd(&row_cnt[key_row[i]], 1);   // slot inside the row's group
        __syncthreads();
        // exclusive scan of the row populations (rows <= 1024 = one lane each)
        const int lane = tid & 63, wave = tid >> 6;
        const int cnt = tid < rows ? row_cnt[tid] : 0;
        int incl = cnt;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d, 64);
            if (lane >= d) incl += up;
        }
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int off = incl - cnt;
        for (int w = 0; w < wave; ++w) off += wave_tot[w];
        if (tid < rows && cnt > 0) row_start[tid] = off;      // first sorted position of the row
        __syncthreads();
        for (int i = tid; i < n; i += GB_NT) members[row_start[key_row[i]] + order[i]] = i;
        __syncthreads();
        for (int i = tid; i < n; i += GB_NT) {
            const int ri = key_row[i];
            const double vi = key_v[i], hi = key_h[i];
            const int di = key_id[i];
            const int m0 = row_start[ri], m1 = m0 + row_cnt[ri];
            int rank = m0;
            for (int m = m0; m < m1; ++m) {
                const int j = members[m];
                rank += key_less(ri, key_v[j], key_h[j], key_id[j], j, ri, vi, hi, di, i) ? 1 : 0;
            }
            order[rank] = i;      // the slots kept in order[] were all read before the barrier above
        }
        __syncthreads();
    } else {
        // a document too long for the LDS scratch: all lanes read key j together (broadcast), n * n comparisons
        for (int i = tid; i < n; i += GB_NT) {
            const int ri = key_row[i];
            const double vi = key_v[i], hi = key_h[i];
            const int di = key_id[i];
            int rank = 0;
#pragma unroll 4
            for (int j = 0; j < n; ++j) {
                rank += key_less(key_row[j], key_v[j], key_h[j], key_id[j], j, ri, vi, hi, di, i) ? 1 : 0;
            }
            order[rank] = i;
        }
        __syncthreads();

        // ---- first sorted position of every row
        for (int p = tid; p < n; p += GB_NT) {
            const int r = key_row[order[p]];
            if (p == 0 || key_row[order[p - 1]] != r) row_start[r] = p;
        }
        __syncthreads();
    }

    // ---- one lane per row replays the sequential placement (:508-557)
    const size_t cell0 = (size_t)doc * rows * cols;
    for (int r = tid; r < rows; r += GB_NT) {
        int p = row_start[r];
        if (p < 0) continue;
        int32_t* gr = g + r * cols;
        for (; p < n; ++p) {
            const int i = order[p];
            if (key_row[i] != r) break;
            int col = key_col[i];
            const int dict_id = key_id[i];
            while (col < cols && gr[col] != 0) ++col;
            bool write = true;
            if (a.strategy == 0) {
                if (col == cols) write = false;  // :519 overlap, token skipped
            } else {
                if (col == cols) {                // :532-539 shift left from the first free cell
                    col -= 1;
                    int ptr = 0;
                    while (ptr < cols && gr[ptr] != 0) ++ptr;
                    if (ptr == cols) ptr = 0;
                    for (int c = ptr; c < cols - 1; ++c) gr[c] = gr[c + 1];
                    gr[cols - 1] = 0;             // what the reference pulls in from its buffer half
                }
            }
            if (write) {
                gr[col] = dict_id;
                a.labels[cell0 + r * cols + col] = (int8_t)(in_lds ? t_ecid[i] : a.tok_ecid[t0 + i]);  // int8 wrap
                if (a.strategy != 0) {
                    int px, py;
                    if (in_lds) {
                        px = t_px[i];
                        py = t_py[i];
                    } else {
                        const int y_top = a.tok_box[(t0 + i) * 4 + 1], y_bottom = a.tok_box[(t0 + i) * 4 + 3];
                        px = a.tok_box[(t0 + i) * 4 + 0];
                        py = (int)((double)y_top + (double)(y_bottom - y_top) / 2.0);
                    }
                    a.ps_x[cell0 + r * cols + col] = px;
                    a.ps_y[cell0 + r * cols + col] = py;
                }
                a.tok_place[(t0 + p) * 2 + 0] = r;
                a.tok_place[(t0 + p) * 2 + 1] = col;
            } else {
                a.tok_place[(t0 + p) * 2 + 0] = -1;
                a.tok_place[(t0 + p) * 2 + 1] = -1;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < rows * cols; i += GB_NT) a.grid[cell0 + i] = g[i];
    if (in_lds) {
        for (int i = tid; i < n; i += GB_NT) a.tok_order[t0 + i] = order[i];
    }
    if (tid == 0) a.status[doc] = 0;
}

}  // namespace

extern "C" size_t cutie_grid_build_workspace_bytes(int n_tokens) {
    size_t t = (size_t)(n_tokens > 0 ? n_tokens : 1);
    return t * (8 + 8 + 4 + 4) + 64;
}

extern "C" int cutie_grid_build(const int32_t* tok_box, const int32_t* tok_id, const int32_t* tok_ecid,
                                const int32_t* doc_extent, const int32_t* doc_off, int n_docs,
                                int n_tokens, int rows, int cols, int strategy, int32_t* grid,
                                int8_t* labels, int32_t* ps_x, int32_t* ps_y, int32_t* tok_order,
                                int32_t* tok_place, int32_t* status, void* workspace,
                                size_t workspace_bytes, cutie_stream_t stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    CUTIE_CHECK_ARG(n_docs >= 0 && n_tokens >= 0, "grid_build: negative sizes");
    CUTIE_CHECK_ARG(rows >= 1 && cols >= 1 && rows <= 1024, "grid_build: rows=%d cols=%d out of range", rows, cols);
    CUTIE_CHECK_ARG(strategy >= 0 && strategy <= 2, "grid_build: positional_mapping_strategy %d (0, 1 or 2)", strategy);
    // Strategy 2 ("expand grid table if overlap", data_loader_json.py:541-550) searches for a free slot
    // from column cols-1 AFTER the strategy-1 left shift (:532-539) has freed exactly that column, so
    // it never moves and `cols` never grows: its outputs are strategy 1's.  The reference itself
    // confirms it on every fixture document (tests/golden/strategy2_cases.json).
    if (strategy == 2) strategy = 1;
    const size_t lds_bytes = ((size_t)rows * cols + 2 * (size_t)rows + 2) * sizeof(int32_t);
    CUTIE_CHECK_ARG(lds_bytes + 64 <= 160 * 1024, "grid_build: %dx%d grid needs %zu B of LDS (max 163776)", rows,
                    cols, lds_bytes);
    if (n_docs == 0) return CUTIE_OK;
    CUTIE_CHECK_ARG(grid && labels && ps_x && ps_y && tok_order && tok_place && status && doc_off && doc_extent,
                    "grid_build: null pointer");
    CUTIE_CHECK_ARG(n_tokens == 0 || (tok_box && tok_id && tok_ecid), "grid_build: null token arrays");
    CUTIE_CHECK_ARG(cutie::aligned16(tok_box), "grid_build: tok_box must be 16-byte aligned");
    if (workspace_bytes < cutie_grid_build_workspace_bytes(n_tokens) || !workspace) {
        cutie::set_error("grid_build: workspace %zu < %zu", workspace_bytes,
                         cutie_grid_build_workspace_bytes(n_tokens));
        return CUTIE_E_WORKSPACE;
    }
    GridArgs a;
    a.tok_box = tok_box; a.tok_id = tok_id; a.tok_ecid = tok_ecid;
    a.doc_extent = doc_extent; a.doc_off = doc_off;
    a.rows = rows; a.cols = cols; a.strategy = strategy;
    a.grid = grid; a.labels = labels; a.ps_x = ps_x; a.ps_y = ps_y;
    a.tok_order = tok_order; a.tok_place = tok_place; a.status = status;
    const size_t t = (size_t)(n_tokens > 0 ? n_tokens : 1);
    char* ws = (char*)(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
    a.key_v = (double*)ws;
    a.key_h = a.key_v + t;
    a.key_row = (int32_t*)(a.key_h + t);
    a.key_col = a.key_row + t;

    // per-token scratch in LDS: 48 bytes a token, up to 2048 tokens a document or what is left
    const size_t grid_part = (lds_bytes + 7) & ~(size_t)7;
    size_t cap = (160 * 1024 - 64 - grid_part) / 48;       // 64: the kernel's static wave totals
    if (cap > 2048) cap = 2048;
    a.cap = (int)cap;
    const size_t lds_total = grid_part + cap * 48;
    CUTIE_ENSURE_LDS(grid_build_kernel, lds_total);
    hipLaunchKernelGGL(grid_build_kernel, dim3(n_docs), dim3(GB_NT), lds_total, stream, a);
    CUTIE_CHECK_LAUNCH();
    return CUTIE_OK;
}
