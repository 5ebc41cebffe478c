/* cutie_hip.h - C ABI of libcutie_hip.so: the MI355X (gfx950) kernels behind CUTIE's hot path.
 *
 * The reference (vsymbol/CUTIE, TensorFlow 1.12) has no FFI: its boundary is the Python API
 * between the drivers and model_framework.Model / DataLoader.  The
 * Python mirror of that API lives in cutie_amd/ and calls the entry points below through
 * ctypes.  Each entry point cites the reference call site whose arithmetic it replaces
 * (paths relative to /root/reference).
 *
 * Conventions
 *   - every function returns 0 on success or a negative CUTIE_E_* code; the message of the
 *     last failure on the calling thread is cutie_last_error().  Nothing throws or aborts.
 *   - the caller owns every buffer (device pointers unless marked "host"); functions that
 *     need scratch take a workspace and publish its size through *_workspace_bytes().
 *   - every launch goes to the hipStream_t passed in (as void*), asynchronously.
 *   - activations are NHWC with an explicit leading dimension: element (b,h,w,c) of a view
 *     lives at base[((b*H + h)*W + w)*ld + c]; a concat is a wider buffer whose producers
 *     write channel slices (model_framework.py:294-296 needs no kernel).
 *   - dtype: CUTIE_F32, CUTIE_BF16 or CUTIE_F16 activations; statistics, losses, parameters and
 *     gradients are always f32; ids/labels are int32 (labels in the grid are int8).
 */
#ifndef CUTIE_HIP_H
#define CUTIE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CUTIE_F32 0
#define CUTIE_BF16 1
#define CUTIE_F16 2  /* IEEE half activations; training uses the loss scale of K12 below */

#define CUTIE_OK 0
#define CUTIE_E_ARG (-1)      /* bad shape / pointer / alignment / unsupported option */
#define CUTIE_E_HIP (-2)      /* a HIP runtime call failed */
#define CUTIE_E_WORKSPACE (-3)/* workspace too small */

#define CUTIE_MAX_SEGS 6

/* Activation codes: 0 none, 1 ReLU, 2 sigmoid.  Backward entry points that take a layer's forward activation `a` accept
 * the code OR-ed with CUTIE_ACT_FROM_Y: the operand is then the layer's NORMALISED output y as the one-launch form of
 * cutie_conv_in_fwd stores it (xhat = y; the ReLU pattern is read out of y, see that entry point). */
#define CUTIE_ACT_FROM_Y 16

typedef void* cutie_stream_t; /* hipStream_t */

int cutie_version(void);
const char* cutie_last_error(void);

/* ---------------------------------------------------------------------------------------
 * K0  token -> grid positional mapping.   Replaces DataLoader._positional_mapping,
 *     data_loader_json.py:437-577 (bit-exact, fp64 keys, stable order).  strategy 0 = skip a token
 *     whose row tail is full, 1 = shift the row left, 2 = the reference's "expand", whose outputs are
 *     those of 1 on every input (its search starts on the cell the shift has just freed).
 *
 * tok_box    int32 [T,4]  x_left,y_top,x_right,y_bottom per token, documents concatenated
 * tok_id     int32 [T]    dict id (word_to_index[text], :469)
 * tok_ecid   int32 [T]    entity_id*num_classes + class_id (:510)
 * doc_extent int32 [B,4]  left,top,right,bottom of the document (:467)
 * doc_off    int32 [B+1]  token range of document b is [doc_off[b], doc_off[b+1])
 * outputs    grid int32 [B,rows,cols], labels int8 [B,rows,cols], ps_x/ps_y int32 [B,rows,cols]
 *            tok_order int32 [T]: for document b, tok_order[doc_off[b]+k] = index (within the
 *                 document) of the k-th token in placement order (the stable sort of :507)
 *            tok_place int32 [T,2]: (row,col) where that k-th token was written, (-1,-1) if
 *                 strategy 0 dropped it
 *            status int32 [B]: 0 ok; 1 = some token maps outside the grid (the reference
 *                 raises IndexError or writes into its cropped buffer half); such a document
 *                 gets an all-zero grid / labels / ps and -1 in its tok_order / tok_place.
 *            Every element of every output is written by the call (nothing to pre-fill).
 * workspace  cutie_grid_build_workspace_bytes(T) bytes.
 * limits     (rows*cols + 2*rows + 2)*4 + 64 bytes must fit the 160 KB LDS (about 40000 cells), rows <= 1024.
 */
size_t cutie_grid_build_workspace_bytes(int n_tokens);
int cutie_grid_build(const int32_t* tok_box, const int32_t* tok_id, const int32_t* tok_ecid,
                     const int32_t* doc_extent, const int32_t* doc_off, int n_docs, int n_tokens,
                     int rows, int cols, int strategy,
                     int32_t* grid, int8_t* labels, int32_t* ps_x, int32_t* ps_y,
                     int32_t* tok_order, int32_t* tok_place, int32_t* status,
                     void* workspace, size_t workspace_bytes, cutie_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K1  embedding gather.   Replaces tf.nn.embedding_lookup in Model.embed,
 *     model_framework.py:43-54 (dropout keep_prob 1 is the identity).
 * out[cell, 0:E] = table[ids[cell], 0:E].  An id outside [0,V) gives a zero row and bumps
 * *oob_count (int32, caller-zeroed) - TF's GPU gather returns zeros, its CPU gather raises.
 * E must be a multiple of 4.
 */
int cutie_embed_fwd(const int32_t* ids, int64_t n_cells, const float* table, int V, int E,
                    void* out, int ld_out, int out_dtype, int32_t* oob_count,
                    cutie_stream_t stream);

/* backward of K1: dtable[ids[cell], :] += dout[cell, :]  (f32 accumulate; dtable is NOT
 * cleared here).  This is the scatter-add TF applies to the IndexedSlices gradient.
 * Same-address float atomics are what bounds it, so ids 0 and 1 ([PAD] = empty cell, [UNK]) are summed per
 * workgroup in registers and every other id per workgroup in an LDS table before anything reaches the table.
 * workspace (nullable, cutie_embed_bwd_workspace_bytes(n_cells, E) bytes, 16-byte aligned, clobbered): with it the
 * workgroups' partial rows for ids 0 and 1 are stored and folded in a fixed order (no atomics on the two hot rows,
 * the same bits every run); without it they are added atomically, one add per workgroup and element. */
size_t cutie_embed_bwd_workspace_bytes(int64_t n_cells, int E);
int cutie_embed_bwd(const int32_t* ids, int64_t n_cells, const void* dout, int ld_dout,
                    int dout_dtype, float* dtable, int V, int E,
                    void* workspace, size_t workspace_bytes, cutie_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K2-K5  stride-1 SAME (dilated) cross-correlation as an implicit GEMM on MFMA, with the
 *     bias + ReLU epilogue and the instance-norm statistics.   Replaces tf.nn.conv2d /
 *     tf.nn.atrous_conv2d + bias_add + relu in Model.conv / dilate_conv / convolution,
 *     model_framework.py:168-204,414-422, and their tf.gradients data-gradient.
 *
 * out[b,h,w,n] = act( bias[n] + sum over segments s, taps (i,j), channels c of
 *        x_s[b, h + sgn_s*(i - kh_s/2)*dil_s, w + sgn_s*(j - kw_s/2)*dil_s, c] * w_s[(i*kw_s+j)][n][c] )
 * with zero padding, sgn = +1 (forward) or -1 (flip != 0: the data gradient, fed with the
 * layer's output gradient and weights packed [tap][cin][cout]).  Several segments give the
 * gradient of a tensor with several consumers in one pass (no read-modify-write).
 * stats (nullable) f32 [B][stats_tiles][n_out][2]: per 256-pixel tile, the sum and sum of squares
 * of (value stored - shift), shift[n] = act(bias[n]) (0 without a bias) - the instance-norm moments of
 * model_framework.py:423 as partial sums about the shift of K6 below; hand the same bias and activation
 * code to whoever consumes them.
 * Tile t of a sample STORES slot t (no float atomics: reruns are bit-identical); slots the launch
 * does not own are left alone, so the caller zeroes the array ONCE, when it allocates it (every later
 * launch of the same shape stores the same slots again).  stats_tiles
 * >= cutie_conv_stats_tiles(H, W).  The consumers below add the slots in slot order.
 */
int cutie_conv_stats_tiles(int H, int W);
typedef struct {
    const void* x;        /* activations view, dtype of the call                          */
    const void* w;        /* packed weights [kh*kw][n rows][cin] in the call's dtype       */
    int64_t w_tap_stride; /* elements between taps (n_total*cin of the packed tensor)      */
    int ld_x;             /* leading dimension of x in elements                            */
    int cin;              /* channels contracted; multiple of 32 (f32) / 64 (bf16) - pad   */
    int kh, kw, dil;
    int flip;             /* 0 forward taps, 1 negated shifts (data gradient)              */
} cutie_conv_seg;

/* relu: activation code - 0 none, 1 ReLU, 2 sigmoid (the logits layer under use_ghm,
 * model_framework.py:173-175; generic kernel only).  The same code goes to cutie_in_bwd_apply. */
int cutie_conv_fwd(const cutie_conv_seg* segs /* host */, int n_segs,
                   int B, int H, int W, int n_out,
                   void* out, int ld_out, int out_dtype,
                   const float* bias, int relu, float* stats, int stats_tiles,
                   int dtype, cutie_stream_t stream);

/* Model.convolution as one call (model_framework.py:414-424): y = instance_norm(act(conv(x, W) + b)), i.e.
 * cutie_conv_fwd into `a` followed by cutie_in_apply (K6 below) into `y`.  With a `sync` area (as for
 * cutie_conv_dgrad_in_bwd below: cutie_conv_sync_bytes(B, stats_tiles, n_out) bytes, 64-byte aligned, zeroed ONCE, one
 * per layer and shape, word 0 counts waits that ran out) and where the wide MFMA kernel applies and a sample's tiles
 * are balanced, ONE launch does it: the tiles publish their moments as tagged granules, meet, and store y straight
 * from the accumulators - `a` is then NOT written, so pass a `sync` area only when nobody needs `a` afterwards
 * unless the backward entry points are then given y with CUTIE_ACT_FROM_Y.  *fused (host int, nullable) receives 1
 * when the one-launch form ran, else 0.  y is rounded once from the f32 activation in that form and from the stored
 * 16-bit activation in the two-call form: equal to one unit in the last place, not bit for bit.  In the one-launch form y
 * also carries the ReLU pattern: every a == 0 element of a (sample, channel) plane holds exactly
 * y0 = round16((0 - mean) * inv), and an a > 0 element whose y rounds onto y0 is stored one ulp above it.  The choice
 * depends on the layer and the grid shape only, never on the batch size.  `stats` is written either way. */
int cutie_conv_in_fwd(const cutie_conv_seg* segs /* host */, int n_segs,
                      int B, int H, int W, int n_out,
                      void* a, int ld_a, void* y, int ld_y,
                      const float* bias, int relu, float* stats, int stats_tiles, float eps,
                      void* sync, size_t sync_bytes, int32_t* late_count /* nullable, see cutie_conv_dgrad_in_bwd */,
                      int* fused /* host, nullable */,
                      int dtype, cutie_stream_t stream);

/* The data gradient of a layer that ends in the instance norm of model_framework.py:423, together with the
 * first half of that norm's gradient: out = dy as cutie_conv_fwd(flip segments, no bias / activation) gives
 * it, and red[b][t][c] = (sum dy, sum dy*xhat) exactly as cutie_in_bwd_stats (K6 below) defines it for
 * (dy = out, a, stats), ready for cutie_in_bwd_apply.  On the wide MFMA kernel the sums are taken in the
 * store pass of the epilogue from the rounded dy (tile t of a sample STORES slot t; slots the launch does
 * not own are left alone - zero `red` once when it is allocated); otherwise the conv is followed by
 * cutie_in_bwd_stats over all red_tiles slots.  red_tiles >= cutie_conv_stats_tiles(H, W).
 * CUTIE_FUSE_NORM_BWD=0 in the environment WHEN THE LIBRARY IS LOADED selects the two-pass form (A/B
 * measurements; the library reads its five switches - CUTIE_CONV_WIDE, CUTIE_FUSE_NORM_BWD, CUTIE_CONV_LPT,
 * CUTIE_RENDEZVOUS, each "0" to turn the named path off, and CUTIE_CONV_DUO (0: no 4-wave conv kernel, 1 = default:
 * for 128-channel tiles and short 64-channel 1x1 launches, 2: for the 256-channel layers too) - once at load, never
 * per call; none of them changes results beyond the rounding differences documented at the entry points;
 * CUTIE_CONV_RING and CUTIE_CONV_PATCH are read too but select nothing outside -DCW_WITH_RING / -DCW_WITH_PATCH variant
 * builds). */
int cutie_conv_dgrad_in_stats(const cutie_conv_seg* segs /* host */, int n_segs,
                              int B, int H, int W, int n_out,
                              void* out, int ld_out, int out_dtype,
                              const void* a, int ld_a, int a_dtype,
                              const float* stats, int stats_tiles, float eps,
                              float* red, int red_tiles,
                              const float* shift_bias, int shift_act /* the shift of `stats`, see K6 */,
                              int dtype, cutie_stream_t stream);

/* The data gradient of such a layer AND the whole gradient of its instance norm and activation, i.e.
 * cutie_conv_dgrad_in_stats followed by cutie_in_bwd_apply (K6 below), asked for as one call:
 *   dz = act'(a) * inv * (dy - mean(dy) - xhat * mean(dy * xhat)),  dbias_part[b*dbias_tiles + t][c] = partial sums of dz.
 * Where the wide MFMA kernel applies, the tiles of a sample take about equally long and at most 128 workgroups
 * make up a sample, ONE launch does it: every tile publishes its two sums as tagged 8-byte granules (stored and swept in 16-byte pairs) in `sync`,
 * sweeps the granules of its sample's other tiles until they carry the launch's tag (a bounded wait), folds them
 * and stores dz straight from the tile - dy is never written and the pass over dy, a and dz does not run (the
 * same bits as the two-call form: same operands, same order).  Otherwise the call runs the two-call form through
 * `dy` and `red`.  Either way `dy` must be a valid [B,H,W,>=n_out] scratch and `red` a valid slot array.
 *   sync: cutie_conv_sync_bytes(B, red_tiles, n_out) bytes, 64-byte aligned, zeroed ONCE by the caller and then left
 *         to the library, one area per (layer, shape): a launch's tag is the previous launch's + 1.  Word 0 counts
 *         waits that ran out (0.2 s: a sample's workgroups were not resident together) - the results of such a
 *         launch are invalid; read it at a synchronisation point.  NULL selects the two-call form.  The library
 *         remembers (on the host) the (B, red_tiles, n_out) an area was last used with and clears the area itself, on
 *         the call's stream, when a call brings another geometry.
 *   late_count (nullable): a device int32 of the caller's that is bumped together with word 0 - ONE word for all the
 *         launches of a train step, cleared by the caller when the step starts and handed to cutie_opt_step /
 *         cutie_adam_step / cutie_loss_scale_update as `invalid`: a step in which any tile gave up waiting is then
 *         skipped on the device (parameters, moments and packed operands untouched, counted in the loss-scale state),
 *         whether or not the host ever looks.
 *   dbias_part (nullable): dbias_tiles >= cutie_conv_stats_tiles(H, W); tile t of a sample STORES row t, rows the
 *         launch does not own are left alone (zero the array once when it is allocated), as for `red`.
 * CUTIE_RENDEZVOUS=0 in the environment when the library is loaded selects the two-call form everywhere. */
size_t cutie_conv_sync_bytes(int B, int red_tiles, int n_out);
/* Measurement hook, no counterpart in the reference (SURVEY.md 8d asks for the clock the chip actually sustains under the
 * MFMA kernels, whose peak is quoted at 2.4 GHz).  probe: three 8-byte device words, zeroed by the caller, or NULL to switch
 * the hook off.  While it is set, workgroup 0 of every launch of the wide conv kernel ADDS {shader-clock cycles, 100 MHz ticks,
 * 1} of its K loop to them (three atomics per launch): held clock = 100 MHz x cycles / ticks.  Process-wide; set and read it
 * at synchronisation points. */
int cutie_conv_clock_probe(void* probe);
int cutie_conv_dgrad_in_bwd(const cutie_conv_seg* segs /* host */, int n_segs,
                            int B, int H, int W, int n_out,
                            void* dy, int ld_dy,
                            const void* a, int ld_a, const float* stats, int stats_tiles, float eps,
                            float* red, int red_tiles, int relu,
                            const float* shift_bias /* with `relu`: the shift of `stats`, see K6 */,
                            void* dz, int ld_dz, float* dbias_part, int dbias_tiles,
                            void* sync, size_t sync_bytes, int32_t* late_count /* nullable */,
                            int dtype, cutie_stream_t stream);

/* weight gradient: dw[(i*kw+j)][c][n] (= HWIO, model_framework.py:416) =
 *   sum over b,h,w of x[b, h+(i-kh/2)*dil, w+(j-kw/2)*dil, c] * dz[b,h,w,n].
 * Deterministic: K is split over `splits` row groups into f32 slabs in the workspace, then
 * summed in a fixed order; `accumulate` != 0 adds into dw instead of overwriting. */
size_t cutie_conv_wgrad_workspace_bytes(int B, int H, int W, int cin, int cout, int kh, int kw);
int cutie_conv_wgrad(const void* x, int ld_x, int cin, const void* dz, int ld_dz, int cout,
                     int B, int H, int W, int kh, int kw, int dil,
                     float* dw, int accumulate, int dtype,
                     void* workspace, size_t workspace_bytes, cutie_stream_t stream);

#define CUTIE_MAX_BATCH 32
/* the weight gradients of up to CUTIE_MAX_BATCH layers in one call (the filter gradients of tf.gradients,
 * main_train_json.py:152, for several layers of model_framework.py:414-424 at once): layers that run on the same kernel
 * instance share ONE launch, so 256 CUs are filled by 256 / (sum of the layers' output tiles) K splits instead of
 * 256 / (one layer's tiles) - a third of the slab bytes and of the reduction reads for three 256->256 layers - and all
 * reductions are one launch.  Same results as cutie_conv_wgrad up to the order of the f32 slab additions (fixed for a
 * given set of layers: deterministic).  descs is a HOST array (device pointers inside), copied into the kernel
 * arguments.  workspace: cutie_conv_wgrad_batched_workspace_bytes(descs, n, dtype) bytes, 256-byte aligned. */
typedef struct cutie_wgrad_desc {
    const void* x;  int ld_x, cin;
    const void* dz; int ld_dz, cout;
    int B, H, W, kh, kw, dil;
    float* dw;
    int accumulate;
} cutie_wgrad_desc;
size_t cutie_conv_wgrad_batched_workspace_bytes(const cutie_wgrad_desc* descs /* host */, int n, int dtype);
int cutie_conv_wgrad_batched(const cutie_wgrad_desc* descs /* host */, int n, int dtype,
                             void* workspace, size_t workspace_bytes, cutie_stream_t stream);

/* packs f32 HWIO weights [kh*kw][cin][cout] (the layout of '<layer>/weights') into the two
 * operand layouts the conv kernels read: w_fwd [tap][cout][cin_pad] and w_dgrad
 * [tap][cin][cout_pad] (either may be NULL), zero padded, in `dtype`. */
int cutie_pack_weights(const float* w_hwio, int taps, int cin, int cout,
                       void* w_fwd, int cin_pad, void* w_dgrad, int cout_pad,
                       int dtype, cutie_stream_t stream);

/* the same for up to CUTIE_MAX_BATCH layers in ONE launch (the per-step repack after Adam:
 * main_train_json.py:156 applies the update, the next forward needs the new operands) */
typedef struct cutie_pack_desc {
    const float* w_hwio;
    void* w_fwd;        /* nullable */
    void* w_dgrad;      /* nullable */
    int taps, cin, cout, cin_pad, cout_pad;
} cutie_pack_desc;
int cutie_pack_weights_batched(const cutie_pack_desc* descs, int n, int dtype, cutie_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K6  instance norm, center=False scale=False, eps 1e-6, biased variance.   Replaces
 *     tf.contrib.layers.instance_norm at model_framework.py:423 and its gradient.
 * stats f32 [B][stats_tiles][C][2] = partial (sum, sum of squares) of (a - shift) over H*W as produced by
 * cutie_conv_fwd or cutie_in_stats; every consumer adds the slots of a (b, c) in slot order.
 * The shift: shift[c] = act(shift_bias[c]) with act = the activation code (0 none, 1 ReLU, 2 sigmoid), 0 when
 * shift_bias is NULL.  It makes mean = shift + S1/N, var = S2/N - (S1/N)^2 well conditioned in f32 for a
 * channel whose bias dominates it (mean >> std); a channel of an instance-normalised layer's input that is
 * nearly constant for another reason still cancels.  Producer and consumers of one `stats` array must be
 * given the same (shift_bias, act); cutie_conv_fwd uses its own bias and activation.
 *   apply:     y = (a - mean) * rsqrt(var + eps)
 *   bwd_stats: red[b][t][c] = (sum dy, sum dy*xhat) over the t-th of red_tiles slices of H*W
 *   bwd_apply: dz = [a > 0] * inv * (dy - red0/HW - xhat*red1/HW);  dbias_part[b*dbias_tiles + t][c]
 *              = sum of dz over the t-th slice of H*W
 *              (relu = 1 applies the ReLU mask of model_framework.py:173, relu = 2 the sigmoid
 *              derivative a*(1-a) of :175, 0 neither; dbias_part f32 [B*dbias_tiles][C], nullable;
 *              cutie_sum_rows folds its B*dbias_tiles rows into d(biases) in a fixed order)
 * Every producer STORES its slot (nothing to zero, no float atomics).  The *_tiles of an OUTPUT is the
 * number of workgroups per sample the kernel runs with: cutie_in_tiles(B, HW, C) is the count that
 * keeps the chip busy; 1 is valid and slow.
 */
int cutie_in_tiles(int B, int HW, int C);
int cutie_in_stats(const void* a, int ld_a, int a_dtype, int B, int HW, int C, float* stats,
                   int stats_tiles, const float* shift_bias, int shift_act, cutie_stream_t stream);
int cutie_in_apply(const void* a, int ld_a, int a_dtype, const float* stats, int stats_tiles,
                   float eps, void* y, int ld_y, int y_dtype, int B, int HW, int C,
                   const float* shift_bias, int shift_act, cutie_stream_t stream);
int cutie_in_bwd_stats(const void* dy, int ld_dy, int dy_dtype, const void* a, int ld_a,
                       int a_dtype, const float* stats, int stats_tiles, float eps, float* red,
                       int red_tiles, int B, int HW, int C, const float* shift_bias, int shift_act,
                       cutie_stream_t stream);
int cutie_in_bwd_apply(const void* dy, int ld_dy, int dy_dtype, const void* a, int ld_a,
                       int a_dtype, const float* stats, int stats_tiles, float eps,
                       const float* red, int red_tiles, int relu,
                       void* dz, int ld_dz, int dz_dtype, float* dbias_part, int dbias_tiles,
                       int B, int HW, int C, const float* shift_bias /* act = relu */,
                       cutie_stream_t stream);
/* out[c] (+)= sum over b of part[b][c] */
int cutie_sum_rows(const float* part, int B, int C, float* out, int accumulate, cutie_stream_t stream);
/* the same for up to CUTIE_MAX_BATCH (part, out, C) triples in ONE launch: the bias gradients of
 * every layer at the end of backward */
typedef struct cutie_sum_rows_desc {
    const float* part;
    float* out;
    int C;
} cutie_sum_rows_desc;
int cutie_sum_rows_batched(const cutie_sum_rows_desc* descs, int n, int B, int accumulate, cutie_stream_t stream);

/* K8  Model.global_pool, model_framework.py:309-315: spatial mean broadcast back to HxW
 * (reduce_mean + nearest-neighbour resize), written into a channel slice.
 * scratch (nullable): B*CUTIE_POOL_TILES*C*2 floats, clobbered; with it the sums are taken by
 * the wide-grid statistics pass (slot partials) and the broadcast is a store pass that adds the
 * slots in order; without it one workgroup per (sample, 128 channels) does both.  Both orders
 * are fixed: reruns are bit-identical. */
#define CUTIE_POOL_TILES 32
int cutie_spatial_mean_bcast(const void* x, int ld_x, int x_dtype, void* out, int ld_out,
                             int out_dtype, int B, int HW, int C, float* scratch,
                             cutie_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K9-K11  per-cell softmax cross-entropy with hard negative mining.   Replaces
 *     Model.softmax (model_framework.py:318-320) and CUTIE.build_loss
 *     (model_cutie.py:107-130; duplicate at model_cutie2.py:114-147).
 * fwd:    probs[n,:] = softmax(logits[n,:]) (nullable), ce[n] = logsumexp - logits[n,label]
 *         a label outside [0,C) bumps *bad_labels (caller-zeroed) and gives ce = NaN.
 * hnm:    Nfg = #{label != 0}, num_bg = min(Nbg, ratio*Nfg); selects the num_bg largest
 *         background ce (ties: lower index first, as tf.nn.top_k); weight[n] = 1/Nfg (fg),
 *         1/num_bg (selected bg), 0 otherwise;  loss_out[0] = mean(fg ce) + mean(selected)
 *         (NaN when either set is empty, as reduce_mean of an empty tensor), loss_out[1] =
 *         Nfg, loss_out[2] = num_bg.  A three-level radix select (11 + 11 + 10 bits of the CE key) over a
 *         wide grid of workgroups, whose histograms and counters live in the workspace
 *         (>= cutie_hnm_workspace_bytes(n_cells), 16-byte aligned, clobbered; no pre-fill needed).
 * bwd:    dlogits[n,c] = weight[n] * (probs[n,c] - [c == label[n]])
 */
int cutie_softmax_ce_fwd(const void* logits, int ld, int dtype, const int32_t* labels,
                         int64_t n_cells, int C, float* ce, float* probs,
                         int32_t* bad_labels, cutie_stream_t stream);
size_t cutie_hnm_workspace_bytes(int64_t n_cells);
int cutie_hnm_weights(const float* ce, const int32_t* labels, int64_t n_cells, int ratio,
                      float* weight, float* loss_out, void* workspace, size_t workspace_bytes,
                      cutie_stream_t stream);
/* cutie_hnm_weights without a workspace argument and without any allocation.  No counterpart in the
 * reference: it is for callers that cannot allocate device memory - code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc.  Same kernels, same results bit for bit; the select's work
 * area (histograms, counters, per-chunk words: 348 KB) is a device array of the library's code object, one copy
 * per GPU for the life of the process.  n_cells <= CUTIE_HNM_STATIC_MAX_CELLS, else CUTIE_E_ARG.  The current
 * device must be the one the pointers and the stream belong to.  Every call on a GPU shares that GPU's array
 * (each call clears what it uses first): calls that overlap on one GPU - two streams, two threads - race.
 * The first call on a device resolves the array's address (hipGetSymbolAddress): make it outside graph capture. */
#define CUTIE_HNM_STATIC_MAX_CELLS (4 * 1024 * 1024)
int cutie_hnm_weights_static(const float* ce, const int32_t* labels, int64_t n_cells, int ratio,
                             float* weight, float* loss_out, cutie_stream_t stream);
int cutie_softmax_ce_bwd(const float* probs, const int32_t* labels, const float* weight,
                         int64_t n_cells, int C, void* dlogits, int ld, int dtype,
                         const float* loss_scale /* nullable, see K12 */,
                         const float* elem_weight /* nullable f32 [n_cells][C]: GHM, below */,
                         cutie_stream_t stream);

/* The logits head in a handful of small launches (16-bit activations, C <= 32 classes, ReLU).   Replaces, for the
 * last layer of every CUTIE graph, Model.conv(1, 1, C) + bias + ReLU + instance norm (model_cutie_aspp.py:72-73,
 * model_framework.py:414-424), Model.softmax (:318-320) and CUTIE.build_loss (model_cutie.py:107-130) - the same
 * arithmetic as cutie_conv_fwd + cutie_in_apply + cutie_softmax_ce_fwd + cutie_hnm_weights + cutie_softmax_ce_bwd +
 * cutie_in_bwd_stats + cutie_in_bwd_apply on that layer, with the pre-norm activation kept in f32.
 * fwd:  a[n][0..C) = relu(x[n][0..cin) . w_hwio[cin][C] + bias) (MFMA, 16-bit operands, f32 accumulate; f32 out, ld_a a
 *       multiple of 4, columns C..ld_a of `a` and `logits` receive zeros); stats [B][stats_tiles][C][2] = per
 *       256-pixel tile (sum, sum of squares) of a - relu(bias) (K6's shifted moments; stats_tiles >=
 *       cutie_head_tiles(HW), the other slots are left alone: zero the array once);
 *       logits = instance_norm(a), probs[n][0..C) = softmax(logits[n]) (nullable), and with labels (int8 as the loader's
 *       grid, data_loader_json.py:455, or int32: label_bytes 1 / 4) ce[n] and *bad_labels as cutie_softmax_ce_fwd, plus
 *       level 0 of the hard-negative select in the workspace.  The softmax and the CE are wavefront reductions: eight
 *       lanes share a cell, four classes each.  labels == NULL: inference (ce, bad_labels, workspace unused).
 * bwd:  the rest of the select (ties to the lower index, as tf.nn.top_k), weight[n] and loss_out[0..3) as
 *       cutie_hnm_weights; red [B][cutie_head_chunks(HW)][C][2] = partial (sum dy, sum dy * logits) with
 *       dy = S * weight * (probs - onehot); dz[n][c] = [a > 0] * inv * (dy - mean(dy) - logits * mean(dy * logits)) in
 *       dz_dtype (ld_dz a multiple of 4; columns beyond C of a group of four that starts below C receive zeros, the
 *       others are left alone); dbias_part (nullable) [B * cutie_head_chunks(HW)][C] partial sums of dz.
 *       Must follow cutie_head_fwd on the same stream with the same workspace (>= cutie_head_workspace_bytes(B, HW)
 *       bytes, 16-byte aligned, clobbered; no pre-fill).  Deterministic: every sum is taken in a fixed order. */
int cutie_head_tiles(int HW);
int cutie_head_chunks(int HW);
size_t cutie_head_workspace_bytes(int B, int HW);
int cutie_head_fwd(const void* x, int ld_x, int dtype, int B, int HW, int cin,
                   const float* w_hwio, const float* bias, int C, float eps,
                   float* a, int ld_a, float* stats, int stats_tiles,
                   float* logits, int ld_logits, float* probs,
                   const void* labels /* nullable */, int label_bytes, float* ce, int32_t* bad_labels,
                   void* workspace, size_t workspace_bytes, cutie_stream_t stream);
int cutie_head_bwd(const float* a, int ld_a, const float* stats, int stats_tiles,
                   const float* logits, int ld_logits, const float* probs,
                   const void* labels, int label_bytes, const float* ce,
                   int B, int HW, int C, float eps, const float* bias, int ratio,
                   const float* loss_scale /* nullable */,
                   float* weight, float* red, float* loss_out,
                   void* dz, int ld_dz, int dz_dtype, float* dbias_part /* nullable */,
                   void* workspace, size_t workspace_bytes, cutie_stream_t stream);

/* Gradient harmonising weights (use_ghm = 1).   Replaces calc_ghm_weights, main_train_json.py:86-119,
 * and the product cls_logits * ghm_weights of model_cutie.py:110.
 *   grad = |logit - onehot| per (cell, class); bin i: i/bins <= grad < (i+1)/bins (last edge + eps);
 *   acc_sum[i] = momentum*acc_sum[i] + (1-momentum)*count[i] for the M non-empty bins;
 *   weights[n][c] = (n_cells*C) / acc_sum[bin] / M   (1/M where grad lies outside every bin);
 *   weighted_logits[n][c] = logit * weights[n][c]   (nullable) - the tensor the CE is taken of.
 * state: cutie_ghm_state_bytes(bins) bytes, 8-byte aligned, zeroed once by the caller; it carries
 * acc_sum from step to step (the reference's module-level list).  Feed weighted_logits to
 * cutie_softmax_ce_fwd and `weights` to cutie_softmax_ce_bwd as elem_weight (the weights are a
 * constant input of the graph, no gradient flows into them).  bins <= 256. */
size_t cutie_ghm_state_bytes(int bins);
int cutie_ghm_weights(const void* logits, int ld, int dtype, const int32_t* labels, int64_t n_cells,
                      int C, int bins, float momentum, float eps, void* state, size_t state_bytes,
                      float* weights, float* weighted_logits, cutie_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * K12,K14  L2 regulariser, global-norm clip and Adam over flat f32 buffers.   Replaces
 *     l2_regularizer (model_framework.py:427-432), tf.clip_by_global_norm(.., 10.0) and
 *     tf.train.AdamOptimizer.apply_gradients (main_train_json.py:152-156).
 * l2:    g += wd * w;  *reg_loss += wd * sum(w^2)/2            (reg_loss caller-zeroed)
 * sumsq: *out += sum(x^2)                                       (out caller-zeroed)
 *        both reduce in two stages over a fixed grid through `scratch` (CUTIE_REDUCE_SCRATCH
 *        floats, clobbered) in a fixed order: bit-identical on every data-parallel replica.
 * adam:  scale = clip / max(sqrt(*gnorm_sq), clip);  g' = g*scale;
 *        m = b1*m + (1-b1)*g';  v = b2*v + (1-b2)*g'^2;  p -= lr_t * m / (sqrt(v) + eps)
 *        with lr_t = lr*sqrt(1-b2^t)/(1-b1^t) computed by the caller (TF's form).
 *        A non-finite *gnorm_sq leaves p, m and v untouched (the step is skipped); with `loss_scale` and t > 0 the
 *        bias correction then follows the number of steps that were NOT skipped (state[3] counts the skipped ones).
 *
 * Loss scale (no counterpart in the reference, which trains in f32; it is what lets the CUTIE_F16
 * path keep the gradients the reference computes instead of flushing them to zero): a device
 * array `loss_scale` = {S, 1/S, good steps in a row, steps skipped, ...} of CUTIE_LOSS_SCALE_STATE floats.  cutie_softmax_ce_bwd
 * multiplies dlogits by S, so every activation gradient, weight gradient and bias gradient of the
 * backward pass carries S; cutie_l2_add (g = g/S + wd*w) and cutie_sumsq (sum/S^2) take it out
 * again when they are handed the same array.  NULL means S = 1 everywhere.
 * cutie_loss_scale_update, after the Adam step: a non-finite *gnorm_sq multiplies S by `backoff`
 * (>= min_scale), `interval` finite steps in a row multiply it by `growth` (<= max_scale).  All of
 * it stays on the device: no step waits for the host.
 */
#define CUTIE_REDUCE_SCRATCH 1024 /* floats of caller scratch for the two-stage reductions */
int cutie_l2_add(const float* w, float* g, int64_t n, float wd, float* reg_loss, float* scratch,
                 const float* loss_scale /* nullable */, cutie_stream_t stream);
int cutie_sumsq(const void* x, int64_t n, int dtype, float* out, float* scratch,
                const float* loss_scale /* nullable */, cutie_stream_t stream);
#define CUTIE_LOSS_SCALE_STATE 8 /* floats: {S, 1/S, good steps in a row, steps skipped, of them invalid, -, -, -} */
int cutie_loss_scale_init(float* state /* device, CUTIE_LOSS_SCALE_STATE floats */, float scale, cutie_stream_t stream);
/* `invalid` (nullable, device int32): the step-wide late_count of the conv entry points.  Non-zero = the step's
 * results are not to be trusted: Adam leaves p, m, v untouched, the update counts the step as skipped (state[3],
 * state[4]) and leaves S alone. */
int cutie_loss_scale_update(float* state, const float* gnorm_sq, float backoff, float growth,
                            int interval, float min_scale, float max_scale, const int32_t* invalid,
                            cutie_stream_t stream);
int cutie_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr_t,
                    float beta1, float beta2, float eps, const float* gnorm_sq, float clip,
                    float grad_scale, int t /* the caller's step count behind lr_t; 0 = take lr_t as it is */,
                    const float* loss_scale /* nullable: lr_t is re-based to step t - (steps the device skipped) */,
                    const int32_t* invalid /* nullable */, cutie_stream_t stream);

/* The same tail - L2 term, global-norm clip, Adam, the repack of the conv operands - as TWO launches over the flat
 * buffers (main_train_json.py:152-156; l2_regularizer model_framework.py:427-432).  The gradient buffer is left as
 * backward wrote it (no L2 term added in place, still loss-scaled):
 *   norm:  partials[0..1023]    = per-workgroup sums of (g/S + wd*w)^2 over [0, n_l2), (g/S)^2 over [n_l2, n_dense),
 *                                 plus (rows/S)^2 over `rows` (nullable; the embedding's IndexedSlices rows, one per
 *                                 grid cell, in rows_dtype - how tf.clip_by_global_norm sees that gradient);
 *          partials[1024..2047] = per-workgroup sums of w^2 over [0, n_l2).   CUTIE_OPT_SCRATCH floats.
 *   step:  every workgroup folds the partials in one fixed order (+ *extra_sq, nullable: a share of the squared norm
 *          that was summed elsewhere, e.g. over data-parallel ranks), so all of them hold the same bits;
 *          g' = (g/S + wd_i*w) * grad_scale * clip / max(|g| * grad_scale, clip), wd_i = wd on [0, n_l2) else 0;
 *          Adam as cutie_adam_step.  `layers` (host array, device pointers inside; w_hwio must point INTO p, in
 *          ascending order, inside [0, n_l2)) are walked in 64 x 64 (cin x cout) tiles whose updated values also go,
 *          in `dtype`, to w_dgrad [tap][cin][cout_pad] and transposed to w_fwd [tap][cout][cin_pad]
 *          (cutie_pack_weights_batched's layouts; no separate repack).  Elements of [0, n_l2) outside every layer are
 *          padding and are left alone.  *gnorm_sq_out = the folded squared norm, *reg_out = reg_scale * sum w^2
 *          (both nullable).  g[zero_from .. n) is cleared after use (zero_from < 0: nothing) - the scatter-added
 *          embedding gradient starts the next step at zero without a fill of its own.
 *          A non-finite norm or *invalid != 0 leaves p, m, v and the packed operands untouched (g[zero_from .. n) is
 *          cleared all the same). */
#define CUTIE_OPT_SCRATCH 2048
int cutie_opt_norm(const float* w, const float* g, int64_t n_l2, int64_t n_dense, float wd,
                   const void* rows, int64_t n_rows, int rows_dtype, const float* loss_scale /* nullable */,
                   float* partials, cutie_stream_t stream);
int cutie_opt_step(float* p, float* g, float* m, float* v, int64_t n, int64_t n_l2, float wd, float reg_scale,
                   const struct cutie_pack_desc* layers /* host */, int n_layers, int dtype,
                   float lr_t, float beta1, float beta2, float eps, float clip, float grad_scale, int t,
                   const float* partials, const float* extra_sq /* nullable */,
                   float* loss_scale /* nullable */, const int32_t* invalid /* nullable */,
                   float* gnorm_sq_out, float* reg_out, int64_t zero_from, cutie_stream_t stream);

/* K13  Model.positional_sampling, model_framework.py:73-86 (tf.batch_gather):
 * out[b, cell, :] = feat[b, idx[b, cell], :].  An index outside [0, n_src) is clamped into it and bumps *oob_count
 * (int32, nullable, caller-zeroed): tf.batch_gather raises on such an index, the caller raises from the counter. */
int cutie_batch_gather(const void* feat, int ld_feat, int n_src, const int32_t* idx, int n_cells,
                       int B, int C, void* out, int ld_out, int dtype, int32_t* oob_count,
                       cutie_stream_t stream);

/* backward of K13 (the gradient tf.gradients derives for tf.batch_gather):
 * dfeat[b, idx[b, cell], :] += dout[b, cell, :], idx clamped as in the forward; dfeat is a dense
 * f32 [B, n_src, C] accumulator that the caller zeroes (float atomics; the row of index 0, where
 * every token-less grid cell points, is pre-reduced per workgroup). */
int cutie_batch_scatter_add(const void* dout, int ld_dout, int dtype, const int32_t* idx, int n_cells,
                            int n_src, int B, int C, float* dfeat, cutie_stream_t stream);

/* dtype conversion of a [n_rows, C] view (used for the image input and f32 <-> bf16 / f16 views) */
int cutie_convert(const void* src, int ld_src, int src_dtype, void* dst, int ld_dst,
                  int dst_dtype, int64_t n_rows, int C, cutie_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CUTIE_HIP_H */
