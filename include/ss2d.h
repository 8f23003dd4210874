/* ss2d.h -- C ABI of libss2d_hip.so: the MI355X (gfx950) implementation of Mamba-UNet's SS2D hot path.
 *
 * Boundary being replaced (the reference has no FFI layer; its operator boundary is Python):
 *   - `selective_scan_fn` of the third-party CUDA package mamba_ssm, imported at
 *     /root/reference/code/networks/mamba_sys.py:17-20 and called at :436-442
 *     (semantics text embedded at :62-144);
 *   - the chain of torch ops of SS2D.forward_corev0 / SS2D.forward, mamba_sys.py:396-461, :552-565.
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller; the library never allocates or frees
 *     device memory and never synchronises; all work is enqueued on `stream` (a hipStream_t);
 *   - tensors are contiguous, row-major, fp32 unless `dtype` says otherwise;
 *   - return 0 on success; <0 = rejected argument (see ss2d_last_error()); >0 = hipError_t;
 *   - no C++ exception crosses the boundary; the only global state is a thread-local error string.
 */
#ifndef SS2D_H
#define SS2D_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SS2D_ABI_VERSION 1
#define SS2D_DTYPE_F32 0
#define SS2D_DTYPE_BF16 1   /* every `void *` tensor (activations and their gradients) is stored as bf16; arithmetic,
                               parameters (`float *`), scan state/checkpoints and the atomically accumulated
                               gradients (dA, dB, dC, dD, ddelta_bias, dw, dbias) stay fp32; vector path needs
                               8-byte aligned pointers and L % 4 == 0, otherwise a scalar path is used */
#define SS2D_SCAN_CHUNK 128 /* positions per state checkpoint */
#define SS2D_MAX_DSTATE 16

int ss2d_abi_version(void);
const char *ss2d_last_error(void);

/* Number of state checkpoints per (batch, channel) row that the forward writes for the backward:
 * ceil(L / SS2D_SCAN_CHUNK).  Checkpoint buffer = batch*dim*chunks*dstate floats. */
int ss2d_scan_num_chunks(int L);
size_t ss2d_scan_ckpt_bytes(int batch, int dim, int L, int dstate);

/* selective_scan_fn forward -- replaces mamba_ssm.selective_scan_fn as called at mamba_sys.py:436-442.
 *   u, delta, z, out : (batch, dim, L)      A : (dim, dstate) fp32, real
 *   B, C             : (batch, ngroups, dstate, L), dim % ngroups == 0, group(d) = d / (dim/ngroups)
 *   D, delta_bias    : (dim) fp32, nullable        z : nullable (out *= silu(z))
 *   last_state       : (batch, dim, dstate) fp32, nullable
 *   ckpt             : nullable; state entering every SS2D_SCAN_CHUNK-long chunk, (batch, dim, chunks, dstate)
 *   delta_softplus   : delta <- softplus(delta + delta_bias) (torch threshold 20), as at mamba_sys.py:68-69 */
int ss2d_selective_scan_fwd(int dtype, const void *u, const void *delta, const float *A, const void *B,
                            const void *C, const float *D, const void *z, const float *delta_bias, void *out,
                            float *last_state, float *ckpt, int batch, int dim, int L, int dstate, int ngroups,
                            int delta_softplus, void *stream);

/* selective_scan_fn backward (the autograd half of the same boundary).
 *   du, ddelta, dz : (batch, dim, L)   dB, dC : (batch, ngroups, dstate, L) fp32 ALWAYS
 *   dA : (dim, dstate)   dD, ddelta_bias : (dim)     (dD / dz / ddelta_bias nullable with D / z / delta_bias)
 *   dA, dD, ddelta_bias, dB, dC are ACCUMULATED INTO with float atomics: the caller zero-fills them.
 *   ckpt: the buffer the forward filled (required). */
int ss2d_selective_scan_bwd(int dtype, const void *u, const void *delta, const float *A, const void *B,
                            const void *C, const float *D, const void *z, const float *delta_bias,
                            const void *dout, const float *ckpt, void *du, void *ddelta, float *dA, float *dB,
                            float *dC, float *dD, void *dz, float *ddelta_bias, int batch, int dim, int L,
                            int dstate, int ngroups, int delta_softplus, void *stream);

/* Direction-aware variants of the two calls above: the scan reads and writes the UN-permuted tensors in place.
 *   rev_mask : bit g set = group g walks its rows backwards (scan position l <-> memory index L-1-l) for u, delta,
 *              B, C, z, out and, in _bwd, dout, du, ddelta, dB, dC, dz.  This is how the two reversed directions of
 *              CrossScan (mamba_sys.py:404, the torch.flip half of xs) and the un-flip of :446 run without any
 *              flipped copy in HBM.  Checkpoints stay in scan order.
 *   u_groups : u (and, in _bwd, dout) hold only u_groups groups per batch: (batch, u_groups * dim/ngroups, L);
 *              scan group g uses group g / (ngroups / u_groups).  With the four SS2D directions ordered
 *              (row-major, row-major reversed, column-major, column-major reversed), u_groups = 2 lets a direction
 *              and its reversal share one copy of the input.  du is still written per scan group (batch, dim, L).
 *   bc_bg_stride : elements between consecutive (batch, group) blocks of B and of C; 0 = dense (dstate * L).  Lets
 *              the scan read B / C in place from the x_proj output x_dbl (batch, K, R + 2N, L) of mamba_sys.py:407-417
 *              (B = x_dbl + R*L, C = x_dbl + (R+N)*L, stride (R + 2N) * L) instead of from contiguous copies.
 *              dB / dC stay dense.
 * ss2d_selective_scan_fwd/_bwd are these with u_groups = ngroups, rev_mask = 0, dense B / C.
 * ngroups <= 32 when rev_mask != 0. */
int ss2d_selective_scan_dir_fwd(int dtype, const void *u, const void *delta, const float *A, const void *B,
                                const void *C, const float *D, const void *z, const float *delta_bias, void *out,
                                float *last_state, float *ckpt, int batch, int dim, int L, int dstate, int ngroups,
                                int delta_softplus, int u_groups, unsigned rev_mask, long long bc_bg_stride, void *stream);
int ss2d_selective_scan_dir_bwd(int dtype, const void *u, const void *delta, const float *A, const void *B,
                                const void *C, const float *D, const void *z, const float *delta_bias,
                                const void *dout, const float *ckpt, void *du, void *ddelta, float *dA, float *dB,
                                float *dC, float *dD, void *dz, float *ddelta_bias, int batch, int dim, int L,
                                int dstate, int ngroups, int delta_softplus, int u_groups, unsigned rev_mask,
                                long long bc_bg_stride, void *stream);

/* The same backward with a caller-provided workspace for the partial dB / dC sums of the workgroups (the sums over
 * the channels of a group, mamba_sys.py:425-426 share B / C across a direction's d_inner channels):
 *   workspace : >= ss2d_scan_bwd_workspace_bytes(...) bytes of device memory, 16-byte aligned, contents irrelevant;
 *               with it (and dstate == 16, z == NULL, L % 4 == 0) dB / dC are WRITTEN (plain stores + one reduction
 *               pass: deterministic, need no zero-fill); NULL / too small / other shapes = float atomics as above.
 *   dA, dD, ddelta_bias are accumulated with atomics either way (caller zero-fills).
 * ss2d_scan_bwd_workspace_bytes returns 0 when no workspace would be used. */
size_t ss2d_scan_bwd_workspace_bytes(int batch, int dim, int L, int dstate, int ngroups);
int ss2d_selective_scan_dir_bwd_ws(int dtype, const void *u, const void *delta, const float *A, const void *B,
                                   const void *C, const float *D, const void *z, const float *delta_bias,
                                   const void *dout, const float *ckpt, void *du, void *ddelta, float *dA, float *dB,
                                   float *dC, float *dD, void *dz, float *ddelta_bias, int batch, int dim, int L,
                                   int dstate, int ngroups, int delta_softplus, int u_groups, unsigned rev_mask,
                                   long long bc_bg_stride, void *workspace, size_t workspace_bytes, void *stream);

/* ss2d_selective_scan_dir_bwd_ws for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  NO counterpart in the reference: it exists only for those callers.
 * The workspace is a fixed SS2D_STATIC_WS_BYTES array inside the library's code object (one copy per GPU, held until the
 * process ends); the call allocates nothing and frees nothing.  Same arguments, outputs and zero-fill contract as
 * ss2d_selective_scan_dir_bwd_ws with a workspace (dB / dC written; dA, dD, ddelta_bias accumulated: caller zero-fills).
 * Rejected with a negative code: shapes for which ss2d_scan_bwd_workspace_bytes() is 0 (dstate != 16, L % 4 != 0), a z
 * gate, and shapes that need more than SS2D_STATIC_WS_BYTES.  All calls on one GPU share the array: calls that overlap
 * in time on one GPU (different streams) race; calls on one stream are ordered and safe. */
#define SS2D_STATIC_WS_BYTES (64u << 20)
int ss2d_selective_scan_dir_bwd_static(int dtype, const void *u, const void *delta, const float *A, const void *B,
                                       const void *C, const float *D, const void *z, const float *delta_bias,
                                       const void *dout, const float *ckpt, void *du, void *ddelta, float *dA, float *dB,
                                       float *dC, float *dD, void *dz, float *ddelta_bias, int batch, int dim, int L,
                                       int dstate, int ngroups, int delta_softplus, int u_groups, unsigned rev_mask,
                                       long long bc_bg_stride, void *stream);

/* dt_proj folded into the scan (mamba_sys.py:417-424 + :436-442): the (batch, K*D, L) `delta` tensor
 * is never materialised.  x_dbl (batch, ngroups, dt_rank + 2*dstate, L) is the x_proj output with rows [dts | B | C] per
 * (batch, group) block, read in place three times; dt_w (dim, dt_rank) fp32 (= dt_projs_weight (K, D, R) in scan-group
 * order): delta[b, d, l] = sum_r dt_w[d, r] * dts[b, g(d), r, l], then softplus(delta + delta_bias[d]) as in SS2D.
 * Other arguments as ss2d_selective_scan_dir_fwd / _bwd_ws (delta_softplus = 1, no z, dstate == 16, L % 4 == 0,
 * dt_rank <= SS2D_MAX_DT_RANK; the backward needs dt_rank 6 or 12 and a workspace).  _bwd still WRITES ddelta
 * (batch, dim, L): the gradients of dt_w and of the dts rows are its contractions with dts / dt_w (caller). */
#define SS2D_MAX_DT_RANK 16
int ss2d_selective_scan_core_fwd(int dtype, const void *u, const void *x_dbl, const float *dt_w, const float *A, const float *D,
                                 const float *delta_bias, void *out, float *ckpt, int batch, int dim, int L, int dstate,
                                 int ngroups, int dt_rank, int u_groups, unsigned rev_mask, void *stream);
int ss2d_selective_scan_core_bwd(int dtype, const void *u, const void *x_dbl, const float *dt_w, const float *A, const float *D,
                                 const float *delta_bias, const void *dout, const float *ckpt, void *du, void *ddelta, float *dA,
                                 float *dB, float *dC, float *dD, float *ddelta_bias, int batch, int dim, int L, int dstate,
                                 int ngroups, int dt_rank, int u_groups, unsigned rev_mask, void *workspace,
                                 size_t workspace_bytes, void *stream);
/* _packed: the same backward with dB / dC written where their consumer wants them: as rows [dt_rank, dt_rank + 2*dstate) of
 * every (batch, group) block of d_x_dbl, the gradient of the x_proj output (same layout and `dtype` as x_dbl; rows
 * [0, dt_rank) -- d_dts -- are the caller's to fill, e.g. with ss2d_chan_gemm writing in place).  No fp32 dB / dC tensors
 * and no packing pass (ss2d_xdbl_grad_pack).  No counterpart in the reference (autograd of torch.split, mamba_sys.py:417). */
int ss2d_selective_scan_core_bwd_packed(int dtype, const void *u, const void *x_dbl, const float *dt_w, const float *A,
                                        const float *D, const float *delta_bias, const void *dout, const float *ckpt, void *du,
                                        void *ddelta, float *dA, void *d_x_dbl, float *dD, float *ddelta_bias, int batch, int dim,
                                        int L, int dstate, int ngroups, int dt_rank, int u_groups, unsigned rev_mask,
                                        void *workspace, size_t workspace_bytes, void *stream);

/* CrossScan (mamba_sys.py:403-404): x (B, D, H, W) -> xs (B, 4, D, H*W); _bwd is its adjoint
 * (sums the four directions back into dx). */
int ss2d_cross_scan_fwd(int dtype, const void *x, void *xs, int B, int D, int H, int W, void *stream);
int ss2d_cross_scan_bwd(int dtype, const void *dxs, void *dx, int B, int D, int H, int W, void *stream);

/* CrossMerge (mamba_sys.py:445-449): ys (B, 4, D, H*W) -> y; channel_last=1 writes (B, H, W, D)
 * (what :449 produces), 0 writes (B, D, H*W).  _bwd scatters dy to the four directions. */
int ss2d_cross_merge_fwd(int dtype, const void *ys, void *y, int B, int D, int H, int W, int channel_last,
                         void *stream);
int ss2d_cross_merge_bwd(int dtype, const void *dy, void *dys, int B, int D, int H, int W, int channel_last,
                         void *stream);

/* The same two kernels with a selectable layout of the direction axis, for use with ss2d_selective_scan_dir_*:
 *   SS2D_DIRS_REFERENCE     (b, 4, D, L): slots row-major, column-major, row-major reversed, column-major reversed
 *                           (what mamba_sys.py:403-404 builds and :445-448 consumes);
 *   SS2D_DIRS_MIRROR_FREE_4 (b, 4, D, L): slots row-major, row-major, column-major, column-major, nothing stored
 *                           reversed (the scan's rev_mask does the reversal in place); slot pairs in scan-group order;
 *   SS2D_DIRS_MIRROR_FREE_2 (b, 2, D, L): slots row-major, column-major (one copy per orientation, shared by a
 *                           direction and its reversal through u_groups = 2).
 * ss2d_dirs_scatter: x (B,D,H,W) or channel-last (B,H,W,D) -> every slot of xs;  ss2d_dirs_gather: y = sum over the
 * slots of ys (same summation order as the reference: row-major pair first), y (B,D,H*W) or channel-last. */
#define SS2D_DIRS_REFERENCE 0
#define SS2D_DIRS_MIRROR_FREE_4 1
#define SS2D_DIRS_MIRROR_FREE_2 2
int ss2d_dirs_scatter(int dtype, int layout, const void *x, void *xs, int B, int D, int H, int W, int x_channel_last,
                      void *stream);
int ss2d_dirs_gather(int dtype, int layout, const void *ys, void *y, int B, int D, int H, int W, int y_channel_last,
                     void *stream);
/* y (B, D, H, W) = sum of the four direction slots of ys4 (SS2D_DIRS_MIRROR_FREE_4) and the two orientation slots of ys2
 * (SS2D_DIRS_MIRROR_FREE_2) in one pass: autograd's sum of the two gradients that reach x in SS2D's backward -- du of the
 * selective scan (mamba_sys.py:436-442) through CrossScan (:403-404) and the input gradient of x_proj (:407). */
int ss2d_dirs_gather6(int dtype, const void *ys4, const void *ys2, void *y, int B, int D, int H, int W, void *stream);
/* The same three with a row pitch: the rows of the direction-side tensor (H*W positions each) lie `row_pitch` >= H*W
 * elements apart; the image side is unchanged.  For a caller whose H*W is odd (the 7 x 7 stage of a 224 x 224 input,
 * mamba_sys.py:403 with L = 49): it pads the rows to a multiple of 8, zero-fills the pad once, and runs x_proj, dt_proj and
 * the scan (:407-442) on the padded length with their aligned fast paths -- u = 0 in the pad keeps the state of a reversed
 * scan at zero until the first real position, and a forward scan reaches the pad after the last one. */
int ss2d_dirs_scatter_ld(int dtype, int layout, const void *x, void *xs, int B, int D, int H, int W, int x_channel_last,
                         long long row_pitch, void *stream);
int ss2d_dirs_gather_ld(int dtype, int layout, const void *ys, void *y, int B, int D, int H, int W, int y_channel_last,
                        long long row_pitch, void *stream);
int ss2d_dirs_gather6_ld(int dtype, const void *ys4, const void *ys2, void *y, int B, int D, int H, int W, long long row_pitch,
                         void *stream);
/* ... with y channel-last (B, H, W, D): dy of a channel-last convolution output (ss2d_dwconv3x3_silu_cl_bwd below). */
int ss2d_dirs_gather6_cl_ld(int dtype, const void *ys4, const void *ys2, void *y, int B, int D, int H, int W, long long row_pitch,
                            void *stream);

/* Depthwise 3x3 conv (pad 1, bias) + SiLU (mamba_sys.py:299-308, :558-559).
 *   x: channel_last_in ? (B, H, W, ldx) with the D channels first (ldx >= D lets it read the x half of
 *      the in_proj output xz directly, ldx = 2*D) : (B, D, H, W);  w (D, 3, 3); bias (D) nullable;
 *   y (B, D, H, W).  _bwd: dy (B, D, H, W) -> dx (same layout as x, row stride lddx), dw, dbias
 *   (accumulated into with float atomics: caller zero-fills dw/dbias). */
int ss2d_dwconv3x3_silu_fwd(int dtype, const void *x, const float *w, const float *bias, void *y, int B, int D,
                            int H, int W, int channel_last_in, int ldx, void *stream);
int ss2d_dwconv3x3_silu_bwd(int dtype, const void *x, const float *w, const float *bias, const void *dy,
                            void *dx, float *dw, float *dbias, int B, int D, int H, int W, int channel_last_in,
                            int ldx, int lddx, void *stream);
/* The same convolution with channel-last tensors on both sides (mamba_sys.py:299-308, :558-559 without the permute of
 * :558 and without its inverse in the backward): x (B, H, W, ldx), y (B, H, W, ldy), dy (B, H, W, lddy), dx (B, H, W, lddx),
 * the D channels first in every pixel, pitches >= D.  D and the pitches multiples of 4, pointers aligned to 4 elements
 * (w, bias to 16 bytes); one image below 2 GB.  dw / dbias accumulated with float atomics (caller zero-fills). */
int ss2d_dwconv3x3_silu_cl_fwd(int dtype, const void *x, const float *w, const float *bias, void *y, int B, int D,
                               int H, int W, int ldx, int ldy, void *stream);
int ss2d_dwconv3x3_silu_cl_bwd(int dtype, const void *x, const float *w, const float *bias, const void *dy, void *dx,
                               float *dw, float *dbias, int B, int D, int H, int W, int ldx, int lddy, int lddx,
                               void *stream);

/* dt_proj forward (mamba_sys.py:420, weight :319-327): out[b,g,d,l] = sum_r w[g,d,r] * x[b,g,r,l], R = dt_rank in
 * {6, 12} (stages 0 and 1 of Mamba-UNet; other ranks: use a library GEMM).  x is read in place from the first R rows
 * of every (b, g) block of the x_proj output (x_bg_stride elements between blocks, >= R*L); w (G, D, R) fp32;
 * out (batch, G, D, L) dense.  L and x_bg_stride multiples of 4. */
int ss2d_dt_proj_fwd(int dtype, const void *x, const float *w, void *out, int batch, int ngroups, int D, int R, int L,
                     long long x_bg_stride, void *stream);

/* Parameter transforms in front of the scan (mamba_sys.py:428-431; dt_projs_weight of :420), one launch each way:
 *   A[g*Dm+d][n] = -exp(A_logs[k*Dm+d][n]),  D[g*Dm+d] = Ds[k*Dm+d],  bias[g*Dm+d] = dt_bias[k*Dm+d],
 *   w[g][d][r] = dt_w[k][d][r],   all fp32, K direction groups of Dm channels, N states, rank R.
 * regroup = 0: g = k (the reference's order).  regroup = 1 (K = 4): scan group g = 2*orientation + reversed takes the
 * reference direction k = 2*reversed + orientation, the order ss2d_selective_scan_dir_* is used with.
 * _bwd: gradients w.r.t. the outputs (scan-group order; any of dA/dD/dbias/dw may be null = zero) -> gradients
 * w.r.t. the parameters (reference order), dA_logs = dA * A with A as produced by _fwd.  Outputs are overwritten. */
int ss2d_scan_params_fwd(const float *A_logs, const float *Ds, const float *dt_bias, const float *dt_w, float *A,
                         float *D, float *bias, float *w, int K, int Dm, int N, int R, int regroup, void *stream);
int ss2d_scan_params_bwd(const float *A, const float *dA, const float *dD, const float *dbias, const float *dw,
                         float *dA_logs, float *dDs, float *ddt_bias, float *ddt_w, int K, int Dm, int N, int R,
                         int regroup, void *stream);

/* Every parameter-derived operand of one fused-direction SS2D call (mamba_sys.py:407, :420, :428-431) in one launch
 * each way: A, D, bias, w as ss2d_scan_params_fwd with K = 4, regroup = 1, and from x_w = x_proj_weight (4, Cx, Dm) fp32
 * (Cx = R + 2N, reference direction order) and dt_w, in `dtype` with rows padded with zeros to their leading dimension
 * (the operand layout of ss2d_chan_gemm):
 *   xw  (2, 2Cx, ldD): xw[o][rev*Cx + c][d] = x_w[2*rev + o][c][d]    (x_proj forward: both directions of orientation o)
 *   xwT (2, Dm, ldC):  xwT[o][d][rev*Cx + c] = the same element        (input gradient of x_proj)
 *   dtwT (4, R, ldD):  dtwT[g][r][d] = dt_w[k(g)][d][r]                (d_dts = dt_w^T . ddelta)
 *   dtw_io (4, Dm, ldR; nullable): dt_w[k(g)][d][r]                    (delta as a GEMM: ranks other than 6 / 12)
 * _bwd: as ss2d_scan_params_bwd, plus dx_w (4, Cx, Dm) fp32 (reference order) from d_xw (2, 2Cx, Dm) fp32, the weight
 * gradient of the stacked x_proj (null = zero).  Outputs are overwritten. */
int ss2d_block_params_fwd(int dtype, const float *A_logs, const float *Ds, const float *dt_bias, const float *dt_w,
                          const float *x_w, float *A, float *D, float *bias, float *w, void *xw, void *xwT, void *dtwT,
                          void *dtw_io, int Dm, int N, int R, int ldD, int ldC, int ldR, void *stream);
int ss2d_block_params_bwd(const float *A, const float *dA, const float *dD, const float *dbias, const float *dw,
                          const float *d_xw, float *dA_logs, float *dDs, float *ddt_bias, float *ddt_w, float *dx_w,
                          int Dm, int N, int R, void *stream);

/* Gradient of the x_proj output x_dbl (batch, K, R + 2N, L) (mamba_sys.py:407-417; the autograd of its `torch.split`)
 * assembled in one pass: rows [0, R) of every (batch, k) block from d_dts (nblocks, R, L) in `dtype`, rows [R, R+N) and
 * [R+N, R+2N) from the scan backward's fp32 dB / dC (nblocks, N, L), cast to `dtype`.  nblocks = batch * K. */
int ss2d_xdbl_grad_pack(int dtype, const void *d_dts, const float *dB, const float *dC, void *out, long long nblocks,
                        int R, int N, int L, void *stream);

/* Weight gradient of the bias-free in_proj / out_proj Linears (mamba_sys.py:298, :337; used :555, :562; the
 * reference gets it from autograd of nn.Linear): C[m][n] += sum_r A[r][m] * B[r][n] with A = dL/d(output)
 * (rows x M, row stride lda), B = the Linear's input (rows x N, row stride ldb), rows = batch*H*W tokens,
 * C = dL/dweight (M x N fp32, row stride ldc), ACCUMULATED INTO with float atomics: caller zero-fills.
 * Products and sums are fp32 in both storage types: bf16 products are exact on v_mfma_f32_32x32x16_bf16; fp32 operands
 * enter the same instruction as exact three-way bf16 splits, six products each (16-byte aligned rows), or
 * v_mfma_f32_32x32x2_f32 for other layouts. */
int ss2d_linear_wgrad(int dtype, const void *A, const void *B, float *C, long long rows, int M, int N, int lda,
                      int ldb, int ldc, void *stream);

/* Fused out_norm + gate (mamba_sys.py:336/:452 LayerNorm(D) eps, :561 `y * F.silu(z)`):
 *   out[r][d] = ((y[r][d] - mean_r) * rstd_r * gamma[d] + beta[d]) * silu(z[r][d]),  rows = batch*H*W tokens.
 *   y, out, dout, dy: (rows, D) dense; z: row stride ldz (the z half of the in_proj output is read in place,
 *   ldz = 2*D); dz: row stride lddz.  mean/rstd (rows, fp32) are written by _fwd when non-NULL and read by _bwd.
 *   dgamma, dbeta (D, fp32) are ACCUMULATED INTO with float atomics: caller zero-fills.  D % 4 == 0, D <= 1536. */
int ss2d_outnorm_gate_fwd(int dtype, const void *y, const void *z, const float *gamma, const float *beta, void *out,
                          float *mean, float *rstd, long long rows, int D, int ldz, float eps, void *stream);
int ss2d_outnorm_gate_bwd(int dtype, const void *dout, const void *y, const void *z, const float *gamma,
                          const float *beta, const float *mean, const float *rstd, void *dy, void *dz, float *dgamma,
                          float *dbeta, long long rows, int D, int ldz, int lddz, void *stream);
/* The same pair with a per-sample factor folded in: out (forward) and dout (backward) of row r are multiplied by
 * row_scale[r / rows_per_scale] (fp32; null = 1).  This is where the block's stochastic-depth factor mask / keep
 * (timm DropPath around the branch, mamba_sys.py:581-585) is applied: out_proj has no bias, so scaling its input equals
 * scaling its output, and the pass that multiplied the branch (and the one that multiplied its gradient) disappears. */
int ss2d_outnorm_gate_scaled_fwd(int dtype, const void *y, const void *z, const float *gamma, const float *beta,
                                 const float *row_scale, long long rows_per_scale, void *out, float *mean, float *rstd,
                                 long long rows, int D, int ldz, float eps, void *stream);
int ss2d_outnorm_gate_scaled_bwd(int dtype, const void *dout, const void *y, const void *z, const float *gamma,
                                 const float *beta, const float *mean, const float *rstd, const float *row_scale,
                                 long long rows_per_scale, void *dy, void *dz, float *dgamma, float *dbeta, long long rows,
                                 int D, int ldz, int lddz, void *stream);

/* LayerNorm over the last axis, eps inside the sqrt: the `ln_1` of VSSBlock (mamba_sys.py:573-585) and the norms
 * of PatchMerging2D / PatchExpand / FinalPatchExpand_X4 (:196, :226, :249).  x, dx: (rows, D) in `in_dtype`;
 * out, dout: (rows, D) in `out_dtype` (fp32->fp32, bf16->bf16, or fp32->bf16: normalise the fp32 residual stream
 * straight into the bf16 operand of the next GEMM).  mean/rstd (rows, fp32) written by _fwd when non-NULL, read by
 * _bwd.  dgamma, dbeta (D, fp32) ACCUMULATED INTO with float atomics: caller zero-fills.  D % 4 == 0, D <= 1536. */
int ss2d_layernorm_fwd(int in_dtype, int out_dtype, const void *x, const float *gamma, const float *beta, void *out,
                       float *mean, float *rstd, long long rows, int D, float eps, void *stream);
int ss2d_layernorm_bwd(int in_dtype, int out_dtype, const void *dout, const void *x, const float *gamma,
                       const float *mean, const float *rstd, void *dx, float *dgamma, float *dbeta, long long rows,
                       int D, void *stream);
/* The same with a second gradient of x added to dx as it is written: `add` (rows x D, x's dtype) is the gradient that
 * arrives over the residual connection around the norm (mamba_sys.py:583-585: the block input feeds ln_1 and the sum);
 * autograd would add the two in a pass of its own.  dx may alias neither x nor add. */
int ss2d_layernorm_bwd_add(int in_dtype, int out_dtype, const void *dout, const void *x, const float *gamma,
                           const float *mean, const float *rstd, const void *add, void *dx, float *dgamma, float *dbeta,
                           long long rows, int D, void *stream);

/* The same LayerNorm over the rows of a ps x ps pixel un-shuffle of x: x (batch, H, W, ps*ps*D) as the expand Linear
 * of PatchExpand / FinalPatchExpand_X4 leaves it (mamba_sys.py:233-265), out / dout (batch, H*ps, W*ps, D) =
 * norm(rearrange(x, 'b h w (p1 p2 c) -> b (h p1) (w p2) c')) (:244, :262); mean / rstd per output row; dx in the
 * layout of x.  The rearranging copy becomes address arithmetic inside the kernel. */
int ss2d_layernorm_unshuffle_fwd(int in_dtype, int out_dtype, const void *x, const float *gamma, const float *beta,
                                 void *out, float *mean, float *rstd, long long batch, int H, int W, int ps, int D,
                                 float eps, void *stream);
int ss2d_layernorm_unshuffle_bwd(int in_dtype, int out_dtype, const void *dout, const void *x, const float *gamma,
                                 const float *mean, const float *rstd, void *dx, float *dgamma, float *dbeta,
                                 long long batch, int H, int W, int ps, int D, void *stream);

/* LayerNorm(4C) over the 2x2 neighbourhoods of x (batch, H, W, C), channel-last: row (b, i, j) of out (batch*H/2*W/2, 4C) is
 * x[b,2i,2j] | x[b,2i+1,2j] | x[b,2i,2j+1] | x[b,2i+1,2j+1], normalised -- PatchMerging2D's gather + norm
 * (mamba_sys.py:213-221) without the four strided slices and the concatenation.  Even H, W; C % 4 == 0, 4C <= 1536.
 * _bwd writes dx (batch, H, W, C) completely and ACCUMULATES dgamma / dbeta (4C, fp32; caller zero-fills). */
int ss2d_layernorm_merge_fwd(int in_dtype, int out_dtype, const void *x, const float *gamma, const float *beta, void *out,
                             float *mean, float *rstd, long long batch, int H, int W, int C, float eps, void *stream);
int ss2d_layernorm_merge_bwd(int in_dtype, int out_dtype, const void *dout, const void *x, const float *gamma,
                             const float *mean, const float *rstd, void *dx, float *dgamma, float *dbeta, long long batch,
                             int H, int W, int C, void *stream);

/* LayerNorm + segmentation head in one pass: the tail of VSSM.up_x4 (mamba_sys.py:859-866) = FinalPatchExpand_X4's
 * rearrange + norm (:262-265), permute to (B, C, H', W') and the bias-free 1x1 Conv2d(C -> classes) `output` (:789):
 *   logits[b, k, h', w'] = sum_c w_head[k][c] * LN(unshuffle(x))[b, h', w', c]       H' = H*ps, W' = W*ps
 * x (batch, H, W, ps*ps*D) in in_dtype, w_head (classes, D) fp32, logits / dlogits (batch, classes, H', W') in
 * out_dtype; D <= 128, classes <= 16.  The normalised tensor is never materialised.  _bwd: dx in the layout of x;
 * dgamma, dbeta, dw_head (classes, D) ACCUMULATED INTO (caller zero-fills). */
int ss2d_ln_head_fwd(int in_dtype, int out_dtype, const void *x, const float *gamma, const float *beta,
                     const float *w_head, void *logits, float *mean, float *rstd, long long batch, int H, int W, int ps,
                     int D, int classes, float eps, void *stream);
int ss2d_ln_head_bwd(int in_dtype, int out_dtype, const void *dlogits, const void *x, const float *gamma,
                     const float *beta, const float *w_head, const float *mean, const float *rstd, void *dx,
                     float *dgamma, float *dbeta, float *dw_head, long long batch, int H, int W, int ps, int D,
                     int classes, void *stream);

/* Statistics of the segmentation losses in one pass over the logits: softmax over the class axis, the per-class sums of
 * DiceLoss over the WHOLE batch (code/utils/losses.py:179-215: intersect, y_sum, z_sum of _dice_loss, one-hot of :186-192)
 * and the summed cross entropy of nn.CrossEntropyLoss as composed in train_fully_supervised_2D_VIM.py:153-157 and
 * train_Semi_Mamba_UNet.py:424-446.  logits (B, C, HW) in dtype (NCHW), labels (B, HW) integers of label_bytes (1, 4, 8);
 * stats (3C+1) fp32 WRITTEN: [0,C) sum p_c*[y==c], [C,2C) sum [y==c], [2C,3C) sum p_c^2, [3C] sum -log p_y.
 * workspace: ss2d_seg_loss_workspace_floats(C) floats.  Deterministic (fixed summation order, no atomics).
 * _bwd: dlogits (same layout and dtype as logits, WRITTEN) from gstats = dL/dstats (3C+1, device):
 * dlogit_c = p_c*(q_c - sum_k p_k q_k) + gstats[3C]*(p_c - [y==c]),  q_c = gstats[c]*[y==c] + 2*gstats[2C+c]*p_c.
 * A label outside [0, C) contributes to no statistic (and gets no cross-entropy gradient).  C <= 16. */
/* Weight gradients of x_proj (mamba_sys.py:407) and dt_proj (:420), whose operands are channel-major (rows = channels,
 * positions contiguous):  out[g*sog + i*soi + j*soj] += sum_b sum_l P[b*spb + g*spg + i*L + l] * Q[b*sqb + g*sqg + j*L + l]
 * for i < Mb (the wide operand), j < Ns; P, Q in dtype, out fp32 ACCUMULATED INTO (caller zero-fills; float atomics).
 * L, the strides and the base addresses must be multiples of 4 elements.  MFMA (bf16: exact products, fp32 sums; fp32
 * operands as exact three-way bf16 splits, six products each). */
int ss2d_chan_wgrad(int dtype, const void *P, const void *Q, float *out, int B, int G, int Mb, int Ns, int L, long long spb,
                    long long spg, long long sqb, long long sqg, long long sog, long long soi, long long soj, void *stream);

/* x_proj (mamba_sys.py:407), dt_proj (:420) and their input gradients as ONE product over channel-major operands:
 *   Y[b*syb + g*syg + m*L + l] = sum_{k<K} W[g*swg + m*ldw + k] * X[b*sxb + g*sxg + k*L + l],   m < M, l < L
 * (x_proj: W = x_proj_weight, X = xs; its dX: W = weight^T, X = d_x_dbl; dt_proj: W = dt_projs_weight, X = the dts rows of
 * x_dbl; its d_dts: W = weight^T, X = d_delta).  W, X, Y in dtype, fp32 accumulation on the bf16 MFMA (fp32 operands enter
 * it as exact three-way bf16 splits, six products each: an fp32 GEMM's accuracy), Y WRITTEN.  X and Y rows are L apart and
 * may be row ranges of wider blocks (strides).  L, the strides of X / Y and their base addresses must be multiples of 4
 * elements (fp32: X, Y 16-byte aligned); ldw >= K and the group stride swg multiples of 8 (fp32: 4), W 16-byte aligned,
 * and the columns K .. ldw-1 of W must hold zeros. */
int ss2d_chan_gemm(int dtype, const void *W, const void *X, void *Y, int B, int G, int M, int K, int ldw, int L, long long swg,
                   long long sxb, long long sxg, long long syb, long long syg, void *stream);

/* The token-major Linears (in_proj / out_proj, mamba_sys.py:298, :337, applied :556, :563; PatchMerging2D.reduction :196,
 * the expand Linears :243, :261) on the bf16 MFMA; A, B, C in dtype (bf16, or fp32 as exact three-way bf16 splits with six
 * products each -- an fp32 GEMM's accuracy):
 *   b_k_major = 0:  C[t*ldc + n] = sum_k A[t*lda + k] * B[n*ldb + k]     (F.linear(x, weight): A = x, B = weight (out, in))
 *   b_k_major = 1:  C[t*ldc + n] = sum_k A[t*lda + k] * B[k*ldb + n]     (its input gradient: A = dy, B = weight)
 * t < T, n < N; fp32 accumulation, C WRITTEN.  N, K and the leading dimensions multiples of 8 (fp32: 4), operands 16-byte
 * aligned. */
int ss2d_token_gemm(int dtype, const void *A, const void *B, void *C, long long T, int N, int K, int lda, int ldb, int ldc,
                    int b_k_major, void *stream);
/* The same forward product over a CONCATENATED input, without the concatenation, and with the bias:
 *   C (T x N) = [A | A2] (T x (K1 + K2)) * W^T + bias      W (N x K) row-major with leading dimension ldw, K = K1 + K2
 * -- `self.concat_back_dim[inx](torch.cat([x, x_downsample[3 - inx]], -1))` (mamba_sys.py:835, the nn.Linear(2*dim, dim) of
 * :774-778).  bias (N) fp32 is added before the result is rounded / stored; nullable.  A2 may be null with K1 == K (a plain
 * biased Linear).  K1 must be a multiple of the kernel's k chunk (bf16: 32 for K <= 256, else 64; fp32: 32); the other
 * conditions are those of ss2d_token_gemm. */
int ss2d_token_gemm_cat(int dtype, const void *A, const void *A2, const void *W, const float *bias, void *C, long long T, int N,
                        int K, int K1, int lda, int lda2, int ldw, int ldc, void *stream);
/* C (T x N) = R + A (T x K) * W^T with the product rounded to `dtype` before the sum -- bit for bit what the Linear followed
 * by `x + y` gives: out_proj and the residual connection of a VSSBlock (mamba_sys.py:562, :585) in one pass.  R (T x N, row
 * stride ldr) in `dtype`; C may be R.  Same alignment rules as ss2d_token_gemm. */
int ss2d_token_gemm_res(int dtype, const void *A, const void *W, const void *R, void *C, long long T, int N, int K, int lda,
                        int ldw, int ldr, int ldc, void *stream);

/* Column sums out[j] = sum_r x[r*ldx + j] (fp32) of a (rows, M) matrix in dtype: autograd's bias gradient of the Linears
 * that have a bias (`concat_back_dim`, mamba_sys.py:774-778, applied :835).  Deterministic (per-workgroup partial rows in
 * `workspace`, ss2d_colsum_workspace_floats(M) floats, added in a fixed order); out is WRITTEN. */
size_t ss2d_colsum_workspace_floats(int M);
int ss2d_colsum(int dtype, const void *x, float *out, float *workspace, long long rows, int M, long long ldx, void *stream);

#define SS2D_SEG_LOSS_MAX_CLASSES 16
size_t ss2d_seg_loss_workspace_floats(int C);
int ss2d_seg_loss_stats(int dtype, const void *logits, const void *labels, int label_bytes, float *stats, float *workspace,
                        int B, int C, long long HW, void *stream);
int ss2d_seg_loss_bwd(int dtype, const void *logits, const void *labels, int label_bytes, const float *gstats, void *dlogits,
                      int B, int C, long long HW, void *stream);

#ifdef __cplusplus
}
#endif
#endif
