// Weight gradient of a bias-free Linear over a long token axis, for gfx950 MFMA:
//     C[m][n] += sum_r A[r][m] * B[r][n]          A: rows x M (row stride lda), B: rows x N (ldb), C: M x N fp32
// This is dW of SS2D's in_proj / out_proj (/root/reference/code/networks/mamba_sys.py:298, :337; applied :555,
// :562): rows = batch*H*W tokens (75 264 at stage 0) against an M x N output of only 384 x 96 -- a shape
// rocBLAS/hipBLASLt run at 10 % of their own forward rate (no split over the token axis: measured 356 us fp32
// for a 144 MB read at stage 0).
//
// Layout: both operands are token-major, i.e. the contraction index is the SLOW axis of both.  That is exactly
// the operand layout of v_mfma_f32_32x32x2_f32 (A: lane = m + 32k, B: lane = n + 32k, k = 0..1): a wave reads
// two token rows with plain coalesced 128-byte runs straight into MFMA operands -- no LDS, no transposes.
// A wave owns a 96 x 96 tile of C (3 x 3 MFMA tiles, 144 accumulator registers) for one slab of the token
// axis; token slabs are spread over the whole chip and the partial tiles are added into C with one
// fp32 atomic per element (C is zero-filled by the caller, like the other accumulated gradients of the ABI).
// bf16 operands are widened on load (exact) and go through the same fp32 MFMA: the kernel stays bound by its
// reads, and the gradient keeps fp32 products and sums.
#include <cstdlib>
#include "ss2d_common.h"
#include "../../include/ss2d.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int WT = 96;    // wave tile edge = 3 MFMA tiles
constexpr int UNR = 8;    // token-row pairs per pipeline stage

struct WgradParams {
    const void *A, *B;
    float *C;
    long long rows;
    int M, N, lda, ldb, ldc, tiles_m, tiles_n, tiles_pad, ksplit;
    long long slab;  // token rows per slab (even)
};

// One pipeline stage = 2*UNR token rows.  Addresses of out-of-range columns are clamped (their products only
// reach accumulator rows/columns that are never written back), so column masks are not needed.  Rows past the
// slab end (TAIL stage only) are clamped too and zeroed with a bitwise AND -- not a select on the loaded value,
// which LLVM would turn back into a branch with one wait per load.
template <bool TAIL, typename IO>
__device__ __forceinline__ void load_stage_impl(const IO *__restrict__ A, const IO *__restrict__ B, const WgradParams &p,
                                                long long r, long long r1, int half, const int (&acol)[3],
                                                const int (&bcol)[3], float (&a)[UNR][3], float (&b)[UNR][3]) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
        long long row = r + 2 * u + half;
        unsigned keep = 0xffffffffu;
        if (TAIL) {
            keep = row < r1 ? 0xffffffffu : 0u;
            row = row < r1 ? row : r1 - 1;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float av = ld1(A + row * p.lda + acol[i]);
            const float bv = ld1(B + row * p.ldb + bcol[i]);
            a[u][i] = TAIL ? __builtin_bit_cast(float, __builtin_bit_cast(unsigned, av) & keep) : av;
            b[u][i] = TAIL ? __builtin_bit_cast(float, __builtin_bit_cast(unsigned, bv) & keep) : bv;
        }
    }
}
template <typename IO>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // make everything derived from it scalar
    const long long w = (long long)blockIdx.x * 4 + wave;
    const int tile = (int)(w % p.tiles_pad), ks = (int)(w / p.tiles_pad);
    if (tile >= p.tiles_m * p.tiles_n) return;  // padding waves (no barrier in this kernel)
    const int tm = tile / p.tiles_n, tn = tile % p.tiles_n;
    const int m0 = tm * WT, n0 = tn * WT;
    const long long r0 = (long long)ks * p.slab;
    const long long r1 = (r0 + p.slab < p.rows) ? r0 + p.slab : p.rows;
    const IO *__restrict__ A = static_cast<const IO *>(p.A);
    const IO *__restrict__ B = static_cast<const IO *>(p.B);
    const int half = lane >> 5, l32 = lane & 31;
    int acol[3], bcol[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        acol[i] = m0 + 32 * i + l32; acol[i] = acol[i] < p.M ? acol[i] : p.M - 1;
        bcol[i] = n0 + 32 * i + l32; bcol[i] = bcol[i] < p.N ? bcol[i] : p.N - 1;
    }
    f16v acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;

    // Full stages of 2*UNR rows are double-buffered: the loads of one stage are in flight while the other stage's
    // 72 MFMAs issue.  Every trip issues the same loads on every path (stages past the end re-read the last full
    // stage and are simply not multiplied): a conditional load inside the loop would make the compiler's
    // s_waitcnt bookkeeping fall back to vmcnt(0) at the join and serialise memory against the MFMAs.
    constexpr int S = 2 * UNR;
    const long long nfull = (r1 - r0) / S;
    float a0[UNR][3], b0[UNR][3], a1[UNR][3], b1[UNR][3];
#define SS2D_MFMA_STAGE(a, b)                                                                                      \
    _Pragma("unroll") for (int u = 0; u < UNR; ++u) _Pragma("unroll") for (int i = 0; i < 3; ++i)                  \
        _Pragma("unroll") for (int j = 0; j < 3; ++j)                                                              \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], b[u][j], acc[i][j], 0, 0, 0);
    if (nfull > 0) {
        const long long rlast = r0 + (nfull - 1) * S;
        load_stage_impl<false>(A, B, p, r0, r1, half, acol, bcol, a0, b0);
        for (long long st = 0; st < nfull; st += 2) {
            const long long ra = r0 + (st + 1) * S, rb = r0 + (st + 2) * S;
            load_stage_impl<false>(A, B, p, ra < rlast ? ra : rlast, r1, half, acol, bcol, a1, b1);
            __builtin_amdgcn_sched_barrier(0);  // keep the loads ahead of the MFMAs they are meant to overlap
            SS2D_MFMA_STAGE(a0, b0)
            __builtin_amdgcn_sched_barrier(0);
            load_stage_impl<false>(A, B, p, rb < rlast ? rb : rlast, r1, half, acol, bcol, a0, b0);
            __builtin_amdgcn_sched_barrier(0);
            if (st + 1 < nfull) { SS2D_MFMA_STAGE(a1, b1) }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (r0 + nfull * S < r1) {  // ragged end of the slab: one masked stage
        load_stage_impl<true>(A, B, p, r0 + nfull * S, r1, half, acol, bcol, a0, b0);
        SS2D_MFMA_STAGE(a0, b0)
    }
#undef SS2D_MFMA_STAGE
    // accumulator register e of a 32x32 tile: row = 8*(e/4) + 4*(lane/32) + e%4, column = lane%32
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int n = n0 + 32 * j + l32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + 32 * i + 8 * (e >> 2) + 4 * half + (e & 3);
                if (m < p.M && n < p.N) atomicAdd(p.C + (size_t)m * p.ldc + n, acc[i][j][e]);
            }
        }
}

}  // namespace

extern "C" int ss2d_linear_wgrad(int dtype, const void *A, const void *B, float *C, long long rows, int M, int N,
                                 int lda, int ldb, int ldc, void *stream) {
    if (dtype != SS2D_DTYPE_F32 && dtype != SS2D_DTYPE_BF16) { ss2d_set_error("linear_wgrad: unknown dtype %d", dtype); return -1; }
    if (!A || !B || !C) { ss2d_set_error("linear_wgrad: null pointer"); return -2; }
    if (rows <= 0 || M <= 0 || N <= 0) { ss2d_set_error("linear_wgrad: non-positive size"); return -3; }
    if (lda < M || ldb < N || ldc < N) { ss2d_set_error("linear_wgrad: leading dimension smaller than the row length"); return -4; }
    WgradParams p;
    p.A = A; p.B = B; p.C = C; p.rows = rows; p.M = M; p.N = N; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.tiles_m = (M + WT - 1) / WT; p.tiles_n = (N + WT - 1) / WT;
    const int tiles = p.tiles_m * p.tiles_n;
    p.tiles_pad = (tiles + 3) / 4 * 4;
    // one wave per SIMD on every CU (1024 waves) when the token axis is long enough; a slab is at least 128 rows
    long long target_waves = 1024;
    if (const char *e = getenv("SS2D_WGRAD_WAVES")) { const long long v = atoll(e); if (v >= 4 && v <= 65536) target_waves = v; }  // tuning knob
    long long ks = (target_waves + p.tiles_pad - 1) / p.tiles_pad;
    const long long max_ks = (rows + 127) / 128;
    if (ks > max_ks) ks = max_ks;
    if (ks < 1) ks = 1;
    long long slab = (rows + ks - 1) / ks;
    slab = (slab + 1) / 2 * 2;
    ks = (rows + slab - 1) / slab;
    p.slab = slab; p.ksplit = (int)ks;
    const long long nwg = ks * (p.tiles_pad / 4);
    if (nwg > 0x7fffffffLL) { ss2d_set_error("linear_wgrad: grid too large"); return -6; }
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SS2D_DTYPE_BF16) hipLaunchKernelGGL(wgrad_kernel<bf16>, dim3((unsigned)nwg), dim3(256), 0, s, p);
    else hipLaunchKernelGGL(wgrad_kernel<float>, dim3((unsigned)nwg), dim3(256), 0, s, p);
    const int rc = (int)hipGetLastError();
    if (rc) ss2d_set_error("linear_wgrad: launch failed: %s", hipGetErrorString((hipError_t)rc));
    return rc;
}
