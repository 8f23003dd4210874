// Weight gradients of the two projections that sit between the depthwise convolution and the scan, for gfx950 (MFMA):
//   x_proj   (mamba_sys.py:407)  dW[k][c][d] = sum_{b,l} d_x_dbl[b][k][c][l] * xs[b][k][d][l]
//   dt_proj  (mamba_sys.py:420)  dW[k][d][r] = sum_{b,l} d_delta[b][k][d][l] * dts[b][k][r][l]
// Both operands are CHANNEL-major (a row = one channel, positions contiguous), so the contraction index l is the fast
// axis of both: exactly the A / B operand layout of the MFMA instructions (a lane holds consecutive k of one row), no
// transposes.  As library GEMMs these are B*K small products with K = L (49 ... 3136) followed by a sum
// over the batch: 30-60 us each at every stage, whatever the size (tools/proj_gemm_bench.py); here the batch is part of
// the contraction.
//
//   out[g][i][j] += sum_b sum_l P[b][g][i][l] * Q[b][g][j][l]        P: the wide operand (rows i), Q: the narrow one
//
// A workgroup (4 waves) owns one 32-row tile of P, up to 3 tiles of Q (96 rows) and a slab of steps; a step is 512 bytes
// of every row (256 positions in bf16, 128 in fp32) of one batch entry, brought into LDS with coalesced 16-byte reads
// (a first version read its MFMA operands straight from memory, one row per lane: 64 cache lines per load instruction,
// 3 us per step).  Each wave multiplies a quarter of the step's positions out of LDS (one ds_read_b128 per operand and
// MFMA K-step, rows 528 bytes apart: conflict-free) on v_mfma_f32_32x32x16_bf16 (exact products, fp32 sums); fp32 operands
// enter it as three-way bf16 splits, six products each (mfma_step).  At the end the four waves add their tiles in LDS by turns and the workgroup sends one set of
// fp32 atomics (out is ACCUMULATED INTO: the caller zero-fills).
#include "ss2d_common.h"
#include "../../include/ss2d.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));

struct CmParams {
    const void *P, *Q;
    float *out;
    long long spb, spg, sqb, sqg;   // element strides of the (batch, group) blocks; rows inside a block are L apart
    long long sog, soi, soj;        // out[g*sog + i*soi + j*soj]
    int B, G, Mb, Ns, L;
    int spr;                        // steps per row = ceil(L / positions per step)
    long long steps, slab;          // B * spr steps in all, `slab` of them per workgroup
    int big_tiles, small_groups, slabs4;   // grid = G * big_tiles * small_groups * slabs4 workgroups
};

template <int NT, typename IO>
__device__ __forceinline__ void mfma_step(const u4 (&a)[4], const u4 (&b)[NT][4], f16v (&acc)[NT]) {
    if (sizeof(IO) == 2) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < NT; ++j)
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[q]), __builtin_bit_cast(bf16x8, b[j][q]), acc[j], 0, 0, 0);
    } else {
        // fp32 on the bf16 MFMA: the lane's 16 consecutive positions are two k-steps of 8; every value enters as the exact
        // sum of three bf16 values, six products kept (split3_bf16, ss2d_common.h) -- 12 MFMAs at the bf16 rate in place of
        // 16 v_mfma_f32_32x32x2_f32 at 1/16 of it
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const Split4 alo = split3_bf16(a[2 * ks]), ahi = split3_bf16(a[2 * ks + 1]);
            const bf16x8 a1 = __builtin_bit_cast(bf16x8, u4{alo.h1.x, alo.h1.y, ahi.h1.x, ahi.h1.y});
            const bf16x8 a2 = __builtin_bit_cast(bf16x8, u4{alo.h2.x, alo.h2.y, ahi.h2.x, ahi.h2.y});
            const bf16x8 a3 = __builtin_bit_cast(bf16x8, u4{alo.h3.x, alo.h3.y, ahi.h3.x, ahi.h3.y});
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const Split4 blo = split3_bf16(b[j][2 * ks]), bhi = split3_bf16(b[j][2 * ks + 1]);
                const bf16x8 b1 = __builtin_bit_cast(bf16x8, u4{blo.h1.x, blo.h1.y, bhi.h1.x, bhi.h1.y});
                const bf16x8 b2 = __builtin_bit_cast(bf16x8, u4{blo.h2.x, blo.h2.y, bhi.h2.x, bhi.h2.y});
                const bf16x8 b3 = __builtin_bit_cast(bf16x8, u4{blo.h3.x, blo.h3.y, bhi.h3.x, bhi.h3.y});
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b2, acc[j], 0, 0, 0);      // the small terms first
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a3, b1, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b3, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b1, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b2, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[j], 0, 0, 0);
            }
        }
    }
}

constexpr int ROWB = 512, ROWP = ROWB + 16;      // bytes of a row per step, and its padded stride in LDS

template <int NT, typename IO, bool V16>
__global__ __launch_bounds__(256) void cm_wgrad_kernel(const CmParams p) {
    extern __shared__ __align__(16) unsigned char lds[];        // [(1 + NT) * 32 rows][ROWP]; reused for the final tile sum
    constexpr int EPS = ROWB / sizeof(IO);                       // positions per step: 128 (fp32) or 256 (bf16)
    constexpr int ROWS = (1 + NT) * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, half = lane >> 5;
    int blk = xcd_remap(blockIdx.x, gridDim.x);   // neighbours in the tile order share cache lines / operands: one XCD's L2
    const int sl = blk % p.slabs4; blk /= p.slabs4;
    const int sg = blk % p.small_groups; blk /= p.small_groups;
    const int bt = blk % p.big_tiles;
    const int g = blk / p.big_tiles;
    const long long t0 = (long long)sl * p.slab;
    const long long t1 = t0 + p.slab < p.steps ? t0 + p.slab : p.steps;
    f16v acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

    // staging: piece = 16 bytes (V16) or 8 bytes of one row; consecutive threads take consecutive pieces of a row
    constexpr int PB = V16 ? 16 : 8, PPR = ROWB / PB, NPIECE = ROWS * PPR / 256, EPP = PB / sizeof(IO);
    const IO *__restrict__ Pg = static_cast<const IO *>(p.P) + (size_t)g * p.spg;
    const IO *__restrict__ Qg = static_cast<const IO *>(p.Q) + (size_t)g * p.sqg;
    const IO *rowp[NPIECE];          // row base of each of this thread's pieces at batch 0 (nullptr-free: rows are clamped)
    long long rowb[NPIECE];          // its batch stride
    int pcol[NPIECE], poff[NPIECE];
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) {
        const int id = k * 256 + tid, r = id / PPR, c = id - r * PPR;
        pcol[k] = c * EPP;
        poff[k] = r * ROWP + c * PB;
        if (r < 32) {
            int row = bt * 32 + r;
            row = row < p.Mb ? row : p.Mb - 1;               // clamped rows feed accumulator rows that are never written
            rowp[k] = Pg + (size_t)row * p.L;
            rowb[k] = p.spb;
        } else {
            int row = sg * NT * 32 + (r - 32);
            row = row < p.Ns ? row : p.Ns - 1;
            rowp[k] = Qg + (size_t)row * p.L;
            rowb[k] = p.sqb;
        }
    }
    // MFMA operand reads of this lane: row l32 of each tile, 16 bytes per K-step inside the wave's quarter of the step
    const int kbase = wave * (ROWB / 4) + half * (ROWB / 8);    // byte offset of the lane's 64-byte run inside the row
    for (long long t = t0; t < t1; ++t) {
        const int b = (int)(t / p.spr), st = (int)(t - (long long)b * p.spr);
        const int pos0 = st * EPS;
        // unconditional reads (address clamped, value masked): a read under a branch costs a full wait at the join
        typedef unsigned piece_t __attribute__((ext_vector_type(V16 ? 4 : 2)));
        piece_t v[NPIECE];
#pragma unroll
        for (int k = 0; k < NPIECE; ++k) {
            const int e0 = pos0 + pcol[k];
            const bool ok = e0 < p.L;                             // a piece is inside the row as a whole (L % EPP == 0)
            const piece_t tv = *reinterpret_cast<const piece_t *>(rowp[k] + (size_t)b * rowb[k] + (ok ? e0 : 0));
            const unsigned m = ok ? ~0u : 0u;
            v[k] = tv & m;
        }
        __syncthreads();                                          // the previous step's MFMA reads are done
#pragma unroll
        for (int k = 0; k < NPIECE; ++k) *reinterpret_cast<piece_t *>(lds + poff[k]) = v[k];
        __syncthreads();
        u4 a[4], bq[NT][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            a[q] = *reinterpret_cast<const u4 *>(lds + l32 * ROWP + kbase + 16 * q);
#pragma unroll
            for (int j = 0; j < NT; ++j) bq[j][q] = *reinterpret_cast<const u4 *>(lds + ((1 + j) * 32 + l32) * ROWP + kbase + 16 * q);
        }
        mfma_step<NT, IO>(a, bq, acc);
    }
    __syncthreads();
    // accumulator register e of a 32x32 tile: row = 8*(e/4) + 4*(lane/32) + e%4, column = lane%32.  The waves add their
    // tiles in LDS by turns (plain stores, then read-add-write), then one set of atomics per workgroup.
    float *tile = reinterpret_cast<float *>(lds);
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float *tp = tile + j * 1024 + (8 * (e >> 2) + 4 * half + (e & 3)) * 32 + l32;
                    *tp = w == 0 ? acc[j][e] : *tp + acc[j][e];
                }
        }
        __syncthreads();
    }
    for (int i = tid; i < NT * 1024; i += 256) {
        // consecutive threads -> consecutive addresses of out: the narrow index is the fast one unless the wide one has stride 1
        const int j = i >> 10, hi = (i >> 5) & 31, lo = i & 31;
        const int r = p.soi == 1 ? lo : hi, c = p.soi == 1 ? hi : lo;
        const int bi = bt * 32 + r, sj = (sg * NT + j) * 32 + c;
        if (bi < p.Mb && sj < p.Ns) atomicAdd(p.out + (size_t)g * p.sog + (size_t)bi * p.soi + (size_t)sj * p.soj, tile[j * 1024 + r * 32 + c]);
    }
}

template <typename IO, bool V16>
void launch_cm(const CmParams &p, int nt, unsigned nwg, hipStream_t s) {
    // dynamic LDS: (1 + nt) * 32 padded rows (<= 66 KB: two workgroups per CU); above 64 KB the attribute must be raised
    const size_t lds = (size_t)(1 + nt) * 32 * ROWP;
    const void *k = nt == 1 ? (const void *)cm_wgrad_kernel<1, IO, V16> : nt == 2 ? (const void *)cm_wgrad_kernel<2, IO, V16> : (const void *)cm_wgrad_kernel<3, IO, V16>;
    static bool attr[3][SS2D_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SS2D_MAX_DEVICES) dev = 0;
    if (!attr[nt - 1][dev]) {
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { ss2d_launch_rc_ref() = (int)hipErrorInvalidValue; return; }
        attr[nt - 1][dev] = true;
    }
    CmParams pc = p;
    void *args[] = {&pc};
    const hipError_t e = hipLaunchKernel(k, dim3(nwg), dim3(256), args, lds, s);
    if (e != hipSuccess && ss2d_launch_rc_ref() == 0) ss2d_launch_rc_ref() = (int)e;
}

}  // namespace

extern "C" int ss2d_chan_wgrad(int dtype, const void *P, const void *Q, float *out, int B, int G, int Mb, int Ns, int L, long long spb,
                               long long spg, long long sqb, long long sqg, long long sog, long long soi, long long soj, void *stream) {
    if (dtype != SS2D_DTYPE_F32 && dtype != SS2D_DTYPE_BF16) { ss2d_set_error("chan_wgrad: unknown dtype %d", dtype); return -1; }
    if (!P || !Q || !out) { ss2d_set_error("chan_wgrad: null required pointer"); return -2; }
    if (B <= 0 || G <= 0 || Mb <= 0 || Ns <= 0 || L <= 0) { ss2d_set_error("chan_wgrad: non-positive size"); return -3; }
    const uintptr_t amask = dtype == SS2D_DTYPE_BF16 ? 7 : 15;
    if (L % 4 || ((spb | spg | sqb | sqg) % 4) || (((uintptr_t)P | (uintptr_t)Q) & amask)) {
        ss2d_set_error("chan_wgrad: L, the block strides and the base addresses must be multiples of 4 elements"); return -4; }
    CmParams p;
    p.P = P; p.Q = Q; p.out = out; p.spb = spb; p.spg = spg; p.sqb = sqb; p.sqg = sqg; p.sog = sog; p.soi = soi; p.soj = soj;
    p.B = B; p.G = G; p.Mb = Mb; p.Ns = Ns; p.L = L;
    const int per_step = dtype == SS2D_DTYPE_BF16 ? 256 : 128;      // 512 bytes of every row
    p.spr = (L + per_step - 1) / per_step;
    p.steps = (long long)B * p.spr;
    p.big_tiles = (Mb + 31) / 32;
    const int small_tiles = (Ns + 31) / 32;
    const int nt = small_tiles >= 3 ? 3 : small_tiles;
    p.small_groups = (small_tiles + nt - 1) / nt;
    // about 512 workgroups (2 per CU), at least 4 steps each: the epilogue (LDS turns + one atomic per output element) is a
    // fixed cost per workgroup
    const long long tiles = (long long)G * p.big_tiles * p.small_groups;
    long long slabs = (512 + tiles - 1) / tiles;
    const long long max_slabs = (p.steps + 3) / 4;
    if (slabs > max_slabs) slabs = max_slabs;
    if (slabs < 1) slabs = 1;
    p.slab = (p.steps + slabs - 1) / slabs;
    p.slabs4 = (int)((p.steps + p.slab - 1) / p.slab);               // slabs per tile (one workgroup each)
    const long long nwg = tiles * p.slabs4;
    if (nwg > 0x7fffffffLL) { ss2d_set_error("chan_wgrad: grid too large"); return -6; }
    hipStream_t s = (hipStream_t)stream;
    // 16-byte reads need 16-byte aligned rows: always for fp32 here, for bf16 when everything is a multiple of 8 elements
    const bool v16 = dtype == SS2D_DTYPE_F32 || (L % 8 == 0 && ((spb | spg | sqb | sqg) % 8) == 0 && (((uintptr_t)P | (uintptr_t)Q) & 15) == 0);
    if (dtype == SS2D_DTYPE_BF16) { if (v16) launch_cm<bf16, true>(p, nt, (unsigned)nwg, s); else launch_cm<bf16, false>(p, nt, (unsigned)nwg, s); }
    else launch_cm<float, true>(p, nt, (unsigned)nwg, s);
    const int rc = ss2d_take_launch_rc();
    if (rc) ss2d_set_error("chan_wgrad: launch failed: %s", hipGetErrorString((hipError_t)rc));
    return rc;
}
