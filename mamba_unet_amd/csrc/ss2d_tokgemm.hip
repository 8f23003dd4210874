// The token-major Linears of the SS2D path (in_proj / out_proj, mamba_sys.py:298, :337; the patch merge / expand Linears
// :196, :243, :261), forward and input gradient, bf16 on MFMA for gfx950:
//     C[t][n] = sum_k A[t][k] * Bm[n][k]      (B_KMAJOR = false: forward,  A = x (T x in),  Bm = weight (out x in))
//     C[t][n] = sum_k A[t][k] * Bm[k][n]      (B_KMAJOR = true:  dX,       A = dy (T x out), Bm = weight (out x in))
// These are skinny products (K or N is 96 ... 1536 against T = 1176 ... 75264 tokens) that the BLAS library runs at
// 1.5-2.4 TB/s of their operand bytes at the large-token stages (tools/proj_gemm_bench.py, scratch timings in DESIGN.md);
// they are bound by memory, not by the MFMA.
//   * a workgroup (4 waves) owns 128 tokens x 96 output columns; wave w multiplies tokens 32w .. 32w+31; K is walked in
//     chunks of 32 or 64 that are copied to LDS as they lie in memory (16-byte pieces), the next chunk in flight in registers;
//   * the A operand (tokens x k, k contiguous) and the forward's weight rows are plain ds_read_b128 fragments; the dX
//     weight (k x n, n contiguous) is read back with ds_read_b64_tr_b16;
//   * the 128 x 96 result goes through LDS once more so that the stores are 16-byte pieces of whole rows.
// Needs K, N and the leading dimensions multiples of 8 and 16-byte aligned operands; the caller (ss2d_ops._Linear) uses the
// BLAS library otherwise, and for fp32.
#include "ss2d_common.h"
#include "../../include/ss2d.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct TgParams {
    const bf16 *A, *A2, *B;     // A2: the token operand of k >= K1 (the skip half of a concatenated input), or A again
    const float *bias;          // (N) fp32, added before the result is rounded; nullable
    const bf16 *R;              // (T x N, row stride ldr) nullable: C = R + round(A * W^T): the residual sum behind out_proj
    bf16 *C;
    long long T;
    int N, K, K1, lda, lda2, ldb, ldc, ldr, tiles_n;
};

// a + b over eight bf16 values in four registers, in fp32, rounded once: what `x + y` of two bf16 tensors gives
__device__ __forceinline__ u32x4 add_bf16x8(u32x4 a, u32x4 b) {
    typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f2 s = f2{__builtin_bit_cast(float, a[i] << 16) + __builtin_bit_cast(float, b[i] << 16),
                        __builtin_bit_cast(float, a[i] & 0xffff0000u) + __builtin_bit_cast(float, b[i] & 0xffff0000u)};
        o[i] = __builtin_bit_cast(unsigned, __builtin_convertvector(s, bf16x2v));
    }
    return o;
}

constexpr int GT = 128, GN = 96;                          // tokens, columns per workgroup
constexpr int BROW_NN = 2 * GN;                           // dX weight chunk [k][96 n]: 192 B = 48 banks, conflict-free for the transposed read
constexpr int gstage(bool kmajor, int gk) { return GT * (2 * gk + 16) + (kmajor ? gk * BROW_NN : GN * (2 * gk + 16)); }
constexpr int CROW = 2 * GN + 16;                         // result image [128 t][96 n] bf16
static_assert(GT * CROW <= 2 * gstage(true, 32) && GT * CROW <= 2 * gstage(false, 32) && 2 * gstage(false, 64) <= 65536, "LDS budget");

// GK: k per chunk.  32 for a short contraction (K <= 256: 35 KB of LDS, four workgroups per CU -- a workgroup is mostly
// prologue and epilogue there), 64 for a long one (half the barriers).
template <bool B_KMAJOR, int GK>
__global__ __launch_bounds__(256) void tok_gemm_kernel(const TgParams p) {
    constexpr int AROW = 2 * GK + 16, BROW_NT = 2 * GK + 16, ABYTES = GT * AROW;     // bytes per LDS row of the A chunk [128 t][GK k] / the forward's weight chunk [96 n][GK k]
    constexpr int STAGE = gstage(B_KMAJOR, GK);
    constexpr int APR = GK / 8, NAP = GT * APR / 256;       // 16-byte pieces per A row, per thread
    constexpr int NBP = (B_KMAJOR ? GK * 12 : GN * APR) / 256 + (((B_KMAJOR ? GK * 12 : GN * APR) % 256) ? 1 : 0), NBTOT = B_KMAJOR ? GK * 12 : GN * APR;
    __shared__ __align__(16) unsigned char lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int blk = xcd_remap(blockIdx.x, gridDim.x);           // the column tiles of one token block run back to back on one XCD
    const int nt = blk % p.tiles_n;
    const long long t0 = (long long)(blk / p.tiles_n) * GT;
    const int n0 = nt * GN;
    // this thread's pieces of a chunk: A 128 rows x 8 pieces -> 4; B 768 pieces -> 3
    u32x4 pa[NAP], pb[NBP];
    const int arow = tid / APR, acol = (tid % APR) * 8;    // piece j: token row arow + (256 / APR) j, k = acol
    int brow[NBP], bcol[NBP];
    bool bon[NBP];
#pragma unroll
    for (int j = 0; j < NBP; ++j) {
        const int id = j * 256 + tid;
        bon[j] = id < NBTOT;
        const int ic = bon[j] ? id : 0;
        if (B_KMAJOR) { brow[j] = ic / 12; bcol[j] = (ic - brow[j] * 12) * 8; }     // [GK k][96 n]: row = k, col = n
        else { brow[j] = ic / APR; bcol[j] = (ic - brow[j] * APR) * 8; }             // [96 n][GK k]: row = n, col = k
    }
    auto load_chunk = [&](int k0) {
        // never a guarded load: the address is clamped to element 0 and the value masked
        // K1 is a multiple of GK: a chunk lies in one of the two token operands (scalar choice)
        const bool second = k0 >= p.K1;
        const bf16 *__restrict__ Ap = second ? p.A2 : p.A;
        const long long lda = second ? p.lda2 : p.lda;
        const int ka = (second ? k0 - p.K1 : k0) + acol;
#pragma unroll
        for (int j = 0; j < NAP; ++j) {
            const long long t = t0 + arow + (256 / APR) * j;
            const bool ok = t < p.T && k0 + acol < p.K;
            const u32x4 v = *reinterpret_cast<const u32x4 *>(Ap + (ok ? t * lda + ka : 0));
            pa[j] = v & (ok ? ~0u : 0u);
        }
#pragma unroll
        for (int j = 0; j < NBP; ++j) {
            bool ok;
            long long off;
            if (B_KMAJOR) { ok = bon[j] && k0 + brow[j] < p.K && n0 + bcol[j] < p.N; off = (long long)(k0 + brow[j]) * p.ldb + n0 + bcol[j]; }
            else { ok = bon[j] && n0 + brow[j] < p.N && k0 + bcol[j] < p.K; off = (long long)(n0 + brow[j]) * p.ldb + k0 + bcol[j]; }
            const u32x4 v = *reinterpret_cast<const u32x4 *>(p.B + (ok ? off : 0));
            pb[j] = v & (ok ? ~0u : 0u);
        }
    };
    auto put_chunk = [&](int s) {
        unsigned char *as = lds + s * STAGE, *bs = as + ABYTES;
#pragma unroll
        for (int j = 0; j < NAP; ++j) *reinterpret_cast<u32x4 *>(as + (arow + (256 / APR) * j) * AROW + acol * 2) = pa[j];
#pragma unroll
        for (int j = 0; j < NBP; ++j)
            if (bon[j]) *reinterpret_cast<u32x4 *>(bs + brow[j] * (B_KMAJOR ? BROW_NN : BROW_NT) + bcol[j] * 2) = pb[j];
    };
    f16v acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    const int aoff = (32 * wave + l32) * AROW + 16 * half;                 // + 32 bytes per k-step
    const int boff_nt = l32 * BROW_NT + 16 * half;                          // + 32 rows per column tile, + 32 bytes per k-step
    // transposed read (dX): 16-lane group gq covers columns 16*(gq & 1) .. +15 of a 32-column tile and the k rows
    // 8*(gq >> 1) .. +7 of a k-step in two blocks of 4; lane 4q + pp addresses k row q, columns 4pp .. 4pp+3
    const int gq = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int boff_nn = (8 * (gq >> 1) + q) * BROW_NN + (16 * (gq & 1) + 4 * pp) * 2;
    const int nchunks = (p.K + GK - 1) / GK;
    load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
        const int s = c & 1;
        put_chunk(s);                                        // stage s was last read two chunks ago, behind the previous barrier
        load_chunk((c + 1 < nchunks ? c + 1 : c) * GK);      // in flight during the products (the last one is a harmless repeat)
        __syncthreads();
        const unsigned char *as = lds + s * STAGE, *bs = as + ABYTES;
#pragma unroll
        for (int ks = 0; ks < GK / 16; ++ks) {
            const bf16x8 af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(as + aoff + 32 * ks));
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                bf16x8 bf;
                if (B_KMAJOR) {
                    typedef s16x4 __attribute__((address_space(3))) *lp;
                    const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(bs + boff_nn + 16 * ks * BROW_NN + 64 * j));
                    const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(bs + boff_nn + (16 * ks + 4) * BROW_NN + 64 * j));
                    bf = __builtin_bit_cast(bf16x8, s16x8{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w});
                } else {
                    bf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(bs + boff_nt + 32 * j * BROW_NT + 32 * ks));
                }
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf, af, acc[j], 0, 0, 0);      // the TRANSPOSED block: rows = columns n, columns = tokens
            }
        }
    }
    // the residual pieces this thread will add to its six output pieces, requested ahead of the epilogue's two barriers
    u32x4 rv[6];
    if (p.R) {
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const int id = j * 256 + tid, row = id / 12, pc = id - row * 12;
            const long long t = t0 + row;
            const int n = n0 + pc * 8;
            rv[j] = *reinterpret_cast<const u32x4 *>(p.R + ((t < p.T && n < p.N) ? t * p.ldr + n : 0));
        }
    }
    __syncthreads();
    // accumulator register e of a 32x32 block: row (= column n of the result) 8*(e/4) + 4*(lane/32) + e%4, column (= token)
    // lane%32 -- four consecutive n of one token per register quad: one 8-byte LDS store
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (p.bias) {       // N % 8 == 0: the four columns are inside N together
                const int n = n0 + 32 * j + 8 * g + 4 * half;
                const f4 bv = *reinterpret_cast<const f4 *>(p.bias + (n < p.N ? n : 0));
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][4 * g + i] += bv[i];
            }
            const bf16x4 v = {(__bf16)acc[j][4 * g], (__bf16)acc[j][4 * g + 1], (__bf16)acc[j][4 * g + 2], (__bf16)acc[j][4 * g + 3]};
            *reinterpret_cast<bf16x4 *>(lds + (32 * wave + l32) * CROW + (32 * j + 8 * g + 4 * half) * 2) = v;
        }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int id = j * 256 + tid, row = id / 12, pc = id - row * 12;
        const long long t = t0 + row;
        const int n = n0 + pc * 8;
        if (t < p.T && n < p.N) {
            u32x4 v = *reinterpret_cast<const u32x4 *>(lds + row * CROW + pc * 16);
            if (p.R) v = add_bf16x8(rv[j], v);
            *reinterpret_cast<u32x4 *>(p.C + t * p.ldc + n) = v;
        }
    }
}


// ---- fp32 operands (the reference's own precision): the same tiles on the bf16 MFMA, every fp32 value taken as the exact
// sum of three bf16 values (split3_bf16, ss2d_common.h) and every 16-k product as six MFMAs.
//   * the token operand never visits LDS: a wave's 32 tokens x 32 k of a chunk are its own MFMA operand, so lane (token
//     l32, half h) reads the 64 contiguous bytes k = 16h .. 16h+15 of its row (a whole cache line per row and chunk) and
//     splits them in registers -- the k order inside a chunk is free as long as both operands agree: k-step s of half h is
//     k = 16h + 8s .. + 7;
//   * the weight chunk (shared by the four waves) is split on its way to LDS, into three images of the bf16 kernel's layouts;
//   * 36 MFMAs per wave between barriers; LDS is the 51 KB fp32 result image (three workgroups per CU).
// Needs K, N and the leading dimensions multiples of 4 and 16-byte aligned operands.
struct TgParamsF {
    const float *A, *A2, *B;
    const float *bias;
    const float *R;             // residual (see TgParams)
    float *C;
    long long T;
    int N, K, K1, lda, lda2, ldb, ldc, ldr, tiles_n;
};

constexpr int FK = 32;                                    // k per chunk
constexpr int FROW = 2 * FK + 16;                         // bytes per LDS row of an image of the forward's weight chunk [96 n][32 k]
constexpr int fimg_b(bool kmajor) { return kmajor ? FK * BROW_NN : GN * FROW; }
constexpr int fstage(bool kmajor) { return 3 * fimg_b(kmajor); }
constexpr int FCROW = 4 * GN + 16;                        // result image [128 t][96 n] fp32
constexpr int flds(bool kmajor) { return 2 * fstage(kmajor) > GT * FCROW ? 2 * fstage(kmajor) : GT * FCROW; }

template <bool B_KMAJOR>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void tok_gemm_f32_kernel(const TgParamsF p) {
    constexpr int IMG_B = fimg_b(B_KMAJOR), STAGE = fstage(B_KMAJOR), BROW = B_KMAJOR ? BROW_NN : FROW;
    __shared__ __align__(16) unsigned char lds[flds(B_KMAJOR)];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = blk % p.tiles_n;
    const long long t0 = (long long)(blk / p.tiles_n) * GT;
    const int n0 = nt * GN;
    const long long trow = t0 + 32 * wave + l32;           // this lane's token
    const bool trow_ok = trow < p.T;
    u32x4 ra[4], pb[3];
    // the weight chunk: 768 16-byte pieces, [96 n] x 8 or [32 k] x 24, three per thread
    int brow[3], bcol[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int id = j * 256 + tid;
        if (B_KMAJOR) { brow[j] = id / 24; bcol[j] = (id - brow[j] * 24) * 4; }     // row = k, col = n
        else { brow[j] = id >> 3; bcol[j] = (id & 7) * 4; }                          // row = n, col = k
    }
    auto load_chunk = [&](int k0) {
        // never a guarded load: the address is clamped to element 0 and the value masked
        const bool second = k0 >= p.K1;                    // K1 is a multiple of 32: a chunk lies in one of the two token operands
        const float *__restrict__ Ap = second ? p.A2 : p.A;
        const long long lda = second ? p.lda2 : p.lda;
        const int ka = (second ? k0 - p.K1 : k0) + 16 * half;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = trow_ok && k0 + 16 * half + 4 * i < p.K;
            const u32x4 v = *reinterpret_cast<const u32x4 *>(Ap + (ok ? trow * lda + ka + 4 * i : 0));
            ra[i] = v & (ok ? ~0u : 0u);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            bool ok;
            long long off;
            if (B_KMAJOR) { ok = k0 + brow[j] < p.K && n0 + bcol[j] < p.N; off = (long long)(k0 + brow[j]) * p.ldb + n0 + bcol[j]; }
            else { ok = n0 + brow[j] < p.N && k0 + bcol[j] < p.K; off = (long long)(n0 + brow[j]) * p.ldb + k0 + bcol[j]; }
            const u32x4 v = *reinterpret_cast<const u32x4 *>(p.B + (ok ? off : 0));
            pb[j] = v & (ok ? ~0u : 0u);
        }
    };
    auto put_chunk = [&](int s) {
        unsigned char *bs = lds + s * STAGE;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const Split4 sp = split3_bf16(pb[j]);
            unsigned char *d = bs + brow[j] * BROW + bcol[j] * 2;
            *reinterpret_cast<u32x2 *>(d) = sp.h1;
            *reinterpret_cast<u32x2 *>(d + IMG_B) = sp.h2;
            *reinterpret_cast<u32x2 *>(d + 2 * IMG_B) = sp.h3;
        }
    };
    f16v acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    const int boff_nt = l32 * FROW + 32 * half;                             // + 32 rows per column tile, + 16 bytes per k-step
    // transposed read (dX): 16-lane group gq covers columns 16*(gq & 1) .. +15 of a 32-column tile and the k rows
    // 16*(gq >> 1) + 8s .. +7 of k-step s in two blocks of 4; lane 4q + pp addresses k row q, columns 4pp .. 4pp+3
    const int gq = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int boff_nn = (16 * (gq >> 1) + q) * BROW_NN + (16 * (gq & 1) + 4 * pp) * 2;
    const int nchunks = (p.K + FK - 1) / FK;
    load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
        const int s = c & 1;
        put_chunk(s);                                        // stage s was last read two chunks ago, behind the previous barrier
        u32x4 af[2][3];                                      // [k-step][rank]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const Split4 lo = split3_bf16(ra[2 * ks]), hi = split3_bf16(ra[2 * ks + 1]);
            af[ks][0] = u32x4{lo.h1.x, lo.h1.y, hi.h1.x, hi.h1.y};
            af[ks][1] = u32x4{lo.h2.x, lo.h2.y, hi.h2.x, hi.h2.y};
            af[ks][2] = u32x4{lo.h3.x, lo.h3.y, hi.h3.x, hi.h3.y};
        }
        load_chunk((c + 1 < nchunks ? c + 1 : c) * FK);      // in flight during the products (the last one is a harmless repeat)
        __syncthreads();
        const unsigned char *bs = lds + s * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                bf16x8 bf[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    if (B_KMAJOR) {
                        typedef s16x4 __attribute__((address_space(3))) *lp;
                        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(bs + i * IMG_B + boff_nn + 8 * ks * BROW_NN + 64 * j));
                        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(bs + i * IMG_B + boff_nn + (8 * ks + 4) * BROW_NN + 64 * j));
                        bf[i] = __builtin_bit_cast(bf16x8, s16x8{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w});
                    } else {
                        bf[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(bs + i * IMG_B + boff_nt + 32 * j * FROW + 16 * ks));
                    }
                }
                const bf16x8 a1 = __builtin_bit_cast(bf16x8, af[ks][0]), a2 = __builtin_bit_cast(bf16x8, af[ks][1]), a3 = __builtin_bit_cast(bf16x8, af[ks][2]);
                // the TRANSPOSED block (rows = columns n, columns = tokens); the small terms first
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[1], a2, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[2], a1, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[0], a3, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[1], a1, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[0], a2, acc[j], 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[0], a1, acc[j], 0, 0, 0);
            }
        }
    }
    __syncthreads();
    // accumulator register e of a 32x32 block: row (= column n of the result) 8*(e/4) + 4*(lane/32) + e%4, column (= token)
    // lane%32 -- four consecutive n of one token per register quad: one 16-byte LDS store
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f4 v = f4{acc[j][4 * g], acc[j][4 * g + 1], acc[j][4 * g + 2], acc[j][4 * g + 3]};
            if (p.bias) {       // N % 4 == 0: the four columns are inside N together
                const int n = n0 + 32 * j + 8 * g + 4 * half;
                v += *reinterpret_cast<const f4 *>(p.bias + (n < p.N ? n : 0));
            }
            *reinterpret_cast<f4 *>(lds + (32 * wave + l32) * FCROW + (32 * j + 8 * g + 4 * half) * 4) = v;
        }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 12; ++j) {
        const int id = j * 256 + tid, row = id / 24, pc = id - row * 24;
        const long long t = t0 + row;
        const int n = n0 + pc * 4;
        if (t < p.T && n < p.N) {
            f4 v = *reinterpret_cast<const f4 *>(lds + row * FCROW + pc * 16);
            if (p.R) v = *reinterpret_cast<const f4 *>(p.R + t * p.ldr + n) + v;
            *reinterpret_cast<f4 *>(p.C + t * p.ldc + n) = v;
        }
    }
}

}  // namespace

namespace {
int tg_run(const char *name, int dtype, const void *A, const void *A2, const void *B, const float *bias, const void *R, void *C, long long T,
           int N, int K, int K1, int lda, int lda2, int ldb, int ldr, int ldc, int b_k_major, void *stream) {
    if (dtype != SS2D_DTYPE_BF16 && dtype != SS2D_DTYPE_F32) { ss2d_set_error("%s: unknown dtype %d", name, dtype); return -1; }
    const bool f32 = dtype == SS2D_DTYPE_F32;
    const int q = f32 ? 4 : 8;                                 // elements per 16 bytes
    if (!A || !B || !C || (K1 < K && !A2)) { ss2d_set_error("%s: null required pointer", name); return -2; }
    if (T <= 0 || N <= 0 || K <= 0 || K1 <= 0 || K1 > K || lda < K1 || (K1 < K && lda2 < K - K1) || ldc < N || ldb < (b_k_major ? N : K)) {
        ss2d_set_error("%s: bad size", name); return -3; }
    if (R && ldr < N) { ss2d_set_error("%s: bad size", name); return -3; }
    if (N % q || K % q || lda % q || ldb % q || ldc % q || (K1 < K && lda2 % q) || (R && ldr % q) ||
        ((((uintptr_t)A | (uintptr_t)(K1 < K ? A2 : A) | (uintptr_t)B | (uintptr_t)C | (uintptr_t)bias | (uintptr_t)R) & 15) != 0)) {
        ss2d_set_error("%s: N, K, the leading dimensions must be multiples of %d and the operands 16-byte aligned", name, q); return -4; }
    const bool k64 = K > 256;
    const int kchunk = f32 ? FK : k64 ? 64 : 32;
    if (K1 < K && K1 % kchunk) { ss2d_set_error("%s: K1 must be a multiple of the k chunk (%d)", name, kchunk); return -4; }
    const int tiles_n = (N + GN - 1) / GN;
    const long long nwg = ((T + GT - 1) / GT) * tiles_n;
    if (nwg > 0x7fffffffLL) { ss2d_set_error("%s: grid too large", name); return -6; }
    hipStream_t s = (hipStream_t)stream;
    if (f32) {
        TgParamsF p;
        p.A = static_cast<const float *>(A); p.A2 = static_cast<const float *>(K1 < K ? A2 : A); p.B = static_cast<const float *>(B);
        p.bias = bias; p.C = static_cast<float *>(C); p.R = static_cast<const float *>(R); p.ldr = ldr;
        p.T = T; p.N = N; p.K = K; p.K1 = K1; p.lda = lda; p.lda2 = K1 < K ? lda2 : lda; p.ldb = ldb; p.ldc = ldc; p.tiles_n = tiles_n;
        if (b_k_major) ss2d_launch((tok_gemm_f32_kernel<true>), dim3((unsigned)nwg), dim3(256), 0, s, p);
        else ss2d_launch((tok_gemm_f32_kernel<false>), dim3((unsigned)nwg), dim3(256), 0, s, p);
    } else {
        TgParams p;
        p.A = static_cast<const bf16 *>(A); p.A2 = static_cast<const bf16 *>(K1 < K ? A2 : A); p.B = static_cast<const bf16 *>(B);
        p.bias = bias; p.C = static_cast<bf16 *>(C); p.R = static_cast<const bf16 *>(R); p.ldr = ldr;
        p.T = T; p.N = N; p.K = K; p.K1 = K1; p.lda = lda; p.lda2 = K1 < K ? lda2 : lda; p.ldb = ldb; p.ldc = ldc; p.tiles_n = tiles_n;
        if (b_k_major) { if (k64) ss2d_launch((tok_gemm_kernel<true, 64>), dim3((unsigned)nwg), dim3(256), 0, s, p); else ss2d_launch((tok_gemm_kernel<true, 32>), dim3((unsigned)nwg), dim3(256), 0, s, p); }
        else { if (k64) ss2d_launch((tok_gemm_kernel<false, 64>), dim3((unsigned)nwg), dim3(256), 0, s, p); else ss2d_launch((tok_gemm_kernel<false, 32>), dim3((unsigned)nwg), dim3(256), 0, s, p); }
    }
    const int rc = ss2d_take_launch_rc();
    if (rc) ss2d_set_error("%s: launch failed: %s", name, hipGetErrorString((hipError_t)rc));
    return rc;
}
}  // namespace

extern "C" int ss2d_token_gemm(int dtype, const void *A, const void *B, void *C, long long T, int N, int K, int lda, int ldb,
                               int ldc, int b_k_major, void *stream) {
    return tg_run("token_gemm", dtype, A, nullptr, B, nullptr, nullptr, C, T, N, K, K, lda, 0, ldb, 0, ldc, b_k_major, stream);
}

// C (T x N) = R + A (T x K) * W^T, the product rounded to the storage type before the sum (what a Linear followed by `x + y`
// gives): out_proj and the residual connection of a VSSBlock (mamba_sys.py:562, :585) in one pass.  C may be R.
extern "C" int ss2d_token_gemm_res(int dtype, const void *A, const void *W, const void *R, void *C, long long T, int N, int K,
                                   int lda, int ldw, int ldr, int ldc, void *stream) {
    if (!R) { ss2d_set_error("token_gemm_res: null required pointer"); return -2; }
    return tg_run("token_gemm_res", dtype, A, nullptr, W, nullptr, R, C, T, N, K, K, lda, 0, ldw, ldr, ldc, 0, stream);
}

// C (T x N) = [A | A2] (T x (K1 + K2)) * W^T + bias: a Linear over a concatenated input without the concatenation, the bias
// added in fp32 before the result is rounded.  A2 may be null with K1 == K (a plain biased Linear).
extern "C" int ss2d_token_gemm_cat(int dtype, const void *A, const void *A2, const void *W, const float *bias, void *C, long long T,
                                   int N, int K, int K1, int lda, int lda2, int ldw, int ldc, void *stream) {
    return tg_run("token_gemm_cat", dtype, A, A2, W, bias, nullptr, C, T, N, K, K1, lda, lda2, ldw, 0, ldc, 0, stream);
}
