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
// axis; token slabs are spread over the whole chip, the four partial tiles of a workgroup are summed in LDS and
// added into C with one fp32 atomic per element and workgroup (C is zero-filled by the caller, like the other
// accumulated gradients of the ABI).
// bf16 operands take v_mfma_f32_32x32x16_bf16 (second kernel below): bf16 products are exact in fp32, so the
// gradient keeps fp32 products and sums, at 16x the MFMA rate; that kernel is bound by its reads
// (stage 0 in_proj: 27 us of loop for 72 MB, plus ~10 us of reduction and atomics).
#include <cstdlib>
#include <type_traits>
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
    int M, N, lda, ldb, ldc, tiles_m, tiles_n, ksplit;
    long long slab;  // token rows per slab (even)
};

// The four waves of a workgroup own the same 96 x 96 tile of C and four consecutive token slabs.  Their partial
// tiles are summed in LDS and leave as ONE set of global atomics per workgroup: the memory-side
// atomic rate (~1.3 TB/s of added bytes chip-wide) otherwise bounds the kernel -- 1024 waves x 9216 elements took
// about as long as reading the operands.
__device__ __forceinline__ void tile_of_wave(const WgradParams &p, int wave, int &m0, int &n0, long long &r0, long long &r1) {
    const int tiles = p.tiles_m * p.tiles_n;
    // the tiles of one slab group run back to back on one XCD: they read the same B rows, which then come from that L2
    const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int tile = lid % tiles;
    const long long ks = (long long)(lid / tiles) * 4 + wave;
    m0 = (tile / p.tiles_n) * WT; n0 = (tile % p.tiles_n) * WT;
    r0 = ks * p.slab;
    r1 = (r0 + p.slab < p.rows) ? r0 + p.slab : p.rows;   // r1 <= r0 for a slab past the end: that wave adds zeros
}

__device__ __forceinline__ void flush_tile(const WgradParams &p, float *tile, const f16v (&acc)[3][3], int m0, int n0, int lane, int wave) {
    const int half = lane >> 5, l32 = lane & 31;
    // The waves take turns (plain LDS stores, then read-add-write): float atomics on LDS cost 40 us here.
    // accumulator register e of a 32x32 tile: row = 8*(e/4) + 4*(lane/32) + e%4, column = lane%32
#ifdef SS2D_WGRAD_NOTURNS
    if (acc[0][0][0] == 12345.678f) tile[lane] = acc[1][1][1] + acc[2][2][2] + acc[0][2][5] + acc[2][0][7] + acc[1][0][3] + acc[0][1][9] + acc[1][2][11] + acc[2][1][13];
    return;
#endif
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float *t = tile + (32 * i + 8 * (e >> 2) + 4 * half + (e & 3)) * WT + 32 * j + l32;
                        *t = w == 0 ? acc[i][j][e] : *t + acc[i][j][e];
                    }
        }
        __syncthreads();
    }
#ifndef SS2D_WGRAD_NOATOMICS   // (timing experiments only)
    for (int i = threadIdx.x; i < WT * WT; i += 256) {
        const int m = m0 + i / WT, n = n0 + i % WT;
        if (m < p.M && n < p.N) atomicAdd(p.C + (size_t)m * p.ldc + n, tile[i]);
    }
#endif
}

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
    __shared__ float ctile[WT * WT];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // make everything derived from it scalar
    int m0, n0;
    long long r0, r1;
    tile_of_wave(p, wave, m0, n0, r0, r1);
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
    flush_tile(p, ctile, acc, m0, n0, lane, wave);
}

// ---- bf16 operands: v_mfma_f32_32x32x16_bf16.  The products of two bf16 values are exact in fp32, so this is
// the same arithmetic as widening to fp32 (fp32 sums, another order) at 16x the MFMA rate: the fp32 form above is
// MFMA-bound for bf16 inputs (64 cycles per 2 tokens per 32x32 block), this one is bound by its reads.
// Operand layout (cdna_hip_programming.md, A/B operand lane maps): lane (r = lane & 31, h = lane >> 5) holds
// k = 8h .. 8h+7 of row/column r, i.e. 8 consecutive tokens of one column: eight 2-byte reads at stride lda that
// land in the halves of four registers (global_load_short_d16 / _d16_hi), each a 64-byte run per 32 lanes.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int KG = 16;    // tokens per MFMA

// Full group: token row r + j is workgroup-uniform, so a read is a scalar row base plus one of six loop-invariant
// 32-bit lane offsets (aoff/boff = ((8h) * ld + column) * 2 bytes).  The 48 two-byte reads of a group land
// zero-extended in 48 registers (`Raw`) and are packed into the 24 operand registers (`Frag`) once they have
// arrived.  (global_load_short_d16 / _d16_hi would pack for free, but with SRAM ECC on they clear the other half
// of the register instead of preserving it.)  One raw group is in flight per wave while the previous one's MFMAs
// issue; two waves per SIMD (<= 256 registers with the 144 accumulators) keep ~100 KB per CU in flight.
struct Raw { unsigned a[3][8], b[3][8]; };
struct Frag { u32x4 a[3], b[3]; };

__device__ __forceinline__ void load_group(const bf16 *__restrict__ A, const bf16 *__restrict__ B, const WgradParams &p,
                                           long long r, const unsigned (&aoff)[3], const unsigned (&boff)[3], Raw &w) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const char *ra = reinterpret_cast<const char *>(A + (r + j) * p.lda);   // uniform: SGPR pair
        const char *rb = reinterpret_cast<const char *>(B + (r + j) * p.ldb);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            w.a[i][j] = *reinterpret_cast<const unsigned short *>(ra + aoff[i]);
            w.b[i][j] = *reinterpret_cast<const unsigned short *>(rb + boff[i]);
        }
    }
}
__device__ __forceinline__ void pack_group(const Raw &w, Frag &f) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f.a[i][q] = w.a[i][2 * q] | (w.a[i][2 * q + 1] << 16);
            f.b[i][q] = w.b[i][2 * q] | (w.b[i][2 * q + 1] << 16);
        }
}

// Ragged last group of a slab: rows past the end are clamped and zeroed with a bit mask.
__device__ __forceinline__ void load_group_tail(const bf16 *__restrict__ A, const bf16 *__restrict__ B, const WgradParams &p,
                                                long long r, long long r1, int h8, const int (&acol)[3], const int (&bcol)[3],
                                                bf16x8 (&a)[3], bf16x8 (&b)[3]) {
    u32x4 keep = u32x4{0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        long long row = r + h8 + j;
        if (row >= r1) keep[j >> 1] &= (j & 1) ? 0x0000ffffu : 0xffff0000u;
        row = row < r1 ? row : r1 - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            a[i][j] = A[row * p.lda + acol[i]];
            b[i][j] = B[row * p.ldb + bcol[i]];
        }
    }
    asm volatile("" : "+v"(keep));   // opaque: keeps the reads unconditional (see load_stage_impl)
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        a[i] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, a[i]) & keep);
        b[i] = __builtin_bit_cast(bf16x8, __builtin_bit_cast(u32x4, b[i]) & keep);
    }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void wgrad_bf16_kernel(const WgradParams p) {
    __shared__ float ctile[WT * WT];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    int m0, n0;
    long long r0, r1;
    tile_of_wave(p, wave, m0, n0, r0, r1);
    const bf16 *__restrict__ A = static_cast<const bf16 *>(p.A);
    const bf16 *__restrict__ B = static_cast<const bf16 *>(p.B);
    const int half = lane >> 5, l32 = lane & 31, h8 = 8 * half;
    int acol[3], bcol[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        acol[i] = m0 + 32 * i + l32; acol[i] = acol[i] < p.M ? acol[i] : p.M - 1;
        bcol[i] = n0 + 32 * i + l32; bcol[i] = bcol[i] < p.N ? bcol[i] : p.N - 1;
    }
    unsigned aoff[3], boff[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        aoff[i] = (unsigned)(h8 * p.lda + acol[i]) * 2u;
        boff[i] = (unsigned)(h8 * p.ldb + bcol[i]) * 2u;
    }
    f16v acc[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.0f;
    const long long nfull = (r1 - r0) / KG;
    if (nfull > 0) {   // every trip issues the same reads (the trip past the end re-reads the last group, unused)
        const long long rlast = r0 + (nfull - 1) * KG;
        Raw w;
        Frag f;
        load_group(A, B, p, r0, aoff, boff, w);
        for (long long g = 0; g < nfull; ++g) {
            pack_group(w, f);
            const long long rn = r0 + (g + 1) * KG;
            __builtin_amdgcn_sched_barrier(0);
            load_group(A, B, p, rn < rlast ? rn : rlast, aoff, boff, w);
            __builtin_amdgcn_sched_barrier(0);  // keep the reads ahead of the MFMAs they are meant to overlap
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 3; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, f.a[i]), __builtin_bit_cast(bf16x8, f.b[j]), acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (r0 + nfull * KG < r1) {  // ragged end of the slab: one masked group
        bf16x8 a0[3], b0[3];
        load_group_tail(A, B, p, r0 + nfull * KG, r1, h8, acol, bcol, a0, b0);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0[i], b0[j], acc[i][j], 0, 0, 0);
    }
    flush_tile(p, ctile, acc, m0, n0, lane, wave);
}

// ---- bf16 operands, LDS-staged.  The kernel above fetches its MFMA operands with 48 two-byte reads per 16 tokens and
// lane (the contraction index -- the token -- is the slow axis of both matrices), and its four waves each hold the whole
// 96 x 96 tile for a quarter of the tokens, so they meet in LDS by turns at the end.  Timing experiments
// (-DSS2D_WGRAD_NOTURNS / -DSS2D_WGRAD_NOATOMICS): of 41 us at the stage-0 in_proj shape the main loop is 17, the turns
// 15, the atomics 9.  Here:
//   * the workgroup copies a chunk of 64 token rows x (128 | 96) columns of the two operands to LDS as it lies in memory
//     (16-byte pieces) and the waves read it back with ds_read_b64_tr_b16, which delivers 4 consecutive tokens of one
//     column per lane: both MFMA operands without a transposing pass.  The A rows (256 bytes) are padded to 320 bytes
//     (4 consecutive token rows x 64 bytes start at banks 0 / 16 / 32 / 48), the B rows of 192 bytes = 48 banks fall on
//     0 / 48 / 32 / 16: both conflict-free;
//   * wave w owns output rows 32w .. 32w+31 of the 128 x 96 tile for ALL tokens of the workgroup's slab: 48 accumulator
//     registers instead of 144, no cross-wave sum; its three 32 x 32 results leave as fp32 atomics straight from the
//     accumulator layout (32 consecutive columns per instruction);
//   * the next chunk is in flight in registers during the MFMAs; one barrier per chunk.
// Needs M, N, lda, ldb multiples of 8 and 16-byte aligned operands (the host falls back to the kernel above otherwise).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
constexpr int TKC = 64, TM = 128, TN = 96;                  // tokens per chunk, tile rows (A columns), tile columns (B columns)
constexpr int TAROW = 2 * TM + 64, TBROW = 2 * TN;          // bytes per LDS row
constexpr int TSTAGE = TKC * (TAROW + TBROW);

__global__ __launch_bounds__(256) void wgrad_bf16_tr_kernel(const WgradParams p) {
    __shared__ __align__(16) unsigned char lds[2 * TSTAGE];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = p.tiles_m * p.tiles_n;
    const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int tile = lid % tiles;
    const int m0 = (tile / p.tiles_n) * TM, n0 = (tile % p.tiles_n) * TN;
    const long long r0 = (long long)(lid / tiles) * p.slab;
    long long r1 = r0 + p.slab;
    r1 = r1 < p.rows ? r1 : p.rows;
    const bf16 *__restrict__ A = static_cast<const bf16 *>(p.A);
    const bf16 *__restrict__ B = static_cast<const bf16 *>(p.B);
    // this thread's pieces of a chunk: 4 of A (64 rows x 16 pieces), 3 of B (64 rows x 12 pieces)
    // TWO chunks are in flight in registers (set = stage = chunk parity): with one, every chunk waited a full memory round
    // trip -- a workgroup is alone on its CU (about one per CU is launched) and the MFMAs of a chunk take a fraction of that.
    // The operands come through buffer descriptors that end with the slab (r1 rows): a row at or past the slab end reads
    // zeros by the range check, and a column past M / N is sent out of range by its offset.  Nothing touches a loaded
    // value before its ds_write: the first version zeroed those pieces with an AND on the value, the compiler gathered the
    // ANDs of BOTH sets at the loop latch behind one s_waitcnt vmcnt(0), and every trip waited for the loads it had just
    // issued (main loop 13-20 us whatever the shape).  The chunk's row offset is a scalar (soffset), so a chunk costs no
    // address arithmetic at all.  (Host: (rows + 4 chunks) * ld * 2 bytes < 2^31.)
    u32x4 pa[2][4], pb[2][3];
    int brow[3], bcol[3], aofs[4], bofs[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { const int id = j * 256 + tid; brow[j] = id / 12; bcol[j] = (id - brow[j] * 12) * 8; }
    const int arow = tid >> 4, acol = (tid & 15) * 8;          // piece j: row arow + 16 j
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, (unsigned)(r1 > 0 ? r1 * p.lda * 2 : 0));
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(B, (unsigned)(r1 > 0 ? r1 * p.ldb * 2 : 0));
#pragma unroll
    for (int j = 0; j < 4; ++j) aofs[j] = m0 + acol < p.M ? ((arow + 16 * j) * p.lda + m0 + acol) * 2 : SS2D_BUF_OOR;
#pragma unroll
    for (int j = 0; j < 3; ++j) bofs[j] = n0 + bcol[j] < p.N ? (brow[j] * p.ldb + n0 + bcol[j]) * 2 : SS2D_BUF_OOR;
    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, 1> S1{};
    auto load_chunk = [&](auto set, long long rc) {              // rows at or past r1 (also whole chunks) read as zeros
        constexpr int S = decltype(set)::value;
        const int sa = (int)(rc * p.lda * 2), sb = (int)(rc * p.ldb * 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) pa[S][j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, aofs[j], sa, 0);
#pragma unroll
        for (int j = 0; j < 3; ++j) pb[S][j] = __builtin_amdgcn_raw_buffer_load_b128(rsB, bofs[j], sb, 0);
    };
    auto put_chunk = [&](auto set) {
        constexpr int S = decltype(set)::value;
        unsigned char *as = lds + S * TSTAGE, *bs = as + TKC * TAROW;
#pragma unroll
        for (int j = 0; j < 4; ++j) *reinterpret_cast<u32x4 *>(as + (arow + 16 * j) * TAROW + acol * 2) = pa[S][j];
#pragma unroll
        for (int j = 0; j < 3; ++j) *reinterpret_cast<u32x4 *>(bs + brow[j] * TBROW + bcol[j] * 2) = pb[S][j];
    };
    f16v acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    // transposed read: 16-lane group gq covers columns 16*(gq & 1) .. +15 of a 32-column block and the tokens
    // 8*(gq >> 1) .. +7 of a 16-token step in two blocks of 4; lane 4q + pp addresses token row q, columns 4pp .. 4pp+3
    const int gq = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int trow = 8 * (gq >> 1) + q, tcol = (16 * (gq & 1) + 4 * pp) * 2;
    const int aoff = trow * TAROW + 64 * wave + tcol, boff = trow * TBROW + tcol;
    const bool live = m0 + 32 * wave < p.M;                     // a wave whose rows lie past M multiplies nothing
    const long long nchunks = r1 > r0 ? (r1 - r0 + TKC - 1) / TKC : 0;
    // (Requesting the fragments of k-step ks + 1 ahead of the MFMAs of k-step ks changes nothing: a chunk costs 0.86 us
    // because the 28 KB it pulls through L2 arrive at ~9 TB/s chip-wide, tools/wgrad_floor_probe.py -- every shape of the
    // model moves the same 101 / 50 MB of tile operands, and its atomics cost about as much again.)
    auto products = [&](auto set) {
        constexpr int s = decltype(set)::value;
        if (live) {
            typedef s16x4 __attribute__((address_space(3))) *lp;
            const unsigned char *as = lds + s * TSTAGE + aoff, *bs = lds + s * TSTAGE + TKC * TAROW + boff;
#pragma unroll
            for (int ks = 0; ks < TKC / 16; ++ks) {
                const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(as + 16 * ks * TAROW));
                const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(as + (16 * ks + 4) * TAROW));
                const bf16x8 af = __builtin_bit_cast(bf16x8, s16x8{a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w});
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(bs + 16 * ks * TBROW + 64 * j));
                    const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(bs + (16 * ks + 4) * TBROW + 64 * j));
                    const bf16x8 bf = __builtin_bit_cast(bf16x8, s16x8{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w});
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[j], 0, 0, 0);
                }
            }
        }
    };
    // (the barrier keeps set 0's loads in front of set 1's: interleaved, the first put of every trip would wait for both)
    if (nchunks > 0) { load_chunk(S0, r0); __builtin_amdgcn_sched_barrier(0); load_chunk(S1, r0 + TKC); }
    // two chunks per trip, nothing under a branch (an odd count ends with a chunk of zeros); a stage was last read two chunks
    // ago, behind the previous barrier
    for (long long c = 0; c < nchunks; c += 2) {
        put_chunk(S0);
        load_chunk(S0, r0 + (c + 2) * TKC);
        __syncthreads();
        products(S0);
        put_chunk(S1);
        load_chunk(S1, r0 + (c + 3) * TKC);
        __syncthreads();
        products(S1);
    }
#ifndef SS2D_WGRAD_NOATOMICS   // (timing experiments only)
    // accumulator register e of a 32x32 block: row = 8*(e/4) + 4*(lane/32) + e%4, column = lane%32
    if (live && nchunks > 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int n = n0 + 32 * j + l32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + 32 * wave + 8 * (e >> 2) + 4 * half + (e & 3);
                if (m < p.M && n < p.N) atomicAdd(p.C + (size_t)m * p.ldc + n, acc[j][e]);
            }
        }
    }
#else   // keeps the products alive without the flush (a value that never occurs)
    if (live && nchunks > 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (acc[j][e] == 1.2345e30f) atomicAdd(p.C, acc[j][e]);
    }
#endif
}


// ---- fp32 operands, LDS-staged, on the bf16 MFMA: the kernel above with every fp32 value split on its way to LDS into the
// exact sum of three bf16 values (split3_bf16, ss2d_common.h: three images of the chunk) and six MFMAs per product --
// fp32 products and sums at 3/8 of the MFMA cycles of v_mfma_f32_32x32x2_f32 (wgrad_kernel<float>), and operands read as
// whole 16-byte pieces.  A chunk is 16 tokens (one MFMA k-step); two stages of three images are 48 KB of LDS and the launch
// aims at three workgroups per CU: with one, the splitting (VALU), the MFMAs and the memory round trips of a workgroup add
// up instead of overlapping.  Needs M, N, lda, ldb multiples of 4 and 16-byte aligned operands.
constexpr int FTKC = 16, FTIMG_A = FTKC * TAROW, FTIMG_B = FTKC * TBROW, FTSTAGE = 3 * (FTIMG_A + FTIMG_B);

__global__ __launch_bounds__(256) void wgrad_f32_tr_kernel(const WgradParams p) {
    __shared__ __align__(16) unsigned char lds[2 * FTSTAGE];
    const int tid = threadIdx.x, lane = tid & 63, l32 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = p.tiles_m * p.tiles_n;
    const int lid = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int tile = lid % tiles;
    const int m0 = (tile / p.tiles_n) * TM, n0 = (tile % p.tiles_n) * TN;
    const long long r0 = (long long)(lid / tiles) * p.slab;
    long long r1 = r0 + p.slab;
    r1 = r1 < p.rows ? r1 : p.rows;
    const float *__restrict__ A = static_cast<const float *>(p.A);
    const float *__restrict__ B = static_cast<const float *>(p.B);
    // this thread's pieces of a chunk: 2 of A (16 rows x 32 pieces of 4 floats), 1.5 of B (16 rows x 24 pieces); two chunks in
    // flight in registers (set = stage = chunk parity)
    constexpr int NA = FTKC * 32 / 256, NB = (FTKC * 24 + 255) / 256;
    u32x4 pa[2][NA], pb[2][NB];
    int brow[NB], bcol[NB], aofs[NA], bofs[NB];
    bool bon[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int id = j * 256 + tid;
        bon[j] = id < FTKC * 24;
        brow[j] = (bon[j] ? id : 0) / 24; bcol[j] = ((bon[j] ? id : 0) - brow[j] * 24) * 4;
    }
    const int arow = tid >> 5, acol = (tid & 31) * 4;          // piece j: row arow + 8 j
    // operands through descriptors that end with the slab, columns past M / N sent out of range by their offset, the
    // chunk's rows chosen by a scalar offset: see wgrad_bf16_tr_kernel
    const __amdgpu_buffer_rsrc_t rsA = make_rsrc(A, (unsigned)(r1 > 0 ? r1 * p.lda * 4 : 0));
    const __amdgpu_buffer_rsrc_t rsB = make_rsrc(B, (unsigned)(r1 > 0 ? r1 * p.ldb * 4 : 0));
#pragma unroll
    for (int j = 0; j < NA; ++j) aofs[j] = m0 + acol < p.M ? ((arow + 8 * j) * p.lda + m0 + acol) * 4 : SS2D_BUF_OOR;
#pragma unroll
    for (int j = 0; j < NB; ++j) bofs[j] = (bon[j] && n0 + bcol[j] < p.N) ? (brow[j] * p.ldb + n0 + bcol[j]) * 4 : SS2D_BUF_OOR;
    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, 1> S1{};
    auto load_chunk = [&](auto set, long long rc) {              // rows at or past r1 (also whole chunks) read as zeros
        constexpr int S = decltype(set)::value;
        const int sa = (int)(rc * p.lda * 4), sb = (int)(rc * p.ldb * 4);
#pragma unroll
        for (int j = 0; j < NA; ++j) pa[S][j] = __builtin_amdgcn_raw_buffer_load_b128(rsA, aofs[j], sa, 0);
#pragma unroll
        for (int j = 0; j < NB; ++j) pb[S][j] = __builtin_amdgcn_raw_buffer_load_b128(rsB, bofs[j], sb, 0);
    };
    auto put_chunk = [&](auto set) {
        constexpr int S = decltype(set)::value;
        unsigned char *as = lds + S * FTSTAGE, *bs = as + 3 * FTIMG_A;
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            const Split4 sp = split3_bf16(pa[S][j]);
            unsigned char *d = as + (arow + 8 * j) * TAROW + acol * 2;
            *reinterpret_cast<u32x2 *>(d) = sp.h1;
            *reinterpret_cast<u32x2 *>(d + FTIMG_A) = sp.h2;
            *reinterpret_cast<u32x2 *>(d + 2 * FTIMG_A) = sp.h3;
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            if (!bon[j]) continue;
            const Split4 sp = split3_bf16(pb[S][j]);
            unsigned char *d = bs + brow[j] * TBROW + bcol[j] * 2;
            *reinterpret_cast<u32x2 *>(d) = sp.h1;
            *reinterpret_cast<u32x2 *>(d + FTIMG_B) = sp.h2;
            *reinterpret_cast<u32x2 *>(d + 2 * FTIMG_B) = sp.h3;
        }
    };
    f16v acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    const int gq = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;          // transposed read: see wgrad_bf16_tr_kernel
    const int trow = 8 * (gq >> 1) + q, tcol = (16 * (gq & 1) + 4 * pp) * 2;
    const int aoff = trow * TAROW + 64 * wave + tcol, boff = trow * TBROW + tcol;
    const bool live = m0 + 32 * wave < p.M;
    const long long nchunks = r1 > r0 ? (r1 - r0 + FTKC - 1) / FTKC : 0;
    auto products = [&](auto set) {
        constexpr int s = decltype(set)::value;
        if (live) {
            typedef s16x4 __attribute__((address_space(3))) *lp;
            const unsigned char *as = lds + s * FTSTAGE + aoff, *bs = lds + s * FTSTAGE + 3 * FTIMG_A + boff;
#pragma unroll
            for (int ks = 0; ks < FTKC / 16; ++ks) {
                bf16x8 af[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(as + i * FTIMG_A + 16 * ks * TAROW));
                    const s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(as + i * FTIMG_A + (16 * ks + 4) * TAROW));
                    af[i] = __builtin_bit_cast(bf16x8, s16x8{a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w});
                }
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    bf16x8 bf[3];
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(bs + i * FTIMG_B + 16 * ks * TBROW + 64 * j));
                        const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(bs + i * FTIMG_B + (16 * ks + 4) * TBROW + 64 * j));
                        bf[i] = __builtin_bit_cast(bf16x8, s16x8{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w});
                    }
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[1], acc[j], 0, 0, 0);      // the small terms first
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[2], bf[0], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[2], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1], bf[0], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[1], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0], bf[0], acc[j], 0, 0, 0);
                }
            }
        }
    };
    if (nchunks > 0) { load_chunk(S0, r0); __builtin_amdgcn_sched_barrier(0); load_chunk(S1, r0 + FTKC); }
    for (long long c = 0; c < nchunks; c += 2) {                 // see wgrad_bf16_tr_kernel
        put_chunk(S0);
        load_chunk(S0, r0 + (c + 2) * FTKC);
        __syncthreads();
        products(S0);
        put_chunk(S1);
        load_chunk(S1, r0 + (c + 3) * FTKC);
        __syncthreads();
        products(S1);
    }
    if (live && nchunks > 0) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int n = n0 + 32 * j + l32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + 32 * wave + 8 * (e >> 2) + 4 * half + (e & 3);
                if (m < p.M && n < p.N) atomicAdd(p.C + (size_t)m * p.ldc + n, acc[j][e]);
            }
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
    static const bool env_direct = getenv("SS2D_WGRAD_DIRECT") != nullptr;   // A/B against the direct-read bf16 form, read once
    static const bool env_widen = getenv("SS2D_WGRAD_WIDEN") != nullptr;     // A/B against the fp32-MFMA form, read once
    const int q16 = dtype == SS2D_DTYPE_BF16 ? 8 : 4;            // elements per 16 bytes
    // (fp32 with SS2D_WGRAD_WIDEN: the fp32-MFMA form, for A/B)
    // (the LDS-staged kernels address their operands with 32-bit byte offsets: rows + 4 chunks of the widest operand < 2^31 bytes)
    const long long widest = (rows + 4 * TKC) * (long long)(lda > ldb ? lda : ldb) * (dtype == SS2D_DTYPE_BF16 ? 2 : 4);
    const bool use_tr = !env_widen && !(env_direct && dtype == SS2D_DTYPE_BF16) && M % q16 == 0 && N % q16 == 0 && lda % q16 == 0 && ldb % q16 == 0 &&
                        ((((uintptr_t)A | (uintptr_t)B) & 15) == 0) && widest < (1ll << 31);
    long long target_waves = 1024;
    static const long long env_waves = ss2d_env_ll("SS2D_WGRAD_WAVES", 0);   // tuning knob, read once
    if (env_waves >= 4 && env_waves <= 65536) target_waves = env_waves;
    else if (use_tr && dtype == SS2D_DTYPE_F32) target_waves = 3072;                 // three workgroups per CU (wgrad_f32_tr_kernel: 1024 / 2048 / 3072 -> 74 / 69 / 64 us at stage 0)
    else if (use_tr && (double)rows * (M + N) * 2.0 > 150e6) target_waves = 1536;   // 246 MB final expand: 60 us against 78
    long long nwg;
    if (use_tr) {
        // 128 x 96 tiles, one token slab per workgroup (its four waves split the tile's rows, not the tokens): about
        // target_waves / 4 workgroups, at least 256 token rows each
        p.tiles_m = (M + TM - 1) / TM; p.tiles_n = (N + TN - 1) / TN;
        const int tiles = p.tiles_m * p.tiles_n;
        long long ks = (target_waves / 4 + tiles - 1) / tiles;
        const long long max_ks = (rows + 255) / 256;
        if (ks > max_ks) ks = max_ks;
        if (ks < 1) ks = 1;
        p.slab = (rows + ks - 1) / ks;
        ks = (rows + p.slab - 1) / p.slab;
        p.ksplit = (int)ks;
        nwg = ks * tiles;
    } else {
    p.tiles_m = (M + WT - 1) / WT; p.tiles_n = (N + WT - 1) / WT;
    const int tiles = p.tiles_m * p.tiles_n;
    // token slabs: 4 per workgroup (one per wave), about `target_waves` waves in all, at least 128 rows each.
    // One wave per SIMD on every CU: the fp32 form cannot hold more (298 registers); the bf16 form could hold two
    // (242), but twice the workgroups flush twice the atomics and measured 5-10 % slower.
    long long ks = (target_waves + tiles - 1) / tiles;
    const long long max_ks = (rows + 127) / 128;
    if (ks > max_ks) ks = max_ks;
    ks = (ks + 3) / 4 * 4;
    long long slab = (rows + ks - 1) / ks;
    slab = (slab + 1) / 2 * 2;
    p.slab = slab; p.ksplit = (int)ks;
    nwg = (ks / 4) * tiles;
    }
    if (nwg > 0x7fffffffLL) { ss2d_set_error("linear_wgrad: grid too large"); return -6; }
    hipStream_t s = (hipStream_t)stream;
    if (use_tr && dtype == SS2D_DTYPE_F32) ss2d_launch(wgrad_f32_tr_kernel, dim3((unsigned)nwg), dim3(256), 0, s, p);
    else if (use_tr) ss2d_launch(wgrad_bf16_tr_kernel, dim3((unsigned)nwg), dim3(256), 0, s, p);
    else if (dtype == SS2D_DTYPE_BF16 && !env_widen) ss2d_launch(wgrad_bf16_kernel, dim3((unsigned)nwg), dim3(256), 0, s, p);   // env: A/B against the fp32-MFMA form
    else if (dtype == SS2D_DTYPE_BF16) ss2d_launch(wgrad_kernel<bf16>, dim3((unsigned)nwg), dim3(256), 0, s, p);
    else ss2d_launch(wgrad_kernel<float>, dim3((unsigned)nwg), dim3(256), 0, s, p);
    const int rc = ss2d_take_launch_rc();
    if (rc) ss2d_set_error("linear_wgrad: launch failed: %s", hipGetErrorString((hipError_t)rc));
    return rc;
}
