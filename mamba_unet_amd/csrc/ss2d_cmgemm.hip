// The projections between the depthwise convolution and the scan, and their input gradients, for gfx950 (bf16 MFMA):
//   x_proj   (mamba_sys.py:407)  x_dbl[b][k][c][l]  = sum_d W[k][c][d]  * xs[b][k][d][l]
//            its input gradient  dxs[b][k][d][l]    = sum_c W[k][c][d]  * d_x_dbl[b][k][c][l]        (W given transposed)
//   dt_proj  (mamba_sys.py:420)  delta[b][k][d][l]  = sum_r Wd[k][d][r] * dts[b][k][r][l]
//            its input gradient  d_dts[b][k][r][l]  = sum_d Wd[k][d][r] * d_delta[b][k][d][l]        (Wd given transposed)
// All four are   Y[b][g] (M x L) = W[g] (M x K) . X[b][g] (K x L)   with X and Y CHANNEL-major (a row = one channel, the L
// positions contiguous) -- the layout the scan wants on both sides.  For the MFMA the contraction index k must be the
// fast index of both operands inside a lane; W has it (rows of K), X has the positions there instead.  The library runs
// these as B*G small transposed GEMMs at 30-60 us each whatever the size (tools/proj_gemm_bench.py).  Here:
//   * a workgroup (4 waves) owns 96 rows of W and 128 positions of one (b, g); K is walked in chunks of 64 or 32;
//   * the X chunk is brought in as it lies in memory (rows of 256 bytes, coalesced 16-byte pieces) and read back with
//     ds_read_b64_tr_b16: per 16 lanes a 4 (k) x 16 (position) block, delivered position-per-lane with the four k in a
//     lane's register pair -- the B operand of v_mfma_f32_32x32x16_bf16 without a transposing pass;
//   * LDS rows are 320 bytes apart (4 consecutive k rows x 64 bytes of one 32-lane half fall on 64 different banks);
//   * the next chunk's pieces are in flight (registers) while the current one is multiplied; one barrier per chunk;
//   * the 96 x 128 result goes through LDS once more so that the stores are whole 256-byte rows.
#include <type_traits>
#include "ss2d_common.h"
#include "../../include/ss2d.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

struct CgParams {
    const bf16 *W, *X;
    bf16 *Y;
    long long swg, sxb, sxg, syb, syg;   // element strides: W per group; X, Y per batch entry and per group (rows are L apart)
    int B, G, M, K, ldw, L;              // W rows are ldw >= K apart; ldw % 8 == 0 and columns K .. ldw-1 hold zeros
    int mblocks, ntiles;
};

constexpr int BM = 96, BN = 128;
constexpr int XROW = 2 * BN + 64;                            // bytes between LDS rows of the X chunk
constexpr int YROW = 2 * BN + 16;                            // result image [96][128] bf16
constexpr int cg_stage_bytes(int kc) { return kc * XROW + BM * (2 * kc + 16); }

// V16: rows of X / Y are 16-byte aligned (L % 8 == 0): 16-byte pieces; otherwise 8-byte pieces (L % 4 == 0).
// KC: rows of K per chunk.  64 for a long contraction (x_proj forward: fewer barriers and waits per byte), 32 for a short
// one (the input gradients, dt_proj: half the LDS, four workgroups per CU instead of two -- these are bound by their
// stores).  Measured both ways at every stage (tools/chan_gemm_bench.py).
template <bool V16, int KC>
__global__ __launch_bounds__(256) void cm_gemm_kernel(const CgParams p) {
    constexpr int WROW = 2 * KC + 16, XBYTES = KC * XROW, STAGE = cg_stage_bytes(KC);
    constexpr int WPR = KC / 8, NWP = (BM * WPR + 255) / 256;    // 16-byte pieces per W row, per thread
    static_assert(BM * YROW <= 2 * STAGE, "the result image reuses the chunk buffers");
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, half = lane >> 5;
    int blk = xcd_remap(blockIdx.x, gridDim.x);   // neighbours in the tile order share cache lines / operands: one XCD's L2
    const int nt = blk % p.ntiles; blk /= p.ntiles;
    const int mb = blk % p.mblocks; blk /= p.mblocks;
    const int g = blk % p.G, b = blk / p.G;
    const int l0 = nt * BN, m0 = mb * BM, L = p.L;
    const bf16 *__restrict__ Xg = p.X + (size_t)b * p.sxb + (size_t)g * p.sxg;
    const bf16 *__restrict__ Wg = p.W + (size_t)g * p.swg + (size_t)m0 * p.ldw;
    bf16 *__restrict__ Yg = p.Y + (size_t)b * p.syb + (size_t)g * p.syg + (size_t)m0 * L;
    const int nchunks = (p.K + KC - 1) / KC;

    // this thread's pieces of a chunk (row / column fixed over the chunks)
    constexpr int XPB = V16 ? 16 : 8, XPR = 2 * BN / XPB, NXP = KC * XPR / 256, XE = XPB / 2;   // piece bytes, pieces per row, per thread
    typedef unsigned xpiece_t __attribute__((ext_vector_type(V16 ? 4 : 2)));
    // TWO chunks in flight in registers (set = stage = chunk parity): a workgroup of the long contractions (x_proj forward:
    // 12 / 24 chunks at the 14 x 14 / 7 x 7 stages) walked them one memory round trip at a time
    constexpr bool DEEP = KC == 64;        // the short contractions (KC = 32: 1-4 chunks) keep one chunk ahead and their occupancy
    xpiece_t xr[DEEP ? 2 : 1][NXP];
    u32x4 wr[DEEP ? 2 : 1][NWP];
    constexpr std::integral_constant<int, 0> S0{};
    constexpr std::integral_constant<int, 1> S1{};
    // Operands through buffer descriptors: X ends behind its K rows (a chunk row past K reads zeros by the range check), W behind
    // the block's rows; a column past L / a k past ldw is sent out of range by its offset.  Nothing touches a loaded value
    // before its ds_write -- the first version zeroed those pieces with an AND on the value, and in the two-deep form the
    // compiler gathered the ANDs of both register sets behind one s_waitcnt vmcnt(0) per trip (see ss2d_wgrad.hip).
    // (Host: K * L and 96 * ldw elements stay below 2^30.)
    const __amdgpu_buffer_rsrc_t rsX = make_rsrc(Xg, (unsigned)p.K * (unsigned)L * 2u);
    const int wrows = p.M - m0 < BM ? p.M - m0 : BM;
    const __amdgpu_buffer_rsrc_t rsW = make_rsrc(Wg, (unsigned)wrows * (unsigned)p.ldw * 2u);
    int xofs[NXP], wofs[NWP], wk[NWP];
#pragma unroll
    for (int j = 0; j < NXP; ++j) {
        const int id = j * 256 + tid, row = id / XPR, pc = id - row * XPR;
        const int col = l0 + pc * XE;
        xofs[j] = col < L ? (row * L + col) * 2 : SS2D_BUF_OOR;
    }
#pragma unroll
    for (int j = 0; j < NWP; ++j) {
        const int id = j * 256 + tid, row = id / WPR, pc = id - row * WPR;
        wk[j] = pc * 8;
        wofs[j] = row < wrows ? (row * p.ldw + pc * 8) * 2 : SS2D_BUF_OOR;
    }
    auto load_chunk = [&](auto set, int c) {                 // a chunk past K reads as zeros
        constexpr int S = DEEP ? decltype(set)::value : 0;
        const int sx = c * KC * L * 2, sw = c * KC * 2;
#pragma unroll
        for (int j = 0; j < NXP; ++j) {
            if constexpr (V16) xr[S][j] = __builtin_bit_cast(xpiece_t, __builtin_amdgcn_raw_buffer_load_b128(rsX, xofs[j], sx, 0));
            else xr[S][j] = __builtin_bit_cast(xpiece_t, __builtin_amdgcn_raw_buffer_load_b64(rsX, xofs[j], sx, 0));
        }
#pragma unroll
        for (int j = 0; j < NWP; ++j)
            wr[S][j] = __builtin_amdgcn_raw_buffer_load_b128(rsW, c * KC + wk[j] < p.ldw ? wofs[j] : SS2D_BUF_OOR, sw, 0);
    };
    auto put_chunk = [&](auto set, int stage) {
        constexpr int S = DEEP ? decltype(set)::value : 0;
        unsigned char *xs = lds + stage * STAGE, *ws = xs + XBYTES;
#pragma unroll
        for (int j = 0; j < NXP; ++j) {
            const int id = j * 256 + tid, row = id / XPR, pc = id - row * XPR;
            *reinterpret_cast<xpiece_t *>(xs + row * XROW + pc * XPB) = xr[S][j];
        }
#pragma unroll
        for (int j = 0; j < NWP; ++j) {
            const int id = j * 256 + tid, row = id / WPR, pc = id - row * WPR;
            if (BM * WPR % 256 == 0 || row < BM) *reinterpret_cast<u32x4 *>(ws + row * WROW + pc * 16) = wr[S][j];
        }
    };

    f16v acc[3];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;

    // transposed read of the B operand: 16-lane group gq covers positions 32*wave + 16*(gq & 1) .. +15 and the k rows
    // 8*(gq >> 1) .. +7 of a k-step in two blocks of 4; lane 4q + p of the group addresses row q, positions 4p .. 4p+3
    const int gq = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int xoff = (8 * (gq >> 1) + q) * XROW + (32 * wave + 16 * (gq & 1) + 4 * pp) * 2;
    const int woff = l32 * WROW + 16 * half;

    auto products = [&](int stage) {
        const unsigned char *xs = lds + stage * STAGE, *ws = xs + XBYTES;
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            typedef s16x4 __attribute__((address_space(3))) *lp;
            const s16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(xs + xoff + (16 * ks) * XROW));
            const s16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(xs + xoff + (16 * ks + 4) * XROW));
            const bf16x8 bf = __builtin_bit_cast(bf16x8, s16x8{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w});
#pragma unroll
            for (int mt = 0; mt < 3; ++mt) {
                const bf16x8 af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(ws + mt * 32 * WROW + woff + 32 * ks));
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc[mt], 0, 0, 0);
            }
        }
    };
    if constexpr (DEEP) {
        load_chunk(S0, 0);
        __builtin_amdgcn_sched_barrier(0);     // set 0's loads in front of set 1's: interleaved, the first put of every trip waits for both
        load_chunk(S1, 1);
        // two chunks per trip, nothing under a branch (an odd count ends with a chunk of zeros); a stage was last read two
        // chunks ago, behind the previous barrier
        for (int c = 0; c < nchunks; c += 2) {
            put_chunk(S0, 0);
            load_chunk(S0, c + 2);
            __syncthreads();
            products(0);
            put_chunk(S1, 1);
            load_chunk(S1, c + 3);
            __syncthreads();
            products(1);
        }
    } else {
        load_chunk(S0, 0);
        for (int c = 0; c < nchunks; ++c) {
            const int st = c & 1;
            put_chunk(S0, st);                                // stage st was last read two chunks ago, behind the previous barrier
            load_chunk(S0, c + 1);                            // in flight during the products (past K: zeros, never used)
            __syncthreads();
            products(st);
        }
    }
    __syncthreads();
    // accumulator register e of a 32x32 tile: row (= m) 8*(e/4) + 4*(lane/32) + e%4, column (= position) lane%32
#pragma unroll
    for (int mt = 0; mt < 3; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = 32 * mt + 8 * (e >> 2) + 4 * half + (e & 3);
            *reinterpret_cast<bf16 *>(lds + m * YROW + (32 * wave + l32) * 2) = (bf16)acc[mt][e];
        }
    __syncthreads();
    constexpr int NYP = BM * XPR / 256;
#pragma unroll
    for (int j = 0; j < NYP; ++j) {
        const int id = j * 256 + tid, row = id / XPR, pc = id - row * XPR;
        const int col = l0 + pc * XE;
        if (m0 + row < p.M && col < L)
            *reinterpret_cast<xpiece_t *>(Yg + (size_t)row * L + col) = *reinterpret_cast<const xpiece_t *>(lds + row * YROW + pc * XPB);
    }
}


// ---- fp32 operands: the same products on the bf16 MFMA, every fp32 value taken as the exact sum of three bf16 values
// (split3_bf16, ss2d_common.h), six MFMAs per 16-k product.
//   * X never visits LDS: with 4-byte elements a lane can read its own MFMA operand -- position l32 of the wave's 32, the
//     k rows 8h .. 8h+7 of a k-step -- as eight 4-byte reads that are 128 contiguous bytes per row across the lanes, and
//     splits it in registers;
//   * the W chunk [96 m][32 k] (shared by the four waves) is split on its way to LDS, into three images;
//   * the result leaves straight from the accumulators: 32 consecutive positions of one row per store instruction.
struct CgParamsF {
    const float *W, *X;
    float *Y;
    long long swg, sxb, sxg, syb, syg;
    int B, G, M, K, ldw, L;              // ldw % 4 == 0 and columns K .. ldw-1 hold zeros
    int mblocks, ntiles;
};
constexpr int FKC = 32, FWROW = 2 * FKC + 16;

// MT: 32-row tiles of W per workgroup (1 for the 6 ... 24 rows of d_dts: a 96-row block would be 3/4 padding there)
template <int MT>
__global__ __launch_bounds__(256) void cm_gemm_f32_kernel(const CgParamsF p) {
    constexpr int FBM = 32 * MT, FWIMG = FBM * FWROW, FWSTAGE = 3 * FWIMG;
    __shared__ __align__(16) unsigned char lds[2 * FWSTAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l32 = lane & 31, half = lane >> 5;
    int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int nt = blk % p.ntiles; blk /= p.ntiles;
    const int mb = blk % p.mblocks; blk /= p.mblocks;
    const int g = blk % p.G, b = blk / p.G;
    const int m0 = mb * FBM, L = p.L;
    const int pos = nt * BN + 32 * wave + l32;              // this lane's position
    const bool pos_ok = pos < L;
    const float *__restrict__ Xl = p.X + (size_t)b * p.sxb + (size_t)g * p.sxg + (pos_ok ? pos : 0);
    const float *__restrict__ Wg = p.W + (size_t)g * p.swg + (size_t)m0 * p.ldw;
    float *__restrict__ Yg = p.Y + (size_t)b * p.syb + (size_t)g * p.syg + (size_t)m0 * L;
    const int nchunks = (p.K + FKC - 1) / FKC;
    unsigned xr[2][8];                                       // [k-step][k row 8h + i]
    u32x4 wr[MT];                                            // the W chunk: 32 MT rows x 8 pieces of 16 bytes, MT per thread
    const int wrow = tid >> 3, wcol = (tid & 7) * 4;         // piece j: row wrow + 32 j, k = wcol .. wcol + 3
    auto load_chunk = [&](int c) {                           // a chunk past K reads as zeros
        // never a guarded load: the address is clamped to element 0 and the value masked
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = c * FKC + 16 * ks + 8 * half + i;
                const bool ok = pos_ok && k < p.K;
                const unsigned v = *reinterpret_cast<const unsigned *>(Xl + (ok ? (size_t)k * L : 0));
                xr[ks][i] = v & (ok ? ~0u : 0u);
            }
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int row = wrow + 32 * j, k = c * FKC + wcol;
            const bool ok = m0 + row < p.M && k < p.ldw;
            const u32x4 v = *reinterpret_cast<const u32x4 *>(Wg + (ok ? (size_t)row * p.ldw + k : 0));
            wr[j] = v & (ok ? ~0u : 0u);
        }
    };
    auto put_chunk = [&](int stage) {
        unsigned char *ws = lds + stage * FWSTAGE;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const Split4 sp = split3_bf16(wr[j]);
            unsigned char *d = ws + (wrow + 32 * j) * FWROW + wcol * 2;
            *reinterpret_cast<u32x2 *>(d) = sp.h1;
            *reinterpret_cast<u32x2 *>(d + FWIMG) = sp.h2;
            *reinterpret_cast<u32x2 *>(d + 2 * FWIMG) = sp.h3;
        }
    };
    f16v acc[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.0f;
    const int woff = l32 * FWROW + 16 * half;
    load_chunk(0);
    for (int c = 0; c < nchunks; ++c) {
        const int st = c & 1;
        put_chunk(st);                                        // stage st was last read two chunks ago, behind the previous barrier
        u32x4 xf[2][3];                                       // [k-step][rank]
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const Split4 lo = split3_bf16(u32x4{xr[ks][0], xr[ks][1], xr[ks][2], xr[ks][3]});
            const Split4 hi = split3_bf16(u32x4{xr[ks][4], xr[ks][5], xr[ks][6], xr[ks][7]});
            xf[ks][0] = u32x4{lo.h1.x, lo.h1.y, hi.h1.x, hi.h1.y};
            xf[ks][1] = u32x4{lo.h2.x, lo.h2.y, hi.h2.x, hi.h2.y};
            xf[ks][2] = u32x4{lo.h3.x, lo.h3.y, hi.h3.x, hi.h3.y};
        }
        load_chunk(c + 1);                                    // in flight during the products (past K: zeros, never used)
        __syncthreads();
        const unsigned char *ws = lds + st * FWSTAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 x1 = __builtin_bit_cast(bf16x8, xf[ks][0]), x2 = __builtin_bit_cast(bf16x8, xf[ks][1]), x3 = __builtin_bit_cast(bf16x8, xf[ks][2]);
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const unsigned char *wp = ws + mt * 32 * FWROW + woff + 32 * ks;
                const bf16x8 w1 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(wp));
                const bf16x8 w2 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(wp + FWIMG));
                const bf16x8 w3 = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4 *>(wp + 2 * FWIMG));
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x2, acc[mt], 0, 0, 0);      // the small terms first
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w3, x1, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x3, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w2, x1, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x2, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, x1, acc[mt], 0, 0, 0);
            }
        }
    }
    // accumulator register e of a 32x32 tile: row (= m) 8*(e/4) + 4*(lane/32) + e%4, column (= position) lane%32
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = 32 * mt + 8 * (e >> 2) + 4 * half + (e & 3);
            if (pos_ok && m0 + m < p.M) Yg[(size_t)m * L + pos] = acc[mt][e];
        }
}

}  // namespace

extern "C" int ss2d_chan_gemm(int dtype, const void *W, const void *X, void *Y, int B, int G, int M, int K, int ldw, int L,
                              long long swg, long long sxb, long long sxg, long long syb, long long syg, void *stream) {
    if (dtype != SS2D_DTYPE_BF16 && dtype != SS2D_DTYPE_F32) { ss2d_set_error("chan_gemm: unknown dtype %d", dtype); return -1; }
    const bool f32 = dtype == SS2D_DTYPE_F32;
    if (!W || !X || !Y) { ss2d_set_error("chan_gemm: null required pointer"); return -2; }
    if (B <= 0 || G <= 0 || M <= 0 || K <= 0 || L <= 0 || ldw < K) { ss2d_set_error("chan_gemm: bad size"); return -3; }
    if (L % 4 || ((sxb | sxg | syb | syg) % 4) || (((uintptr_t)X | (uintptr_t)Y) & (f32 ? 15 : 7))) {
        ss2d_set_error("chan_gemm: L, the block strides of X / Y and their base addresses must be multiples of 4 elements"); return -4; }
    const int wq = f32 ? 4 : 8;
    if (ldw % wq || swg % wq || ((uintptr_t)W & 15)) { ss2d_set_error("chan_gemm: W rows must be 16-byte aligned (ldw, group stride multiples of %d)", wq); return -4; }
    const int fmt = M <= 32 ? 1 : M <= 64 ? 2 : 3;                 // fp32: 32-row tiles per workgroup
    const int bm = f32 ? 32 * fmt : BM;
    const int mblocks = (M + bm - 1) / bm, ntiles = (L + BN - 1) / BN;
    const long long nwg = (long long)B * G * mblocks * ntiles;
    if (nwg > 0x7fffffffLL) { ss2d_set_error("chan_gemm: grid too large"); return -6; }
    if (!f32 && ((long long)(K + 256) * L >= (1ll << 30) || (long long)BM * ldw >= (1ll << 30))) {   // 32-bit byte offsets inside a (b, g) block
        ss2d_set_error("chan_gemm: a (batch, group) block of X of 2 GB or more"); return -6; }
    if (f32) {
        CgParamsF p;
        p.W = static_cast<const float *>(W); p.X = static_cast<const float *>(X); p.Y = static_cast<float *>(Y);
        p.swg = swg; p.sxb = sxb; p.sxg = sxg; p.syb = syb; p.syg = syg;
        p.B = B; p.G = G; p.M = M; p.K = K; p.ldw = ldw; p.L = L; p.mblocks = mblocks; p.ntiles = ntiles;
        if (fmt == 1) ss2d_launch(cm_gemm_f32_kernel<1>, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, p);
        else if (fmt == 2) ss2d_launch(cm_gemm_f32_kernel<2>, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, p);
        else ss2d_launch(cm_gemm_f32_kernel<3>, dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, p);
        const int rc = ss2d_take_launch_rc();
        if (rc) ss2d_set_error("chan_gemm: launch failed: %s", hipGetErrorString((hipError_t)rc));
        return rc;
    }
    CgParams p;
    p.W = static_cast<const bf16 *>(W); p.X = static_cast<const bf16 *>(X); p.Y = static_cast<bf16 *>(Y);
    p.swg = swg; p.sxb = sxb; p.sxg = sxg; p.syb = syb; p.syg = syg;
    p.B = B; p.G = G; p.M = M; p.K = K; p.ldw = ldw; p.L = L;
    p.mblocks = mblocks; p.ntiles = ntiles;
    const bool v16 = L % 8 == 0 && ((sxb | sxg | syb | syg) % 8) == 0 && (((uintptr_t)X | (uintptr_t)Y) & 15) == 0;
    const bool k64 = K > 128;
    const void *k = v16 ? (k64 ? (const void *)cm_gemm_kernel<true, 64> : (const void *)cm_gemm_kernel<true, 32>)
                        : (k64 ? (const void *)cm_gemm_kernel<false, 64> : (const void *)cm_gemm_kernel<false, 32>);
    static bool attr[4][SS2D_MAX_DEVICES] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SS2D_MAX_DEVICES) dev = 0;
    const int lds = 2 * cg_stage_bytes(k64 ? 64 : 32), ai = 2 * v16 + k64;
    if (!attr[ai][dev]) {
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) {
            ss2d_set_error("chan_gemm: cannot raise the dynamic LDS limit to %d", lds); return -7; }
        attr[ai][dev] = true;
    }
    void *args[] = {&p};
    const hipError_t e = hipLaunchKernel(k, dim3((unsigned)nwg), dim3(256), args, lds, (hipStream_t)stream);
    if (e != hipSuccess) { ss2d_set_error("chan_gemm: launch failed: %s", hipGetErrorString(e)); return (int)e; }
    return 0;
}
