// Selective scan (S6 recurrence) forward for gfx950 -- the `selective_scan_fn` half of the SS2D hot path
// (/root/reference/code/networks/mamba_sys.py:436-442; semantics :62-144).
//
// Decomposition ("X16"): a wavefront holds 4 rows of 16 lanes; a row is one (batch, channel) sequence,
// each lane owns T=8 consecutive positions, so a wave advances 4 channels by 128 positions per chunk.
//   * thread-local recurrence over the lane's 8 items, states handled two at a time as float2
//     (v_pk_mul_f32 / v_pk_fma_f32);
//   * 4-step Hillis-Steele scan of the (decay, state) pairs across the 16 lanes of a row with DPP
//     row_shr modifiers folded into v_fmac/v_mul (no LDS, no cross-row traffic);
//   * second local pass from the lane's true incoming state, fused with y += C*h;
//   * the state leaving a chunk is moved to lane 0 of the row (row_shl:15, zero elsewhere) for the next chunk.
// B and C of the (batch, group) are shared by every channel of the group: the workgroup stages the
// chunk's B/C tile once in LDS, interleaving state pairs so one ds_read_b128 feeds two packed FMAs.
// The product of exp(delta*A) over a lane's items is exp(A * sum(delta)), so the lane aggregate costs
// one exp instead of seven multiplies.
#include <cstdlib>
#include "ss2d_common.h"
#include "../../include/ss2d.h"

namespace {

#ifndef SS2D_FWD_PRIO
#define SS2D_FWD_PRIO 1     // 0: off, 1: priority falls with progress (-2 % measured), 2: by phase (no gain)
#endif
// tuning builds: -DSS2D_FWD_WPE=4 asks the register allocator for 4 waves per SIMD (128 registers)
#ifdef SS2D_FWD_WPE
#define SS2D_FWD_WAVES __attribute__((amdgpu_waves_per_eu(SS2D_FWD_WPE, SS2D_FWD_WPE)))
#else
#define SS2D_FWD_WAVES
#endif
constexpr int T = 8;          // items per lane
constexpr int CHUNK = 128;    // positions per row per chunk = 16 lanes * T
constexpr int NP = 8;         // state pairs (dstate <= 16; odd/short dstate padded with inert states)
static_assert(CHUNK == SS2D_SCAN_CHUNK, "checkpoint spacing is part of the ABI");

struct ScanFwdParams {
    const void *u, *delta, *B, *C, *z;   // IO type (fp32 or bf16)
    const float *A, *D, *bias;
    void *out;                            // IO type
    float *last, *ckpt;
    int batch, dim, L, N, G, dpg, nchunks, softplus, wgs_per_group, nwg;
    int Gu;              // u has Gu groups per batch; scan group g reads u group g / (G / Gu)
    unsigned rev_mask;   // bit g: group g runs over its rows in reverse memory order
    long long bc_stride; // elements between consecutive (batch, group) blocks of B and of C (N*L when dense)
    const float *dtw;    // DT kernels: dt_proj weight (G * dpg, R) in scan-group order; `delta` then points at the dts rows
    int R;               // (R, L) of every (batch, group) block (same block stride as B / C), and delta is computed here
};
constexpr int MAXR = SS2D_MAX_DT_RANK;

// Inclusive scan of (a, h) over the 16 lanes of each DPP row, two independent states interleaved so
// that every DPP read is >= 2 instructions behind the write of its source (hardware hazard).
__device__ __forceinline__ void row_scan_pair(float &hx, float &hy, float &ax, float &ay) {
#define SS2D_STEP(n)                                                                   \
    "v_fmac_f32_dpp %0, %0, %2 row_shr:" #n " row_mask:0xf bank_mask:0xf\n\t"         \
    "v_fmac_f32_dpp %1, %1, %3 row_shr:" #n " row_mask:0xf bank_mask:0xf\n\t"         \
    "v_mul_f32_dpp %2, %2, %2 row_shr:" #n " row_mask:0xf bank_mask:0xf\n\t"          \
    "v_mul_f32_dpp %3, %3, %3 row_shr:" #n " row_mask:0xf bank_mask:0xf\n\t"
    asm volatile("s_nop 1\n\t" SS2D_STEP(1) SS2D_STEP(2) SS2D_STEP(4)
                 "v_fmac_f32_dpp %0, %0, %2 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %1, %1, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 : "+v"(hx), "+v"(hy), "+v"(ax), "+v"(ay));
#undef SS2D_STEP
}

// FULL: dstate == 16 known at compile time (the only value Mamba-UNet uses): no per-pair bounds checks
// DIRS: direction-aware build (rev_mask / u_groups honoured); false compiles the mirrored paths away
#ifdef SS2D_FWD_PROBE
// timing probes (debug builds only: make variant SRC=ss2d_scan_fwd DEFS=-DSS2D_FWD_PROBE, tools/probe_fwd.py)
__device__ unsigned long long *g_fwd_probe;
#define FWD_NPROBE 8
#define FWD_PROBE(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                          pacc[i] += (unsigned)(t_ - tprev); tprev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define FWD_PROBE(i) do { } while (0)
#endif

// DT: delta is not read but computed in the kernel from the rank-R rows of the x_proj output (mamba_sys.py:417-420:
//     delta[d, l] = sum_r dt_w[d, r] * dts[r, l]): the chunk's (R, 128) tile is staged in LDS beside B / C, the 16
//     weight rows of the workgroup's channels once; 8R FMAs per lane and chunk instead of a (B, K*D, L) tensor.
// PACK: sequences of at most 64 positions (one chunk: the 7 x 7 and 8 x 8 stages), TWO channels per DPP row -- lanes 0-7 of
//     a row walk one channel, lanes 8-15 the next (a wave advances 8 channels): the row scan is cut between lanes 7 and 8
//     (lane 8's aggregate decay is zeroed going in, its incoming state coming out).  With one channel per row 9 of 16 lanes
//     idle at L = 56 and the kernel costs what it costs at L = 128.
template <int NW, bool HAS_Z, bool VEC4, bool FULL, typename IO, bool DIRS, bool DT = false, bool PACK = false>
__global__ __launch_bounds__(NW * 64) SS2D_FWD_WAVES void scan_fwd_kernel(const ScanFwdParams p) {
    static_assert(!PACK || (VEC4 && FULL && !HAS_Z && !DT), "the packed form exists for the SS2D call pattern only");
    constexpr int CPR = PACK ? 2 : 1;      // channels per DPP row
    // [buffer][B|C][(pair*4 + quarter)*16 + seg] ; one f4 = (n,t),(n+1,t),(n,t+1),(n+1,t+1)
    __shared__ f4 tile[2][2][NP * 4 * 16];
    __shared__ __align__(16) float dtile[DT ? 2 : 1][DT ? MAXR : 1][DT ? CHUNK : 4];
    __shared__ float wtile[DT ? NW * 4 : 1][DT ? MAXR : 1];
    __shared__ f2 A2s[NW * 4 * CPR][NP];
#ifdef SS2D_FWD_PROBE
    unsigned pacc[FWD_NPROBE] = {};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tprev, rstart = __builtin_amdgcn_s_memrealtime();
#endif

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, seg = lane & 15;
    // PACK: `row` counts channel slots of the wave (0-7), `sg` the lane's segment inside its channel (0-7)
    const int row = PACK ? (lane >> 4) * 2 + (seg >> 3) : lane >> 4, sg = PACK ? seg & 7 : seg;
    int blk = xcd_remap(blockIdx.x, p.nwg);
    const int cb = blk % p.wgs_per_group;
    blk /= p.wgs_per_group;
    const int g = blk % p.G, b = blk / p.G;
    const int c_in_g = cb * (NW * 4 * CPR) + wave * (4 * CPR) + row;
    const bool row_ok = c_in_g < p.dpg;
    const int c = g * p.dpg + (row_ok ? c_in_g : 0);
    const int L = p.L, N = FULL ? 16 : p.N;

    const size_t rowoff = ((size_t)b * p.dim + c) * (size_t)L;
    const bool rev = DIRS && ((p.rev_mask >> g) & 1u);
    const size_t urowoff = DIRS ? (((size_t)b * p.Gu + g / (p.G / p.Gu)) * p.dpg + (row_ok ? c_in_g : 0)) * (size_t)L : rowoff;
    const IO *__restrict__ urow = static_cast<const IO *>(p.u) + urowoff;
    const IO *__restrict__ drow = static_cast<const IO *>(p.delta) + rowoff;
    const IO *__restrict__ zrow = HAS_Z ? static_cast<const IO *>(p.z) + rowoff : nullptr;
    IO *__restrict__ orow = static_cast<IO *>(p.out) + rowoff;
    const size_t bcs = DIRS ? (size_t)p.bc_stride : N * (size_t)L;   // the plain path keeps its dense addressing
    const IO *__restrict__ Bg = static_cast<const IO *>(p.B) + (size_t)(b * p.G + g) * bcs;
    const IO *__restrict__ Cg = static_cast<const IO *>(p.C) + (size_t)(b * p.G + g) * bcs;

    // FAST builds store through buffer resources: rows and positions that do not exist are dropped by the range check
    // instead of being branched around.  (A store under a branch makes the compiler's count of outstanding memory
    // operations path dependent; it then waits for vmcnt(0) before the prefetched tile is used -- that is, for the
    // stores issued a moment ago -- which the probes showed as 11 % of a wave's time.)
    constexpr bool FAST = VEC4 && FULL && !HAS_Z;
    const int c0_in_g = cb * (NW * 4 * CPR) + __builtin_amdgcn_readfirstlane(wave) * (4 * CPR);
    const int nrow = __builtin_amdgcn_readfirstlane(p.dpg - c0_in_g < 0 ? 0 : (p.dpg - c0_in_g > 4 * CPR ? 4 * CPR : p.dpg - c0_in_g));
    const size_t wrow0 = (size_t)b * p.dim + (size_t)g * p.dpg + c0_in_g;
    const __amdgpu_buffer_rsrc_t out_rs = make_rsrc(static_cast<IO *>(p.out) + wrow0 * L, FAST ? (unsigned)(nrow * L * (int)sizeof(IO)) : 0u);
    const __amdgpu_buffer_rsrc_t ck_rs = make_rsrc(p.ckpt ? p.ckpt + wrow0 * p.nchunks * 16 : nullptr,
                                                    FAST && p.ckpt ? (unsigned)(nrow * p.nchunks * 64) : 0u);
    constexpr unsigned OOB = 0xffffff00u;

    // A * log2(e) of the row's 16 states lives in LDS (one ds_read_b64 per pair and chunk): 16 registers that pay for the
    // B/C tile prefetch below without giving up the third wave per SIMD
    f2 carry[NP];
    if (sg < NP) {
        const float a0 = (2 * sg < N) ? p.A[(size_t)c * N + 2 * sg] : 0.0f;
        const float a1 = (2 * sg + 1 < N) ? p.A[(size_t)c * N + 2 * sg + 1] : 0.0f;
        A2s[wave * (4 * CPR) + row][sg] = f2{a0, a1} * SS2D_LOG2E;           // visible behind the first barrier
    }
#pragma unroll
    for (int np = 0; np < NP; ++np) carry[np] = f2{0.0f, 0.0f};
    const float Dc = p.D ? p.D[c] : 0.0f;
    const float bias = p.bias ? p.bias[c] : 0.0f;
    const IO *__restrict__ dtsg = static_cast<const IO *>(p.delta) + (size_t)(b * p.G + g) * bcs;   // DT: the dts rows of this (b, g)
    if (DT && seg < p.R) wtile[wave * 4 + row][seg] = p.dtw[(size_t)c * p.R + seg];                // visible behind the first barrier

    // software prefetch: the next chunk's u / delta are requested as soon as the current chunk's are consumed, into the
    // same registers.  VEC4: the address is clamped into the row instead of the load being guarded (a guarded load is
    // sunk into its branch with a full wait); what a clamped load returns past L is real data of the same row and is
    // switched off by dt = 0 below.
    float uv[T], dn[T];
    auto fetch_row = [&](const IO *__restrict__ rowp, int q0, float (&v)[T]) {
        if (VEC4) {
            const int pa = q0 < L - 4 ? q0 : L - 4, pb = q0 + 4 < L - 4 ? q0 + 4 : L - 4;
            f4 a = ld4(rowp + (unsigned)(rev ? L - 4 - pa : pa)), b = ld4(rowp + (unsigned)(rev ? L - 4 - pb : pb));
            if (rev) { a = f4{a.w, a.z, a.y, a.x}; b = f4{b.w, b.z, b.y, b.x}; }
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
            load8d<VEC4, DIRS>(rowp, q0, L, row_ok, rev, v);
        }
    };
    fetch_row(urow, sg * T, uv);
    if (!DT) fetch_row(drow, sg * T, dn);

    // B/C tile item `it` = (state pair, 4 positions): two rows of each matrix
    auto tile_fetch = [&](int base_, int it, f4 (&lo)[2], f4 (&hi)[2]) {
        const int np = it >> 5, q4 = it & 31;
        const int pos = base_ + q4 * 4;
        const int n0 = 2 * np, n1 = 2 * np + 1;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const IO *__restrict__ M = m ? Cg : Bg;
            lo[m] = f4{0, 0, 0, 0};
            hi[m] = f4{0, 0, 0, 0};
            if (VEC4 && FULL) {   // clamped, unguarded (positions past L meet du = 0 and are never stored)
                const int pc = pos < L - 4 ? pos : L - 4;
                const unsigned off = (unsigned)(rev ? L - 4 - pc : pc);
                lo[m] = ld4(M + (size_t)n0 * L + off);
                hi[m] = ld4(M + (size_t)n1 * L + off);
                if (rev) { lo[m] = f4{lo[m].w, lo[m].z, lo[m].y, lo[m].x}; hi[m] = f4{hi[m].w, hi[m].z, hi[m].y, hi[m].x}; }
            } else if (DIRS) {
                if (n0 < N) lo[m] = load4m<VEC4>(M + (size_t)n0 * L, pos, L, rev);
                if (n1 < N) hi[m] = load4m<VEC4>(M + (size_t)n1 * L, pos, L, rev);
            } else if (VEC4) {
                if (pos < L) {
                    if (n0 < N) lo[m] = ld4(M + (size_t)n0 * L + pos);
                    if (n1 < N) hi[m] = ld4(M + (size_t)n1 * L + pos);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (pos + e < L) {
                        if (n0 < N) lo[m][e] = ld1(M + (size_t)n0 * L + pos + e);
                        if (n1 < N) hi[m][e] = ld1(M + (size_t)n1 * L + pos + e);
                    }
                }
            }
        }
    };
    auto tile_put = [&](int buf_, int it, const f4 (&lo)[2], const f4 (&hi)[2]) {
        // slot of (pair, quarter j, segment): segments of quarters 2, 3 are stored 4 places on (sg ^ 4).  ds_write_b128
        // serves 8 consecutive lanes per pass over 32 banks (8 slots of 16 bytes): lanes 2k and 2k+1 -- same segment,
        // quarters 0 and 2 -- would meet in one slot (10 % of the LDS cycles); with the swizzle the 8 lanes take the slots
        // 0 4 1 5 2 6 3 7.  The readers (16 lanes = the 16 slots of a 256-byte line, permuted) stay conflict-free.
        const int np = it >> 5, q4 = it & 31, j0 = (q4 & 1) * 2, sg = (q4 >> 1) ^ (j0 * 2);
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            tile[buf_][m][(np * 4 + j0) * 16 + sg] = f4{lo[m].x, hi[m].x, lo[m].y, hi[m].y};
            tile[buf_][m][(np * 4 + j0 + 1) * 16 + sg] = f4{lo[m].z, hi[m].z, lo[m].w, hi[m].w};
        }
    };
    // NW == 4: a thread stages exactly one tile item per chunk and requests it a chunk ahead (the probes put 18 % of a
    // wave's time into this round trip when it is taken at the top of the chunk)
    constexpr bool TPRE = NW == 4 && VEC4 && FULL;
    f4 tlo[2], thi[2];
    if (TPRE) tile_fetch(0, tid, tlo, thi);

    // FAST: a chunk's outputs are stored by the NEXT iteration, just before it requests its prefetches.  Stores issued at
    // the end of an iteration would be the youngest memory operations when the next one waits for its tile and inputs,
    // and whenever the compiler's vmcnt bookkeeping is conservative (it merges the loop entry with the back edge) that
    // wait becomes a wait for the stores' acknowledgement.
    f4 ya = f4{0, 0, 0, 0}, yb = f4{0, 0, 0, 0};
    unsigned oa = OOB, ob = OOB;
    auto put_y = [&]() {
        if (sizeof(IO) == 4) {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, ya), out_rs, oa, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, yb), out_rs, ob, 0, 0);
        } else {
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, __builtin_convertvector(ya, bf16x4)), out_rs, oa, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, __builtin_convertvector(yb, bf16x4)), out_rs, ob, 0, 0);
        }
    };

    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        const int buf = chunk & 1;
        const int base = chunk * CHUNK;
#if SS2D_FWD_PRIO == 2
        __builtin_amdgcn_s_setprio(3);      // experiment: the short memory-issuing part of a chunk goes first, the long arithmetic part yields
#elif SS2D_FWD_PRIO == 1
        {   // a wave's issue priority falls as it advances: the SIMD arbitrates oldest-first, which lets the waves of a
            // round leave one after the other (107 / 145 / 190 us) and the SIMD run half empty in between; this way
            // they advance together and newly arrived waves catch up.  301 -> 295 us.
            const int qd = (4 * chunk) / p.nchunks;
            if (qd == 0) __builtin_amdgcn_s_setprio(3); else if (qd == 1) __builtin_amdgcn_s_setprio(2); else if (qd == 2) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
        }
#endif
        FWD_PROBE(0);
        // ---- stage the chunk's B/C tile (shared by all waves of the workgroup)
        if (TPRE) {
            tile_put(buf, tid, tlo, thi);
        } else {
            for (int it = tid; it < NP * 32; it += NW * 64) {
                f4 lo[2], hi[2];
                tile_fetch(base, it, lo, hi);
                tile_put(buf, it, lo, hi);
            }
        }
        FWD_PROBE(1);
        if (DT) {   // the (R, 128) tile of dts rows
            for (int it = tid; it < p.R * 32; it += NW * 64) {
                const int r = it >> 5, q4 = it & 31;
                *reinterpret_cast<f4 *>(&dtile[buf][r][q4 * 4]) = load4m<VEC4>(dtsg + (size_t)r * L, base + q4 * 4, L, rev);
            }
        }
        // ---- this lane's 8 positions of u / delta
        const int p0 = base + sg * T;
        float dt[T], du[T];
        float S = 0.0f;
        f2 y2[T];
        // valid positions of this lane in this chunk (0 for a padding row): positions past L act as the identity (a = 1, b = 0)
        const int nv = row_ok ? L - p0 : 0;
        auto finish_dt = [&]() {
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float x = (DT ? dt[t] : dn[t]) + bias, sg_unused;
                if (p.softplus) x = softplus_sig<false>(x, sg_unused);
                x = t < nv ? x : 0.0f;
                dt[t] = x;
                du[t] = x * uv[t];
                S += x;
            }
            // D*u enters through both halves of the packed accumulator (exact halving; saves zeroing the second half)
#pragma unroll
            for (int t = 0; t < T; ++t) y2[t] = splat2(uv[t]) * f2{0.5f * Dc, 0.5f * Dc};
            if (FAST) put_y();
            // u / delta of the next chunk, into the registers just consumed (unconditional: behind the last chunk the
            // clamped address re-reads the row's tail, and a conditional load would bring back a register copy per value)
            fetch_row(urow, p0 + CHUNK, uv);
            if (!DT) fetch_row(drow, p0 + CHUNK, dn);
        };
        if (DT) {
#pragma unroll
            for (int t = 0; t < T; ++t) dt[t] = 0.0f;
        }
        if (!DT) finish_dt();
        FWD_PROBE(2);

        __syncthreads();  // tile[buf] complete; the other buffer is free for the next chunk's staging
        FWD_PROBE(3);
        if (TPRE) tile_fetch(base + CHUNK, tid, tlo, thi);      // unconditional, clamped (see fetch_row)

        if (DT) {   // delta = dt_w[c, :] . dts[:, l] for this lane's 8 positions
            const float *__restrict__ wr = wtile[wave * 4 + row];
#pragma unroll 2
            for (int r = 0; r < p.R; ++r) {
                const float w = wr[r];
                const f4 a = *reinterpret_cast<const f4 *>(&dtile[buf][r][seg * T]), b4 = *reinterpret_cast<const f4 *>(&dtile[buf][r][seg * T + 4]);
                dt[0] = fmaf(w, a.x, dt[0]); dt[1] = fmaf(w, a.y, dt[1]); dt[2] = fmaf(w, a.z, dt[2]); dt[3] = fmaf(w, a.w, dt[3]);
                dt[4] = fmaf(w, b4.x, dt[4]); dt[5] = fmaf(w, b4.y, dt[5]); dt[6] = fmaf(w, b4.z, dt[6]); dt[7] = fmaf(w, b4.w, dt[7]);
            }
            finish_dt();
        }

        if (FAST) {   // state entering this chunk, for the backward: lane 0 of every row, 64 bytes
            const unsigned co = sg == 0 ? (unsigned)((row * p.nchunks + chunk) * 64) : OOB;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f4{carry[2 * q].x, carry[2 * q].y, carry[2 * q + 1].x, carry[2 * q + 1].y}),
                                                       ck_rs, co + 16 * q, 0, 0);
        } else if (p.ckpt != nullptr && seg == 0 && row_ok) {
            float *ck = p.ckpt + (((size_t)b * p.dim + c) * p.nchunks + chunk) * N;
            if (FULL) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<f4 *>(ck + 4 * q) = f4{carry[2 * q].x, carry[2 * q].y, carry[2 * q + 1].x, carry[2 * q + 1].y};
            } else {
#pragma unroll
                for (int np = 0; np < NP; ++np) {
                    if (2 * np < N) ck[2 * np] = carry[np].x;
                    if (2 * np + 1 < N) ck[2 * np + 1] = carry[np].y;
                }
            }
        }
        FWD_PROBE(4);
#if SS2D_FWD_PRIO == 2
        __builtin_amdgcn_s_setprio(0);
#endif
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            if (FULL || 2 * np < N) {  // wave-uniform
                f2 a[T], bb[T];
                const f2 A2np = A2s[wave * (4 * CPR) + row][np];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f4 Bq = tile[buf][0][(np * 4 + j) * 16 + (sg ^ ((j >> 1) * 4))];
                    bb[2 * j] = splat2(du[2 * j]) * f2{Bq.x, Bq.y};
                    bb[2 * j + 1] = splat2(du[2 * j + 1]) * f2{Bq.z, Bq.w};
                }
#pragma unroll
                for (int t = 0; t < T; ++t) a[t] = exp2_2(splat2(dt[t]) * A2np);
                // pass 1: lane-local recurrence (lane 0 of the row starts from the carried state)
                f2 h = carry[np];                                  // zero outside lane 0 of the row (see below)
#pragma unroll
                for (int t = 0; t < T; ++t) h = fma2(a[t], h, bb[t]);
                f2 at = exp2_2(splat2(S) * A2np);
                float hx = h.x, hy = h.y, ax = at.x, ay = at.y;
                if (PACK) { ax = seg == 8 ? 0.0f : ax; ay = seg == 8 ? 0.0f : ay; }   // nothing crosses from lane 7 into lane 8
                row_scan_pair(hx, hy, ax, ay);
                // state entering this lane = inclusive result of the lane to the left (carry for lane 0)
                h.x = dpp_mov<DPP_ROW_SHR(1), 0xf, false>(carry[np].x, hx);
                h.y = dpp_mov<DPP_ROW_SHR(1), 0xf, false>(carry[np].y, hy);
                if (PACK) { h.x = seg == 8 ? 0.0f : h.x; h.y = seg == 8 ? 0.0f : h.y; }   // the second channel starts from state 0 (one chunk)
                // pass 2: true states, fused with the C contraction
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f4 Cq = tile[buf][1][(np * 4 + j) * 16 + (sg ^ ((j >> 1) * 4))];
                    h = fma2(a[2 * j], h, bb[2 * j]);
                    y2[2 * j] = fma2(f2{Cq.x, Cq.y}, h, y2[2 * j]);
                    h = fma2(a[2 * j + 1], h, bb[2 * j + 1]);
                    y2[2 * j + 1] = fma2(f2{Cq.z, Cq.w}, h, y2[2 * j + 1]);
                }
                // lane 0 takes the state leaving lane 15; row_shl:15 finds no source for the other lanes, which bound_ctrl
                // zeroes: the carry is ready to start pass 1 without a select
                carry[np].x = dpp_mov<DPP_ROW_SHL(15), 0xf, true>(h.x, h.x);
                carry[np].y = dpp_mov<DPP_ROW_SHL(15), 0xf, true>(h.y, h.y);
                // Pin the partial outputs here: otherwise LLVM sinks the whole y accumulation into the
                // (conditional) store block at the end of the chunk and keeps every C and h value of all 8
                // state pairs alive (256 VGPRs + AGPR spills).  Also keeps the pairs from being interleaved.
#pragma unroll
                for (int t = 0; t < T; ++t) asm volatile("" : "+v"(y2[t]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        FWD_PROBE(5);
        float yv[T];
#pragma unroll
        for (int t = 0; t < T; ++t)   // (plain C is vectorised into v_pk_add_f32 behind three register moves per pair of sums)
            asm("v_add_f32 %0, %1, %2" : "=v"(yv[t]) : "v"(y2[t].x), "v"(y2[t].y));
        if (HAS_Z) {
            float zv[T];
            load8d<VEC4, DIRS>(zrow, p0, L, row_ok, rev, zv);
#pragma unroll
            for (int t = 0; t < T; ++t) yv[t] *= zv[t] * sigmoidf_fast(zv[t]);
        }
        if (FAST) {   // handed to the next iteration, which stores it ahead of its prefetches (see put_y)
            ya = rev ? f4{yv[3], yv[2], yv[1], yv[0]} : f4{yv[0], yv[1], yv[2], yv[3]};
            yb = rev ? f4{yv[7], yv[6], yv[5], yv[4]} : f4{yv[4], yv[5], yv[6], yv[7]};
            oa = p0 < L ? (unsigned)(row * L + (rev ? L - 4 - p0 : p0)) * (unsigned)sizeof(IO) : OOB;
            ob = p0 + 4 < L ? (unsigned)(row * L + (rev ? L - 8 - p0 : p0 + 4)) * (unsigned)sizeof(IO) : OOB;
        } else {
            store8d<VEC4, DIRS>(orow, p0, L, row_ok, rev, yv);
        }
        FWD_PROBE(6);
    }
    if (FAST) put_y();
#ifdef SS2D_FWD_PROBE
    if (g_fwd_probe && lane == 0) {
        unsigned long long *q = g_fwd_probe + ((size_t)blockIdx.x * NW + wave) * (FWD_NPROBE + 1);
        for (int i = 0; i < FWD_NPROBE; ++i) q[i] = pacc[i];
        q[FWD_NPROBE - 2] = rstart;                      // slots 6 / 7: start / end on the 100 MHz device-wide clock
        q[FWD_NPROBE - 1] = __builtin_amdgcn_s_memrealtime();
        q[FWD_NPROBE] = __builtin_amdgcn_s_memtime() - tstart;
    }
#endif
    if (p.last != nullptr && seg == 0 && row_ok) {
#pragma unroll
        for (int np = 0; np < NP; ++np) {
            if (2 * np < N) p.last[((size_t)b * p.dim + c) * N + 2 * np] = carry[np].x;
            if (2 * np + 1 < N) p.last[((size_t)b * p.dim + c) * N + 2 * np + 1] = carry[np].y;
        }
    }
}

template <int NW, typename IO, bool DIRS>
int launch_fwd(const ScanFwdParams &p, bool has_z, bool vec4, hipStream_t s) {
    dim3 grid(p.nwg), block(NW * 64);
    const bool full = p.N == 16;
    if (has_z) {
        if (vec4) ss2d_launch((scan_fwd_kernel<NW, true, true, false, IO, DIRS>), grid, block, 0, s, p);
        else ss2d_launch((scan_fwd_kernel<NW, true, false, false, IO, DIRS>), grid, block, 0, s, p);
    } else if (vec4 && full) {
        ss2d_launch((scan_fwd_kernel<NW, false, true, true, IO, DIRS>), grid, block, 0, s, p);  // the SS2D call pattern
    } else {
        if (vec4) ss2d_launch((scan_fwd_kernel<NW, false, true, false, IO, DIRS>), grid, block, 0, s, p);
        else ss2d_launch((scan_fwd_kernel<NW, false, false, false, IO, DIRS>), grid, block, 0, s, p);
    }
    return ss2d_take_launch_rc();
}
// the fused-delta form: 16 states, no z, 4-element rows, direction-aware addressing (the x_proj output is strided)
template <typename IO>
int launch_fwd_dt(int nw, const ScanFwdParams &p, hipStream_t s) {
    dim3 grid(p.nwg);
    if (nw == 4) ss2d_launch((scan_fwd_kernel<4, false, true, true, IO, true, true>), grid, dim3(256), 0, s, p);
    else if (nw == 2) ss2d_launch((scan_fwd_kernel<2, false, true, true, IO, true, true>), grid, dim3(128), 0, s, p);
    else ss2d_launch((scan_fwd_kernel<1, false, true, true, IO, true, true>), grid, dim3(64), 0, s, p);
    return ss2d_take_launch_rc();
}

// two channels per DPP row (L <= 64, one chunk): 16 states, no z, 4-element rows, four waves = 32 channels per workgroup
template <typename IO, bool DIRS>
int launch_fwd_pack(const ScanFwdParams &p, hipStream_t s) {
    ss2d_launch((scan_fwd_kernel<4, false, true, true, IO, DIRS, false, true>), dim3(p.nwg), dim3(256), 0, s, p);
    return ss2d_take_launch_rc();
}

template <typename IO, bool DIRS>
int launch_fwd_nw(int nw, const ScanFwdParams &p, bool has_z, bool vec4, hipStream_t s) {
    return nw == 4 ? launch_fwd<4, IO, DIRS>(p, has_z, vec4, s) : nw == 2 ? launch_fwd<2, IO, DIRS>(p, has_z, vec4, s)
                                                                           : launch_fwd<1, IO, DIRS>(p, has_z, vec4, s);
}

}  // namespace

#ifdef SS2D_FWD_PROBE
extern "C" int ss2d_debug_set_fwd_probe(void *buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fwd_probe), &buf, sizeof(buf)); }
#endif

extern "C" int ss2d_scan_num_chunks(int L) { return (L + CHUNK - 1) / CHUNK; }

extern "C" size_t ss2d_scan_ckpt_bytes(int batch, int dim, int L, int dstate) {
    return (size_t)batch * dim * ss2d_scan_num_chunks(L) * dstate * sizeof(float);
}

extern "C" int ss2d_selective_scan_fwd(int dtype, const void *u, const void *delta, const float *A, const void *B,
                                       const void *C, const float *D, const void *z, const float *delta_bias,
                                       void *out, float *last_state, float *ckpt, int batch, int dim, int L,
                                       int dstate, int ngroups, int delta_softplus, void *stream) {
    return ss2d_selective_scan_dir_fwd(dtype, u, delta, A, B, C, D, z, delta_bias, out, last_state, ckpt, batch, dim, L,
                                       dstate, ngroups, delta_softplus, ngroups, 0u, 0, stream);
}

static int scan_fwd_impl(int dtype, const void *u, const void *delta, const float *A, const void *B, const void *C, const float *D,
                         const void *z, const float *delta_bias, void *out, float *last_state, float *ckpt, int batch, int dim, int L,
                         int dstate, int ngroups, int delta_softplus, int u_groups, unsigned rev_mask, long long bc_bg_stride,
                         const float *dt_w, int dt_rank, void *stream);

extern "C" int ss2d_selective_scan_dir_fwd(int dtype, const void *u, const void *delta, const float *A, const void *B,
                                           const void *C, const float *D, const void *z, const float *delta_bias,
                                           void *out, float *last_state, float *ckpt, int batch, int dim, int L,
                                           int dstate, int ngroups, int delta_softplus, int u_groups,
                                           unsigned rev_mask, long long bc_bg_stride, void *stream) {
    return scan_fwd_impl(dtype, u, delta, A, B, C, D, z, delta_bias, out, last_state, ckpt, batch, dim, L, dstate, ngroups,
                         delta_softplus, u_groups, rev_mask, bc_bg_stride, nullptr, 0, stream);
}

extern "C" int ss2d_selective_scan_core_fwd(int dtype, const void *u, const void *x_dbl, const float *dt_w, const float *A,
                                            const float *D, const float *delta_bias, void *out, float *ckpt, int batch, int dim,
                                            int L, int dstate, int ngroups, int dt_rank, int u_groups, unsigned rev_mask,
                                            void *stream) {
    if (!x_dbl || !dt_w) { ss2d_set_error("selective_scan_core_fwd: null required pointer"); return -2; }
    if (dt_rank <= 0 || dt_rank > SS2D_MAX_DT_RANK) { ss2d_set_error("selective_scan_core_fwd: dt_rank %d not in [1, %d]", dt_rank, SS2D_MAX_DT_RANK); return -4; }
    if (dstate != 16 || L % 4) { ss2d_set_error("selective_scan_core_fwd: needs dstate == 16 and L %% 4 == 0 (got %d, %d)", dstate, L); return -4; }
    const size_t es = dtype == SS2D_DTYPE_BF16 ? 2 : 4;
    const char *base = static_cast<const char *>(x_dbl);
    const long long stride = (long long)(dt_rank + 2 * dstate) * L;
    return scan_fwd_impl(dtype, u, x_dbl, A, base + (size_t)dt_rank * L * es, base + (size_t)(dt_rank + dstate) * L * es, D, nullptr,
                         delta_bias, out, nullptr, ckpt, batch, dim, L, dstate, ngroups, 1, u_groups, rev_mask, stride, dt_w, dt_rank, stream);
}

static int scan_fwd_impl(int dtype, const void *u, const void *delta, const float *A, const void *B, const void *C, const float *D,
                         const void *z, const float *delta_bias, void *out, float *last_state, float *ckpt, int batch, int dim, int L,
                         int dstate, int ngroups, int delta_softplus, int u_groups, unsigned rev_mask, long long bc_bg_stride,
                         const float *dt_w, int dt_rank, void *stream) {
    if (dtype != SS2D_DTYPE_F32 && dtype != SS2D_DTYPE_BF16) { ss2d_set_error("selective_scan_fwd: unknown dtype %d", dtype); return -1; }
    if (!u || !delta || !A || !B || !C || !out) { ss2d_set_error("selective_scan_fwd: null required pointer"); return -2; }
    if (batch <= 0 || dim <= 0 || L <= 0 || dstate <= 0 || ngroups <= 0) { ss2d_set_error("selective_scan_fwd: non-positive size"); return -3; }
    if (dstate > SS2D_MAX_DSTATE) { ss2d_set_error("selective_scan_fwd: dstate %d > %d", dstate, SS2D_MAX_DSTATE); return -4; }
    if (dim % ngroups) { ss2d_set_error("selective_scan_fwd: dim %d not divisible by ngroups %d", dim, ngroups); return -5; }
    if (u_groups <= 0 || ngroups % u_groups) { ss2d_set_error("selective_scan_fwd: u_groups %d must divide ngroups %d", u_groups, ngroups); return -5; }
    if (ngroups > 32 && rev_mask) { ss2d_set_error("selective_scan_fwd: rev_mask needs ngroups <= 32"); return -5; }
    const long long bc_dense = (long long)dstate * L;
    if (bc_bg_stride == 0) bc_bg_stride = bc_dense;
    if (bc_bg_stride < bc_dense) { ss2d_set_error("selective_scan_fwd: bc_bg_stride %lld < dstate*L", bc_bg_stride); return -5; }
    ScanFwdParams p;
    p.Gu = u_groups; p.rev_mask = rev_mask; p.bc_stride = bc_bg_stride;
    p.dtw = dt_w; p.R = dt_rank;
    p.u = u; p.delta = delta; p.A = A; p.B = B; p.C = C;
    p.D = D; p.z = z; p.bias = delta_bias; p.out = out; p.last = last_state; p.ckpt = ckpt;
    p.batch = batch; p.dim = dim; p.L = L; p.N = dstate; p.G = ngroups; p.dpg = dim / ngroups;
    p.nchunks = ss2d_scan_num_chunks(L); p.softplus = delta_softplus;
    int nw = p.dpg >= 16 ? 4 : (p.dpg >= 8 ? 2 : 1);
    static const int env_nw = (int)ss2d_env_ll("SS2D_FWD_NW", 0);   // tuning knob, read once
    if (env_nw == 1 || env_nw == 2 || env_nw == 4) nw = env_nw;
    const uintptr_t amask = dtype == SS2D_DTYPE_BF16 ? 7 : 15;  // 4 elements
    const bool vec4 = (L % 4 == 0) && (bc_bg_stride % 4 == 0) &&
                      ((((uintptr_t)u | (uintptr_t)delta | (uintptr_t)B | (uintptr_t)C | (uintptr_t)out | (uintptr_t)z) & amask) == 0);
    // sequences that fit half a DPP row (the 7 x 7 / 8 x 8 stages): two channels per row, 32 per workgroup
    static const bool env_nopack = getenv("SS2D_FWD_NO_PACK") != nullptr;      // A/B switch, read once
    const bool pack = !env_nopack && !dt_w && !z && !last_state && dstate == 16 && vec4 && L <= CHUNK / 2 && nw == 4 && p.dpg >= 32;
    const int cpw = pack ? 32 : nw * 4;                          // channels per workgroup
    p.wgs_per_group = (p.dpg + cpw - 1) / cpw;
    const long long nwg = (long long)batch * ngroups * p.wgs_per_group;
    if (nwg > 0x7fffffffLL) { ss2d_set_error("selective_scan_fwd: grid too large"); return -6; }
    p.nwg = (int)nwg;
    hipStream_t s = (hipStream_t)stream;
    const bool dirs = rev_mask != 0 || u_groups != ngroups || bc_bg_stride != bc_dense;
    if (pack) {
        const int rc = dtype == SS2D_DTYPE_BF16 ? (dirs ? launch_fwd_pack<bf16, true>(p, s) : launch_fwd_pack<bf16, false>(p, s))
                                                : (dirs ? launch_fwd_pack<float, true>(p, s) : launch_fwd_pack<float, false>(p, s));
        if (rc) ss2d_set_error("selective_scan_fwd: launch failed: %s", hipGetErrorString((hipError_t)rc));
        return rc;
    }
    if (dt_w) {
        if (!vec4) { ss2d_set_error("selective_scan_core_fwd: u, x_dbl and out must be aligned to 4 elements"); return -7; }
        const int rc = dtype == SS2D_DTYPE_BF16 ? launch_fwd_dt<bf16>(nw, p, s) : launch_fwd_dt<float>(nw, p, s);
        if (rc) ss2d_set_error("selective_scan_core_fwd: launch failed: %s", hipGetErrorString((hipError_t)rc));
        return rc;
    }
    int rc = dtype == SS2D_DTYPE_BF16
                 ? (dirs ? launch_fwd_nw<bf16, true>(nw, p, z != nullptr, vec4, s) : launch_fwd_nw<bf16, false>(nw, p, z != nullptr, vec4, s))
                 : (dirs ? launch_fwd_nw<float, true>(nw, p, z != nullptr, vec4, s) : launch_fwd_nw<float, false>(nw, p, z != nullptr, vec4, s));
    if (rc) ss2d_set_error("selective_scan_fwd: launch failed: %s", hipGetErrorString((hipError_t)rc));
    return rc;
}
