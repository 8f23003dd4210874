// Selective scan backward, state-row layout with CHUNK-PIPELINED waves, for gfx950 -- the fast path of
// ss2d_selective_scan_{,dir_}bwd / _core_bwd for dstate == 16 without z on sequences of many chunks (the form the
// large stages of Mamba-UNet issue, /root/reference/code/networks/mamba_sys.py:436-442; forward semantics :62-144).
//
// Lane layout and per-channel arithmetic are those of ss2d_scan_bwd_sr.hip: a wavefront works on ONE (batch, channel)
// sequence at a time, its 4 DPP rows own 4 states each (2 packed pairs), the 16 lanes of a row 8 consecutive positions
// each (a 128-position chunk x 16 states per wave), dB / dC are sums over the channels and accumulate in 64 registers.
// What differs is how the chunks and channels of a workgroup's rows are dealt to its 8 waves:
//   * ss2d_scan_bwd_sr.hip: all waves work on the SAME chunk and split its channels; once per chunk they meet at a
//     barrier, add their 8 register tiles through LDS and store one tile: 15 % of the kernel (barrier waits included).
//   * here a chunk belongs to ONE wave, which walks ALL channels of the part for it and stores its dB / dC tile straight
//     from registers: no cross-wave sum, no workgroup barrier inside the kernel.  The 8 waves work on 8 consecutive chunks
//     at once, skewed by a channel: the adjoint r_l = a_l (dy_l C_l + r_{l+1}) of channel k enters chunk c from chunk
//     c + 1, so (c, k) may start when the wave of chunk c + 1 has finished channel k.  The carry (16 floats per channel)
//     travels through LDS; a wave publishes "item s, channels done" in a monotonic LDS word after every channel and its
//     successor reads that word before it reads the carry (LDS executes the instructions of all waves of a CU in issue
//     order, so no fence beyond the compiler's is needed).  Every dependency points to a chunk further right, all 8 waves
//     of a workgroup are resident together, hence the wait terminates; it is bounded anyway (a wave that gives up poisons
//     dA with NaN instead of hanging the GPU).
//   * nchunks is rarely a multiple of 8: the first R = nchunks % 8 chunks (the rightmost, which depend on nothing) are run
//     in "split" form -- wave w takes the w-th eighth of the channels through these R chunks -- and their 8 register tiles
//     are added into the (pre-zeroed) tile with float atomics; the remaining 8 m chunks are pipelined, m per wave.
//     Config 2 (L = 3136: 25 chunks, 72 rows per workgroup): 9 channels of chunk 24, then 3 whole chunks per wave.
// B / C of a chunk are staged by the owning wave in its private 16 KB LDS tile.  dA / dD / ddelta_bias are accumulated
// per row in LDS (float atomics: several waves hold the same channel at different chunks) and leave as one global
// atomic per value and workgroup.  Inputs of channel k + 1 are requested at the top of channel k and staged (softplus,
// delta * u, dout -> the wave's LDS strip) behind its state pairs, so that the strip is long written when k + 1 reads it.
#include <cstdlib>
#include "ss2d_scan.h"
#include "../../include/ss2d.h"

namespace {

constexpr int T = 8;
constexpr int CHUNK = 128;
constexpr int NW = 8;
constexpr int ROWS_MAX = SS2D_SRP_ROWS_MAX;
constexpr int TILE = 2 * 16 * CHUNK;        // floats: [B|C][pair][item pair][seg] f4, as in ss2d_scan_fwd.hip
constexpr int STG = 3 * CHUNK;              // per wave: softplus(delta), delta*u, dout of one channel-chunk
constexpr int ROWF = 36;                    // per row of the workgroup: 16 dA | dD | ddelta_bias (sums) | bias | D | 16 A (parameters)
constexpr int LDS_FLOATS = NW * TILE + NW * STG + 2 * ROWS_MAX * 16 + ROWS_MAX * ROWF + 16;
constexpr int SPIN_LIMIT = 1 << 22;         // x ~0.1 us: far beyond any legitimate wait
static_assert(LDS_FLOATS * 4 <= 160 * 1024, "one workgroup per CU");
static_assert(CHUNK == SS2D_SCAN_CHUNK, "checkpoint spacing is part of the ABI");

#ifdef SS2D_SRP_PROBE
// timing probes (debug builds only: make variant SRC=ss2d_scan_bwd_srp NAME=probep DEFS=-DSS2D_SRP_PROBE, tools/probe_bwd.py --pipe)
__device__ unsigned long long *g_srp_probe;
#define SRP_NPROBE 16
#define SRP_PROBE(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                          pacc[i] += (unsigned)(t_ - tprev); tprev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SRP_PROBE(i) do { } while (0)
#endif

// A copy the register allocator cannot coalesce away.  A prefetched value that is both consumed now and kept for later
// (`keep = loaded; loaded = next load`) otherwise gets its copy at the loop latch, on the NEW load -- i.e. a wait for a
// load that was issued moments ago.
__device__ __forceinline__ float hard_copy(float v) { float r; asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(v)); return r; }
__device__ __forceinline__ f2 hard_copy(f2 v) { return f2{hard_copy(v.x), hard_copy(v.y)}; }
__device__ __forceinline__ f4 hard_copy(f4 v) { return f4{hard_copy(v.x), hard_copy(v.y), hard_copy(v.z), hard_copy(v.w)}; }

template <int LANE> __device__ __forceinline__ float lane_value(float v) {   // value of lane LANE (v_readlane: scalar)
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), LANE));
}

struct Item {          // one (part, chunk, channel range) of a wave's schedule; every member is wave-uniform
    int valid, q, t, chunk, k_lo, k_hi, rw0, bg, rev, atomic_out;
    int dep_mode;      // 0: the carry comes from this wave's previous item or is zero; 1: from dep_wave; 2: from the split phase
    int dep_wave, dep_item, cb;
    size_t r0, urow0;  // (batch, channel) row of the part's first channel in delta / du / ddelta / ckpt, and in u / dout
};
struct Chan {          // where one channel-chunk's inputs live
    size_t rowU, rowD;
    int rw, chunk, rev, bg, valid;
};

template <bool VEC, typename IO, bool DIRS, bool WS, int DTR = 0>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void scan_bwd_srp_kernel(const ScanBwdParams p) {
    extern __shared__ __align__(16) float smem[];
#ifdef SS2D_SRP_PROBE
    unsigned pacc[SRP_NPROBE] = {};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tprev;
#endif
    const int tid = threadIdx.x, lane = tid & 63, row = lane >> 4, seg = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Lane-derived LDS addresses are rebuilt from this opaque copy of the lane id where they are used: values hoisted out
    // of the channel loop would be spilled around it (it needs every register), and a spill reload waits for ALL
    // outstanding memory operations -- the prefetches included.
    auto olane = [&]() { int t_ = lane; asm volatile("" : "+v"(t_)); return t_; };
    f4 *tile = reinterpret_cast<f4 *>(smem) + wave * (TILE / 4);
    float *st = smem + NW * TILE + wave * STG;
    float *carry = smem + NW * TILE + NW * STG;                  // [chunk parity][row of the workgroup][16 states]
    float *rows = carry + 2 * ROWS_MAX * 16;                     // [row of the workgroup][ROWF]
    int *progress = reinterpret_cast<int *>(rows + ROWS_MAX * ROWF);   // [NW] progress words, [8] = "a wave gave up waiting"

    const int L = p.L, nchunks = p.nchunks, R = p.split, M = (nchunks - R) >> 3, ipp = R + M;
    const long long total = (long long)p.batch * p.dim;
    const long long lo = (long long)blockIdx.x * p.rows_per_wg;
    const long long hi = lo + p.rows_per_wg < total ? lo + p.rows_per_wg : total;
    const int nrows = (int)(hi - lo);
    const int bg0 = (int)(lo / p.dpg);
    const int nparts = (int)((hi - 1) / p.dpg) - bg0 + 1;
    const unsigned rowbytes = (unsigned)L * sizeof(IO);
    const size_t bcs = DIRS ? (size_t)p.bc_stride : 16 * (size_t)L;

    // ---- per-row records: zero sums, and the channel's parameters (read from LDS by every wave that meets the channel)
    for (int i = tid; i < nrows * ROWF; i += NW * 64) {
        const int r = i / ROWF, j = i - r * ROWF, c = (int)((lo + r) % p.dim);
        float v = 0.0f;
        if (j == 18) v = p.bias ? p.bias[c] : 0.0f;
        else if (j == 19) v = p.D ? p.D[c] : 0.0f;
        else if (j >= 20) v = p.A[(size_t)c * 16 + (j - 20)];
        rows[i] = v;
    }
    if (tid < 16) progress[tid] = 0;
    if (WS && R > 0) {   // the tiles of the split chunks are sums of 8 waves' atomics: zero them first (L % 4 == 0 with a workspace)
        for (int q = 0; q < nparts; ++q) {
            const int bg = bg0 + q, g = bg % p.G;
            const bool rev = DIRS && ((p.rev_mask >> g) & 1u);
            float *dst0 = p.ws + ((size_t)blockIdx.x + bg) * 32 * (size_t)L;
            for (int r = 0; r < R; ++r)
                for (int i = tid; i < 1024; i += NW * 64) {
                    const int nn = i >> 5, pos = (nchunks - 1 - r) * CHUNK + 4 * (i & 31);
                    if (pos < L) *reinterpret_cast<f4 *>(dst0 + (size_t)nn * L + (rev ? L - 4 - pos : pos)) = f4{0.0f, 0.0f, 0.0f, 0.0f};
                }
        }
    }
    __syncthreads();   // (waits for the zero stores too: the atomics that follow come from this workgroup, i.e. the same L2)

    // ---- the schedule of this wave: per part R split items, then M pipelined chunks
    auto decode = [&](int q, int t) {
        Item it;
        it.q = q; it.t = t; it.valid = q < nparts;
        const int bg = bg0 + (it.valid ? q : 0);
        const int b = bg / p.G, g = bg - b * p.G;
        const long long glo = (long long)bg * p.dpg, ghi = glo + p.dpg;
        const long long plo = glo > lo ? glo : lo, phi = ghi < hi ? ghi : hi;
        const int n = (int)(phi - plo);
        it.bg = bg; it.rw0 = (int)(plo - lo); it.r0 = (size_t)plo;
        const int cin0 = (int)(plo - glo);
        it.rev = DIRS && ((p.rev_mask >> g) & 1u);
        const int ug = DIRS ? g / (p.G / p.Gu) : g;
        it.urow0 = DIRS ? ((size_t)(b * p.Gu + ug) * p.dpg + cin0) : (size_t)plo;
        it.cb = (n + NW - 1) / NW;
        it.dep_wave = 0; it.dep_item = 0;
        if (t < R) {
            it.chunk = nchunks - 1 - t;
            it.k_lo = wave * it.cb < n ? wave * it.cb : n;
            it.k_hi = it.k_lo + it.cb < n ? it.k_lo + it.cb : n;
            it.dep_mode = 0; it.atomic_out = 1;
        } else {
            const int j = (t - R) * NW + wave;
            it.chunk = nchunks - 1 - R - j;
            it.k_lo = 0; it.k_hi = n; it.atomic_out = !WS;
            if (j == 0) { it.dep_mode = R > 0 ? 2 : 0; it.dep_item = q * ipp + R - 1; }
            else { it.dep_mode = 1; it.dep_wave = (wave + NW - 1) & (NW - 1); it.dep_item = q * ipp + t - (wave == 0 ? 1 : 0); }
        }
        return it;
    };
    auto next_item = [&](int q, int t) {       // the first non-empty item behind (q, t)
        for (;;) {
            if (++t == ipp) { t = 0; ++q; }
            Item it = decode(q, t);
            if (!it.valid || it.k_lo < it.k_hi) return it;
        }
    };
    auto chan_of = [&](const Item &it, int k) {
        Chan c;
        c.rowU = it.urow0 + k; c.rowD = it.r0 + k; c.rw = it.rw0 + k; c.chunk = it.chunk; c.rev = it.rev; c.bg = it.bg; c.valid = it.valid;
        return c;
    };
    Item cur = decode(0, 0);
    if (cur.k_lo >= cur.k_hi) cur = next_item(0, 0);

    if (cur.valid) {
        typedef const float __attribute__((address_space(4))) *cfp;
        constexpr int NR = DTR > 0 ? DTR : 1;

        // u / delta / dout at this lane's two positions, requested TWO channels ahead (HBM latency under load is longer than
        // one channel); `pf_row` says which channel the registers hold, and a mismatch at the consumer reloads them
        f2 nu, nd = f2{0.0f, 0.0f}, ny;
        f4 nck;                                        // the next channel's checkpoint, requested one channel ahead
        float nw[NR];                                  // DTR: the next channel's dt_proj weight row (scalar loads)
        f2 dtsv[NR];                                   // DTR: dts at this lane's two positions (per item)
        auto input_load = [&](const Chan &c) {
            const int ps_ = c.chunk * CHUNK + 8 * seg + 2 * row;
            nu = buf_load2m<VEC, IO>(make_rsrc(static_cast<const IO *>(p.u) + c.rowU * (size_t)L, rowbytes), ps_, L, c.rev);
            if (DTR == 0) nd = buf_load2m<VEC, IO>(make_rsrc(static_cast<const IO *>(p.delta) + c.rowD * (size_t)L, rowbytes), ps_, L, c.rev);
            ny = buf_load2m<VEC, IO>(make_rsrc(static_cast<const IO *>(p.dout) + c.rowU * (size_t)L, rowbytes), ps_, L, c.rev);
        };
        auto dts_load = [&](const Chan &c) {     // DTR: the chunk's dts values at this lane's two positions (all channels of the item share them)
            if (DTR > 0) {
                const int ps_ = c.chunk * CHUNK + 8 * seg + 2 * row;
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    dtsv[r] = buf_load2m<VEC, IO>(make_rsrc(static_cast<const IO *>(p.delta) + (size_t)c.bg * bcs + (size_t)r * L, rowbytes), ps_, L, c.rev);
            }
        };
        auto ckpt_load = [&](const Chan &c) {
            nck = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(make_rsrc(p.ckpt + (c.rowD * nchunks + c.chunk) * 16, 64), 16 * row, 0, 0));
            if (DTR > 0) {
                const cfp wp = (cfp)(uintptr_t)(p.dtw + (size_t)(c.rowD % p.dim) * DTR);
#pragma unroll
                for (int r = 0; r < NR; ++r) nw[r] = wp[r];
            }
        };
        f2 o_u, o_sg;       // u and sigmoid at the own two positions of the staged channel; softplus(delta) and dout are re-read from the strip
        auto stage = [&](const Chan &c) {            // nu / nd / ny (+ nw, dtsv) -> strip, o_*
            const int ol = olane(), ps_ = c.chunk * CHUNK + 8 * (ol & 15) + 2 * (ol >> 4);
            f2 cd = nd;
            if (DTR > 0) {
#pragma unroll
                for (int r = 0; r < NR; ++r) cd = fma2(splat2(nw[r]), dtsv[r], cd);
            }
            const float bias = rows[c.rw * ROWF + 18];
            float x0 = cd.x + bias, x1 = cd.y + bias, sg0, sg1;
            {   // no branch here: this runs between the state pairs and the reductions of their results, and LLVM sinks the 64
                // scalar FMAs that build ddu / ddt below the first branch it finds (keeping dL/dh, B and w of both pairs alive)
                const float y0 = softplus_sig<true>(x0, sg0), y1 = softplus_sig<true>(x1, sg1);
                const bool sp = p.softplus != 0;
                x0 = sp ? y0 : x0; x1 = sp ? y1 : x1; sg0 = sp ? sg0 : 1.0f; sg1 = sp ? sg1 : 1.0f;
            }
            const f2 dt_ = f2{ps_ < L ? x0 : 0.0f, ps_ + 1 < L ? x1 : 0.0f};     // positions past L are the identity
            o_u = hard_copy(nu); o_sg = f2{sg0, sg1};                             // u, dout: zero past L (buffer range check)
            const int sto = 32 * (ol >> 4) + 2 * (ol & 15);
            *reinterpret_cast<f2 *>(st + sto) = dt_;
            *reinterpret_cast<f2 *>(st + CHUNK + sto) = dt_ * nu;
            *reinterpret_cast<f2 *>(st + 2 * CHUNK + sto) = ny;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        };

        // The 8 positions of this lane's segment (staged by the 4 rows of the wave) and the channel's A row are read from LDS at
        // the END of the previous channel: their latency hides behind its tail instead of standing in front of the pairs.
        f2 dtp[4], dup[4], dyp[4];                            // pairs of neighbouring positions (see pk_*_sel)
        f4 Av;
        auto strip_read = [&](int rw) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int ol = olane(), sta = 2 * (ol & 15);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                dtp[j] = *reinterpret_cast<const f2 *>(st + sta + 32 * j);
                dup[j] = *reinterpret_cast<const f2 *>(st + CHUNK + sta + 32 * j);
                dyp[j] = *reinterpret_cast<const f2 *>(st + 2 * CHUNK + sta + 32 * j);
            }
            Av = *reinterpret_cast<const f4 *>(rows + rw * ROWF + 20 + 4 * (ol >> 4));
        };
        // ---- prologue: the first channel of the first item
        {
            const Chan c0 = chan_of(cur, cur.k_lo);
            dts_load(c0);
            input_load(c0);
            ckpt_load(c0);
            stage(c0);
            input_load(chan_of(cur, cur.k_lo + 1));
            strip_read(cur.rw0 + cur.k_lo);
        }
        int s_idx = cur.q * ipp + cur.t;

        for (;;) {
            int nq = cur.q, nt = cur.t;      // the wave's next item, looked up in the last channels of this one
            bool more = false;
            const int base = cur.chunk * CHUNK;
            const bool rev = cur.rev;
            SRP_PROBE(12);
            // ---- B / C tile of this chunk -> the wave's private LDS tile
            {
                const __amdgpu_buffer_rsrc_t rsB = make_rsrc(static_cast<const IO *>(p.B) + (size_t)cur.bg * bcs, 16u * L * sizeof(IO));
                const __amdgpu_buffer_rsrc_t rsC = make_rsrc(static_cast<const IO *>(p.C) + (size_t)cur.bg * bcs, 16u * L * sizeof(IO));
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f4 lo4[2][2], hi4[2][2];
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int qq = 0; qq < 2; ++qq) {
                            const int idx = lane + 64 * (qq + 2 * h), tnp = idx >> 5, tq4 = idx & 31;
                            const int pos = base + tq4 * 4;
                            lo4[m][qq] = buf_load4m<VEC, IO>(m ? rsC : rsB, (unsigned)(2 * tnp) * L, pos, L, rev);
                            hi4[m][qq] = buf_load4m<VEC, IO>(m ? rsC : rsB, (unsigned)(2 * tnp + 1) * L, pos, L, rev);
                        }
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int qq = 0; qq < 2; ++qq) {
                            const int idx = lane + 64 * (qq + 2 * h), tnp = idx >> 5, tq4 = idx & 31, tj0 = (tq4 & 1) * 2, tsg = tq4 >> 1;
                            f4 *td = tile + m * 512 + (tnp * 4 + tj0) * 16 + tsg;
                            td[0] = f4{lo4[m][qq].x, hi4[m][qq].x, lo4[m][qq].y, hi4[m][qq].y};
                            td[16] = f4{lo4[m][qq].z, hi4[m][qq].z, lo4[m][qq].w, hi4[m][qq].w};
                        }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }

            SRP_PROBE(11);
            f2 accB[2][T], accC[2][T];
#pragma unroll
            for (int pi = 0; pi < 2; ++pi)
#pragma unroll
                for (int t = 0; t < T; ++t) { accB[pi][t] = f2{0.0f, 0.0f}; accC[pi][t] = f2{0.0f, 0.0f}; }

            const bool first = cur.chunk == nchunks - 1;
            const float *cin = carry + ((cur.chunk + 1) & 1) * (ROWS_MAX * 16) + cur.rw0 * 16;   // adjoint entering from chunk + 1
            float *cout = carry + (cur.chunk & 1) * (ROWS_MAX * 16) + cur.rw0 * 16;
            int dwv = cur.dep_mode == 2 ? 0 : cur.dep_wave, dcnt = 0;   // split dependency: wave k / cb, channels of its block done

            for (int k = cur.k_lo; k < cur.k_hi; ++k) {
                SRP_PROBE(10);
                // ---- the next channel (k + 1 of this item, or the first of the wave's next item) and the one behind it
                Chan c1 = chan_of(cur, k + 1), c2 = chan_of(cur, k + 2);
                if (k + 2 >= cur.k_hi) {      // scalar work only: no load sits under this branch
                    const Item nx = next_item(cur.q, cur.t);
                    const bool lastk = k + 1 == cur.k_hi;
                    if (lastk) { more = nx.valid; nq = nx.q; nt = nx.t; }
                    if (nx.valid) {
                        if (lastk) { c1 = chan_of(nx, nx.k_lo); c2 = chan_of(nx, nx.k_lo + 1); }    // every item has >= 2 channels (host)
                        else c2 = chan_of(nx, nx.k_lo);
                    } else {                  // prefetches behind the very last channel go to a valid row and are not used
                        if (lastk) c1 = chan_of(cur, cur.k_lo);
                        c2 = chan_of(cur, cur.k_lo);
                    }
                }
                const f4 ck = hard_copy(nck);
                ckpt_load(c1);
                if (DTR > 0 && k + 1 == cur.k_hi) dts_load(c1);      // the strip already holds this channel: dtsv is free for the next item
                // the predecessor's progress word, read early: in the steady state it is already far enough
                int need = 0, pv = 0;
                if (cur.dep_mode) {
                    if (cur.dep_mode == 2) { if (dcnt == cur.cb) { dcnt = 0; ++dwv; } ++dcnt; }
                    need = cur.dep_item * 256 + (cur.dep_mode == 2 ? dcnt : k + 1);
                    pv = __builtin_amdgcn_readfirstlane(__hip_atomic_load(progress + dwv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                }
                SRP_PROBE(0);
                const float *rowrec = rows + (cur.rw0 + k) * ROWF;
                const f2 c_u = o_u, c_sg = o_sg;
                SRP_PROBE(1);
                if (cur.dep_mode) {     // the carry of (chunk + 1, k) must have been written
                    int spins = 0;
                    while (pv < need) {
                        __builtin_amdgcn_s_sleep(2);
                        pv = __builtin_amdgcn_readfirstlane(__hip_atomic_load(progress + dwv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                        if (++spins > SPIN_LIMIT || __hip_atomic_load(progress + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                            __hip_atomic_store(progress + 8, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                            break;
                        }
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                SRP_PROBE(13);
                f4 rcv = *reinterpret_cast<const f4 *>(cin + k * 16 + 4 * (olane() >> 4));
                if (first) rcv = f4{0.0f, 0.0f, 0.0f, 0.0f};
                float S = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) S += dtp[j].x + dtp[j].y;

                float ddt[T], ddu[T];                                   // sums over this row's 4 states
                f2 dAp[2], rleft[2];
#pragma unroll
                for (int t = 0; t < T; ++t) { ddt[t] = 0.0f; ddu[t] = 0.0f; }

#pragma unroll
                for (int pi = 0; pi < 2; ++pi) {
                    const int np = 2 * row + pi;
                    const f4 *__restrict__ tB = tile + (np * 4) * 16 + seg, *__restrict__ tC = tB + 512;
                    const f2 A2 = f2{Av[2 * pi], Av[2 * pi + 1]} * SS2D_LOG2E;
                    const f2 hin = f2{ck[2 * pi], ck[2 * pi + 1]};
                    const f2 rcar = f2{rcv[2 * pi], rcv[2 * pi + 1]};
                    f2 a[T], h[T];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { a[2 * j] = exp2_2(pk_mul_lo(dtp[j], A2)); a[2 * j + 1] = exp2_2(pk_mul_hi(dtp[j], A2)); }
                    const f2 at = exp2_2(splat2(S) * A2);
                    f2 hstart;
                    f4 Cq[4];
                    {   // ---- forward states of the chunk: lane-local pass, row scan, true pass
                        f2 bb[T];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const f4 Bq = tB[j * 16];
                            bb[2 * j] = pk_mul_lo(dup[j], f2{Bq.x, Bq.y});
                            bb[2 * j + 1] = pk_mul_hi(dup[j], f2{Bq.z, Bq.w});
                        }
                        f2 hl = f2{seg == 0 ? hin.x : 0.0f, seg == 0 ? hin.y : 0.0f};
#pragma unroll
                        for (int t = 0; t < T; ++t) hl = fma2(a[t], hl, bb[t]);
                        float hx = hl.x, hy = hl.y, ax = at.x, ay = at.y;
#pragma unroll
                        for (int j = 0; j < 4; ++j) Cq[j] = tC[j * 16];   // requested here: they arrive behind the row scan
                        row_scan_fwd_pair(hx, hy, ax, ay);
                        hstart.x = dpp_mov<DPP_ROW_SHR(1), 0xf, false>(hin.x, hx);
                        hstart.y = dpp_mov<DPP_ROW_SHR(1), 0xf, false>(hin.y, hy);
                        f2 hh = hstart;
#pragma unroll
                        for (int t = 0; t < T; ++t) { hh = fma2(a[t], hh, bb[t]); h[t] = hh; }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    SRP_PROBE(2 + 3 * pi);
                    // ---- adjoint r_l = a_l * (dy_l * C_l + r_{l+1}): lane-local suffix pass, row scan, true pass
                    f2 rr;
                    f4 Bn, Cn;
                    {
                        f2 rl = f2{seg == 15 ? rcar.x : 0.0f, seg == 15 ? rcar.y : 0.0f};
#pragma unroll
                        for (int j = 3; j >= 0; --j) {
                            rl = a[2 * j + 1] * pk_fma_hi(dyp[j], f2{Cq[j].z, Cq[j].w}, rl);
                            rl = a[2 * j] * pk_fma_lo(dyp[j], f2{Cq[j].x, Cq[j].y}, rl);
                        }
                        float rx = rl.x, ry = rl.y, ax = at.x, ay = at.y;
                        Bn = tB[3 * 16]; Cn = tC[3 * 16];                 // first operands of the true pass, behind the row scan
                        row_scan_rev_pair(rx, ry, ax, ay);
                        rr.x = dpp_mov<DPP_ROW_SHL(1), 0xf, false>(rcar.x, rx);
                        rr.y = dpp_mov<DPP_ROW_SHL(1), 0xf, false>(rcar.y, ry);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    SRP_PROBE(3 + 3 * pi);
                    f2 dAacc = f2{0.0f, 0.0f};
#pragma unroll
                    for (int j = 3; j >= 0; --j) {
                        // B and C of the pair are re-read (keeping them alive costs 32 registers), one item pair ahead
                        const f4 Bq = Bn, Cj = Cn;
                        if (j > 0) { Bn = tB[(j - 1) * 16]; Cn = tC[(j - 1) * 16]; }
                        {   // t = 2j + 1
                            const int t = 2 * j + 1;
                            const f2 Bt = f2{Bq.z, Bq.w}, Ct = f2{Cj.z, Cj.w};
                            const f2 dh = pk_fma_hi(dyp[j], Ct, rr);             // dL/dh_t
                            rr = a[t] * dh;                                       // a_t * dL/dh_t, flows to t-1
                            pk_fmac_sel<1>(accC[pi][t], dyp[j], h[t]);
                            pk_fmac_sel<1>(accB[pi][t], dup[j], dh);
                            ddu[t] = fmaf(dh.x, Bt.x, fmaf(dh.y, Bt.y, ddu[t]));
                            const f2 w = rr * h[t - 1];                           // dL/da_t * a_t
                            ddt[t] = fmaf(w.x, A2.x, fmaf(w.y, A2.y, ddt[t]));   // log2 units; scaled by ln2 below
                            dAacc = fma2(w, splat2(dtp[j].y), dAacc);
                        }
                        {   // t = 2j
                            const int t = 2 * j;
                            const f2 Bt = f2{Bq.x, Bq.y}, Ct = f2{Cj.x, Cj.y};
                            const f2 dh = pk_fma_lo(dyp[j], Ct, rr);
                            rr = a[t] * dh;
                            pk_fmac_sel<0>(accC[pi][t], dyp[j], h[t]);
                            pk_fmac_sel<0>(accB[pi][t], dup[j], dh);
                            ddu[t] = fmaf(dh.x, Bt.x, fmaf(dh.y, Bt.y, ddu[t]));
                            const f2 hp = (j > 0) ? h[t - 1] : hstart;
                            const f2 w = rr * hp;
                            ddt[t] = fmaf(w.x, A2.x, fmaf(w.y, A2.y, ddt[t]));
                            dAacc = fma2(w, splat2(dtp[j].x), dAacc);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    SRP_PROBE(4 + 3 * pi);
                    dAp[pi] = dAacc;
                    rleft[pi] = rr;                                           // lane 0: adjoint leaving the chunk on the left
                }
                // ---- stage the next channel (its loads were issued a channel ago; every row has read this channel's strip
                // long ago), then request the channel behind it
                f2 c_dt, c_dy;
                {   // this channel's own two positions back from the strip, before the next channel takes it
                    const int ol = olane(), sto = 32 * (ol >> 4) + 2 * (ol & 15);
                    c_dt = *reinterpret_cast<const f2 *>(st + sto);
                    c_dy = *reinterpret_cast<const f2 *>(st + 2 * CHUNK + sto);
                }
                stage(c1);
                input_load(c2);
                // the row's running sums: requested here, needed behind the reductions (every lane reads: no branch)
                float *rec = rows + (cur.rw0 + k) * ROWF;
                const f4 racc = *reinterpret_cast<const f4 *>(rec + 4 * (olane() >> 4));
                const f2 rdb = *reinterpret_cast<const f2 *>(rec + 16);
                SRP_PROBE(8);
                // No branch between the state pairs and the uses of ddu / ddt: LLVM sinks the 64 scalar FMAs that build them
                // below the first branch it finds, which keeps dL/dh, B and w of BOTH pairs alive (96 registers).
                // ---- sums over the 16 states = over the rows; row r ends up with positions 2r, 2r+1 of every segment
                float e0 = ddu[0], e1 = ddu[2], e2 = ddu[4], e3 = ddu[6], q0 = ddu[1], q1 = ddu[3], q2 = ddu[5], q3 = ddu[7];
                rows_reduce4x2(e0, e1, e2, e3, q0, q1, q2, q3);
                float f0 = ddt[0], f1 = ddt[2], f2_ = ddt[4], f3 = ddt[6], g0 = ddt[1], g1 = ddt[3], g2 = ddt[5], g3 = ddt[7];
                rows_reduce4x2(f0, f1, f2_, f3, g0, g1, g2, g3);
                const int ol = olane(), ps = base + 8 * (ol & 15) + 2 * (ol >> 4);     // the two positions this lane finishes
                const float Dc = rowrec[19];
                float ddta = f0 * SS2D_LN2 + e0 * c_u.x, ddtb = g0 * SS2D_LN2 + q0 * c_u.y;
                if (p.softplus) { ddta *= c_sg.x; ddtb *= c_sg.y; }
                ddta = ps < L ? ddta : 0.0f;
                ddtb = ps + 1 < L ? ddtb : 0.0f;
                buf_store2m<VEC, IO>(make_rsrc(static_cast<IO *>(p.du) + (cur.r0 + k) * (size_t)L, rowbytes), ps, L, rev,
                                     f2{e0 * c_dt.x + c_dy.x * Dc, q0 * c_dt.y + c_dy.y * Dc});
                buf_store2m<VEC, IO>(make_rsrc(static_cast<IO *>(p.ddelta) + (cur.r0 + k) * (size_t)L, rowbytes), ps, L, rev, f2{ddta, ddtb});
                SRP_PROBE(9);
                // ---- dA (4 states of this row), dD, ddelta_bias: row sums, accumulated over the chunks in the row's LDS record
                float s0 = dAp[0].x, s1 = dAp[0].y, s2 = dAp[1].x, s3 = dAp[1].y, sD = c_dy.x * c_u.x + c_dy.y * c_u.y, sb = ddta + ddtb;
                row_sum6_to_lane15(s0, s1, s2, s3, sD, sb);
                // plain read-modify-write: every access to a row's record is ordered by the carry chain -- (chunk, k) runs behind
                // (chunk + 1, k), whose record update precedes its progress word
                if ((olane() & 15) == 15) *reinterpret_cast<f4 *>(rec + 4 * (olane() >> 4)) = racc + f4{s0, s1, s2, s3};
                {   // dD / ddelta_bias: the four rows hold four partial sums
                    const float tD = (lane_value<15>(sD) + lane_value<31>(sD)) + (lane_value<47>(sD) + lane_value<63>(sD));
                    const float tb = (lane_value<15>(sb) + lane_value<31>(sb)) + (lane_value<47>(sb) + lane_value<63>(sb));
                    if (olane() == 0) *reinterpret_cast<f2 *>(rec + 16) = rdb + f2{tD, tb};
                }
                if ((olane() & 15) == 0) *reinterpret_cast<f4 *>(cout + k * 16 + 4 * (olane() >> 4)) = f4{rleft[0].x, rleft[0].y, rleft[1].x, rleft[1].y};
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                if (olane() == 0) __hip_atomic_store(progress + wave, s_idx * 256 + (k - cur.k_lo) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                strip_read(c1.rw);      // the next channel's segment values and A row (staged / resident since the kernel's start)
            }

            // ---- this chunk's dB / dC sums: lane (row, seg) holds states 4 row + 2 pi + e at positions base + 8 seg .. + 7.
            // Through the wave's LDS tile (B / C are done with) into [matrix][state][position] order, so that a store
            // instruction covers two whole 512-byte rows instead of 64 scattered 16-byte pieces.
            {
                float *dBg = WS ? p.ws + ((size_t)blockIdx.x + cur.bg) * 32 * (size_t)L : p.dB + (size_t)cur.bg * 16 * (size_t)L;
                float *dCg = WS ? dBg + 16 * (size_t)L : p.dC + (size_t)cur.bg * 16 * (size_t)L;
                float *tl = reinterpret_cast<float *>(tile);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int mat = 0; mat < 2; ++mat)
#pragma unroll
                    for (int pi = 0; pi < 2; ++pi)
#pragma unroll
                        for (int e = 0; e < 2; ++e) {
                            f2 (&acc)[T] = mat ? accC[pi] : accB[pi];
                            float *q = tl + (mat * 16 + 4 * row + 2 * pi + e) * CHUNK + 8 * seg;
                            *reinterpret_cast<f4 *>(q) = f4{acc[0][e], acc[1][e], acc[2][e], acc[3][e]};
                            *reinterpret_cast<f4 *>(q + 4) = f4{acc[4][e], acc[5][e], acc[6][e], acc[7][e]};
                        }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (WS && !cur.atomic_out) {   // plain stores; L % 4 == 0: the four positions are inside the row together
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int i4 = i * 64 + lane, nn = i4 >> 5, pos = base + 4 * (i4 & 31);     // nn = matrix * 16 + state
                        const f4 r = *reinterpret_cast<const f4 *>(tl + 4 * i4);
                        float *dst = (nn >= 16 ? dCg + (size_t)(nn - 16) * L : dBg + (size_t)nn * L);
                        if (pos < L) {
                            if (!rev) *reinterpret_cast<f4 *>(dst + pos) = r;
                            else *reinterpret_cast<f4 *>(dst + (L - 4 - pos)) = f4{r.w, r.z, r.y, r.x};
                        }
                    }
                } else {                        // float atomics, 64 consecutive positions per instruction
                    for (int i = 0; i < 64; ++i) {
                        const int i1 = i * 64 + lane, nn = i1 >> 7, pos = base + (i1 & 127);
                        float *dst = (nn >= 16 ? dCg + (size_t)(nn - 16) * L : dBg + (size_t)nn * L);
                        if (pos < L) atomicAdd(dst + (rev ? L - 1 - pos : pos), tl[i1]);
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the next item's tile goes into the same LDS
                __builtin_amdgcn_wave_barrier();
            }
            if (!more) break;
            cur = decode(nq, nt);
            s_idx = nq * ipp + nt;
        }
    }

#ifdef SS2D_SRP_PROBE
    if (g_srp_probe && lane == 0) {
        unsigned long long *q = g_srp_probe + ((size_t)blockIdx.x * NW + wave) * (SRP_NPROBE + 1);
        for (int i = 0; i < SRP_NPROBE; ++i) q[i] = pacc[i];
        q[SRP_NPROBE] = __builtin_amdgcn_s_memtime() - tstart;
    }
#endif
    // ---- dA / dD / ddelta_bias of the workgroup's rows: sums over the batch -> global atomics
    __syncthreads();
    if (tid == 0 && progress[8]) rows[0] = __builtin_nanf("");    // a wave gave up waiting (cannot happen): make it visible
    __syncthreads();
    for (int i = tid; i < nrows * 16; i += NW * 64) {
        const int c = (int)((lo + (i >> 4)) % p.dim);
        atomicAdd(p.dA + (size_t)c * 16 + (i & 15), rows[(i >> 4) * ROWF + (i & 15)]);
    }
    if (tid < nrows) {
        const int c = (int)((lo + tid) % p.dim);
        if (p.dD) atomicAdd(p.dD + c, rows[tid * ROWF + 16]);
        if (p.dbias) atomicAdd(p.dbias + c, rows[tid * ROWF + 17]);
    }
}

template <bool VEC, typename IO, bool DIRS, bool WS, int DTR = 0>
int launch_srp(const ScanBwdParams &p, int nwg, hipStream_t s) {
    auto kern = scan_bwd_srp_kernel<VEC, IO, DIRS, WS, DTR>;
    static bool attr_set[SS2D_MAX_DEVICES] = {};   // per instantiation and device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SS2D_MAX_DEVICES) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS * 4);
        if (e != hipSuccess) return (int)e;
        attr_set[dev] = true;
    }
    ScanBwdParams pc = p;
    void *args[] = {&pc};
    return (int)hipLaunchKernel(reinterpret_cast<const void *>(kern), dim3(nwg), dim3(NW * 64), args, LDS_FLOATS * 4, s);
}
template <bool VEC, typename IO, bool DIRS>
int launch_srp_ws(const ScanBwdParams &p, int nwg, hipStream_t s) {
    return p.ws ? launch_srp<VEC, IO, DIRS, true>(p, nwg, s) : launch_srp<VEC, IO, DIRS, false>(p, nwg, s);
}

}  // namespace

#ifdef SS2D_SRP_PROBE
extern "C" int ss2d_debug_set_probe_p(void *buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_srp_probe), &buf, sizeof(buf));
}
#endif

// Does the chunk-pipelined kernel pay for this sequence?  It needs at least one pipelined round of 8 chunks, and the
// R = nchunks % 8 split chunks (8 atomic tiles each instead of one plain tile) must stay a small share of the work.
static int gcd_i(int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; }
bool ss2d_scan_bwd_srp_wanted(int nchunks, int rows_per_wg, int dpg) {
    static const int mode = [] { const char *e = getenv("SS2D_BWD_PIPE"); return e ? atoi(e) : -1; }();   // A/B switch: 0 off, 1 wherever it can run
    const int R = nchunks % NW;
    if (rows_per_wg > ROWS_MAX || nchunks < NW) return false;
    // the two-channel lookahead of the input prefetch assumes >= 2 channels in every item, i.e. >= 16 rows in every part
    // (parts are cut at multiples of rows_per_wg and of dpg)
    if (gcd_i(rows_per_wg, dpg) < 2 * NW) return false;
    if (mode == 0) return false;
    if (mode == 1) return true;
    return 4 * R <= nchunks;
}

// The same (rows_per_wg, nwg, p.ws) decisions as ss2d_scan_bwd_sr_launch, which calls this; the dB / dC reduction kernel
// that follows is shared.
int ss2d_scan_bwd_srp_launch(int dtype, ScanBwdParams p, bool vec4, bool dirs, int nwg, hipStream_t s) {
    p.split = p.nchunks % NW;
    const bool h = dtype == SS2D_DTYPE_BF16;
    if (p.dtw) {
        if (p.R == 6) return h ? launch_srp<true, bf16, true, true, 6>(p, nwg, s) : launch_srp<true, float, true, true, 6>(p, nwg, s);
        return h ? launch_srp<true, bf16, true, true, 12>(p, nwg, s) : launch_srp<true, float, true, true, 12>(p, nwg, s);
    }
    if (h) {
        if (dirs) return vec4 ? launch_srp_ws<true, bf16, true>(p, nwg, s) : launch_srp_ws<false, bf16, true>(p, nwg, s);
        return vec4 ? launch_srp_ws<true, bf16, false>(p, nwg, s) : launch_srp_ws<false, bf16, false>(p, nwg, s);
    }
    if (dirs) return vec4 ? launch_srp_ws<true, float, true>(p, nwg, s) : launch_srp_ws<false, float, true>(p, nwg, s);
    return vec4 ? launch_srp_ws<true, float, false>(p, nwg, s) : launch_srp_ws<false, float, false>(p, nwg, s);
}
