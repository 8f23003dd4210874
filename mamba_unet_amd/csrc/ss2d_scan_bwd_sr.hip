// Selective scan backward, "state-row" decomposition, for gfx950 -- the fast path of
// ss2d_selective_scan_{,dir_}bwd for dstate == 16 without z (the only form Mamba-UNet issues,
// /root/reference/code/networks/mamba_sys.py:436-442; forward semantics :62-144).
//
// A wavefront works on ONE (batch, channel) sequence at a time: its 4 DPP rows own 4 states each (2 packed pairs),
// the 16 lanes of a row own 8 consecutive positions each, so a wave covers a 128-position chunk x 16 states, and
// it walks its channels one after the other inside a chunk, the chunks from the end of the sequence to the start.
//   * dB / dC are sums over the channels of a group: with the states on the rows and the channels in time, every
//     lane simply keeps adding into 64 registers (an FMA instead of a multiply: free); once per chunk the 8 waves
//     of the workgroup put their registers into LDS, add them up and write the workgroup's partial tile to the
//     workspace with plain stores (a second kernel adds the partial tiles of a (batch, group)); without a workspace
//     ONE global float atomic per element.
//     (The channel-row layout of ss2d_scan_bwd.hip needs a cross-row butterfly and an LDS read-modify-write per
//     value instead: a quarter of its instructions.)
//   * du / ddelta are sums over the 16 states = over the 4 rows: one v_permlane swap butterfly per channel-chunk
//     for 16 values, after which lane (row r, segment s) holds positions 8s+2r, 8s+2r+1 -- exactly the two positions
//     it loaded, ran the softplus for and staged into the wave's private LDS strip at the start.
//   * B / C of the chunk are staged once per workgroup in LDS (the tile aliases the flush area, which is idle then).
//   * Work is dealt in (batch, channel) rows, not in (batch, group) units: every one of the 256 CUs gets the same
//     number of rows (72 at config 2); a workgroup whose rows straddle a group boundary runs the parts one after
//     the other.  One 512-thread workgroup per CU (LDS-bound on purpose), 2 waves per SIMD.  Inside a chunk the
//     channels are dealt to the waves one at a time (LDS ticket counter), not split evenly up front.
// Where the time goes is measured with the s_memtime probes of -DSS2D_SR_PROBE (make variant, tools/probe_bwd.py).
#include <cstdlib>
#include "ss2d_scan.h"
#include "../../include/ss2d.h"

namespace {

constexpr int T = 8;
constexpr int CHUNK = 128;
#ifndef SS2D_SR_NW
#define SS2D_SR_NW 8
#endif
constexpr int NW = SS2D_SR_NW;              // waves per workgroup (8: one workgroup per CU; 4: two)
constexpr int WG_PER_CU = 8 / NW;
constexpr int NM = 512 / (NW * 64);         // B/C tile: matrices staged per thread
constexpr int NE = 1024 / (NW * 64);        // flush: float4 groups per thread
constexpr int CW_MAX = NW == 8 ? 12 : 10;                  // channels per wave and part
constexpr int REC = 48;                     // per (wave, channel) record in LDS: 16 adjoint carries | 16 dA | 4 rows x {dD, dbias, -, -}
constexpr int SLOT = 2 * 16 * CHUNK;        // floats per wave in the flush area: [dB|dC][state][position]
constexpr int STG = 3 * CHUNK;              // per wave: softplus(delta), delta*u, dout of the current channel-chunk
constexpr int LDS_FLOATS = NW * SLOT + NW * STG + NW * CW_MAX * REC + 4;   // + the channel ticket counter
static_assert(NW == 8 || NW == 4, "tile staging and flush are laid out for 8 or 4 waves");
static_assert(LDS_FLOATS * 4 * WG_PER_CU <= 160 * 1024, "WG_PER_CU workgroups per CU");
static_assert(CHUNK == SS2D_SCAN_CHUNK, "checkpoint spacing is part of the ABI");

// WS: dB/dC partial tiles go to the workspace with plain stores (slot = workgroup + (batch, group) index: unique,
// and consecutive for the workgroups that share a (batch, group)); otherwise float atomics on dB/dC.
#ifdef SS2D_SR_PROBE
// timing probes (debug builds only: tools/probe_bwd.py): per wave, s_memtime deltas summed per phase
__device__ unsigned long long *g_sr_probe;
#define SR_NPROBE 16
#define SR_PROBE(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                         pacc[i] += (unsigned)(t_ - tprev); tprev = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define SR_PROBE(i) do { } while (0)
#endif

// DTR > 0: the fused-delta form (ss2d_selective_scan_core_bwd): p.delta points at the rank-DTR dts rows of the x_proj output
// and delta[c, l] = sum_r dtw[c, r] * dts[r, l] is recomputed for the lane's two positions (2*DTR FMAs per channel-chunk);
// the dts values of a chunk stay in 2*DTR registers, the weight row of a channel arrives through scalar loads.
// PACK: sequences of at most 64 positions (one chunk: the 7 x 7 / 8 x 8 stages).  A wave works on TWO channels at a time: lanes
// 0-7 of every row hold the 64 positions of channel 2k, lanes 8-15 those of channel 2k + 1.  The row scans are cut between
// lanes 7 and 8 (the aggregate decay of the lane on the far side is zeroed going in, its incoming value coming out), the
// per-channel row sums stop at 8 lanes, and before the flush lane s adds lane s + 8's dB / dC sums (the same positions).
// With one channel per wave 9 of 16 lanes idle at L = 56 and a call costs what it costs at L = 128.
template <bool VEC, typename IO, bool DIRS, bool WS, int DTR = 0, bool PACK = false>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void scan_bwd_sr_kernel(const ScanBwdParams p) {
    static_assert(!PACK || (VEC && DTR == 0), "the packed form: 4-element rows, materialised delta");
    extern __shared__ __align__(16) float smem[];
#ifdef SS2D_SR_PROBE
    unsigned pacc[SR_NPROBE] = {};
    unsigned long long tprev = __builtin_amdgcn_s_memtime();
    const unsigned long long tstart = tprev;
#endif
    float *slots = smem;                                        // [NW][SLOT]; slot 0 doubles as the B/C tile
    f4 *tile = reinterpret_cast<f4 *>(smem);                   // [B|C][pair][item pair][seg] f4, as in ss2d_scan_fwd.hip
    const int tid = threadIdx.x, lane = tid & 63, row = lane >> 4, seg = lane & 15;
    const int half = PACK ? seg >> 3 : 0, sg = PACK ? seg & 7 : seg;     // PACK: which of the wave's two channels, segment inside it
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *st = smem + NW * SLOT + wave * STG;
    float *recs = smem + NW * SLOT + NW * STG;                                  // one record per channel of the part
    int *ticket = reinterpret_cast<int *>(smem + NW * SLOT + NW * STG + NW * CW_MAX * REC);
    float *slot = slots + wave * SLOT;

    const int L = p.L, nchunks = p.nchunks;
    const long long total = (long long)p.batch * p.dim;
    const long long lo = (long long)blockIdx.x * p.rows_per_wg;
    const long long hi = lo + p.rows_per_wg < total ? lo + p.rows_per_wg : total;

    for (long long part_lo = lo; part_lo < hi;) {
        // ---- one part = the rows of this workgroup that belong to one (batch, group)
        const int bg = (int)(part_lo / p.dpg);
        const int b = bg / p.G, g = bg - b * p.G;
        long long part_hi = (long long)(bg + 1) * p.dpg;
        if (part_hi > hi) part_hi = hi;
        const int n = (int)(part_hi - part_lo);
        // The n <= NW * CW_MAX channels of the part are dealt to the waves one at a time inside every chunk (wave w starts
        // with channel w, then takes tickets from an LDS counter): the two waves of a SIMD do not advance at the same
        // rate, and with a fixed split the barrier in front of the flush waited 13 % of the kernel time for the slower one.
        const int nu_ = PACK ? (n + 1) >> 1 : n;                   // work units of the part: channels, or pairs of channels
        const int kfirst = wave < nu_ ? wave : 0;                  // a wave without a channel still prefetches a valid row
        const size_t r0 = (size_t)part_lo;                          // (batch, channel) row of the part's first channel
        const int cin0 = (int)(r0 - (size_t)bg * p.dpg);
        const int cdim0 = g * p.dpg + cin0;
        const bool rev = DIRS && ((p.rev_mask >> g) & 1u);
        const int ug = DIRS ? g / (p.G / p.Gu) : g;
        const size_t bcs = DIRS ? (size_t)p.bc_stride : 16 * (size_t)L;
        const __amdgpu_buffer_rsrc_t rsB = make_rsrc(static_cast<const IO *>(p.B) + (size_t)bg * bcs, 16u * L * sizeof(IO));
        const __amdgpu_buffer_rsrc_t rsC = make_rsrc(static_cast<const IO *>(p.C) + (size_t)bg * bcs, 16u * L * sizeof(IO));
        float *__restrict__ dBg = WS ? p.ws + ((size_t)blockIdx.x + bg) * 32 * (size_t)L : p.dB + (size_t)bg * 16 * (size_t)L;
        float *__restrict__ dCg = WS ? dBg + 16 * (size_t)L : p.dC + (size_t)bg * 16 * (size_t)L;
        const size_t urow0 = DIRS ? ((size_t)(b * p.Gu + ug) * p.dpg + cin0) : r0;   // row of u / dout
        // bias / D come through the vector memory path like A (same address in every lane): a scalar load would share the
        // lgkm counter with the LDS traffic and, returning out of order, force a full drain at the next LDS use
        const float *biasp = p.bias ? p.bias : p.A, *Dp = p.D ? p.D : p.A;   // always a valid address
        const bool has_bias = p.bias != nullptr, has_D = p.D != nullptr;
        const unsigned rowbytes = (unsigned)L * sizeof(IO);

        for (int i = tid; i < (PACK ? 2 * nu_ : n) * REC; i += NW * 64) recs[i] = 0.0f;   // visible to all waves behind the first barrier of the chunk loop

        // Everything that comes from memory is requested one step ahead and never under a branch (a load under a
        // branch makes the compiler wait for every outstanding load at the join):
        //   * u / delta / dout of channel k+1 at the top of channel k, its A row / checkpoint / bias / D between the two
        //     state pairs of channel k; behind the last channel of a chunk the first channel of the NEXT chunk;
        //   * the B / C tile of the next chunk before the flush of this one.
        // Rows are addressed through buffer resources: a position outside the row reads 0 and drops the store.
        // tile item of this thread: waves 0-3 stage B, waves 4-7 stage C (wave-uniform, so the resource stays scalar);
        // (pair, 4 positions) from the lane.  `opaque_tid()` keeps these cheap values from being hoisted out of the chunk
        // loop and then spilled around the channel loop, which needs every register.
        auto opaque_tid = [&]() { int t_ = tid; asm volatile("" : "+v"(t_)); return t_; };
        const int tm = NM == 1 ? wave >> 2 : 0;
        auto tile_load = [&](int chunk_, f4 (&lo4)[NM], f4 (&hi4)[NM]) {
            const int tit = opaque_tid() & 255, tnp = tit >> 5, tq4 = tit & 31;
            const int pos = chunk_ * CHUNK + tq4 * 4;
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const __amdgpu_buffer_rsrc_t rs = (tm + m) ? rsC : rsB;
                lo4[m] = buf_load4m<VEC, IO>(rs, (unsigned)(2 * tnp) * L, pos, L, rev);
                hi4[m] = buf_load4m<VEC, IO>(rs, (unsigned)(2 * tnp + 1) * L, pos, L, rev);
            }
        };
        f2 nu, nd = f2{0.0f, 0.0f}, ny;
        f4 nAv, nck;
        float nbias, nD;
        constexpr int NR = DTR > 0 ? DTR : 1;
        float nw[NR];                                  // DTR: dt_proj weight row of the next channel (wave-uniform: scalar registers)
        f2 dtsv[NR];                                   // DTR: dts[r] at this lane's two positions of the current chunk
        typedef const float __attribute__((address_space(4))) *cfp;
        // One descriptor per tensor and part (base = the part's first row, num_records = the part's rows) and a scalar byte
        // offset per channel, so that the channel loop builds no descriptors (seven per channel-chunk were 0.18 scalar
        // instructions per vector instruction, with two waves per SIMD to hide them behind).  The scalar offset is part
        // of the hardware's range check, so a descriptor cannot be one row long: positions past the row end are sent out
        // of range by the vector offset instead (buf_load2s / buf_store2s).
        constexpr int NR_ = DTR > 0 ? DTR : 1;
        const unsigned partbytes = (unsigned)n * rowbytes;
        const __amdgpu_buffer_rsrc_t rsU = make_rsrc(static_cast<const IO *>(p.u) + urow0 * (size_t)L, partbytes);
        const __amdgpu_buffer_rsrc_t rsY = make_rsrc(static_cast<const IO *>(p.dout) + urow0 * (size_t)L, partbytes);
        const __amdgpu_buffer_rsrc_t rsDl = make_rsrc(static_cast<const IO *>(p.delta) + (DTR > 0 ? (size_t)bg * bcs : r0 * (size_t)L),
                                                      DTR > 0 ? (unsigned)NR_ * rowbytes : partbytes);
        const __amdgpu_buffer_rsrc_t rsDu = make_rsrc(static_cast<IO *>(p.du) + r0 * (size_t)L, partbytes);
        const __amdgpu_buffer_rsrc_t rsDd = make_rsrc(static_cast<IO *>(p.ddelta) + r0 * (size_t)L, partbytes);
        const __amdgpu_buffer_rsrc_t rsA = make_rsrc(p.A + (size_t)cdim0 * 16, 64u * n);
        const __amdgpu_buffer_rsrc_t rsCk = make_rsrc(p.ckpt + r0 * (size_t)nchunks * 16, 64u * n * nchunks);
        const __amdgpu_buffer_rsrc_t rsBi = make_rsrc(biasp + cdim0, 4u * n);
        const __amdgpu_buffer_rsrc_t rsDp = make_rsrc(Dp + cdim0, 4u * n);
        // PACK: unit k_ = channels 2 k_ (+ half); the lane's row is chosen by a vector offset on top of the scalar one, and a
        // channel past the part (odd n) is sent out of range: it reads zeros and drops its stores
        const bool odd_out = PACK && (n & 1) && half;              // this half has no channel in the part's last unit
        auto input_load = [&](int k_, int chunk_) {
            const int ps_ = chunk_ * CHUNK + 8 * sg + 2 * row;
            const int so_ = (PACK ? 2 * k_ : k_) * (int)rowbytes;
            const int ve_ = PACK ? ((odd_out && k_ == nu_ - 1) ? SS2D_BUF_OOR : half * (int)rowbytes) : 0;
            nu = buf_load2s<VEC, IO>(rsU, ps_, L, rev, so_, ve_);
            if (DTR == 0) nd = buf_load2s<VEC, IO>(rsDl, ps_, L, rev, so_, ve_);
            ny = buf_load2s<VEC, IO>(rsY, ps_, L, rev, so_, ve_);
            if (DTR > 0) {
                const cfp wp = (cfp)(uintptr_t)(p.dtw + (size_t)(cdim0 + k_) * DTR);
#pragma unroll
                for (int r = 0; r < NR; ++r) nw[r] = wp[r];
            }
        };
        auto dts_load = [&](int chunk_) {             // DTR: the chunk's dts values of this lane (all channels of the part share them)
            if (DTR > 0) {
                const int ps_ = chunk_ * CHUNK + 8 * seg + 2 * row;
#pragma unroll
                for (int r = 0; r < NR; ++r)
                    dtsv[r] = buf_load2s<VEC, IO>(rsDl, ps_, L, rev, r * (int)rowbytes);
            }
        };
        auto param_load = [&](int k_, int chunk_) {
            // (PACK: the second channel of an odd part's last unit lies past num_records and reads zeros)
            const int kc_ = PACK ? 2 * k_ : k_;
            nAv = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsA, 16 * row + 64 * half, 64 * kc_, 0));
            nck = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rsCk, 16 * row + 64 * half * nchunks, 64 * (kc_ * nchunks + chunk_), 0));
            nbias = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsBi, 4 * half, 4 * kc_, 0));
            nD = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsDp, 4 * half, 4 * kc_, 0));
        };
        f4 tlo[NM], thi[NM];
        tile_load(nchunks - 1, tlo, thi);
        dts_load(nchunks - 1);
        input_load(kfirst, nchunks - 1);
        param_load(kfirst, nchunks - 1);

        for (int chunk = nchunks - 1; chunk >= 0; --chunk) {
            const int base = chunk * CHUNK;
            const int chunk_next = chunk > 0 ? chunk - 1 : 0;
            __syncthreads();   // the previous flush has read the slots: slot 0 may become the tile again
            {
                const int tit = opaque_tid() & 255, tnp = tit >> 5, tq4 = tit & 31, tj0 = (tq4 & 1) * 2, tsg = tq4 >> 1;
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    f4 *td = tile + (tm + m) * 512 + (tnp * 4 + tj0) * 16 + tsg;
                    td[0] = f4{tlo[m].x, thi[m].x, tlo[m].y, thi[m].y};
                    td[16] = f4{tlo[m].z, thi[m].z, tlo[m].w, thi[m].w};
                }
                if (tit == 0 && tm == 0) *ticket = NW;      // channels 0 .. NW-1 are taken: wave w starts with channel w
            }
            __syncthreads();
            SR_PROBE(9);

            f2 accB[2][T], accC[2][T];
#pragma unroll
            for (int pi = 0; pi < 2; ++pi)
#pragma unroll
                for (int t = 0; t < T; ++t) { accB[pi][t] = f2{0.0f, 0.0f}; accC[pi][t] = f2{0.0f, 0.0f}; }

            const int ps = base + 8 * sg + 2 * row;              // the two positions this lane stages and finishes
            // staging strip: [array][item pair p][segment] float2 -- the writer (row = p) and the readers (4 pairs of one
            // segment, one ds_read_b64 each) both touch 32 distinct banks per instruction
            const int sto = 32 * row + 2 * seg, sta = 2 * seg;
            // tickets are drawn two channels ahead so that the LDS atomic's return is never waited for
            int tk = 0;
            if (lane == 0) tk = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            int kt = __builtin_amdgcn_readfirstlane(tk);               // this wave's next channel, if there is one
            for (int k = wave; k < nu_;) {
                SR_PROBE(0);
                const bool lastk = kt >= nu_;
                const bool live = !(odd_out && k == nu_ - 1);          // PACK: this lane's channel exists
                const bool ok0 = ps < L && live, ok1 = ps + 1 < L && live;
                const int kn = lastk ? kfirst : kt, chn = lastk ? chunk_next : chunk;
                if (lane == 0) tk = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                const f2 cu = nu, cy = ny;
                f2 cd = nd;
                if (DTR > 0) {
#pragma unroll
                    for (int r = 0; r < NR; ++r) cd = fma2(splat2(nw[r]), dtsv[r], cd);
                }
                const f4 Av = nAv, ck = nck;
                const float bias = has_bias ? nbias : 0.0f, Dc = has_D ? nD : 0.0f;
                input_load(kn, chn);
                // ---- this lane's two positions: softplus, delta*u; shared with the other rows through LDS
                float x0 = cd.x + bias, x1 = cd.y + bias, sg0 = 1.0f, sg1 = 1.0f;
                if (p.softplus) { x0 = softplus_sig<true>(x0, sg0); x1 = softplus_sig<true>(x1, sg1); }
                const float dt0 = ok0 ? x0 : 0.0f, dt1 = ok1 ? x1 : 0.0f;   // positions past L are the identity
                const float u0 = cu.x, u1 = cu.y, dy0 = cy.x, dy1 = cy.y;   // zero past L (buffer range check)
                *reinterpret_cast<f2 *>(st + sto) = f2{dt0, dt1};
                *reinterpret_cast<f2 *>(st + CHUNK + sto) = f2{dt0 * u0, dt1 * u1};
                *reinterpret_cast<f2 *>(st + 2 * CHUNK + sto) = f2{dy0, dy1};
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                f2 dtp[4], dup[4], dyp[4];                            // pairs of neighbouring positions (see pk_*_sel)
                {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        dtp[j] = *reinterpret_cast<const f2 *>(st + sta + 32 * j);
                        dup[j] = *reinterpret_cast<const f2 *>(st + CHUNK + sta + 32 * j);
                        dyp[j] = *reinterpret_cast<const f2 *>(st + 2 * CHUNK + sta + 32 * j);
                    }
                }
                float *rec = recs + (PACK ? 2 * k + half : k) * REC + 4 * row;     // + 0: carries, + 16: dA, + 32: dD / dbias of this row
                const f4 rcv = *reinterpret_cast<const f4 *>(rec);       // adjoint entering from the right
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                float S = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) S += dtp[j].x + dtp[j].y;
                SR_PROBE(1);

                float ddt[T], ddu[T];                                   // sums over this row's 4 states
                f2 dAp[2], rleft[2];
#pragma unroll
                for (int t = 0; t < T; ++t) { ddt[t] = 0.0f; ddu[t] = 0.0f; }

#pragma unroll
                for (int pi = 0; pi < 2; ++pi) {
                    const int np = 2 * row + pi;
                    const f4 *__restrict__ tB = tile + (np * 4) * 16 + sg, *__restrict__ tC = tB + 512;
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
                        f2 hl = f2{sg == 0 ? hin.x : 0.0f, sg == 0 ? hin.y : 0.0f};
#pragma unroll
                        for (int t = 0; t < T; ++t) hl = fma2(a[t], hl, bb[t]);
                        float hx = hl.x, hy = hl.y, ax = at.x, ay = at.y;
                        if (PACK) { ax = seg == 8 ? 0.0f : ax; ay = seg == 8 ? 0.0f : ay; }      // nothing crosses from lane 7 into lane 8
#pragma unroll
                        for (int j = 0; j < 4; ++j) Cq[j] = tC[j * 16];   // requested here: they arrive behind the row scan
                        row_scan_fwd_pair(hx, hy, ax, ay);
                        hstart.x = dpp_mov<DPP_ROW_SHR(1), 0xf, false>(hin.x, hx);
                        hstart.y = dpp_mov<DPP_ROW_SHR(1), 0xf, false>(hin.y, hy);
                        if (PACK) { hstart.x = seg == 8 ? hin.x : hstart.x; hstart.y = seg == 8 ? hin.y : hstart.y; }
                        f2 hh = hstart;
#pragma unroll
                        for (int t = 0; t < T; ++t) { hh = fma2(a[t], hh, bb[t]); h[t] = hh; }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    SR_PROBE(2 + 3 * pi);
                    // ---- adjoint r_l = a_l * (dy_l * C_l + r_{l+1}): lane-local suffix pass, row scan, true pass
                    f2 rr;
                    f4 Bn, Cn;
                    {
                        f2 rl = f2{sg == (PACK ? 7 : 15) ? rcar.x : 0.0f, sg == (PACK ? 7 : 15) ? rcar.y : 0.0f};
#pragma unroll
                        for (int j = 3; j >= 0; --j) {
                            rl = a[2 * j + 1] * pk_fma_hi(dyp[j], f2{Cq[j].z, Cq[j].w}, rl);
                            rl = a[2 * j] * pk_fma_lo(dyp[j], f2{Cq[j].x, Cq[j].y}, rl);
                        }
                        float rx = rl.x, ry = rl.y, ax = at.x, ay = at.y;
                        if (PACK) { ax = seg == 7 ? 0.0f : ax; ay = seg == 7 ? 0.0f : ay; }      // nothing crosses from lane 8 into lane 7
                        Bn = tB[3 * 16]; Cn = tC[3 * 16];                 // first operands of the true pass, behind the row scan
                        row_scan_rev_pair(rx, ry, ax, ay);
                        rr.x = dpp_mov<DPP_ROW_SHL(1), 0xf, false>(rcar.x, rx);
                        rr.y = dpp_mov<DPP_ROW_SHL(1), 0xf, false>(rcar.y, ry);
                        if (PACK) { rr.x = seg == 7 ? rcar.x : rr.x; rr.y = seg == 7 ? rcar.y : rr.y; }
                    }
                    if (pi == 0) param_load(kn, chn);   // in flight during pair 1
                    __builtin_amdgcn_sched_barrier(0);
                    SR_PROBE(3 + 3 * pi);
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
                    SR_PROBE(4 + 3 * pi);
                    dAp[pi] = dAacc;
                    rleft[pi] = rr;                                           // lane 0: adjoint leaving the chunk on the left
                }
                const int kt2 = __builtin_amdgcn_readfirstlane(tk);   // drawn at the top: has long arrived, few LDS operations behind it
                // No branch between the state pairs and the uses of ddu / ddt: LLVM sinks the 64 scalar FMAs that build them
                // below the first branch it finds, which keeps dL/dh, B and w of BOTH pairs alive (96 registers).
                // ---- sums over the 16 states = over the rows; row r ends up with positions 2r, 2r+1 of every segment
                float e0 = ddu[0], e1 = ddu[2], e2 = ddu[4], e3 = ddu[6], o0 = ddu[1], o1 = ddu[3], o2 = ddu[5], o3 = ddu[7];
                rows_reduce4x2(e0, e1, e2, e3, o0, o1, o2, o3);
                float f0 = ddt[0], f1 = ddt[2], f2_ = ddt[4], f3 = ddt[6], g0 = ddt[1], g1 = ddt[3], g2 = ddt[5], g3 = ddt[7];
                rows_reduce4x2(f0, f1, f2_, f3, g0, g1, g2, g3);
                SR_PROBE(12);
                const f2 dtq = *reinterpret_cast<const f2 *>(st + sto), dyq = *reinterpret_cast<const f2 *>(st + 2 * CHUNK + sto);   // re-read: 4 registers
                float ddta = f0 * SS2D_LN2 + e0 * u0, ddtb = g0 * SS2D_LN2 + o0 * u1;
                if (p.softplus) { ddta *= sg0; ddtb *= sg1; }
                ddta = ok0 ? ddta : 0.0f;
                ddtb = ok1 ? ddtb : 0.0f;
                const int sso = (PACK ? 2 * k : k) * (int)rowbytes, sve = PACK ? (live ? half * (int)rowbytes : SS2D_BUF_OOR) : 0;
                buf_store2s<VEC, IO>(rsDu, ps, L, rev, f2{e0 * dtq.x + dyq.x * Dc, o0 * dtq.y + dyq.y * Dc}, sso, sve);
                buf_store2s<VEC, IO>(rsDd, ps, L, rev, f2{ddta, ddtb}, sso, sve);
                SR_PROBE(13);
                // ---- dA (4 states of this row), dD, ddelta_bias: row sums, accumulated over the chunks in the wave's LDS record
                float s0 = dAp[0].x, s1 = dAp[0].y, s2 = dAp[1].x, s3 = dAp[1].y, sD = dyq.x * u0 + dyq.y * u1, sb = ddta + ddtb;
                if (PACK) row_sum6_to_lanes7_15(s0, s1, s2, s3, sD, sb); else row_sum6_to_lane15(s0, s1, s2, s3, sD, sb);
                if (sg == (PACK ? 7 : 15)) {   // plain read-modify-write: the record is private to the wave
                    f4 *qa = reinterpret_cast<f4 *>(rec + 16);
                    f2 *qd = reinterpret_cast<f2 *>(rec + 32);
                    const f4 va = *qa;
                    const f2 vd = *qd;
                    *qa = va + f4{s0, s1, s2, s3};
                    *qd = vd + f2{sD, sb};
                }
                if (sg == 0) *reinterpret_cast<f4 *>(rec) = f4{rleft[0].x, rleft[0].y, rleft[1].x, rleft[1].y};
                k = kt;
                kt = kt2;
            }
            tile_load(chunk_next, tlo, thi);   // arrives during the flush
            dts_load(chunk_next);

            SR_PROBE(11);
            __syncthreads();   // every wave is done with the tile
            SR_PROBE(14);
            if (PACK) {   // lane s takes over lane s + 8's sums: the same 8 positions, the wave's other channel
#pragma unroll
                for (int pi = 0; pi < 2; ++pi)
#pragma unroll
                    for (int t = 0; t < T; ++t) {
                        accB[pi][t].x += dpp_mov<DPP_ROW_SHL(8), 0xf, true>(0.0f, accB[pi][t].x);
                        accB[pi][t].y += dpp_mov<DPP_ROW_SHL(8), 0xf, true>(0.0f, accB[pi][t].y);
                        accC[pi][t].x += dpp_mov<DPP_ROW_SHL(8), 0xf, true>(0.0f, accC[pi][t].x);
                        accC[pi][t].y += dpp_mov<DPP_ROW_SHL(8), 0xf, true>(0.0f, accC[pi][t].y);
                    }
            }
            // ---- this wave's dB / dC sums -> its slot, [matrix][state][position], halves swizzled against bank conflicts
            {
                const int l2 = opaque_tid() & 63, row2 = l2 >> 4, seg2 = l2 & 15, sw = ((seg2 >> 2) & 1) << 2;
                float *s0 = slot + (((4 * row2) * CHUNK + 8 * seg2) ^ sw), *s1 = slot + (((4 * row2) * CHUNK + 8 * seg2 + 4) ^ sw);
#pragma unroll
                for (int pi = 0; pi < 2; ++pi)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        const int o = (2 * pi + s) * CHUNK;
                        *reinterpret_cast<f4 *>(s0 + o) = f4{accB[pi][0][s], accB[pi][1][s], accB[pi][2][s], accB[pi][3][s]};
                        *reinterpret_cast<f4 *>(s1 + o) = f4{accB[pi][4][s], accB[pi][5][s], accB[pi][6][s], accB[pi][7][s]};
                        *reinterpret_cast<f4 *>(s0 + 2048 + o) = f4{accC[pi][0][s], accC[pi][1][s], accC[pi][2][s], accC[pi][3][s]};
                        *reinterpret_cast<f4 *>(s1 + 2048 + o) = f4{accC[pi][4][s], accC[pi][5][s], accC[pi][6][s], accC[pi][7][s]};
                    }
            }
            SR_PROBE(15);
            __syncthreads();
            SR_PROBE(8);
            // ---- add the NW slots: every thread owns 2 x 4 consecutive positions; all 16 reads are issued before the adds
            {
                const int t2 = opaque_tid();
                f4 v[NE][NW];
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    const int i4 = e * (NW * 64) + t2, phys = i4 ^ ((i4 >> 3) & 1);
#pragma unroll
                    for (int w = 0; w < NW; ++w) v[e][w] = *reinterpret_cast<const f4 *>(slots + w * SLOT + 4 * phys);
                }
#pragma unroll
                for (int e = 0; e < NE; ++e) {
                    f4 r = (v[e][0] + v[e][1]) + (v[e][2] + v[e][3]);
                    if (NW == 8) r = r + ((v[e][4 % NW] + v[e][5 % NW]) + (v[e][6 % NW] + v[e][7 % NW]));
                    const int idx = 4 * (e * (NW * 64) + t2);
                    const int mat = idx >> 11, nn = (idx >> 7) & 15, pos = base + (idx & 127);
                    float *dst = (mat ? dCg : dBg) + (size_t)nn * L;
                    if (WS) {   // L % 4 == 0: the four positions are inside the row together
                        if (pos < L) {
                            if (!rev) *reinterpret_cast<f4 *>(dst + pos) = r;
                            else *reinterpret_cast<f4 *>(dst + (L - 4 - pos)) = f4{r.w, r.z, r.y, r.x};
                        }
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (pos + q < L) atomicAdd(dst + (rev ? L - 1 - pos - q : pos + q), r[q]);
                    }
                }
            }
            SR_PROBE(10);
        }
        // ---- dA / dD / ddelta_bias of the part's channels: sums over the batch -> global atomics
        __syncthreads();   // the records of the last chunk are complete (the flush has barriers, but not behind the last record write)
        for (int i = tid; i < n * 16; i += NW * 64) atomicAdd(p.dA + (size_t)cdim0 * 16 + i, recs[(i >> 4) * REC + 16 + (i & 15)]);
        if (tid < n) {
            const float *q = recs + tid * REC + 32;
            if (p.dD) atomicAdd(p.dD + cdim0 + tid, (q[0] + q[4]) + (q[8] + q[12]));
            if (p.dbias) atomicAdd(p.dbias + cdim0 + tid, (q[1] + q[5]) + (q[9] + q[13]));
        }
        __syncthreads();   // before the next part zeroes the records
        part_lo = part_hi;
    }
#ifdef SS2D_SR_PROBE
    if (g_sr_probe && lane == 0) {
        unsigned long long *q = g_sr_probe + ((size_t)blockIdx.x * NW + wave) * (SR_NPROBE + 1);
        for (int i = 0; i < SR_NPROBE; ++i) q[i] = pacc[i];
        q[SR_NPROBE] = __builtin_amdgcn_s_memtime() - tstart;
    }
#endif
}

// dB/dC[bg] = sum of the partial tiles of the workgroups whose rows touch (batch, group) bg
__global__ __launch_bounds__(256) void scan_bwd_sr_reduce_kernel(const float *__restrict__ ws, float *__restrict__ dB, float *__restrict__ dC,
                                                                 int L, int dpg, int rows_per_wg, int n4) {
    const int nblk = (n4 + 255) / 256, bg = blockIdx.x / nblk;
    const int i = (blockIdx.x - bg * nblk) * 256 + threadIdx.x;   // float4 index inside the (dB | dC) tile of 32 * L floats
    if (i >= n4) return;
    const long long r_lo = (long long)bg * dpg, r_hi = r_lo + dpg - 1;
    const int w0 = (int)(r_lo / rows_per_wg), w1 = (int)(r_hi / rows_per_wg);
    f4 acc = f4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int w = w0; w <= w1; ++w) acc += *reinterpret_cast<const f4 *>(ws + ((size_t)w + bg) * 32 * (size_t)L + 4 * (size_t)i);
    const size_t o = 4 * (size_t)i, half = 16 * (size_t)L;
    float *dst = o < half ? dB + (size_t)bg * half + o : dC + (size_t)bg * half + (o - half);
    *reinterpret_cast<f4 *>(dst) = acc;
}

// The same sum written as rows [R, R + 32) of every (batch, group) block of the x_proj output's gradient, in its storage
// type: the B | C rows of a block are one contiguous run of 32 * L elements there, so the fp32 dB / dC tensors and the pass
// that packed them (with d_dts) into that gradient are not needed.
template <typename IO>
__global__ __launch_bounds__(256) void scan_bwd_sr_reduce_packed_kernel(const float *__restrict__ ws, IO *__restrict__ dxdbl, long long bcs, int R,
                                                                        int L, int dpg, int rows_per_wg, int n4) {
    const int nblk = (n4 + 255) / 256, bg = blockIdx.x / nblk;
    const int i = (blockIdx.x - bg * nblk) * 256 + threadIdx.x;
    if (i >= n4) return;
    const long long r_lo = (long long)bg * dpg, r_hi = r_lo + dpg - 1;
    const int w0 = (int)(r_lo / rows_per_wg), w1 = (int)(r_hi / rows_per_wg);
    f4 acc = f4{0.0f, 0.0f, 0.0f, 0.0f};
    for (int w = w0; w <= w1; ++w) acc += *reinterpret_cast<const f4 *>(ws + ((size_t)w + bg) * 32 * (size_t)L + 4 * (size_t)i);
    st4(dxdbl + (size_t)bg * bcs + (size_t)R * L + 4 * (size_t)i, acc);
}

template <bool VEC, typename IO, bool DIRS, bool WS, int DTR = 0, bool PACK = false>
int launch_sr(const ScanBwdParams &p, int nwg, hipStream_t s) {
    auto kern = scan_bwd_sr_kernel<VEC, IO, DIRS, WS, DTR, PACK>;
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

int num_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        n = v;
    }
    return n;
}

}  // namespace

namespace {
// rows per workgroup: the same number for every CU, several resident rounds when there are more than
// NW * CW_MAX rows per CU
long long sr_rows_per_wg(long long total) {
    const int ncu = num_cus() * WG_PER_CU;
    long long per = (total + ncu - 1) / ncu;
    for (int m = 2; per > NW * CW_MAX; ++m) per = (total + (long long)ncu * m - 1) / ((long long)ncu * m);
    return per < NW ? NW : per;
}
template <bool VEC, typename IO, bool DIRS>
int launch_sr_ws(const ScanBwdParams &p, int nwg, hipStream_t s) {
    return p.ws ? launch_sr<VEC, IO, DIRS, true>(p, nwg, s) : launch_sr<VEC, IO, DIRS, false>(p, nwg, s);
}
}  // namespace

// ---- the workspace of ss2d_selective_scan_dir_bwd_static: a fixed array in the code object, for callers that cannot
// allocate.  Every GPU the library runs on holds its own copy at its own address; the name below is only a host-side
// handle for hipGetSymbolAddress, never a device pointer.
static __device__ __align__(256) float scratch[SS2D_STATIC_WS_BYTES / sizeof(float)];

#ifdef SS2D_SR_PROBE
extern "C" int ss2d_debug_set_probe(void *buf) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_sr_probe), &buf, sizeof(buf));
}
#endif

int ss2d_scan_bwd_sr_static_ws(float **ptr) {
    static float *dev_ptr[SS2D_MAX_DEVICES] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return (int)e;
    if (dev < 0 || dev >= SS2D_MAX_DEVICES) return (int)hipErrorInvalidDevice;
    if (!dev_ptr[dev]) {   // once per device, with that device current
        void *q = nullptr;
        e = hipGetSymbolAddress(&q, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return (int)e;
        dev_ptr[dev] = static_cast<float *>(q);
    }
    *ptr = dev_ptr[dev];
    return 0;
}

size_t ss2d_scan_bwd_sr_ws_bytes(int batch, int dim, int L, int ngroups) {
    const long long total = (long long)batch * dim, per = sr_rows_per_wg(total);
    const long long nwg = (total + per - 1) / per;
    return (size_t)(nwg + (long long)batch * ngroups) * 32 * (size_t)L * sizeof(float);
}

int ss2d_scan_bwd_sr_launch(int dtype, ScanBwdParams p, bool vec4, bool dirs, float *ws, size_t ws_bytes, hipStream_t s) {
    const long long total = (long long)p.batch * p.dim, per = sr_rows_per_wg(total);
    p.rows_per_wg = (int)per;
    const long long nwg = (total + per - 1) / per;
    if (nwg > 0x7fffffffLL) return (int)hipErrorInvalidConfiguration;
    // the partial-tile path needs 16-byte rows (L % 4) for its reduction kernel and a big enough, aligned workspace
    p.ws = (ws && (p.L % 4 == 0) && (((uintptr_t)ws | (uintptr_t)p.dB | (uintptr_t)p.dC | (uintptr_t)p.dxdbl) & 15) == 0 &&
            ws_bytes >= ss2d_scan_bwd_sr_ws_bytes(p.batch, p.dim, p.L, p.G)) ? ws : nullptr;
    int rc;
    static const bool env_nopack = getenv("SS2D_BWD_NO_PACK") != nullptr;      // A/B switch, read once
    if (p.dtw) {   // fused delta: only the form SS2D issues (4-element rows, direction-aware, workspace), ranks 6 and 12
        if (!p.ws || !vec4 || (p.R != 6 && p.R != 12)) return (int)hipErrorInvalidValue;
        const bool h = dtype == SS2D_DTYPE_BF16;
        if (p.R == 6) rc = h ? launch_sr<true, bf16, true, true, 6>(p, (int)nwg, s) : launch_sr<true, float, true, true, 6>(p, (int)nwg, s);
        else rc = h ? launch_sr<true, bf16, true, true, 12>(p, (int)nwg, s) : launch_sr<true, float, true, true, 12>(p, (int)nwg, s);
    } else if (p.ws && vec4 && p.nchunks == 1 && p.L <= CHUNK / 2 && !env_nopack) {
        // sequences that fit half a row (the 7 x 7 / 8 x 8 stages): two channels per wave at a time
        if (dtype == SS2D_DTYPE_BF16) rc = dirs ? launch_sr<true, bf16, true, true, 0, true>(p, (int)nwg, s) : launch_sr<true, bf16, false, true, 0, true>(p, (int)nwg, s);
        else rc = dirs ? launch_sr<true, float, true, true, 0, true>(p, (int)nwg, s) : launch_sr<true, float, false, true, 0, true>(p, (int)nwg, s);
    } else if (dtype == SS2D_DTYPE_BF16) {
        if (dirs) rc = vec4 ? launch_sr_ws<true, bf16, true>(p, (int)nwg, s) : launch_sr_ws<false, bf16, true>(p, (int)nwg, s);
        else rc = vec4 ? launch_sr_ws<true, bf16, false>(p, (int)nwg, s) : launch_sr_ws<false, bf16, false>(p, (int)nwg, s);
    } else {
        if (dirs) rc = vec4 ? launch_sr_ws<true, float, true>(p, (int)nwg, s) : launch_sr_ws<false, float, true>(p, (int)nwg, s);
        else rc = vec4 ? launch_sr_ws<true, float, false>(p, (int)nwg, s) : launch_sr_ws<false, float, false>(p, (int)nwg, s);
    }
    if (rc || !p.ws) return rc;
    const int n4 = 8 * p.L;   // float4s per (dB | dC) tile
    const float *wsc = p.ws;
    int L = p.L, dpg = p.dpg, rpw = p.rows_per_wg, n4v = n4;
    if (p.dxdbl) {
        const dim3 grid((unsigned)((n4 + 255) / 256) * (unsigned)(p.batch * p.G));
        if (dtype == SS2D_DTYPE_BF16) ss2d_launch(scan_bwd_sr_reduce_packed_kernel<bf16>, grid, dim3(256), 0, s, wsc, static_cast<bf16 *>(p.dxdbl), p.bc_stride, p.R, L, dpg, rpw, n4v);
        else ss2d_launch(scan_bwd_sr_reduce_packed_kernel<float>, grid, dim3(256), 0, s, wsc, static_cast<float *>(p.dxdbl), p.bc_stride, p.R, L, dpg, rpw, n4v);
        return ss2d_take_launch_rc();
    }
    void *args[] = {&wsc, &p.dB, &p.dC, &L, &dpg, &rpw, &n4v};
    return (int)hipLaunchKernel(reinterpret_cast<const void *>(scan_bwd_sr_reduce_kernel), dim3((unsigned)((n4 + 255) / 256) * (unsigned)(p.batch * p.G)), dim3(256), args, 0, s);
}
