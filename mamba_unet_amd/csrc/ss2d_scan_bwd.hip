// Selective scan backward for gfx950 (autograd half of `selective_scan_fn`,
// /root/reference/code/networks/mamba_sys.py:436-442; forward semantics :62-144).
// GENERAL path (any dstate <= 16, optional z gate); the SS2D call pattern (dstate 16, no z) is dispatched to the
// state-row kernel of ss2d_scan_bwd_sr.hip.
//
// Same lane layout as the forward (ss2d_scan_fwd.hip): a wave = 4 channel rows x 16 lanes x 8 positions.
// Chunks of 128 positions are walked from the END of the sequence to the start:
//   * forward states of the chunk are recomputed from the checkpoint the forward kernel stored
//     (state entering the chunk): local recurrence + DPP row scan, as in the forward;
//   * the adjoint recurrence  r_l = a_l * (dy_l*C_l + r_{l+1})  (r_l = a_l * dL/dh_l) is scanned the other
//     way with row_shl DPP steps; the value leaving the chunk on the left is carried to the next
//     (earlier) chunk through LDS;
//   * dB/dC are sums over the channels of a group: the 4 rows of a wave are summed in registers with a
//     v_permlane16/32_swap butterfly, every wave adds the result into its private LDS tile (plain
//     read-modify-write, bank-swizzled), the workgroup then adds the wave tiles and issues ONE global
//     float atomic per element and chunk;
//   * dA/dD/ddelta_bias are reduced over lanes (DPP row sums), over chunks (LDS / registers) and
//     over batch (global float atomics at the end).
#include <cstdlib>
#include "ss2d_scan.h"
#include "../../include/ss2d.h"

namespace {

constexpr int T = 8;
constexpr int CHUNK = 128;
constexpr int NP = 8;
constexpr int ACC_WAVE_FLOATS = NP * 2 * 4 * 64;  // per wave: [np][mat][item pair][lane], rows already summed

// DIRS: direction-aware build (rev_mask / u_groups honoured); false compiles the mirrored paths away
template <int NW, bool HAS_Z, bool VEC4, typename IO, bool DIRS>
__global__ __launch_bounds__(NW * 64) void scan_bwd_kernel(const ScanBwdParams p) {
    extern __shared__ __align__(16) unsigned char smem[];
    f4 *tile = reinterpret_cast<f4 *>(smem);                                   // [2][NP*4*16] f4 = 16 KB
    float *acc = reinterpret_cast<float *>(smem + 2 * NP * 4 * 16 * sizeof(f4));  // [NW][ACC_WAVE_FLOATS], 16 KB per wave
    f2 *rc = reinterpret_cast<f2 *>(acc + NW * ACC_WAVE_FLOATS);                 // [NW*rounds][NP][4 rows]
    float *dAs = reinterpret_cast<float *>(rc + NW * p.rounds * NP * 4);       // [NW*rounds][4 rows][16]
    float *dXs = dAs + NW * p.rounds * 64;                                     // [NW*rounds][4 rows][2] : dD, dbias

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, row = lane >> 4, seg = lane & 15;
    int blk = xcd_remap(blockIdx.x, p.nwg);
    const int cbi = blk % p.wgs_per_group;
    blk /= p.wgs_per_group;
    const int g = blk % p.G, b = blk / p.G;
    const int L = p.L, N = p.N;
    const bool rev = DIRS && ((p.rev_mask >> g) & 1u);
    const int ug = DIRS ? g / (p.G / p.Gu) : g;
    const size_t bcs = DIRS ? (size_t)p.bc_stride : N * (size_t)L;   // the plain path keeps its dense addressing
    const IO *__restrict__ Bg = static_cast<const IO *>(p.B) + (size_t)(b * p.G + g) * bcs;
    const IO *__restrict__ Cg = static_cast<const IO *>(p.C) + (size_t)(b * p.G + g) * bcs;
    float *__restrict__ dBg = p.dB + (size_t)(b * p.G + g) * N * (size_t)L;
    float *__restrict__ dCg = p.dC + (size_t)(b * p.G + g) * N * (size_t)L;

    // zero the LDS accumulators / carries
    for (int i = tid; i < NW * p.rounds * NP * 4; i += NW * 64) rc[i] = f2{0.0f, 0.0f};
    for (int i = tid; i < NW * p.rounds * (64 + 8); i += NW * 64) dAs[i] = 0.0f;  // dAs and dXs are contiguous

    for (int chunk = p.nchunks - 1; chunk >= 0; --chunk) {
        const int base = chunk * CHUNK;
        __syncthreads();  // previous flush done, acc zeroed, tile free
        for (int it = tid; it < NP * 32; it += NW * 64) {
            const int np = it >> 5, q4 = it & 31;
            const int pos = base + q4 * 4;
            const int n0 = 2 * np, n1 = 2 * np + 1;
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const IO *__restrict__ M = m ? Cg : Bg;
                f4 lo = f4{0, 0, 0, 0}, hi = f4{0, 0, 0, 0};
                if (DIRS) {
                    if (n0 < N) lo = load4m<VEC4>(M + (size_t)n0 * L, pos, L, rev);
                    if (n1 < N) hi = load4m<VEC4>(M + (size_t)n1 * L, pos, L, rev);
                } else if (VEC4) {
                    if (pos < L) {
                        if (n0 < N) lo = ld4(M + (size_t)n0 * L + pos);
                        if (n1 < N) hi = ld4(M + (size_t)n1 * L + pos);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (pos + e < L) {
                            if (n0 < N) lo[e] = ld1(M + (size_t)n0 * L + pos + e);
                            if (n1 < N) hi[e] = ld1(M + (size_t)n1 * L + pos + e);
                        }
                    }
                }
                const int j0 = (q4 & 1) * 2, sg = q4 >> 1;
                tile[m * (NP * 64) + (np * 4 + j0) * 16 + sg] = f4{lo.x, hi.x, lo.y, hi.y};
                tile[m * (NP * 64) + (np * 4 + j0 + 1) * 16 + sg] = f4{lo.z, hi.z, lo.w, hi.w};
            }
        }
        __syncthreads();

        for (int r = 0; r < p.rounds; ++r) {
            const int quad = r * NW + wave;
            const int c_in_g = cbi * p.cb + quad * 4 + row;
            const bool row_ok = (quad * 4 + row < p.cb) && (c_in_g < p.dpg);
            const int c = g * p.dpg + (row_ok ? c_in_g : 0);
            const size_t rowoff = ((size_t)b * p.dim + c) * (size_t)L;
            const int p0 = base + seg * T;
            float uv[T], dt[T], sg[T], dyv[T], du_[T];
            const size_t urowoff = DIRS ? (((size_t)b * p.Gu + ug) * p.dpg + (row_ok ? c_in_g : 0)) * (size_t)L : rowoff;
            load8d<VEC4, DIRS>(static_cast<const IO *>(p.u) + urowoff, p0, L, row_ok, rev, uv);
            load8d<VEC4, DIRS>(static_cast<const IO *>(p.delta) + rowoff, p0, L, row_ok, rev, dt);
            load8d<VEC4, DIRS>(static_cast<const IO *>(p.dout) + urowoff, p0, L, row_ok, rev, dyv);
            const float bias = p.bias ? p.bias[c] : 0.0f;
            const float Dc = p.D ? p.D[c] : 0.0f;
            float S = 0.0f;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float x = dt[t] + bias, s_ = 1.0f;
                if (p.softplus) x = softplus_sig<true>(x, s_);
                const bool ok = row_ok && (p0 + t < L);
                dt[t] = ok ? x : 0.0f;
                sg[t] = s_;
                du_[t] = dt[t] * uv[t];
                S += dt[t];
            }
            const float *ck = p.ckpt + (((size_t)b * p.dim + c) * p.nchunks + chunk) * N;
            const float *Arow = p.A + (size_t)c * N;
            // Private (per wave) dB/dC tile, plain read-modify-write.  Within a (pair, matrix) block of 256
            // floats the slot of (item pair j, row -> (state s, odd/even item hi), seg) is bit-swizzled so that
            // BOTH this per-lane access (lanes = 16 segs x rows) and the flush (lanes = consecutive positions)
            // hit 32 distinct banks:  low5 = seg[1:0] | (seg2^j1)<<2 | (seg3^j0)<<3 | (s^hi)<<4 ; j0<<5 | hi<<6 | j1<<7
            const int lane_base = seg | ((((row & 1) ^ (row >> 1)) & 1) << 4) | ((row >> 1) << 6);
            float *accw = acc + wave * ACC_WAVE_FLOATS;

            float zs[T], zv[T];  // silu(z) and z when gated
            if (HAS_Z) {
                // gate: out = y * silu(z) with y = sum_n C h + D u, so y has to be rebuilt first
                load8d<VEC4, DIRS>(static_cast<const IO *>(p.z) + rowoff, p0, L, row_ok, rev, zv);
                f2 y2[T];
#pragma unroll
                for (int t = 0; t < T; ++t) y2[t] = f2{Dc * uv[t], 0.0f};
#pragma unroll 1
                for (int np = 0; np < NP; ++np) {
                    if (2 * np >= N) break;
                    const f2 A2 = f2{Arow[2 * np], (2 * np + 1 < N) ? Arow[2 * np + 1] : 0.0f} * SS2D_LOG2E;
                    const f2 hin = f2{ck[2 * np], (2 * np + 1 < N) ? ck[2 * np + 1] : 0.0f};
                    f2 a[T], bb[T];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f4 Bq = tile[(np * 4 + j) * 16 + seg];
                        bb[2 * j] = splat2(du_[2 * j]) * f2{Bq.x, Bq.y};
                        bb[2 * j + 1] = splat2(du_[2 * j + 1]) * f2{Bq.z, Bq.w};
                    }
#pragma unroll
                    for (int t = 0; t < T; ++t) a[t] = exp2_2(splat2(dt[t]) * A2);
                    f2 h = f2{seg == 0 ? hin.x : 0.0f, seg == 0 ? hin.y : 0.0f};
#pragma unroll
                    for (int t = 0; t < T; ++t) h = fma2(a[t], h, bb[t]);
                    f2 at = exp2_2(splat2(S) * A2);
                    float hx = h.x, hy = h.y, ax = at.x, ay = at.y;
                    row_scan_fwd_pair(hx, hy, ax, ay);
                    h.x = dpp_mov<DPP_ROW_SHR(1), 0xf, false>(hin.x, hx);
                    h.y = dpp_mov<DPP_ROW_SHR(1), 0xf, false>(hin.y, hy);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f4 Cq = tile[NP * 64 + (np * 4 + j) * 16 + seg];
                        h = fma2(a[2 * j], h, bb[2 * j]);
                        y2[2 * j] = fma2(f2{Cq.x, Cq.y}, h, y2[2 * j]);
                        h = fma2(a[2 * j + 1], h, bb[2 * j + 1]);
                        y2[2 * j + 1] = fma2(f2{Cq.z, Cq.w}, h, y2[2 * j + 1]);
                    }
                }
                float dzv[T];
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const float y = y2[t].x + y2[t].y;
                    const float s_ = sigmoidf_fast(zv[t]);
                    zs[t] = zv[t] * s_;
                    dzv[t] = dyv[t] * y * (s_ * (1.0f + zv[t] * (1.0f - s_)));
                    dyv[t] *= zs[t];
                }
                store8d<VEC4, DIRS>(static_cast<IO *>(p.dz) + rowoff, p0, L, row_ok, rev, dzv);
            }

            f2 ddt2[T], ddu2[T];
#pragma unroll
            for (int t = 0; t < T; ++t) { ddt2[t] = f2{0, 0}; ddu2[t] = f2{0, 0}; }

#pragma unroll 1
            for (int np = 0; np < NP; ++np) {
                if (2 * np >= N) break;
                const f2 A2 = f2{Arow[2 * np], (2 * np + 1 < N) ? Arow[2 * np + 1] : 0.0f} * SS2D_LOG2E;
                const f2 hin = f2{ck[2 * np], (2 * np + 1 < N) ? ck[2 * np + 1] : 0.0f};
                const f2 rcar = rc[(quad * NP + np) * 4 + row];
                f2 a[T], h[T], Bv[T], Cv[T];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const f4 Bq = tile[(np * 4 + j) * 16 + seg];
                    const f4 Cq = tile[NP * 64 + (np * 4 + j) * 16 + seg];
                    Bv[2 * j] = f2{Bq.x, Bq.y}; Bv[2 * j + 1] = f2{Bq.z, Bq.w};
                    Cv[2 * j] = f2{Cq.x, Cq.y}; Cv[2 * j + 1] = f2{Cq.z, Cq.w};
                }
#pragma unroll
                for (int t = 0; t < T; ++t) a[t] = exp2_2(splat2(dt[t]) * A2);
                const f2 at = exp2_2(splat2(S) * A2);
                // ---- forward states of the chunk
                f2 hl = f2{seg == 0 ? hin.x : 0.0f, seg == 0 ? hin.y : 0.0f};
#pragma unroll
                for (int t = 0; t < T; ++t) hl = fma2(a[t], hl, splat2(du_[t]) * Bv[t]);
                float hx = hl.x, hy = hl.y, ax = at.x, ay = at.y;
                row_scan_fwd_pair(hx, hy, ax, ay);
                f2 hstart;
                hstart.x = dpp_mov<DPP_ROW_SHR(1), 0xf, false>(hin.x, hx);
                hstart.y = dpp_mov<DPP_ROW_SHR(1), 0xf, false>(hin.y, hy);
                {
                    f2 hh = hstart;
#pragma unroll
                    for (int t = 0; t < T; ++t) { hh = fma2(a[t], hh, splat2(du_[t]) * Bv[t]); h[t] = hh; }
                }
                // ---- adjoint: lane-local suffix recurrence, row scan, then the true pass
                f2 rl = f2{seg == 15 ? rcar.x : 0.0f, seg == 15 ? rcar.y : 0.0f};
#pragma unroll
                for (int t = T - 1; t >= 0; --t) rl = a[t] * fma2(splat2(dyv[t]), Cv[t], rl);
                float rx = rl.x, ry = rl.y;
                ax = at.x; ay = at.y;
                row_scan_rev_pair(rx, ry, ax, ay);
                f2 rr;
                rr.x = dpp_mov<DPP_ROW_SHL(1), 0xf, false>(rcar.x, rx);
                rr.y = dpp_mov<DPP_ROW_SHL(1), 0xf, false>(rcar.y, ry);
                f2 dAp = f2{0.0f, 0.0f};
                f2 dBprev = f2{0.0f, 0.0f}, dCprev = f2{0.0f, 0.0f};
#pragma unroll
                for (int t = T - 1; t >= 0; --t) {
                    const f2 dh = fma2(splat2(dyv[t]), Cv[t], rr);  // dL/dh_t
                    rr = a[t] * dh;                                  // a_t * dL/dh_t, flows to t-1
                    const f2 dCc = splat2(dyv[t]) * h[t];
                    const f2 dBc = splat2(du_[t]) * dh;
                    ddu2[t] = fma2(dh, Bv[t], ddu2[t]);
                    const f2 hp = (t > 0) ? h[t - 1] : hstart;
                    const f2 w = rr * hp;                            // dL/da_t * a_t
                    ddt2[t] = fma2(w, A2, ddt2[t]);                  // in log2 units; scaled by ln2 below
                    dAp = fma2(w, splat2(dt[t]), dAp);
                    if (t & 1) {
                        dBprev = dBc; dCprev = dCc;
                    } else {
                        // channel sum over the 4 rows for items (t+1, t) x 2 states; row r = (item t+1-(r>>1), state r&1)
#if !defined(SS2D_ABLATE_LDSADD) && !defined(SS2D_ABLATE_BFLY)
                        float sB = dBprev.x, sC = dCprev.x;
                        rows_reduce4x2(sB, dBprev.y, dBc.x, dBc.y, sC, dCprev.y, dCc.x, dCc.y);
                        const int j = t >> 1;
                        const int off = (lane_base ^ ((j >> 1) << 2) ^ ((j & 1) << 3)) + ((j & 1) << 5) + ((j >> 1) << 7);
                        float *aB = accw + (np * 2 + 0) * 256 + off;
                        float *aC = accw + (np * 2 + 1) * 256 + off;
                        if (r == 0) { *aB = sB; *aC = sC; }   // first round of the chunk overwrites: no zero-fill pass
                        else { *aB += sB; *aC += sC; }
#else
                        asm volatile("" ::"v"(dBc.x), "v"(dBc.y), "v"(dCc.x), "v"(dCc.y), "v"(dBprev.x), "v"(dCprev.x));
#endif
                    }
                }
                // value leaving the chunk on the left: carried to the previous chunk's lane 15
                const float cx = dpp_mov<DPP_ROW_ROR(15), 0xf, false>(0.0f, rr.x);  // lane 15 <- lane 0
                const float cy = dpp_mov<DPP_ROW_ROR(15), 0xf, false>(0.0f, rr.y);
                if (seg == 15) rc[(quad * NP + np) * 4 + row] = f2{cx, cy};
                // dA (log2 units were only used for ddt): reduce over the row, accumulate over chunks in LDS
                float sx = dAp.x, sy = dAp.y;
                row_sum2_to_lane15(sx, sy);
                if (seg == 15) {
                    dAs[(quad * 4 + row) * 16 + 2 * np] += sx;
                    dAs[(quad * 4 + row) * 16 + 2 * np + 1] += sy;
                }
            }
            // ---- per-position gradients
            float duo[T], ddo[T], dD_part = 0.0f, db_part = 0.0f;
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const float ddu = ddu2[t].x + ddu2[t].y;
                float ddt = (ddt2[t].x + ddt2[t].y) * SS2D_LN2 + ddu * uv[t];
                duo[t] = ddu * dt[t] + dyv[t] * Dc;
                if (p.softplus) ddt *= sg[t];
                const bool ok = row_ok && (p0 + t < L);
                ddo[t] = ok ? ddt : 0.0f;
                dD_part += dyv[t] * uv[t];
                db_part += ddo[t];
            }
            store8d<VEC4, DIRS>(static_cast<IO *>(p.du) + rowoff, p0, L, row_ok, rev, duo);
            store8d<VEC4, DIRS>(static_cast<IO *>(p.ddelta) + rowoff, p0, L, row_ok, rev, ddo);
            float sD = dD_part, sb = db_part;
            row_sum2_to_lane15(sD, sb);
            if (seg == 15) {
                dXs[(quad * 4 + row) * 2] += sD;
                dXs[(quad * 4 + row) * 2 + 1] += sb;
            }
        }
        __syncthreads();
        // ---- flush dB/dC of this chunk: add the 4 row tiles, one global atomic per element
#if !defined(SS2D_ABLATE_FLUSHALL)
        for (int idx = tid; idx < 2 * 16 * CHUNK; idx += NW * 64) {
            const int mat = idx >> 11, n = (idx >> 7) & 15, pos = idx & 127;
            const int np = n >> 1, s = n & 1, sgm = pos >> 3, t = pos & 7;
            const int j = t >> 1, hi = (t & 1) ^ 1;
            const int off = (np * 2 + mat) * 256 + ((sgm & 3) | ((((sgm >> 2) & 1) ^ (j >> 1)) << 2) | ((((sgm >> 3) & 1) ^ (j & 1)) << 3) |
                                                   ((s ^ hi) << 4) | ((j & 1) << 5) | (hi << 6) | ((j >> 1) << 7));
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += acc[w * ACC_WAVE_FLOATS + off];
#if !defined(SS2D_ABLATE_FLUSH)
            if (n < N && base + pos < L) atomicAdd((mat ? dCg : dBg) + (size_t)n * L + (rev ? L - 1 - base - pos : base + pos), v);
#else
            asm volatile("" ::"v"(v));
#endif
        }
#endif
    }
    __syncthreads();
    // ---- dA / dD / ddelta_bias: sums over batch -> global atomics
    for (int i = tid; i < NW * p.rounds * 64; i += NW * 64) {
        const int n = i & 15, rw = (i >> 4) & 3, quad = i >> 6;
        const int c_in_g = cbi * p.cb + quad * 4 + rw;
        if (quad * 4 + rw < p.cb && c_in_g < p.dpg && n < N) atomicAdd(p.dA + (size_t)(g * p.dpg + c_in_g) * N + n, dAs[i]);
    }
    for (int i = tid; i < NW * p.rounds * 4; i += NW * 64) {
        const int c_in_g = cbi * p.cb + i;  // i = quad*4 + row
        if (i < p.cb && c_in_g < p.dpg) {
            if (p.dD) atomicAdd(p.dD + g * p.dpg + c_in_g, dXs[i * 2]);
            if (p.dbias) atomicAdd(p.dbias + g * p.dpg + c_in_g, dXs[i * 2 + 1]);
        }
    }
}

template <int NW, bool HAS_Z, bool VEC4, typename IO, bool DIRS>
int launch_one(const ScanBwdParams &p, size_t lds, hipStream_t s) {
    auto kern = scan_bwd_kernel<NW, HAS_Z, VEC4, IO, DIRS>;
    static size_t lds_set[SS2D_MAX_DEVICES] = {};   // per instantiation and device: the largest size already granted
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= SS2D_MAX_DEVICES) dev = 0;
    if (lds > lds_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        lds_set[dev] = lds;
    }
    ScanBwdParams pc = p;
    void *args[] = {&pc};
    return (int)hipLaunchKernel(reinterpret_cast<const void *>(kern), dim3(p.nwg), dim3(NW * 64), args, lds, s);
}
template <int NW, typename IO, bool DIRS>
int launch_bwd(const ScanBwdParams &p, bool has_z, bool vec4, size_t lds, hipStream_t s) {
    if (has_z) return vec4 ? launch_one<NW, true, true, IO, DIRS>(p, lds, s) : launch_one<NW, true, false, IO, DIRS>(p, lds, s);
    return vec4 ? launch_one<NW, false, true, IO, DIRS>(p, lds, s) : launch_one<NW, false, false, IO, DIRS>(p, lds, s);
}
template <typename IO, bool DIRS>
int launch_bwd_nw(int nw, const ScanBwdParams &p, bool has_z, bool vec4, size_t lds, hipStream_t s) {
    return nw == 8 ? launch_bwd<8, IO, DIRS>(p, has_z, vec4, lds, s)
         : nw == 4 ? launch_bwd<4, IO, DIRS>(p, has_z, vec4, lds, s)
         : nw == 2 ? launch_bwd<2, IO, DIRS>(p, has_z, vec4, lds, s)
                   : launch_bwd<1, IO, DIRS>(p, has_z, vec4, lds, s);
}

}  // namespace

extern "C" int ss2d_selective_scan_bwd(int dtype, const void *u, const void *delta, const float *A, const void *B,
                                       const void *C, const float *D, const void *z, const float *delta_bias,
                                       const void *dout, const float *ckpt, void *du, void *ddelta, float *dA,
                                       float *dB, float *dC, float *dD, void *dz, float *ddelta_bias, int batch,
                                       int dim, int L, int dstate, int ngroups, int delta_softplus, void *stream) {
    return ss2d_selective_scan_dir_bwd(dtype, u, delta, A, B, C, D, z, delta_bias, dout, ckpt, du, ddelta, dA, dB, dC, dD,
                                       dz, ddelta_bias, batch, dim, L, dstate, ngroups, delta_softplus, ngroups, 0u, 0, stream);
}

extern "C" int ss2d_selective_scan_dir_bwd(int dtype, const void *u, const void *delta, const float *A, const void *B,
                                           const void *C, const float *D, const void *z, const float *delta_bias,
                                           const void *dout, const float *ckpt, void *du, void *ddelta, float *dA,
                                           float *dB, float *dC, float *dD, void *dz, float *ddelta_bias, int batch,
                                           int dim, int L, int dstate, int ngroups, int delta_softplus, int u_groups,
                                           unsigned rev_mask, long long bc_bg_stride, void *stream) {
    return ss2d_selective_scan_dir_bwd_ws(dtype, u, delta, A, B, C, D, z, delta_bias, dout, ckpt, du, ddelta, dA, dB, dC, dD, dz,
                                          ddelta_bias, batch, dim, L, dstate, ngroups, delta_softplus, u_groups, rev_mask,
                                          bc_bg_stride, nullptr, 0, stream);
}

// SS2D_BWD_KERNEL=general keeps every shape on the general kernel (A/B runs); read once
static bool force_general_kernel() {
    static const bool v = [] { const char *e = getenv("SS2D_BWD_KERNEL"); return e && e[0] == 'g'; }();
    return v;
}

extern "C" size_t ss2d_scan_bwd_workspace_bytes(int batch, int dim, int L, int dstate, int ngroups) {
    if (force_general_kernel()) return 0;
    if (batch <= 0 || dim <= 0 || L <= 0 || ngroups <= 0 || dstate != 16 || L % 4) return 0;   // only the 16-state kernel uses one
    return ss2d_scan_bwd_sr_ws_bytes(batch, dim, L, ngroups);
}

extern "C" int ss2d_selective_scan_dir_bwd_static(int dtype, const void *u, const void *delta, const float *A, const void *B,
                                                  const void *C, const float *D, const void *z, const float *delta_bias,
                                                  const void *dout, const float *ckpt, void *du, void *ddelta, float *dA,
                                                  float *dB, float *dC, float *dD, void *dz, float *ddelta_bias, int batch,
                                                  int dim, int L, int dstate, int ngroups, int delta_softplus, int u_groups,
                                                  unsigned rev_mask, long long bc_bg_stride, void *stream) {
    if (z != nullptr || dz != nullptr) { ss2d_set_error("selective_scan_bwd_static: the z gate is not supported by this entry point"); return -7; }
    const size_t need = ss2d_scan_bwd_workspace_bytes(batch, dim, L, dstate, ngroups);
    if (need == 0) { ss2d_set_error("selective_scan_bwd_static: needs dstate == 16 and L %% 4 == 0 (got dstate %d, L %d)", dstate, L); return -7; }
    if (need > SS2D_STATIC_WS_BYTES) {
        ss2d_set_error("selective_scan_bwd_static: this shape needs a %zu-byte workspace, the built-in one holds %u", need, SS2D_STATIC_WS_BYTES);
        return -8;
    }
    if ((((uintptr_t)dB | (uintptr_t)dC | (uintptr_t)A | (uintptr_t)ckpt) & 15) != 0) {
        ss2d_set_error("selective_scan_bwd_static: A, ckpt, dB, dC must be 16-byte aligned"); return -7; }
    float *ws = nullptr;
    const int rc = ss2d_scan_bwd_sr_static_ws(&ws);
    if (rc) { ss2d_set_error("selective_scan_bwd_static: hipGetSymbolAddress failed: %s", hipGetErrorString((hipError_t)rc)); return rc; }
    return ss2d_selective_scan_dir_bwd_ws(dtype, u, delta, A, B, C, D, nullptr, delta_bias, dout, ckpt, du, ddelta, dA, dB, dC, dD,
                                          nullptr, ddelta_bias, batch, dim, L, dstate, ngroups, delta_softplus, u_groups, rev_mask,
                                          bc_bg_stride, ws, SS2D_STATIC_WS_BYTES, stream);
}

static int scan_bwd_impl(int dtype, const void *u, const void *delta, const float *A, const void *B, const void *C, const float *D,
                         const void *z, const float *delta_bias, const void *dout, const float *ckpt, void *du, void *ddelta,
                         float *dA, float *dB, float *dC, float *dD, void *dz, float *ddelta_bias, int batch, int dim, int L,
                         int dstate, int ngroups, int delta_softplus, int u_groups, unsigned rev_mask, long long bc_bg_stride,
                         void *workspace, size_t workspace_bytes, const float *dt_w, int dt_rank, void *d_xdbl, void *stream);

extern "C" int ss2d_selective_scan_dir_bwd_ws(int dtype, const void *u, const void *delta, const float *A, const void *B,
                                              const void *C, const float *D, const void *z, const float *delta_bias,
                                              const void *dout, const float *ckpt, void *du, void *ddelta, float *dA,
                                              float *dB, float *dC, float *dD, void *dz, float *ddelta_bias, int batch,
                                              int dim, int L, int dstate, int ngroups, int delta_softplus, int u_groups,
                                              unsigned rev_mask, long long bc_bg_stride, void *workspace,
                                              size_t workspace_bytes, void *stream) {
    return scan_bwd_impl(dtype, u, delta, A, B, C, D, z, delta_bias, dout, ckpt, du, ddelta, dA, dB, dC, dD, dz, ddelta_bias, batch, dim,
                         L, dstate, ngroups, delta_softplus, u_groups, rev_mask, bc_bg_stride, workspace, workspace_bytes, nullptr, 0, nullptr, stream);
}

static int core_bwd_impl(const char *name, int dtype, const void *u, const void *x_dbl, const float *dt_w, const float *A,
                         const float *D, const float *delta_bias, const void *dout, const float *ckpt, void *du,
                         void *ddelta, float *dA, float *dB, float *dC, void *d_xdbl, float *dD, float *ddelta_bias, int batch,
                         int dim, int L, int dstate, int ngroups, int dt_rank, int u_groups, unsigned rev_mask,
                         void *workspace, size_t workspace_bytes, void *stream) {
    if (!x_dbl || !dt_w) { ss2d_set_error("%s: null required pointer", name); return -2; }
    if (dt_rank != 6 && dt_rank != 12) { ss2d_set_error("%s: dt_rank %d is not 6 or 12", name, dt_rank); return -4; }
    if (dstate != 16 || L % 4) { ss2d_set_error("%s: needs dstate == 16 and L %% 4 == 0 (got %d, %d)", name, dstate, L); return -4; }
    if (force_general_kernel()) { ss2d_set_error("%s: not available with SS2D_BWD_KERNEL=general", name); return -7; }
    const size_t need = ss2d_scan_bwd_workspace_bytes(batch, dim, L, dstate, ngroups);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)) {
        ss2d_set_error("%s: needs a 16-byte aligned workspace of %zu bytes", name, need); return -7; }
    if ((((uintptr_t)dB | (uintptr_t)dC | (uintptr_t)d_xdbl | (uintptr_t)A | (uintptr_t)ckpt) & 15) != 0) {
        ss2d_set_error("%s: A, ckpt, dB, dC / d_x_dbl must be 16-byte aligned", name); return -7; }
    const size_t es = dtype == SS2D_DTYPE_BF16 ? 2 : 4;
    const char *base = static_cast<const char *>(x_dbl);
    const long long stride = (long long)(dt_rank + 2 * dstate) * L;
    return scan_bwd_impl(dtype, u, x_dbl, A, base + (size_t)dt_rank * L * es, base + (size_t)(dt_rank + dstate) * L * es, D, nullptr,
                         delta_bias, dout, ckpt, du, ddelta, dA, dB, dC, dD, nullptr, ddelta_bias, batch, dim, L, dstate, ngroups, 1,
                         u_groups, rev_mask, stride, workspace, workspace_bytes, dt_w, dt_rank, d_xdbl, stream);
}

extern "C" int ss2d_selective_scan_core_bwd(int dtype, const void *u, const void *x_dbl, const float *dt_w, const float *A,
                                            const float *D, const float *delta_bias, const void *dout, const float *ckpt, void *du,
                                            void *ddelta, float *dA, float *dB, float *dC, float *dD, float *ddelta_bias, int batch,
                                            int dim, int L, int dstate, int ngroups, int dt_rank, int u_groups, unsigned rev_mask,
                                            void *workspace, size_t workspace_bytes, void *stream) {
    if (!dB || !dC) { ss2d_set_error("selective_scan_core_bwd: null required pointer"); return -2; }
    return core_bwd_impl("selective_scan_core_bwd", dtype, u, x_dbl, dt_w, A, D, delta_bias, dout, ckpt, du, ddelta, dA, dB, dC, nullptr,
                         dD, ddelta_bias, batch, dim, L, dstate, ngroups, dt_rank, u_groups, rev_mask, workspace, workspace_bytes, stream);
}

extern "C" int ss2d_selective_scan_core_bwd_packed(int dtype, const void *u, const void *x_dbl, const float *dt_w, const float *A,
                                                   const float *D, const float *delta_bias, const void *dout, const float *ckpt,
                                                   void *du, void *ddelta, float *dA, void *d_x_dbl, float *dD, float *ddelta_bias,
                                                   int batch, int dim, int L, int dstate, int ngroups, int dt_rank, int u_groups,
                                                   unsigned rev_mask, void *workspace, size_t workspace_bytes, void *stream) {
    if (!d_x_dbl) { ss2d_set_error("selective_scan_core_bwd_packed: null required pointer"); return -2; }
    return core_bwd_impl("selective_scan_core_bwd_packed", dtype, u, x_dbl, dt_w, A, D, delta_bias, dout, ckpt, du, ddelta, dA, nullptr,
                         nullptr, d_x_dbl, dD, ddelta_bias, batch, dim, L, dstate, ngroups, dt_rank, u_groups, rev_mask, workspace,
                         workspace_bytes, stream);
}

static int scan_bwd_impl(int dtype, const void *u, const void *delta, const float *A, const void *B, const void *C, const float *D,
                         const void *z, const float *delta_bias, const void *dout, const float *ckpt, void *du, void *ddelta,
                         float *dA, float *dB, float *dC, float *dD, void *dz, float *ddelta_bias, int batch, int dim, int L,
                         int dstate, int ngroups, int delta_softplus, int u_groups, unsigned rev_mask, long long bc_bg_stride,
                         void *workspace, size_t workspace_bytes, const float *dt_w, int dt_rank, void *d_xdbl, void *stream) {
    if (dtype != SS2D_DTYPE_F32 && dtype != SS2D_DTYPE_BF16) { ss2d_set_error("selective_scan_bwd: unknown dtype %d", dtype); return -1; }
    if (!u || !delta || !A || !B || !C || !dout || !ckpt || !du || !ddelta || !dA || ((!dB || !dC) && !d_xdbl)) {
        ss2d_set_error("selective_scan_bwd: null required pointer"); return -2; }
    if ((z != nullptr) != (dz != nullptr)) { ss2d_set_error("selective_scan_bwd: z and dz must be given together"); return -2; }
    if (batch <= 0 || dim <= 0 || L <= 0 || dstate <= 0 || ngroups <= 0) { ss2d_set_error("selective_scan_bwd: non-positive size"); return -3; }
    if (dstate > SS2D_MAX_DSTATE) { ss2d_set_error("selective_scan_bwd: dstate %d > %d", dstate, SS2D_MAX_DSTATE); return -4; }
    if (dim % ngroups) { ss2d_set_error("selective_scan_bwd: dim %d not divisible by ngroups %d", dim, ngroups); return -5; }
    if (u_groups <= 0 || ngroups % u_groups) { ss2d_set_error("selective_scan_bwd: u_groups %d must divide ngroups %d", u_groups, ngroups); return -5; }
    if (ngroups > 32 && rev_mask) { ss2d_set_error("selective_scan_bwd: rev_mask needs ngroups <= 32"); return -5; }
    const long long bc_dense = (long long)dstate * L;
    if (bc_bg_stride == 0) bc_bg_stride = bc_dense;
    if (bc_bg_stride < bc_dense) { ss2d_set_error("selective_scan_bwd: bc_bg_stride %lld < dstate*L", bc_bg_stride); return -5; }
    ScanBwdParams p;
    p.Gu = u_groups; p.rev_mask = rev_mask; p.bc_stride = bc_bg_stride;
    p.dtw = dt_w; p.R = dt_rank; p.ws = nullptr; p.rows_per_wg = 0; p.dxdbl = d_xdbl;
    p.u = u; p.delta = delta; p.A = A; p.B = B; p.C = C;
    p.D = D; p.z = z; p.bias = delta_bias; p.dout = dout; p.ckpt = ckpt;
    p.du = du; p.ddelta = ddelta; p.dA = dA; p.dB = dB; p.dC = dC; p.dD = D ? dD : nullptr;
    p.dz = dz; p.dbias = delta_bias ? ddelta_bias : nullptr;
    p.batch = batch; p.dim = dim; p.L = L; p.N = dstate; p.G = ngroups; p.dpg = dim / ngroups;
    p.nchunks = ss2d_scan_num_chunks(L); p.softplus = delta_softplus;
    // channels per workgroup: NW waves x 4 rows x `rounds`
    const int nw = p.dpg >= 32 ? 8 : (p.dpg >= 16 ? 4 : (p.dpg >= 8 ? 2 : 1));
    const int per_round = nw * 4;
    int rounds = 1;
    if (p.dpg > per_round) {
        // fewer, fatter workgroups cut the dB/dC atomic traffic; keep >= ~1 workgroup per CU when possible
        const long long units = (long long)batch * ngroups;
        rounds = 3;
        while (rounds > 1 && units * ((p.dpg + per_round * rounds - 1) / (per_round * rounds)) < 192) --rounds;
    }
    static const int env_rounds = [] { const char *e = getenv("SS2D_BWD_ROUNDS"); return e ? atoi(e) : 0; }();  // tuning knob, read once
    if (env_rounds >= 1 && env_rounds <= 3) rounds = env_rounds;
    p.rounds = rounds; p.cb = per_round * rounds;
    p.wgs_per_group = (p.dpg + p.cb - 1) / p.cb;
    const long long nwg = (long long)batch * ngroups * p.wgs_per_group;
    if (nwg > 0x7fffffffLL) { ss2d_set_error("selective_scan_bwd: grid too large"); return -6; }
    p.nwg = (int)nwg;
    const size_t lds = 2 * NP * 4 * 16 * sizeof(f4) + (size_t)nw * ACC_WAVE_FLOATS * sizeof(float) +
                       (size_t)nw * rounds * NP * 4 * sizeof(f2) + (size_t)nw * rounds * (64 + 8) * sizeof(float);
    const uintptr_t amask = dtype == SS2D_DTYPE_BF16 ? 7 : 15;  // 4 elements
    const bool vec4 = (L % 4 == 0) && (bc_bg_stride % 4 == 0) &&
                      ((((uintptr_t)u | (uintptr_t)delta | (uintptr_t)B | (uintptr_t)C | (uintptr_t)dout | (uintptr_t)du |
                         (uintptr_t)ddelta | (uintptr_t)z | (uintptr_t)dz) & amask) == 0);
    hipStream_t s = (hipStream_t)stream;
    const bool dirs = rev_mask != 0 || u_groups != ngroups || bc_bg_stride != bc_dense;
    // A caller that passes the workspace ss2d_scan_bwd_workspace_bytes() asks for relies on the library WRITING dB / dC
    // (partial tiles + reduction; it then skips their zero-fill).  Where that path cannot be taken although the shape
    // qualifies -- a misaligned A, ckpt, dB, dC or workspace -- the call fails instead of accumulating into whatever
    // dB / dC hold.
    if (workspace && workspace_bytes > 0 && workspace_bytes >= ss2d_scan_bwd_workspace_bytes(batch, dim, L, dstate, ngroups) &&
        ss2d_scan_bwd_workspace_bytes(batch, dim, L, dstate, ngroups) > 0 && z == nullptr &&
        ((((uintptr_t)A | (uintptr_t)ckpt | (uintptr_t)dB | (uintptr_t)dC | (uintptr_t)workspace) & 15) != 0)) {
        ss2d_set_error("selective_scan_bwd: with a workspace, A, ckpt, dB, dC and the workspace must be 16-byte aligned "
                       "(dB / dC are written, not accumulated, on that path)");
        return -7;
    }
    // the SS2D call pattern: 16 states, no gate -> state-row kernel (SS2D_BWD_KERNEL=general keeps the general one: A/B runs)
    // (L <= 2^22: the state-row kernel addresses the <= 96 rows of a part through one descriptor of less than 2^31 bytes)
    if (dstate == 16 && z == nullptr && !force_general_kernel() && L <= (1 << 22) && ((((uintptr_t)A | (uintptr_t)ckpt) & 15) == 0)) {
        const int rc = ss2d_scan_bwd_sr_launch(dtype, p, vec4, dirs, static_cast<float *>(workspace), workspace_bytes, s);
        if (rc) ss2d_set_error("selective_scan_bwd: launch failed: %s", hipGetErrorString((hipError_t)rc));
        return rc;
    }
    if (d_xdbl) { ss2d_set_error("selective_scan_core_bwd_packed: this shape is outside the state-row kernel (L %d)", L); return -7; }
    if (dt_w) { ss2d_set_error("selective_scan_core_bwd: A and ckpt must be 16-byte aligned"); return -7; }
    int rc = dtype == SS2D_DTYPE_BF16
                 ? (dirs ? launch_bwd_nw<bf16, true>(nw, p, z != nullptr, vec4, lds, s) : launch_bwd_nw<bf16, false>(nw, p, z != nullptr, vec4, lds, s))
                 : (dirs ? launch_bwd_nw<float, true>(nw, p, z != nullptr, vec4, lds, s) : launch_bwd_nw<float, false>(nw, p, z != nullptr, vec4, lds, s));
    if (rc) ss2d_set_error("selective_scan_bwd: launch failed: %s", hipGetErrorString((hipError_t)rc));
    return rc;
}
