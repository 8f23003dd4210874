// Fused out_norm + gate of SS2D for gfx950:  out = (LayerNorm_D(y) * gamma + beta) * SiLU(z)
// (/root/reference/code/networks/mamba_sys.py:336 `out_norm`, :452 applied, :561 `y * F.silu(z)`), forward and
// backward.  The reference runs LayerNorm, SiLU and the product as three elementwise passes (plus their three
// backward passes); here one pass reads y and z and writes out, one pass computes dy, dz, dgamma, dbeta.
//
// One wave per token row (D = 192 ... 1536 channels: 4-element chunks strided over the 64 lanes, kept in
// registers), wave-level butterfly sums for the statistics, fp32 arithmetic in both storage types.
// z is read in place from the z half of the in_proj output (row stride ldz), dz is written with row stride lddz.
// dgamma / dbeta: per-lane column sums over the rows a wave walks, then one float atomic per column and wave.
#include "ss2d_common.h"
#include "../../include/ss2d.h"

namespace {

constexpr int MAXK = 6;  // 4 * 64 * MAXK = 1536 channels max

struct NormGateParams {
    const void *y, *z, *dout;
    const float *gamma, *beta;
    void *out, *dy, *dz;
    float *mean, *rstd, *dgamma, *dbeta;
    long long rows;
    int D, ldz, lddz;
    float eps;
    const float *rscale;    // nullable: out (and, in the backward, dout) of row r is multiplied by rscale[r / rps] -- the per-sample
    unsigned rps;           // stochastic-depth factor mask / keep of the block (mamba_sys.py:581-585), applied here, in front of the
};                          // bias-free out_proj, instead of in a pass of its own behind it

__device__ __forceinline__ float wave_sum(float v) { return lanes_sum<64>(v); }

template <int K, typename IO>
__global__ __launch_bounds__(256) void normgate_fwd_kernel(const NormGateParams p) {
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
    const int D = p.D, nchunk = D >> 2;
    const IO *__restrict__ Y = static_cast<const IO *>(p.y);
    const IO *__restrict__ Z = static_cast<const IO *>(p.z);
    IO *__restrict__ O = static_cast<IO *>(p.out);
    f4 g[K], b[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int c = lane + 64 * k;
        g[k] = c < nchunk ? ld4(p.gamma + 4 * c) : f4{0, 0, 0, 0};
        b[k] = c < nchunk ? ld4(p.beta + 4 * c) : f4{0, 0, 0, 0};
    }
    const float invD = 1.0f / (float)D;
    for (long long r = wave0; r < p.rows; r += nwaves) {
        f4 yv[K], zv[K];
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = lane + 64 * k;
            yv[k] = c < nchunk ? ld4(Y + r * D + 4 * c) : f4{0, 0, 0, 0};
            zv[k] = c < nchunk ? ld4(Z + r * p.ldz + 4 * c) : f4{0, 0, 0, 0};
            s += (yv[k].x + yv[k].y) + (yv[k].z + yv[k].w);
        }
        const float mu = wave_sum(s) * invD;
        float q = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = lane + 64 * k;
            if (c < nchunk) {
                const f4 d = yv[k] - mu;
                q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
            }
        }
        const float rstd = __builtin_amdgcn_rsqf(wave_sum(q) * invD + p.eps);
        if (lane == 0 && p.mean) { p.mean[r] = mu; p.rstd[r] = rstd; }
        const float sc = p.rscale ? p.rscale[(unsigned)r / p.rps] : 1.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = lane + 64 * k;
            if (c < nchunk) {
                f4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = (yv[k][e] - mu) * rstd * g[k][e] + b[k][e];
                    const float zz = zv[k][e];
                    o[e] = v * (zz * sigmoidf_fast(zz)) * sc;
                }
                st4(O + r * D + 4 * c, o);
            }
        }
    }
}

template <int K, int NWV, typename IO>
__global__ __launch_bounds__(NWV * 64) void normgate_bwd_kernel(const NormGateParams p) {
    __shared__ __attribute__((aligned(16))) float red[NWV * 2 * K * 256];   // per-wave partial sums of dgamma / dbeta
    const int lane = threadIdx.x & 63;
    const long long wave0 = (long long)blockIdx.x * NWV + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * NWV;
    const int D = p.D, nchunk = D >> 2;
    const IO *__restrict__ Y = static_cast<const IO *>(p.y);
    const IO *__restrict__ Z = static_cast<const IO *>(p.z);
    const IO *__restrict__ G = static_cast<const IO *>(p.dout);
    IO *__restrict__ DY = static_cast<IO *>(p.dy);
    IO *__restrict__ DZ = static_cast<IO *>(p.dz);
    f4 g[K], b[K], ag[K], ab[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int c = lane + 64 * k;
        g[k] = c < nchunk ? ld4(p.gamma + 4 * c) : f4{0, 0, 0, 0};
        b[k] = c < nchunk ? ld4(p.beta + 4 * c) : f4{0, 0, 0, 0};
        ag[k] = f4{0, 0, 0, 0};
        ab[k] = f4{0, 0, 0, 0};
    }
    const float invD = 1.0f / (float)D;
    // The next row of the wave is requested before this one is worked on: a wave walks only a handful of rows (4.6 at the
    // 14 x 14 stage, 18 at 56 x 56) and every trip used to start with an exposed memory round trip -- 21 us whatever the size.
    // Loads are never under a branch: chunks past D and rows past the end read a clamped address and are not used.
    f4 yv[K], zv[K], gv[K];
    float mu = 0.0f, rstd = 0.0f, sc = 1.0f;
    const float *__restrict__ rsc = p.rscale ? p.rscale : p.rstd;      // always a valid address; unused without rscale
    auto fetch = [&](long long r_, f4 (&y_)[K], f4 (&z_)[K], f4 (&g_)[K], float &mu_, float &rs_, float &sc_) {
        const long long rc = r_ < p.rows ? r_ : p.rows - 1;
        mu_ = p.mean[rc];
        rs_ = p.rstd[rc];
        sc_ = rsc[p.rscale ? (unsigned)rc / p.rps : 0u];
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = lane + 64 * k, cc = c < nchunk ? c : nchunk - 1;
            y_[k] = ld4(Y + rc * D + 4 * cc);
            z_[k] = ld4(Z + rc * p.ldz + 4 * cc);
            g_[k] = ld4(G + rc * D + 4 * cc);
        }
    };
    if (wave0 < p.rows) fetch(wave0, yv, zv, gv, mu, rstd, sc);
    for (long long r = wave0; r < p.rows; r += nwaves) {
        f4 yn[K], zn[K], gn[K];
        float mun, rsn, scn;
        fetch(r + nwaves, yn, zn, gn, mun, rsn, scn);
        const float gsc = p.rscale ? sc : 1.0f;
        f4 nh[K], dn[K];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = lane + 64 * k;
            nh[k] = f4{0, 0, 0, 0};
            dn[k] = f4{0, 0, 0, 0};
            if (c < nchunk) {
                f4 dzv;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float n = (yv[k][e] - mu) * rstd;
                    const float v = n * g[k][e] + b[k][e];
                    const float zz = zv[k][e], sg = sigmoidf_fast(zz);
                    const float go = gv[k][e] * gsc;
                    const float dv = go * (zz * sg);
                    dzv[e] = go * v * (sg * (1.0f + zz * (1.0f - sg)));
                    ag[k][e] += dv * n;
                    ab[k][e] += dv;
                    const float d = dv * g[k][e];
                    nh[k][e] = n;
                    dn[k][e] = d;
                    s1 += d;
                    s2 += d * n;
                }
                st4(DZ + r * p.lddz + 4 * c, dzv);
            }
        }
        const float m1 = wave_sum(s1) * invD, m2 = wave_sum(s2) * invD;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = lane + 64 * k;
            if (c < nchunk) {
                f4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rstd * (dn[k][e] - m1 - nh[k][e] * m2);
                st4(DY + r * D + 4 * c, o);
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) { yv[k] = yn[k]; zv[k] = zn[k]; gv[k] = gn[k]; }
        mu = mun;
        rstd = rsn;
        sc = scn;
    }
    // per-lane column sums -> per-wave LDS slots -> one global atomic per column and workgroup (at most 256
    // workgroups: thousands of same-address global atomics would serialise at the memory side)
    wg_colsum2<K, NWV, 64>(red, ag, ab, D, p.dgamma, p.dbeta);
}

int check(const NormGateParams &p, int dtype, const char *name) {
    if (dtype != SS2D_DTYPE_F32 && dtype != SS2D_DTYPE_BF16) { ss2d_set_error("%s: unknown dtype %d", name, dtype); return -1; }
    if (p.rows <= 0 || p.D <= 0) { ss2d_set_error("%s: non-positive size", name); return -3; }
    if (p.D % 4 || p.D > 256 * MAXK) { ss2d_set_error("%s: D %d must be a multiple of 4 and <= %d", name, p.D, 256 * MAXK); return -4; }
    if (p.ldz < p.D || p.ldz % 4) { ss2d_set_error("%s: ldz %d must be >= D and a multiple of 4", name, p.ldz); return -4; }
    return 0;
}

template <typename IO>
void launch(bool bwd, const NormGateParams &p, hipStream_t s) {
    const int K = (p.D / 4 + 63) / 64;
    if (!bwd) {
        long long nwg = (p.rows + 3) / 4;
        if (nwg > 8192) nwg = 8192;
        const dim3 grid((unsigned)nwg), block(256);
        switch (K) {
            case 1: ss2d_launch((normgate_fwd_kernel<1, IO>), grid, block, 0, s, p); break;
            case 2: ss2d_launch((normgate_fwd_kernel<2, IO>), grid, block, 0, s, p); break;
            case 3: ss2d_launch((normgate_fwd_kernel<3, IO>), grid, block, 0, s, p); break;
            case 4: ss2d_launch((normgate_fwd_kernel<4, IO>), grid, block, 0, s, p); break;
            case 5: ss2d_launch((normgate_fwd_kernel<5, IO>), grid, block, 0, s, p); break;
            default: ss2d_launch((normgate_fwd_kernel<6, IO>), grid, block, 0, s, p); break;
        }
        return;
    }
    // backward: at most 256 workgroups (one per CU); 16 waves each where the register budget allows it (K <= 2)
    const int nwv = K <= 2 ? 16 : 4;
    long long nwg = (p.rows + nwv - 1) / nwv;
    if (nwg > 256) nwg = 256;
    const dim3 grid((unsigned)nwg), block(nwv * 64);
    switch (K) {
        case 1: ss2d_launch((normgate_bwd_kernel<1, 16, IO>), grid, block, 0, s, p); break;
        case 2: ss2d_launch((normgate_bwd_kernel<2, 16, IO>), grid, block, 0, s, p); break;
        case 3: ss2d_launch((normgate_bwd_kernel<3, 4, IO>), grid, block, 0, s, p); break;
        case 4: ss2d_launch((normgate_bwd_kernel<4, 4, IO>), grid, block, 0, s, p); break;
        case 5: ss2d_launch((normgate_bwd_kernel<5, 4, IO>), grid, block, 0, s, p); break;
        default: ss2d_launch((normgate_bwd_kernel<6, 4, IO>), grid, block, 0, s, p); break;
    }
}

}  // namespace

static int scale_ok(const float *row_scale, long long rows_per_scale, long long rows, const char *name) {
    if (row_scale && (rows_per_scale <= 0 || rows_per_scale > 0xffffffffLL || rows > 0xffffffffLL)) {
        ss2d_set_error("%s: rows_per_scale %lld / rows %lld out of range", name, rows_per_scale, rows); return -3; }
    return 0;
}

extern "C" int ss2d_outnorm_gate_fwd(int dtype, const void *y, const void *z, const float *gamma, const float *beta,
                                     void *out, float *mean, float *rstd, long long rows, int D, int ldz, float eps,
                                     void *stream) {
    return ss2d_outnorm_gate_scaled_fwd(dtype, y, z, gamma, beta, nullptr, 0, out, mean, rstd, rows, D, ldz, eps, stream);
}

extern "C" int ss2d_outnorm_gate_scaled_fwd(int dtype, const void *y, const void *z, const float *gamma, const float *beta,
                                            const float *row_scale, long long rows_per_scale, void *out, float *mean,
                                            float *rstd, long long rows, int D, int ldz, float eps, void *stream) {
    NormGateParams p{};
    if (scale_ok(row_scale, rows_per_scale, rows, "outnorm_gate_fwd")) return -3;
    p.rscale = row_scale; p.rps = row_scale ? (unsigned)rows_per_scale : 1u;
    if (!y || !z || !gamma || !beta || !out) { ss2d_set_error("outnorm_gate_fwd: null pointer"); return -2; }
    if ((mean == nullptr) != (rstd == nullptr)) { ss2d_set_error("outnorm_gate_fwd: mean and rstd go together"); return -2; }
    p.y = y; p.z = z; p.gamma = gamma; p.beta = beta; p.out = out; p.mean = mean; p.rstd = rstd;
    p.rows = rows; p.D = D; p.ldz = ldz; p.eps = eps;
    int rc = check(p, dtype, "outnorm_gate_fwd");
    if (rc) return rc;
    if (dtype == SS2D_DTYPE_BF16) launch<bf16>(false, p, (hipStream_t)stream); else launch<float>(false, p, (hipStream_t)stream);
    rc = ss2d_take_launch_rc();
    if (rc) ss2d_set_error("outnorm_gate_fwd: launch failed: %s", hipGetErrorString((hipError_t)rc));
    return rc;
}

extern "C" int ss2d_outnorm_gate_bwd(int dtype, const void *dout, const void *y, const void *z, const float *gamma,
                                     const float *beta, const float *mean, const float *rstd, void *dy, void *dz,
                                     float *dgamma, float *dbeta, long long rows, int D, int ldz, int lddz,
                                     void *stream) {
    return ss2d_outnorm_gate_scaled_bwd(dtype, dout, y, z, gamma, beta, mean, rstd, nullptr, 0, dy, dz, dgamma, dbeta, rows, D, ldz, lddz, stream);
}

extern "C" int ss2d_outnorm_gate_scaled_bwd(int dtype, const void *dout, const void *y, const void *z, const float *gamma,
                                            const float *beta, const float *mean, const float *rstd, const float *row_scale,
                                            long long rows_per_scale, void *dy, void *dz, float *dgamma, float *dbeta,
                                            long long rows, int D, int ldz, int lddz, void *stream) {
    NormGateParams p{};
    if (scale_ok(row_scale, rows_per_scale, rows, "outnorm_gate_bwd")) return -3;
    p.rscale = row_scale; p.rps = row_scale ? (unsigned)rows_per_scale : 1u;
    if (!dout || !y || !z || !gamma || !beta || !mean || !rstd || !dy || !dz || !dgamma || !dbeta) {
        ss2d_set_error("outnorm_gate_bwd: null pointer"); return -2; }
    p.dout = dout; p.y = y; p.z = z; p.gamma = gamma; p.beta = beta; p.mean = const_cast<float *>(mean);
    p.rstd = const_cast<float *>(rstd); p.dy = dy; p.dz = dz; p.dgamma = dgamma; p.dbeta = dbeta;
    p.rows = rows; p.D = D; p.ldz = ldz; p.lddz = lddz;
    int rc = check(p, dtype, "outnorm_gate_bwd");
    if (rc) return rc;
    if (lddz < D || lddz % 4) { ss2d_set_error("outnorm_gate_bwd: lddz %d must be >= D and a multiple of 4", lddz); return -4; }
    if (dtype == SS2D_DTYPE_BF16) launch<bf16>(true, p, (hipStream_t)stream); else launch<float>(true, p, (hipStream_t)stream);
    rc = ss2d_take_launch_rc();
    if (rc) ss2d_set_error("outnorm_gate_bwd: launch failed: %s", hipGetErrorString((hipError_t)rc));
    return rc;
}
