// LayerNorm over the channel axis for gfx950, forward and backward: the `ln_1` of every VSSBlock
// (/root/reference/code/networks/mamba_sys.py:573-585, `x + drop_path(self_attention(ln_1(x)))`) and the norms of
// the patch merge / expand layers (:196, :226, :249).  Same structure as ss2d_normgate.hip: one wave per token
// row, the row lives in registers, wave butterfly sums, fp32 arithmetic; dgamma / dbeta go through a
// workgroup-level LDS reduction and one float atomic per column and workgroup (at most 256 workgroups).
// Input and output storage types are independent: under bf16 autocast the fp32 residual stream is normalised
// straight into the bf16 tensor the next GEMM reads (no separate cast pass); dx comes back in the input's type.
#include <cstdlib>
#include "ss2d_common.h"
#include "../../include/ss2d.h"

namespace {

constexpr int MAXK = 6;  // 4 * 64 * MAXK = 1536 channels max

struct LNParams {
    const void *x, *dout;
    const float *gamma, *beta, *mean_in, *rstd_in;
    void *out, *dx;
    const void *add;    // backward (plain rows): a second gradient of x, added to dx as it is written (the residual path)
    float *mean, *rstd, *dgamma, *dbeta;
    long long rows;
    int D;
    float eps;
    int ps, psH, psW;   // ps > 0: rows are gathered through a ps x ps pixel un-shuffle of x (B, psH, psW, ps*ps*D)
                        // ps = -2: rows are the 2x2 neighbourhoods of x (B, psH, psW, D/4) (PatchMerging2D)
};

// Element offset in x (and dx) of normalised row r.  Plain: r * D.  Un-shuffle (the einops pattern
// 'b h w (p1 p2 c) -> b (h p1) (w p2) c' of mamba_sys.py:244, :262): output row (b, h*ps + p1, w*ps + p2) is the
// c-long run ((((b*H + h)*W + w)*ps + p1)*ps + p2) * c of the expand Linear's output -- the rearranging copy (231 MB
// each way in front of the last LayerNorm) becomes address arithmetic.
// Merge (mamba_sys.py:213-219, x0 | x1 | x2 | x3 = the (0,0), (1,0), (0,1), (1,1) pixels of a 2x2 neighbourhood,
// concatenated): the row's base is pixel (2i, 2j); ln_chunk_off adds the pixel of the chunk's quarter.
__device__ __forceinline__ long long ln_row_off(const LNParams &p, long long r) {
    if (p.ps == 0) return r * p.D;
    if (p.ps < 0) {
        const int w2 = p.psW >> 1, h2 = p.psH >> 1;
        const int j = (int)(r % w2);
        const long long t = r / w2;
        const int i = (int)(t % h2);
        const long long b = t / h2;
        return ((b * p.psH + 2 * i) * p.psW + 2 * j) * (long long)(p.D >> 2);
    }
    const int W4 = p.psW * p.ps, H4 = p.psH * p.ps;
    const int w4 = (int)(r % W4);
    const long long t = r / W4;
    const int h4 = (int)(t % H4);
    const long long b = t / H4;
    const int h = h4 / p.ps, p1 = h4 % p.ps, w = w4 / p.ps, p2 = w4 % p.ps;
    return ((((b * p.psH + h) * p.psW + w) * p.ps + p1) * p.ps + p2) * (long long)p.D;
}

// element offset of four-element chunk c of a row, relative to ln_row_off
__device__ __forceinline__ int ln_chunk_off(const LNParams &p, int c) {
    if (p.ps >= 0) return 4 * c;
    const int C = p.D >> 2, e = 4 * c, q = e / C, within = e - q * C;     // quarter q = pixel (q & 1, q >> 1)
    return ((q & 1) * p.psW + (q >> 1)) * C + within;
}

// sum over the LW (64 or 32) lanes that share a row
template <int LW>
__device__ __forceinline__ float wave_sum_ln(float v) { return lanes_sum<LW>(v); }

// LW = lanes per row: 64, or 32 (two rows per wave) when a row has at most 32 four-element chunks (D <= 128)
template <int K, int LW, typename TI, typename TO>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const LNParams p) {
    constexpr int RPW = 64 / LW;
    const int lane = threadIdx.x & (LW - 1);
    const long long wave0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + ((threadIdx.x & 63) / LW);
    const long long nwaves = (long long)gridDim.x * 4 * RPW;
    const int D = p.D, nchunk = D >> 2;
    const TI *__restrict__ X = static_cast<const TI *>(p.x);
    TO *__restrict__ O = static_cast<TO *>(p.out);
    f4 g[K], b[K];
    int co[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int c = lane + LW * k;
        g[k] = c < nchunk ? ld4(p.gamma + 4 * c) : f4{0, 0, 0, 0};
        b[k] = c < nchunk ? ld4(p.beta + 4 * c) : f4{0, 0, 0, 0};
        co[k] = ln_chunk_off(p, c < nchunk ? c : 0);
    }
    const float invD = 1.0f / (float)D;
    for (long long r = wave0; r < p.rows; r += nwaves) {
        f4 xv[K];
        float s = 0.0f;
        const long long xoff = ln_row_off(p, r);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = lane + LW * k;
            xv[k] = c < nchunk ? ld4(X + xoff + co[k]) : f4{0, 0, 0, 0};
            s += (xv[k].x + xv[k].y) + (xv[k].z + xv[k].w);
        }
        const float mu = wave_sum_ln<LW>(s) * invD;
        float q = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = lane + LW * k;
            if (c < nchunk) {
                const f4 d = xv[k] - mu;
                q += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
            }
        }
        const float rstd = __builtin_amdgcn_rsqf(wave_sum_ln<LW>(q) * invD + p.eps);
        if (lane == 0 && p.mean) { p.mean[r] = mu; p.rstd[r] = rstd; }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = lane + LW * k;
            if (c < nchunk) st4(O + r * D + 4 * c, (xv[k] - mu) * rstd * g[k] + b[k]);
        }
    }
}

// ADD: dx = LayerNorm backward + p.add (the gradient arriving over the residual connection around the norm: the block's
// input feeds ln_1 and the residual sum, mamba_sys.py:583-585, and autograd would add the two gradients in a pass of its own)
template <int K, int NWV, int LW, typename TI, typename TO, bool ADD = false>
__global__ __launch_bounds__(NWV * 64) void ln_bwd_kernel(const LNParams p) {
    __shared__ __attribute__((aligned(16))) float red[NWV * 2 * K * 256];
    constexpr int RPW = 64 / LW;
    const int lane = threadIdx.x & (LW - 1);
    const long long wave0 = ((long long)blockIdx.x * NWV + (threadIdx.x >> 6)) * RPW + ((threadIdx.x & 63) / LW);
    const long long nwaves = (long long)gridDim.x * NWV * RPW;
    const int D = p.D, nchunk = D >> 2;
    const TI *__restrict__ X = static_cast<const TI *>(p.x);
    const TO *__restrict__ G = static_cast<const TO *>(p.dout);
    TI *__restrict__ DX = static_cast<TI *>(p.dx);
    f4 g[K], ag[K], ab[K];
    int co[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int c = lane + LW * k;
        co[k] = ln_chunk_off(p, c < nchunk ? c : 0);
        g[k] = c < nchunk ? ld4(p.gamma + 4 * c) : f4{0, 0, 0, 0};
        ag[k] = f4{0, 0, 0, 0};
        ab[k] = f4{0, 0, 0, 0};
    }
    const float invD = 1.0f / (float)D;
    // the next row of the wave (or half-wave) is requested before this one is worked on -- see normgate_bwd_kernel; loads are
    // never under a branch: chunks past D and rows past the end read a clamped address and are not used
    f4 gv[K], xv[K], av[ADD ? K : 1];
    float mu = 0.0f, rstd = 0.0f;
    long long xoff = 0;
    const TI *__restrict__ AD = static_cast<const TI *>(p.add);
    auto fetch = [&](long long r_, f4 (&g_)[K], f4 (&x_)[K], f4 (&a_)[ADD ? K : 1], float &mu_, float &rs_, long long &xo_) {
        const long long rc = r_ < p.rows ? r_ : p.rows - 1;
        mu_ = p.mean_in[rc];
        rs_ = p.rstd_in[rc];
        xo_ = ln_row_off(p, rc);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = lane + LW * k, cc = c < nchunk ? c : nchunk - 1;
            g_[k] = ld4(G + rc * D + 4 * cc);
            x_[k] = ld4(X + xo_ + co[k]);      // co[k] is already clamped to chunk 0 for c >= nchunk
            if (ADD) a_[k] = ld4(AD + xo_ + co[k]);
        }
    };
    if (wave0 < p.rows) fetch(wave0, gv, xv, av, mu, rstd, xoff);
    for (long long r = wave0; r < p.rows; r += nwaves) {
        f4 gn[K], xn[K], an[ADD ? K : 1];
        float mun, rsn;
        long long xon;
        fetch(r + nwaves, gn, xn, an, mun, rsn, xon);
        f4 nh[K], dn[K];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = lane + LW * k;
            nh[k] = f4{0, 0, 0, 0};
            dn[k] = f4{0, 0, 0, 0};
            if (c < nchunk) {
                nh[k] = (xv[k] - mu) * rstd;
                dn[k] = gv[k] * g[k];
                ag[k] += gv[k] * nh[k];
                ab[k] += gv[k];
                const f4 t = dn[k] * nh[k];
                s1 += (dn[k].x + dn[k].y) + (dn[k].z + dn[k].w);
                s2 += (t.x + t.y) + (t.z + t.w);
            }
        }
        const float m1 = wave_sum_ln<LW>(s1) * invD, m2 = wave_sum_ln<LW>(s2) * invD;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = lane + LW * k;
            if (c < nchunk) st4(DX + xoff + co[k], ADD ? (dn[k] - m1 - nh[k] * m2) * rstd + av[k] : (dn[k] - m1 - nh[k] * m2) * rstd);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) { gv[k] = gn[k]; xv[k] = xn[k]; if (ADD) av[k] = an[k]; }
        mu = mun;
        rstd = rsn;
        xoff = xon;
    }
    wg_colsum2<K, NWV, LW>(red, ag, ab, D, p.dgamma, p.dbeta);
}

template <typename TI, typename TO, bool ADD = false>
void launch_ln(bool bwd, const LNParams &p, hipStream_t s) {
    const int nchunk = p.D / 4;
    const bool half = nchunk <= 32;              // two rows per wave
    const int K = half ? 1 : (nchunk + 63) / 64;
    const int rpw = half ? 2 : 1;
    if (!bwd) {
        long long nwg = (p.rows + 4 * rpw - 1) / (4 * rpw);
        if (nwg > 8192) nwg = 8192;
        const dim3 grid((unsigned)nwg), block(256);
        if (half) { ss2d_launch((ln_fwd_kernel<1, 32, TI, TO>), grid, block, 0, s, p); return; }
        switch (K) {
            case 1: ss2d_launch((ln_fwd_kernel<1, 64, TI, TO>), grid, block, 0, s, p); break;
            case 2: ss2d_launch((ln_fwd_kernel<2, 64, TI, TO>), grid, block, 0, s, p); break;
            case 3: ss2d_launch((ln_fwd_kernel<3, 64, TI, TO>), grid, block, 0, s, p); break;
            case 4: ss2d_launch((ln_fwd_kernel<4, 64, TI, TO>), grid, block, 0, s, p); break;
            case 5: ss2d_launch((ln_fwd_kernel<5, 64, TI, TO>), grid, block, 0, s, p); break;
            default: ss2d_launch((ln_fwd_kernel<6, 64, TI, TO>), grid, block, 0, s, p); break;
        }
        return;
    }
    const int nwv = K <= 2 ? 16 : 4;
    long long nwg = (p.rows + nwv * rpw - 1) / (nwv * rpw);
    if (nwg > 256) nwg = 256;
    const dim3 grid((unsigned)nwg), block(nwv * 64);
    if (half) { ss2d_launch((ln_bwd_kernel<1, 16, 32, TI, TO, ADD>), grid, block, 0, s, p); return; }
    switch (K) {
        case 1: ss2d_launch((ln_bwd_kernel<1, 16, 64, TI, TO, ADD>), grid, block, 0, s, p); break;
        case 2: ss2d_launch((ln_bwd_kernel<2, 16, 64, TI, TO, ADD>), grid, block, 0, s, p); break;
        case 3: ss2d_launch((ln_bwd_kernel<3, 4, 64, TI, TO, ADD>), grid, block, 0, s, p); break;
        case 4: ss2d_launch((ln_bwd_kernel<4, 4, 64, TI, TO, ADD>), grid, block, 0, s, p); break;
        case 5: ss2d_launch((ln_bwd_kernel<5, 4, 64, TI, TO, ADD>), grid, block, 0, s, p); break;
        default: ss2d_launch((ln_bwd_kernel<6, 4, 64, TI, TO, ADD>), grid, block, 0, s, p); break;
    }
}

int dispatch(bool bwd, int in_dtype, int out_dtype, const LNParams &p, void *stream, const char *name) {
    if ((in_dtype != SS2D_DTYPE_F32 && in_dtype != SS2D_DTYPE_BF16) || (out_dtype != SS2D_DTYPE_F32 && out_dtype != SS2D_DTYPE_BF16)) {
        ss2d_set_error("%s: unknown dtype %d / %d", name, in_dtype, out_dtype); return -1; }
    if (in_dtype == SS2D_DTYPE_BF16 && out_dtype == SS2D_DTYPE_F32) { ss2d_set_error("%s: bf16 input with fp32 output is not supported", name); return -1; }
    if (p.rows <= 0 || p.D <= 0) { ss2d_set_error("%s: non-positive size", name); return -3; }
    if (p.D % 4 || p.D > 256 * MAXK) { ss2d_set_error("%s: D %d must be a multiple of 4 and <= %d", name, p.D, 256 * MAXK); return -4; }
    hipStream_t s = (hipStream_t)stream;
    if (bwd && p.add) {
        if (in_dtype == SS2D_DTYPE_F32 && out_dtype == SS2D_DTYPE_F32) launch_ln<float, float, true>(bwd, p, s);
        else if (in_dtype == SS2D_DTYPE_F32) launch_ln<float, bf16, true>(bwd, p, s);
        else launch_ln<bf16, bf16, true>(bwd, p, s);
    } else if (in_dtype == SS2D_DTYPE_F32 && out_dtype == SS2D_DTYPE_F32) launch_ln<float, float>(bwd, p, s);
    else if (in_dtype == SS2D_DTYPE_F32) launch_ln<float, bf16>(bwd, p, s);
    else launch_ln<bf16, bf16>(bwd, p, s);
    const int rc = ss2d_take_launch_rc();
    if (rc) ss2d_set_error("%s: launch failed: %s", name, hipGetErrorString((hipError_t)rc));
    return rc;
}

}  // namespace

extern "C" int ss2d_layernorm_fwd(int in_dtype, int out_dtype, const void *x, const float *gamma, const float *beta,
                                  void *out, float *mean, float *rstd, long long rows, int D, float eps, void *stream) {
    if (!x || !gamma || !beta || !out) { ss2d_set_error("layernorm_fwd: null pointer"); return -2; }
    if ((mean == nullptr) != (rstd == nullptr)) { ss2d_set_error("layernorm_fwd: mean and rstd go together"); return -2; }
    LNParams p{};
    p.x = x; p.gamma = gamma; p.beta = beta; p.out = out; p.mean = mean; p.rstd = rstd; p.rows = rows; p.D = D; p.eps = eps;
    return dispatch(false, in_dtype, out_dtype, p, stream, "layernorm_fwd");
}

extern "C" int ss2d_layernorm_bwd(int in_dtype, int out_dtype, const void *dout, const void *x, const float *gamma,
                                  const float *mean, const float *rstd, void *dx, float *dgamma, float *dbeta,
                                  long long rows, int D, void *stream) {
    if (!dout || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta) { ss2d_set_error("layernorm_bwd: null pointer"); return -2; }
    LNParams p{};
    p.dout = dout; p.x = x; p.gamma = gamma; p.mean_in = mean; p.rstd_in = rstd; p.dx = dx; p.dgamma = dgamma; p.dbeta = dbeta;
    p.rows = rows; p.D = D;
    return dispatch(true, in_dtype, out_dtype, p, stream, "layernorm_bwd");
}

extern "C" int ss2d_layernorm_bwd_add(int in_dtype, int out_dtype, const void *dout, const void *x, const float *gamma,
                                      const float *mean, const float *rstd, const void *add, void *dx, float *dgamma,
                                      float *dbeta, long long rows, int D, void *stream) {
    if (!dout || !x || !gamma || !mean || !rstd || !add || !dx || !dgamma || !dbeta) { ss2d_set_error("layernorm_bwd_add: null pointer"); return -2; }
    LNParams p{};
    p.dout = dout; p.x = x; p.gamma = gamma; p.mean_in = mean; p.rstd_in = rstd; p.add = add; p.dx = dx; p.dgamma = dgamma; p.dbeta = dbeta;
    p.rows = rows; p.D = D;
    return dispatch(true, in_dtype, out_dtype, p, stream, "layernorm_bwd_add");
}

// ---- LayerNorm + segmentation head in one pass (the tail of VSSM.up_x4, mamba_sys.py:859-866: FinalPatchExpand_X4's
// norm over the 4x4 un-shuffled rows, permute to (B, C, H, W), 1x1 Conv2d(C -> classes, bias=False)).
//     logits[b, k, h', w'] = sum_c Wh[k][c] * LN(x)[b, h', w', c]
// The normalised tensor (231 MB at 224x224, batch 24) is never written: the forward reads x once and writes the
// 4-channel logits; the backward re-normalises x on the fly, expands dlogits through Wh^T into the LayerNorm
// backward and accumulates dWh next to dgamma / dbeta.  D <= 128 (two rows per wave, a 4-element chunk per lane).
namespace {

struct HeadParams {
    const float *wh;     // (nc, D) fp32
    const void *glog;    // bwd: dlogits (B, nc, H', W') in the out dtype
    void *logits;        // fwd
    float *dwh;          // bwd: (nc, D) fp32, accumulated into
    int nc;
    long long plane;     // H' * W'
};

__device__ __forceinline__ float half_sum32(float v, int) { return lanes_sum<32>(v); }

// value of lane `it` of this lane's 32-lane half (it is wave-uniform): two v_readlane + a select, no LDS
__device__ __forceinline__ float half_bcast(float v, int it, int half) {
    const int iv = __builtin_bit_cast(int, v);
    const int a = __builtin_amdgcn_readlane(iv, it), b = __builtin_amdgcn_readlane(iv, it + 32);
    return __builtin_bit_cast(float, half ? b : a);
}
__device__ __forceinline__ f4 zero_unless(f4 v, bool keep) { return keep ? v : f4{0, 0, 0, 0}; }

// Walks consecutive output rows of the un-shuffled image without a division per row: the 64-bit div / mod chain of
// ln_row_off (three of them, each a branchy software routine) is done once per 32-row run in 32-bit arithmetic,
// then the (b, h, p1, w, p2) counters are stepped.  off() = element offset of the row in x.
struct RowIter {
    int b, h, p1, w, p2;
    __device__ __forceinline__ RowIter(const LNParams &p, long long r) {
        const unsigned W4 = (unsigned)(p.psW * p.ps), H4 = (unsigned)(p.psH * p.ps), ur = (unsigned)r;   // rows < 2^31 (host check)
        const unsigned w4 = ur % W4, t = ur / W4, h4 = t % H4;
        b = (int)(t / H4); h = (int)(h4 / (unsigned)p.ps); p1 = (int)(h4 % (unsigned)p.ps);
        w = (int)(w4 / (unsigned)p.ps); p2 = (int)(w4 % (unsigned)p.ps);
    }
    __device__ __forceinline__ long long off(const LNParams &p) const {
        return ((((long long)(b * p.psH + h) * p.psW + w) * p.ps + p1) * p.ps + p2) * (long long)p.D;
    }
    __device__ __forceinline__ void next(const LNParams &p) {
        if (++p2 == p.ps) { p2 = 0; if (++w == p.psW) { w = 0; if (++p1 == p.ps) { p1 = 0; if (++h == p.psH) { h = 0; ++b; } } } }
    }
};

// A wave walks blocks of 64 consecutive rows, 32 per half-wave, one row per trip; lane l of a half keeps the
// per-row results (mean, rstd, logits) of trip l, so they leave as coalesced 64-byte stores once per block, and
// in the backward the per-row inputs (mean, rstd, dlogits) arrive as coalesced loads and are broadcast from
// registers -- one-lane stores / same-address loads per row cost more than the 231 MB of row data.
template <int NCP, typename TI, typename TO>
__global__ __launch_bounds__(256) void ln_head_fwd_kernel(const LNParams p, const HeadParams hp) {
    constexpr int LW = 32;
    const int lane = threadIdx.x & (LW - 1), half = (threadIdx.x & 63) >> 5;
    const long long wid = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (long long)gridDim.x * 4;
    const int D = p.D, nchunk = D >> 2;
    const TI *__restrict__ X = static_cast<const TI *>(p.x);
    TO *__restrict__ LG = static_cast<TO *>(hp.logits);
    const bool act = lane < nchunk;
    const int cl = act ? lane : nchunk - 1;
    const f4 g = zero_unless(ld4(p.gamma + 4 * cl), act), b = zero_unless(ld4(p.beta + 4 * cl), act);
    f4 wv[NCP];
#pragma unroll
    for (int k = 0; k < NCP; ++k) wv[k] = zero_unless(ld4(hp.wh + (size_t)(k < hp.nc ? k : hp.nc - 1) * D + 4 * cl), act && k < hp.nc);
    const float invD = 1.0f / (float)D;
    for (long long r0 = wid * 64; r0 < p.rows; r0 += nw * 64) {
        float keep[NCP], kmu = 0.0f, krs = 0.0f;
#pragma unroll
        for (int k = 0; k < NCP; ++k) keep[k] = 0.0f;
        RowIter ri(p, r0 + half * 32);
        const int nvalid = (int)((p.rows - (r0 + half * 32)) < 32 ? (p.rows - (r0 + half * 32)) : 32);   // <= 0: nothing stored
        long long xoff = nvalid > 0 ? ri.off(p) : 0;
#pragma unroll 4
        for (int it = 0; it < 32; ++it) {
            const f4 xv = zero_unless(ld4(X + xoff + 4 * cl), act);
            if (it + 1 < nvalid) { ri.next(p); xoff = ri.off(p); }   // past the end: the last row again, never stored
            const float mu = half_sum32((xv.x + xv.y) + (xv.z + xv.w), half) * invD;
            const f4 d = zero_unless(xv - mu, act);
            const float rstd = __builtin_amdgcn_rsqf(half_sum32((d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w), half) * invD + p.eps);
            const f4 y = d * rstd * g + b;      // zero on the inactive lanes (g = b = 0 there)
            if (lane == it) { kmu = mu; krs = rstd; }
#pragma unroll
            for (int k = 0; k < NCP; ++k) {
                const f4 t = y * wv[k];
                const float sk = half_sum32((t.x + t.y) + (t.z + t.w), half);
                if (lane == it) keep[k] = sk;
            }
        }
        const long long r = r0 + half * 32 + lane;   // the row whose results this lane kept
        if (r < p.rows) {
            if (p.mean) { p.mean[r] = kmu; p.rstd[r] = krs; }
            const long long bi = r / hp.plane, pix = r - bi * hp.plane;
#pragma unroll
            for (int k = 0; k < NCP; ++k)
                if (k < hp.nc) st1(LG + (bi * hp.nc + k) * hp.plane + pix, keep[k]);
        }
    }
}

// NWV waves per workgroup: 16 for up to 4 classes; 8 beyond, where 128 registers per lane would spill
template <int NCP, int NWV, typename TI, typename TO>
__global__ __launch_bounds__(NWV * 64) void ln_head_bwd_kernel(const LNParams p, const HeadParams hp) {
    constexpr int LW = 32;
    __shared__ __attribute__((aligned(16))) float red[NWV * 2 * 256];
    const int lane = threadIdx.x & (LW - 1), half = (threadIdx.x & 63) >> 5;
    const long long wid = (long long)blockIdx.x * NWV + (threadIdx.x >> 6), nw = (long long)gridDim.x * NWV;
    const int D = p.D, nchunk = D >> 2;
    const TI *__restrict__ X = static_cast<const TI *>(p.x);
    const TO *__restrict__ GL = static_cast<const TO *>(hp.glog);
    TI *__restrict__ DX = static_cast<TI *>(p.dx);
    const bool act = lane < nchunk;
    const int cl = act ? lane : nchunk - 1;
    const f4 g = zero_unless(ld4(p.gamma + 4 * cl), act), b = zero_unless(ld4(p.beta + 4 * cl), act);
    f4 wv[NCP], aw[NCP], ag[1] = {f4{0, 0, 0, 0}}, ab[1] = {f4{0, 0, 0, 0}};
#pragma unroll
    for (int k = 0; k < NCP; ++k) {
        wv[k] = zero_unless(ld4(hp.wh + (size_t)(k < hp.nc ? k : hp.nc - 1) * D + 4 * cl), act && k < hp.nc);
        aw[k] = f4{0, 0, 0, 0};
    }
    const float invD = 1.0f / (float)D;
    for (long long r0 = wid * 64; r0 < p.rows; r0 += nw * 64) {
        // this lane's row of the block: its mean, rstd and dlogits, read coalesced, broadcast per trip below
        const long long rl = r0 + half * 32 + lane;
        const bool lok = rl < p.rows;
        const long long rlc = lok ? rl : p.rows - 1;
        const long long lbi = rlc / hp.plane, lpix = rlc - lbi * hp.plane;
        const float lmu = p.mean_in[rlc], lrs = p.rstd_in[rlc];
        float lg[NCP];
#pragma unroll
        for (int k = 0; k < NCP; ++k) {
            const float v = ld1(GL + (lbi * hp.nc + (k < hp.nc ? k : hp.nc - 1)) * hp.plane + lpix);
            lg[k] = (lok && k < hp.nc) ? v : 0.0f;      // rows past the end and padded classes contribute nothing
        }
        RowIter ri(p, r0 + half * 32);
        const int nvalid = (int)((p.rows - (r0 + half * 32)) < 32 ? (p.rows - (r0 + half * 32)) : 32);
        long long xnext = nvalid > 0 ? ri.off(p) : 0;
#pragma unroll 2
        for (int it = 0; it < 32; ++it) {
            const bool rok = it < nvalid;
            const long long xoff = xnext;
            const f4 xv = ld4(X + xoff + 4 * cl);
            if (it + 1 < nvalid) { ri.next(p); xnext = ri.off(p); }
            const float mu = half_bcast(lmu, it, half), rstd = half_bcast(lrs, it, half);
            const f4 nh = zero_unless((xv - mu) * rstd, act);
            const f4 y = nh * g + b;
            f4 dout = f4{0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < NCP; ++k) {
                const float gv = half_bcast(lg[k], it, half);
                dout += wv[k] * gv;        // dL/dLN(x) = Wh^T dlogits
                aw[k] += y * gv;           // dWh[k] += dlogits[k] * LN(x)
            }
            const f4 dn = dout * g;
            ag[0] += dout * nh;
            ab[0] += dout;
            const f4 t = dn * nh;
            const float m1 = half_sum32((dn.x + dn.y) + (dn.z + dn.w), half) * invD;
            const float m2 = half_sum32((t.x + t.y) + (t.z + t.w), half) * invD;
            if (act && rok) st4(DX + xoff + 4 * lane, (dn - m1 - nh * m2) * rstd);
        }
    }
    wg_colsum2<1, NWV, LW>(red, ag, ab, D, p.dgamma, p.dbeta);
#pragma unroll
    for (int k = 0; k < NCP; k += 2) {
        __syncthreads();   // the previous pair's sums have been read
        f4 a0[1] = {aw[k]}, a1[1] = {k + 1 < NCP ? aw[k + 1] : f4{0, 0, 0, 0}};
        // rows past nc hold zeros; their (harmless) atomics go to the last real row
        const int k0 = k < hp.nc ? k : hp.nc - 1, k1 = k + 1 < hp.nc ? k + 1 : hp.nc - 1;
        wg_colsum2<1, NWV, LW>(red, a0, a1, D, hp.dwh + (size_t)k0 * D, hp.dwh + (size_t)k1 * D);
    }
}

// ---- up to 4 classes (the ACDC configuration): 8 lanes per row instead of 32.
// With a row of 96 channels on 32 lanes every row costs six 5-step wave reductions (mean, variance, 4 logits) for 4
// elements per lane: the kernels were bound by cross-lane instructions (169 / 235 us for 231 MB).  Here a lane owns
// KC = 3 (4) four-element chunks of its row, a wave works on 8 rows at once, and a sum over the row is 3 DPP adds
// (quad_perm xor 1, xor 2, row_half_mirror).  gamma is folded into the head weight once per lane:
//     logits[k] = rstd * (sum_d x_d gw[k][d] - mu * sum_d gw[k][d]) + sum_d beta_d wh[k][d],   gw = gamma * wh.
// Backward: with gl = dlogits of the row and nh = (x - mu) * rstd, only  anh[k][d] = sum_rows gl[k] nh[d]  and
// sgl[k] = sum_rows gl[k]  are accumulated; dwh = gamma * anh + beta * sgl, dgamma = sum_k wh[k] anh[k],
// dbeta = sum_k wh[k] sgl[k] follow at the end of the kernel.
__device__ __forceinline__ float sum8(float v) {
    v += dpp_mov<0xB1, 0xf, false>(0.0f, v);      // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E, 0xf, false>(0.0f, v);      // quad_perm [2,3,0,1]
    v += dpp_mov<0x141, 0xf, false>(0.0f, v);     // row_half_mirror: lane i <- lane 7 - i of its group of 8
    return v;
}
__device__ __forceinline__ float dot4(f4 a, f4 b) { return (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }

// element offset of normalised row r (32-bit arithmetic: the host guarantees rows < 2^31)
__device__ __forceinline__ long long head_row_off(const LNParams &p, unsigned r) {
    const unsigned W4 = (unsigned)(p.psW * p.ps), H4 = (unsigned)(p.psH * p.ps), ps = (unsigned)p.ps;
    const unsigned w4 = r % W4, t = r / W4, h4 = t % H4, b = t / H4;
    const unsigned h = h4 / ps, p1 = h4 - h * ps, w = w4 / ps, p2 = w4 - w * ps;
    return ((((long long)(b * p.psH + h) * p.psW + w) * ps + p1) * ps + p2) * (long long)p.D;
}

template <int KC, typename TI, typename TO>
__global__ __launch_bounds__(256) void ln_head4_fwd_kernel(const LNParams p, const HeadParams hp) {
    const int l8 = threadIdx.x & 7;
    const long long row0 = (long long)blockIdx.x * 32 + (threadIdx.x >> 3), stride = (long long)gridDim.x * 32;
    const int D = p.D, nchunk = D >> 2;
    const TI *__restrict__ X = static_cast<const TI *>(p.x);
    TO *__restrict__ LG = static_cast<TO *>(hp.logits);
    bool act[KC];
    int cc[KC];
    f4 gw[4][KC];
    float sgw[4], cb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { sgw[k] = 0.0f; cb[k] = 0.0f; }
#pragma unroll
    for (int j = 0; j < KC; ++j) {
        const int c = l8 + 8 * j;
        act[j] = c < nchunk;
        cc[j] = act[j] ? c : 0;
        const f4 g = act[j] ? ld4(p.gamma + 4 * cc[j]) : f4{0, 0, 0, 0}, b = act[j] ? ld4(p.beta + 4 * cc[j]) : f4{0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f4 w = (act[j] && k < hp.nc) ? ld4(hp.wh + (size_t)k * D + 4 * cc[j]) : f4{0, 0, 0, 0};
            gw[k][j] = g * w;
            sgw[k] += (gw[k][j].x + gw[k][j].y) + (gw[k][j].z + gw[k][j].w);
            cb[k] += dot4(b, w);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { sgw[k] = sum8(sgw[k]); cb[k] = sum8(cb[k]); }
    const float invD = 1.0f / (float)D;
    for (long long r = row0; r - (threadIdx.x >> 3) < p.rows; r += stride) {      // whole waves stay in the loop (DPP sums)
        const bool rok = r < p.rows;
        const long long xoff = head_row_off(p, (unsigned)(rok ? r : p.rows - 1));
        f4 xv[KC];
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const f4 t = ld4(X + xoff + 4 * cc[j]);
            xv[j] = act[j] ? t : f4{0, 0, 0, 0};
            s += (xv[j].x + xv[j].y) + (xv[j].z + xv[j].w);
        }
        const float mu = sum8(s) * invD;
        float q = 0.0f, lk[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            const f4 d = act[j] ? xv[j] - mu : f4{0, 0, 0, 0};
            q += dot4(d, d);
#pragma unroll
            for (int k = 0; k < 4; ++k) lk[k] += dot4(d, gw[k][j]);     // sum (x - mu) gw = sum x gw - mu sum gw
        }
        const float rstd = __builtin_amdgcn_rsqf(sum8(q) * invD + p.eps);
#pragma unroll
        for (int k = 0; k < 4; ++k) lk[k] = sum8(lk[k]) * rstd + cb[k];
        if (rok) {
            if (l8 == 0 && p.mean) { p.mean[r] = mu; p.rstd[r] = rstd; }
            const long long bi = r / hp.plane, pix = r - bi * hp.plane;
            const float mine = l8 == 0 ? lk[0] : (l8 == 1 ? lk[1] : (l8 == 2 ? lk[2] : lk[3]));
            if (l8 < hp.nc) st1(LG + (bi * hp.nc + l8) * hp.plane + pix, mine);
        }
    }
    (void)sgw;
}

template <int KC, typename TI, typename TO>
__global__ __launch_bounds__(256) void ln_head4_bwd_kernel(const LNParams p, const HeadParams hp) {
    __shared__ float red[5 * 128];                 // anh[4][D] | sgl[4] (padded)
    __shared__ __align__(16) float wsh[4 * 128];   // the head weight (zero rows past nc): read per row instead of held in 48 registers
    const int l8 = threadIdx.x & 7;
    const long long row0 = (long long)blockIdx.x * 32 + (threadIdx.x >> 3), stride = (long long)gridDim.x * 32;
    const int D = p.D, nchunk = D >> 2;
    const TI *__restrict__ X = static_cast<const TI *>(p.x);
    const TO *__restrict__ GL = static_cast<const TO *>(hp.glog);
    TI *__restrict__ DX = static_cast<TI *>(p.dx);
    for (int i = threadIdx.x; i < 5 * 128; i += 256) red[i] = 0.0f;
    for (int i = threadIdx.x; i < 4 * 128; i += 256) {
        const int k = i >> 7, d = i & 127;
        wsh[i] = (k < hp.nc && d < D) ? hp.wh[(size_t)k * D + d] : 0.0f;
    }
    __syncthreads();
    bool act[KC];
    int cc[KC];
    f4 g[KC], anh[4][KC];
    float sgl[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int j = 0; j < KC; ++j) {
        const int c = l8 + 8 * j;
        act[j] = c < nchunk;
        cc[j] = act[j] ? c : 0;
        g[j] = act[j] ? ld4(p.gamma + 4 * cc[j]) : f4{0, 0, 0, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) anh[k][j] = f4{0, 0, 0, 0};
    }
    const float invD = 1.0f / (float)D;
    // the next row's inputs are requested before the current row is worked on (one trip = one HBM round trip otherwise)
    f4 xn[KC];
    float mun, rsn, gln[4];
    long long xoffn;
    auto fetch = [&](long long r) {
        const long long rc = r < p.rows ? r : p.rows - 1;
        xoffn = head_row_off(p, (unsigned)rc);
        mun = p.mean_in[rc]; rsn = p.rstd_in[rc];
        const long long bi = rc / hp.plane, pix = rc - bi * hp.plane;
#pragma unroll
        for (int k = 0; k < 4; ++k) gln[k] = ld1(GL + (bi * hp.nc + (k < hp.nc ? k : hp.nc - 1)) * hp.plane + pix);
#pragma unroll
        for (int j = 0; j < KC; ++j) xn[j] = ld4(X + xoffn + 4 * cc[j]);
    };
    fetch(row0);
    for (long long r = row0; r - (threadIdx.x >> 3) < p.rows; r += stride) {
        const bool rok = r < p.rows;
        const long long xoff = xoffn;
        const float mu = mun, rstd = rsn;
        float gl[4];
        f4 xc[KC];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            gl[k] = (rok && k < hp.nc) ? gln[k] : 0.0f;
            sgl[k] += gl[k];                        // (every lane of the row counts it: only lane 0's copy is used)
        }
#pragma unroll
        for (int j = 0; j < KC; ++j) xc[j] = xn[j];
        fetch(r + stride);
        f4 nh[KC], dn[KC];
        float s1 = 0.0f, s2 = 0.0f;
        int wo = 0;
        asm volatile("" : "+v"(wo));      // keeps the weight reads inside the loop (hoisted, they are 48 registers again)
#pragma unroll
        for (int j = 0; j < KC; ++j) {
            nh[j] = act[j] ? (xc[j] - mu) * rstd : f4{0, 0, 0, 0};
            f4 dout = *reinterpret_cast<const f4 *>(wsh + wo + 4 * cc[j]) * gl[0];
#pragma unroll
            for (int k = 1; k < 4; ++k) dout += *reinterpret_cast<const f4 *>(wsh + wo + k * 128 + 4 * cc[j]) * gl[k];
#pragma unroll
            for (int k = 0; k < 4; ++k) anh[k][j] += nh[j] * gl[k];
            dn[j] = dout * g[j];
            s1 += (dn[j].x + dn[j].y) + (dn[j].z + dn[j].w);
            s2 += dot4(dn[j], nh[j]);
        }
        const float m1 = sum8(s1) * invD, m2 = sum8(s2) * invD;
#pragma unroll
        for (int j = 0; j < KC; ++j)
            if (act[j] && rok) st4(DX + xoff + 4 * cc[j], (dn[j] - m1 - nh[j] * m2) * rstd);
    }
    // ---- the workgroup's sums: LDS float atomics (once per kernel), then one global atomic per column
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KC; ++j)
        if (act[j])
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int e = 0; e < 4; ++e) atomicAdd(&red[k * 128 + 4 * cc[j] + e], anh[k][j][e]);
    if (l8 == 0)
#pragma unroll
        for (int k = 0; k < 4; ++k) atomicAdd(&red[4 * 128 + k], sgl[k]);
    __syncthreads();
    for (int d = threadIdx.x; d < D; d += 256) {
        const float gd = p.gamma[d], bd = p.beta[d];
        float dg = 0.0f, db = 0.0f;
        for (int k = 0; k < hp.nc; ++k) {
            const float a = red[k * 128 + d], sk = red[4 * 128 + k], w = hp.wh[(size_t)k * D + d];
            atomicAdd(hp.dwh + (size_t)k * D + d, gd * a + bd * sk);      // dWh[k][d] = sum_rows gl[k] * (nh g + b)
            dg += w * a;                                                    // dgamma[d] = sum_rows dout[d] nh[d]
            db += w * sk;                                                   // dbeta[d]  = sum_rows dout[d]
        }
        atomicAdd(p.dgamma + d, dg);
        atomicAdd(p.dbeta + d, db);
    }
}

template <typename TI, typename TO>
int launch_head(bool bwd, const LNParams &p, const HeadParams &hp, hipStream_t s) {
    const int ncp = hp.nc <= 4 ? 4 : (hp.nc <= 10 ? 10 : 16);
    static const bool old_head = getenv("SS2D_HEAD_WIDE_ROWS") != nullptr;   // A/B: the 32-lanes-per-row kernels, read once
    if (ncp == 4 && !old_head) {        // 8 lanes per row
        const int kc = (p.D / 4 + 7) / 8;
        long long nwg = (p.rows + 31) / 32;
        const long long cap = bwd ? 768 : 4096;      // backward: one resident round (144 registers: 3 workgroups per CU)
        if (nwg > cap) nwg = cap;
        const dim3 grid((unsigned)nwg), block(256);
        if (!bwd) {
            if (kc <= 3) ss2d_launch((ln_head4_fwd_kernel<3, TI, TO>), grid, block, 0, s, p, hp);
            else ss2d_launch((ln_head4_fwd_kernel<4, TI, TO>), grid, block, 0, s, p, hp);
        } else {
            if (kc <= 3) ss2d_launch((ln_head4_bwd_kernel<3, TI, TO>), grid, block, 0, s, p, hp);
            else ss2d_launch((ln_head4_bwd_kernel<4, TI, TO>), grid, block, 0, s, p, hp);
        }
        return ss2d_take_launch_rc();
    }
    if (!bwd) {
        long long nwg = (p.rows + 255) / 256;      // 64 rows per wave and trip
        if (nwg > 8192) nwg = 8192;
        const dim3 grid((unsigned)nwg), block(256);
        if (ncp == 4) ss2d_launch((ln_head_fwd_kernel<4, TI, TO>), grid, block, 0, s, p, hp);
        else if (ncp == 10) ss2d_launch((ln_head_fwd_kernel<10, TI, TO>), grid, block, 0, s, p, hp);
        else ss2d_launch((ln_head_fwd_kernel<16, TI, TO>), grid, block, 0, s, p, hp);
    } else {
        const int nwv = ncp == 4 ? 16 : 8;
        long long nwg = (p.rows + nwv * 64 - 1) / (nwv * 64);    // 64 rows per wave and trip
        const long long cap = ncp == 4 ? 256 : 512;              // 4096 waves in all either way
        if (nwg > cap) nwg = cap;
        const dim3 grid((unsigned)nwg), block(nwv * 64);
        if (ncp == 4) ss2d_launch((ln_head_bwd_kernel<4, 16, TI, TO>), grid, block, 0, s, p, hp);
        else if (ncp == 10) ss2d_launch((ln_head_bwd_kernel<10, 8, TI, TO>), grid, block, 0, s, p, hp);
        else ss2d_launch((ln_head_bwd_kernel<16, 8, TI, TO>), grid, block, 0, s, p, hp);
    }
    return ss2d_take_launch_rc();
}

int dispatch_head(bool bwd, int in_dtype, int out_dtype, const LNParams &p, const HeadParams &hp, void *stream, const char *name) {
    if ((in_dtype != SS2D_DTYPE_F32 && in_dtype != SS2D_DTYPE_BF16) || (out_dtype != SS2D_DTYPE_F32 && out_dtype != SS2D_DTYPE_BF16)) {
        ss2d_set_error("%s: unknown dtype %d / %d", name, in_dtype, out_dtype); return -1; }
    if (in_dtype == SS2D_DTYPE_BF16 && out_dtype == SS2D_DTYPE_F32) { ss2d_set_error("%s: bf16 input with fp32 output is not supported", name); return -1; }
    if (p.rows <= 0 || p.D <= 0) { ss2d_set_error("%s: non-positive size", name); return -3; }
    if (p.D % 4 || p.D > 128) { ss2d_set_error("%s: D %d must be a multiple of 4 and <= 128", name, p.D); return -4; }
    if (hp.nc <= 0 || hp.nc > 16) { ss2d_set_error("%s: classes %d must be in 1..16", name, hp.nc); return -4; }
    if (p.rows >= 0x7fffffffLL) { ss2d_set_error("%s: more than 2^31 rows", name); return -6; }
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (in_dtype == SS2D_DTYPE_F32 && out_dtype == SS2D_DTYPE_F32) rc = launch_head<float, float>(bwd, p, hp, s);
    else if (in_dtype == SS2D_DTYPE_F32) rc = launch_head<float, bf16>(bwd, p, hp, s);
    else rc = launch_head<bf16, bf16>(bwd, p, hp, s);
    if (rc) ss2d_set_error("%s: launch failed: %s", name, hipGetErrorString((hipError_t)rc));
    return rc;
}

}  // namespace

// LayerNorm over the rows of a ps x ps pixel un-shuffle of x (PatchExpand / FinalPatchExpand_X4, mamba_sys.py:233-265)
static int check_ps(long long batch, int H, int W, int ps, const char *name) {
    if (batch <= 0 || H <= 0 || W <= 0 || ps <= 0 || ps > 8) { ss2d_set_error("%s: bad un-shuffle geometry (batch %lld, H %d, W %d, ps %d)", name, batch, H, W, ps); return -3; }
    return 0;
}

extern "C" int ss2d_layernorm_unshuffle_fwd(int in_dtype, int out_dtype, const void *x, const float *gamma, const float *beta,
                                            void *out, float *mean, float *rstd, long long batch, int H, int W, int ps,
                                            int D, float eps, void *stream) {
    if (!x || !gamma || !beta || !out) { ss2d_set_error("layernorm_unshuffle_fwd: null pointer"); return -2; }
    if ((mean == nullptr) != (rstd == nullptr)) { ss2d_set_error("layernorm_unshuffle_fwd: mean and rstd go together"); return -2; }
    if (int rc = check_ps(batch, H, W, ps, "layernorm_unshuffle_fwd")) return rc;
    LNParams p{};
    p.x = x; p.gamma = gamma; p.beta = beta; p.out = out; p.mean = mean; p.rstd = rstd; p.D = D; p.eps = eps;
    p.rows = batch * H * ps * W * ps; p.ps = ps; p.psH = H; p.psW = W;
    return dispatch(false, in_dtype, out_dtype, p, stream, "layernorm_unshuffle_fwd");
}

extern "C" int ss2d_layernorm_unshuffle_bwd(int in_dtype, int out_dtype, const void *dout, const void *x, const float *gamma,
                                            const float *mean, const float *rstd, void *dx, float *dgamma, float *dbeta,
                                            long long batch, int H, int W, int ps, int D, void *stream) {
    if (!dout || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta) { ss2d_set_error("layernorm_unshuffle_bwd: null pointer"); return -2; }
    if (int rc = check_ps(batch, H, W, ps, "layernorm_unshuffle_bwd")) return rc;
    LNParams p{};
    p.dout = dout; p.x = x; p.gamma = gamma; p.mean_in = mean; p.rstd_in = rstd; p.dx = dx; p.dgamma = dgamma; p.dbeta = dbeta;
    p.D = D; p.rows = batch * H * ps * W * ps; p.ps = ps; p.psH = H; p.psW = W;
    return dispatch(true, in_dtype, out_dtype, p, stream, "layernorm_unshuffle_bwd");
}

// LayerNorm over the 2x2 neighbourhoods of x (B, H, W, C) -> out (B*H/2*W/2, 4C): the gather of PatchMerging2D
// (mamba_sys.py:213-221: x0 | x1 | x2 | x3, norm) done by the kernel's addressing instead of four strided slices and a cat
static int check_merge(long long batch, int H, int W, int C, const char *name) {
    if (batch <= 0 || H <= 0 || W <= 0 || C <= 0 || (H & 1) || (W & 1) || (C & 3)) {
        ss2d_set_error("%s: needs even H, W and C %% 4 == 0 (batch %lld, H %d, W %d, C %d)", name, batch, H, W, C); return -3; }
    return 0;
}

extern "C" int ss2d_layernorm_merge_fwd(int in_dtype, int out_dtype, const void *x, const float *gamma, const float *beta,
                                        void *out, float *mean, float *rstd, long long batch, int H, int W, int C, float eps,
                                        void *stream) {
    if (!x || !gamma || !beta || !out) { ss2d_set_error("layernorm_merge_fwd: null pointer"); return -2; }
    if ((mean == nullptr) != (rstd == nullptr)) { ss2d_set_error("layernorm_merge_fwd: mean and rstd go together"); return -2; }
    if (int rc = check_merge(batch, H, W, C, "layernorm_merge_fwd")) return rc;
    LNParams p{};
    p.x = x; p.gamma = gamma; p.beta = beta; p.out = out; p.mean = mean; p.rstd = rstd; p.D = 4 * C; p.eps = eps;
    p.rows = batch * (H / 2) * (W / 2); p.ps = -2; p.psH = H; p.psW = W;
    return dispatch(false, in_dtype, out_dtype, p, stream, "layernorm_merge_fwd");
}

extern "C" int ss2d_layernorm_merge_bwd(int in_dtype, int out_dtype, const void *dout, const void *x, const float *gamma,
                                        const float *mean, const float *rstd, void *dx, float *dgamma, float *dbeta,
                                        long long batch, int H, int W, int C, void *stream) {
    if (!dout || !x || !gamma || !mean || !rstd || !dx || !dgamma || !dbeta) { ss2d_set_error("layernorm_merge_bwd: null pointer"); return -2; }
    if (int rc = check_merge(batch, H, W, C, "layernorm_merge_bwd")) return rc;
    LNParams p{};
    p.dout = dout; p.x = x; p.gamma = gamma; p.mean_in = mean; p.rstd_in = rstd; p.dx = dx; p.dgamma = dgamma; p.dbeta = dbeta;
    p.D = 4 * C; p.rows = batch * (H / 2) * (W / 2); p.ps = -2; p.psH = H; p.psW = W;
    return dispatch(true, in_dtype, out_dtype, p, stream, "layernorm_merge_bwd");
}

extern "C" int ss2d_ln_head_fwd(int in_dtype, int out_dtype, const void *x, const float *gamma, const float *beta,
                                const float *w_head, void *logits, float *mean, float *rstd, long long batch, int H, int W,
                                int ps, int D, int classes, float eps, void *stream) {
    if (!x || !gamma || !beta || !w_head || !logits) { ss2d_set_error("ln_head_fwd: null pointer"); return -2; }
    if ((mean == nullptr) != (rstd == nullptr)) { ss2d_set_error("ln_head_fwd: mean and rstd go together"); return -2; }
    if (int rc = check_ps(batch, H, W, ps, "ln_head_fwd")) return rc;
    LNParams p{};
    p.x = x; p.gamma = gamma; p.beta = beta; p.mean = mean; p.rstd = rstd; p.D = D; p.eps = eps;
    p.rows = batch * H * ps * W * ps; p.ps = ps; p.psH = H; p.psW = W;
    HeadParams hp{};
    hp.wh = w_head; hp.logits = logits; hp.nc = classes; hp.plane = (long long)H * ps * W * ps;
    return dispatch_head(false, in_dtype, out_dtype, p, hp, stream, "ln_head_fwd");
}

extern "C" int ss2d_ln_head_bwd(int in_dtype, int out_dtype, const void *dlogits, const void *x, const float *gamma,
                                const float *beta, const float *w_head, const float *mean, const float *rstd, void *dx,
                                float *dgamma, float *dbeta, float *dw_head, long long batch, int H, int W, int ps, int D,
                                int classes, void *stream) {
    if (!dlogits || !x || !gamma || !beta || !w_head || !mean || !rstd || !dx || !dgamma || !dbeta || !dw_head) {
        ss2d_set_error("ln_head_bwd: null pointer"); return -2; }
    if (int rc = check_ps(batch, H, W, ps, "ln_head_bwd")) return rc;
    LNParams p{};
    p.x = x; p.gamma = gamma; p.beta = beta; p.mean_in = mean; p.rstd_in = rstd; p.dx = dx; p.dgamma = dgamma; p.dbeta = dbeta;
    p.D = D; p.rows = batch * H * ps * W * ps; p.ps = ps; p.psH = H; p.psW = W;
    HeadParams hp{};
    hp.wh = w_head; hp.glog = dlogits; hp.dwh = dw_head; hp.nc = classes; hp.plane = (long long)H * ps * W * ps;
    return dispatch_head(true, in_dtype, out_dtype, p, hp, stream, "ln_head_bwd");
}
