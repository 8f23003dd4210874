// LayerNorm over the channel axis for gfx950, forward and backward: the `ln_1` of every VSSBlock
// (/root/reference/code/networks/mamba_sys.py:573-585, `x + drop_path(self_attention(ln_1(x)))`) and the norms of
// the patch merge / expand layers (:196, :226, :249).  Same structure as ss2d_normgate.hip: one wave per token
// row, the row lives in registers, wave butterfly sums, fp32 arithmetic; dgamma / dbeta go through a
// workgroup-level LDS reduction and one float atomic per column and workgroup (at most 256 workgroups).
// Input and output storage types are independent: under bf16 autocast the fp32 residual stream is normalised
// straight into the bf16 tensor the next GEMM reads (no separate cast pass); dx comes back in the input's type.
#include "ss2d_common.h"
#include "../../include/ss2d.h"

namespace {

constexpr int MAXK = 6;  // 4 * 64 * MAXK = 1536 channels max

struct LNParams {
    const void *x, *dout;
    const float *gamma, *beta, *mean_in, *rstd_in;
    void *out, *dx;
    float *mean, *rstd, *dgamma, *dbeta;
    long long rows;
    int D;
    float eps;
    int ps, psH, psW;   // ps > 0: rows are gathered through a ps x ps pixel un-shuffle of x (B, psH, psW, ps*ps*D)
};

// Element offset in x (and dx) of normalised row r.  Plain: r * D.  Un-shuffle (the einops pattern
// 'b h w (p1 p2 c) -> b (h p1) (w p2) c' of mamba_sys.py:244, :262): output row (b, h*ps + p1, w*ps + p2) is the
// c-long run ((((b*H + h)*W + w)*ps + p1)*ps + p2) * c of the expand Linear's output -- the rearranging copy (231 MB
// each way in front of the last LayerNorm) becomes address arithmetic.
__device__ __forceinline__ long long ln_row_off(const LNParams &p, long long r) {
    if (p.ps == 0) return r * p.D;
    const int W4 = p.psW * p.ps, H4 = p.psH * p.ps;
    const int w4 = (int)(r % W4);
    const long long t = r / W4;
    const int h4 = (int)(t % H4);
    const long long b = t / H4;
    const int h = h4 / p.ps, p1 = h4 % p.ps, w = w4 / p.ps, p2 = w4 % p.ps;
    return ((((b * p.psH + h) * p.psW + w) * p.ps + p1) * p.ps + p2) * (long long)p.D;
}

// sum over the LW (64 or 32) lanes that share a row
template <int LW>
__device__ __forceinline__ float wave_sum_ln(float v) {
#pragma unroll
    for (int o = LW / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

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
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int c = lane + LW * k;
        g[k] = c < nchunk ? ld4(p.gamma + 4 * c) : f4{0, 0, 0, 0};
        b[k] = c < nchunk ? ld4(p.beta + 4 * c) : f4{0, 0, 0, 0};
    }
    const float invD = 1.0f / (float)D;
    for (long long r = wave0; r < p.rows; r += nwaves) {
        f4 xv[K];
        float s = 0.0f;
        const long long xoff = ln_row_off(p, r);
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = lane + LW * k;
            xv[k] = c < nchunk ? ld4(X + xoff + 4 * c) : f4{0, 0, 0, 0};
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

template <int K, int NWV, int LW, typename TI, typename TO>
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
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const int c = lane + LW * k;
        g[k] = c < nchunk ? ld4(p.gamma + 4 * c) : f4{0, 0, 0, 0};
        ag[k] = f4{0, 0, 0, 0};
        ab[k] = f4{0, 0, 0, 0};
    }
    const float invD = 1.0f / (float)D;
    for (long long r = wave0; r < p.rows; r += nwaves) {
        const float mu = p.mean_in[r], rstd = p.rstd_in[r];
        const long long xoff = ln_row_off(p, r);
        f4 nh[K], dn[K];
        float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = lane + LW * k;
            nh[k] = f4{0, 0, 0, 0};
            dn[k] = f4{0, 0, 0, 0};
            if (c < nchunk) {
                const f4 gv = ld4(G + r * D + 4 * c);
                nh[k] = (ld4(X + xoff + 4 * c) - mu) * rstd;
                dn[k] = gv * g[k];
                ag[k] += gv * nh[k];
                ab[k] += gv;
                const f4 t = dn[k] * nh[k];
                s1 += (dn[k].x + dn[k].y) + (dn[k].z + dn[k].w);
                s2 += (t.x + t.y) + (t.z + t.w);
            }
        }
        const float m1 = wave_sum_ln<LW>(s1) * invD, m2 = wave_sum_ln<LW>(s2) * invD;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = lane + LW * k;
            if (c < nchunk) st4(DX + xoff + 4 * c, (dn[k] - m1 - nh[k] * m2) * rstd);
        }
    }
    wg_colsum2<K, NWV, LW>(red, ag, ab, D, p.dgamma, p.dbeta);
}

template <typename TI, typename TO>
void launch_ln(bool bwd, const LNParams &p, hipStream_t s) {
    const int nchunk = p.D / 4;
    const bool half = nchunk <= 32;              // two rows per wave
    const int K = half ? 1 : (nchunk + 63) / 64;
    const int rpw = half ? 2 : 1;
    if (!bwd) {
        long long nwg = (p.rows + 4 * rpw - 1) / (4 * rpw);
        if (nwg > 8192) nwg = 8192;
        const dim3 grid((unsigned)nwg), block(256);
        if (half) { hipLaunchKernelGGL((ln_fwd_kernel<1, 32, TI, TO>), grid, block, 0, s, p); return; }
        switch (K) {
            case 1: hipLaunchKernelGGL((ln_fwd_kernel<1, 64, TI, TO>), grid, block, 0, s, p); break;
            case 2: hipLaunchKernelGGL((ln_fwd_kernel<2, 64, TI, TO>), grid, block, 0, s, p); break;
            case 3: hipLaunchKernelGGL((ln_fwd_kernel<3, 64, TI, TO>), grid, block, 0, s, p); break;
            case 4: hipLaunchKernelGGL((ln_fwd_kernel<4, 64, TI, TO>), grid, block, 0, s, p); break;
            case 5: hipLaunchKernelGGL((ln_fwd_kernel<5, 64, TI, TO>), grid, block, 0, s, p); break;
            default: hipLaunchKernelGGL((ln_fwd_kernel<6, 64, TI, TO>), grid, block, 0, s, p); break;
        }
        return;
    }
    const int nwv = K <= 2 ? 16 : 4;
    long long nwg = (p.rows + nwv * rpw - 1) / (nwv * rpw);
    if (nwg > 256) nwg = 256;
    const dim3 grid((unsigned)nwg), block(nwv * 64);
    if (half) { hipLaunchKernelGGL((ln_bwd_kernel<1, 16, 32, TI, TO>), grid, block, 0, s, p); return; }
    switch (K) {
        case 1: hipLaunchKernelGGL((ln_bwd_kernel<1, 16, 64, TI, TO>), grid, block, 0, s, p); break;
        case 2: hipLaunchKernelGGL((ln_bwd_kernel<2, 16, 64, TI, TO>), grid, block, 0, s, p); break;
        case 3: hipLaunchKernelGGL((ln_bwd_kernel<3, 4, 64, TI, TO>), grid, block, 0, s, p); break;
        case 4: hipLaunchKernelGGL((ln_bwd_kernel<4, 4, 64, TI, TO>), grid, block, 0, s, p); break;
        case 5: hipLaunchKernelGGL((ln_bwd_kernel<5, 4, 64, TI, TO>), grid, block, 0, s, p); break;
        default: hipLaunchKernelGGL((ln_bwd_kernel<6, 4, 64, TI, TO>), grid, block, 0, s, p); break;
    }
}

int dispatch(bool bwd, int in_dtype, int out_dtype, const LNParams &p, void *stream, const char *name) {
    if ((in_dtype != SS2D_DTYPE_F32 && in_dtype != SS2D_DTYPE_BF16) || (out_dtype != SS2D_DTYPE_F32 && out_dtype != SS2D_DTYPE_BF16)) {
        ss2d_set_error("%s: unknown dtype %d / %d", name, in_dtype, out_dtype); return -1; }
    if (in_dtype == SS2D_DTYPE_BF16 && out_dtype == SS2D_DTYPE_F32) { ss2d_set_error("%s: bf16 input with fp32 output is not supported", name); return -1; }
    if (p.rows <= 0 || p.D <= 0) { ss2d_set_error("%s: non-positive size", name); return -3; }
    if (p.D % 4 || p.D > 256 * MAXK) { ss2d_set_error("%s: D %d must be a multiple of 4 and <= %d", name, p.D, 256 * MAXK); return -4; }
    hipStream_t s = (hipStream_t)stream;
    if (in_dtype == SS2D_DTYPE_F32 && out_dtype == SS2D_DTYPE_F32) launch_ln<float, float>(bwd, p, s);
    else if (in_dtype == SS2D_DTYPE_F32) launch_ln<float, bf16>(bwd, p, s);
    else launch_ln<bf16, bf16>(bwd, p, s);
    const int rc = (int)hipGetLastError();
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
