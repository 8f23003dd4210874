// Depthwise 3x3 convolution (pad 1, bias) + SiLU for gfx950, forward and backward
// (/root/reference/code/networks/mamba_sys.py:299-308 `conv2d`, :558-559 `act(conv2d(x))`).
//
// The reference permutes the channel-last in_proj output to (B,D,H,W) with a full copy (:558) before
// the conv; here the kernel reads the channel-last tensor directly (row stride ldx lets it read the x
// half of xz in place) and writes (B,D,H,W), so the permute pass disappears.
// A workgroup owns CD (16) channels x (8 rows x 32 cols) of output, small enough for several workgroups per CU; the input tile with its halo sits in LDS
// as [position][channel] (odd stride): channel-last global reads run along d (128-byte runs), the
// compute phase and the (B,D,H,W) writes run along w.
#include "ss2d_common.h"
#include "../../include/ss2d.h"

namespace {

constexpr int TH = 8, TW = 32;  // CD channels per workgroup (template parameter), LDS channel stride CD + 1

struct ConvParams {
    const void *x, *dy;          // IO type (fp32 or bf16)
    const float *w, *bias;
    void *y, *dx;                // IO type
    float *dw, *dbias;           // fp32 accumulators (atomics)
    int B, D, H, W, ldx, lddx, tiles_h, tiles_w, tiles_d;
};

template <int CD>
__device__ __forceinline__ void conv_tile(const ConvParams &p, int &b, int &d0, int &h0, int &w0) {
    int blk = blockIdx.x;
    const int tw = blk % p.tiles_w; blk /= p.tiles_w;
    const int th = blk % p.tiles_h; blk /= p.tiles_h;
    const int td = blk % p.tiles_d; b = blk / p.tiles_d;
    d0 = td * CD; h0 = th * TH; w0 = tw * TW;
}

// loads x on [h0-HALO, h0+TH+HALO) x [w0-HALO, w0+TW+HALO) for CD channels into s[pos*CS + d], zero outside
template <bool CL, int HALO, int CD, typename IO>
__device__ __forceinline__ void load_x_tile(const ConvParams &p, float *s, int b, int d0, int h0, int w0) {
    const IO *__restrict__ x = static_cast<const IO *>(p.x);
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO, CS = CD + 1;
    const int tid = threadIdx.x;
    if (CL) {
        for (int i = tid; i < PH * PW * CD; i += 256) {
            const int d = i % CD, pos = i / CD, ww = pos % PW, hh = pos / PW;
            const int h = h0 - HALO + hh, w = w0 - HALO + ww;
            float v = 0.0f;
            if (d0 + d < p.D && h >= 0 && h < p.H && w >= 0 && w < p.W)
                v = ld1(x + (((size_t)b * p.H + h) * p.W + w) * p.ldx + d0 + d);
            s[pos * CS + d] = v;
        }
    } else {
        for (int i = tid; i < PH * PW * CD; i += 256) {
            const int pos = i % (PH * PW), d = i / (PH * PW), ww = pos % PW, hh = pos / PW;
            const int h = h0 - HALO + hh, w = w0 - HALO + ww;
            float v = 0.0f;
            if (d0 + d < p.D && h >= 0 && h < p.H && w >= 0 && w < p.W)
                v = ld1(x + (((size_t)b * p.D + d0 + d) * p.H + h) * p.W + w);
            s[pos * CS + d] = v;
        }
    }
}

template <bool CL, int CD, typename IO>
__global__ __launch_bounds__(256) void dwconv_silu_fwd_kernel(const ConvParams p) {
    constexpr int PW = TW + 2, CS = CD + 1;
    __shared__ float s[(TH + 2) * PW * CS];
    int b, d0, h0, w0;
    conv_tile<CD>(p, b, d0, h0, w0);
    load_x_tile<CL, 1, CD, IO>(p, s, b, d0, h0, w0);
    __syncthreads();
    const int wl = threadIdx.x & 31, dl = threadIdx.x >> 5;
#pragma unroll 1
    for (int i = 0; i < CD / 8; ++i) {
        const int d = dl + 8 * i;
        if (d0 + d >= p.D) continue;
        float k[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) k[j] = p.w[(size_t)(d0 + d) * 9 + j];
        const float bias = p.bias ? p.bias[d0 + d] : 0.0f;
        if (w0 + wl >= p.W) continue;
#pragma unroll
        for (int h = 0; h < TH; ++h) {
            if (h0 + h >= p.H) break;
            float o = bias;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) o = fmaf(k[a * 3 + c], s[((h + a) * PW + wl + c) * CS + d], o);
            st1(static_cast<IO *>(p.y) + (((size_t)b * p.D + d0 + d) * p.H + h0 + h) * p.W + w0 + wl, o * sigmoidf_fast(o));
        }
    }
}

// Backward: ds = dy * silu'(conv(x)+b) is rebuilt on the output tile plus a 1-wide halo (needs x with a
// 2-wide halo), then dx = full correlation of ds with the flipped kernel, dw/dbias = sums over the tile.
template <bool CL, int CD, typename IO>
__global__ __launch_bounds__(256) void dwconv_silu_bwd_kernel(const ConvParams p) {
    constexpr int XW = TW + 4, XH = TH + 4, GW = TW + 2, GH = TH + 2, CS = CD + 1;
    __shared__ float sx[XH * XW * CS];
    __shared__ float sg[GH * GW * CS];       // ds; reused as the dx staging tile for the channel-last write
    int b, d0, h0, w0;
    conv_tile<CD>(p, b, d0, h0, w0);
    __shared__ float swt[CD * 10];           // 9 taps + bias per channel: read per element below, keep them off the vector memory path
    for (int i = threadIdx.x; i < CD * 10; i += 256) {
        const int d = i / 10, j = i - d * 10;
        float v = 0.0f;
        if (d0 + d < p.D) v = j < 9 ? p.w[(size_t)(d0 + d) * 9 + j] : (p.bias ? p.bias[d0 + d] : 0.0f);
        swt[i] = v;
    }
    load_x_tile<CL, 2, CD, IO>(p, sx, b, d0, h0, w0);
    __syncthreads();
    const int tid = threadIdx.x;
    // ds on the haloed tile; w on the fast lanes so the (B,D,H,W) dy reads run along w
    for (int i = tid; i < GH * GW * CD; i += 256) {
        const int pos = i % (GH * GW), d = i / (GH * GW), ww = pos % GW, hh = pos / GW;
        const int h = h0 - 1 + hh, w = w0 - 1 + ww;
        float g = 0.0f;
        if (d0 + d < p.D && h >= 0 && h < p.H && w >= 0 && w < p.W) {
            float o = swt[d * 10 + 9];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    o = fmaf(swt[d * 10 + a * 3 + c], sx[((hh + a) * XW + ww + c) * CS + d], o);
            const float sgm = sigmoidf_fast(o);
            g = ld1(static_cast<const IO *>(p.dy) + (((size_t)b * p.D + d0 + d) * p.H + h) * p.W + w) * (sgm * (1.0f + o * (1.0f - sgm)));
        }
        sg[pos * CS + d] = g;
    }
    __syncthreads();
    const int wl = tid & 31, dl = tid >> 5;
    float dxr[CD / 8][TH];
#pragma unroll
    for (int i = 0; i < CD / 8; ++i) {
        const int d = dl + 8 * i;
        const bool dok = d0 + d < p.D;
        float k[9], accw[9], accb = 0.0f;
#pragma unroll
        for (int j = 0; j < 9; ++j) { k[j] = swt[d * 10 + j]; accw[j] = 0.0f; }
#pragma unroll
        for (int h = 0; h < TH; ++h) {
            const bool ok = dok && (h0 + h < p.H) && (w0 + wl < p.W);
            // dx[h][w] = sum_{a,c} k[a][c] * ds[h - a + 1][w - c + 1]
            float dxv = 0.0f;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) dxv = fmaf(k[a * 3 + c], sg[((h + 2 - a) * GW + wl + 2 - c) * CS + d], dxv);
            const float g = ok ? sg[((h + 1) * GW + wl + 1) * CS + d] : 0.0f;
            accb += g;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) accw[a * 3 + c] = fmaf(g, sx[((h + 1 + a) * XW + wl + 1 + c) * CS + d], accw[a * 3 + c]);
            if (!CL) {
                if (ok) st1(static_cast<IO *>(p.dx) + (((size_t)b * p.D + d0 + d) * p.H + h0 + h) * p.W + w0 + wl, dxv);
            } else {
                dxr[i][h] = dxv;  // staged through LDS below: the channel-last write runs along d
            }
        }
        // reduce dw / dbias over the 32 lanes that share this channel (half a wave)
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            float v = accw[j];
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
            if (wl == 0 && dok) atomicAdd(p.dw + (size_t)(d0 + d) * 9 + j, v);
        }
        {
            float v = accb;
            for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 32);
            if (wl == 0 && dok && p.dbias) atomicAdd(p.dbias + d0 + d, v);
        }
    }
    if (CL) {
        __syncthreads();  // every read of ds is done: its tile becomes the staging tile
#pragma unroll
        for (int i = 0; i < CD / 8; ++i)
#pragma unroll
            for (int h = 0; h < TH; ++h) sg[(h * TW + wl) * CS + dl + 8 * i] = dxr[i][h];
        __syncthreads();
        for (int i = tid; i < TH * TW * CD; i += 256) {
            const int d = i % CD, pos = i / CD, w = pos & (TW - 1), h = pos >> 5;
            if (d0 + d < p.D && h0 + h < p.H && w0 + w < p.W)
                st1(static_cast<IO *>(p.dx) + (((size_t)b * p.H + h0 + h) * p.W + w0 + w) * p.lddx + d0 + d, sg[(h * TW + w) * CS + d]);
        }
    }
}

constexpr int CONV_CD = 16;

int setup(ConvParams &p, int dtype, int B, int D, int H, int W, int cl, int ldx, const char *name) {
    constexpr int CD = CONV_CD;
    if (dtype != SS2D_DTYPE_F32 && dtype != SS2D_DTYPE_BF16) { ss2d_set_error("%s: unknown dtype %d", name, dtype); return -1; }
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) { ss2d_set_error("%s: non-positive size", name); return -3; }
    if (cl && ldx < D) { ss2d_set_error("%s: ldx %d < D %d", name, ldx, D); return -4; }
    p.B = B; p.D = D; p.H = H; p.W = W; p.ldx = ldx;
    p.tiles_h = (H + TH - 1) / TH; p.tiles_w = (W + TW - 1) / TW; p.tiles_d = (D + CD - 1) / CD;
    const long long n = (long long)B * p.tiles_d * p.tiles_h * p.tiles_w;
    if (n > 0x7fffffffLL) { ss2d_set_error("%s: grid too large", name); return -6; }
    return 0;
}

}  // namespace

extern "C" int ss2d_dwconv3x3_silu_fwd(int dtype, const void *x, const float *w, const float *bias, void *y, int B,
                                       int D, int H, int W, int channel_last_in, int ldx, void *stream) {
    ConvParams p{};
    if (!x || !w || !y) { ss2d_set_error("dwconv3x3_silu_fwd: null pointer"); return -2; }
    int rc = setup(p, dtype, B, D, H, W, channel_last_in, ldx, "dwconv3x3_silu_fwd");
    if (rc) return rc;
    p.x = x; p.w = w; p.bias = bias; p.y = y;
    const unsigned n = (unsigned)((long long)B * p.tiles_d * p.tiles_h * p.tiles_w);
    hipStream_t s = (hipStream_t)stream;
    const bool h = dtype == SS2D_DTYPE_BF16;
    if (channel_last_in) { if (h) hipLaunchKernelGGL((dwconv_silu_fwd_kernel<true, CONV_CD, bf16>), dim3(n), dim3(256), 0, s, p); else hipLaunchKernelGGL((dwconv_silu_fwd_kernel<true, CONV_CD, float>), dim3(n), dim3(256), 0, s, p); }
    else                 { if (h) hipLaunchKernelGGL((dwconv_silu_fwd_kernel<false, CONV_CD, bf16>), dim3(n), dim3(256), 0, s, p); else hipLaunchKernelGGL((dwconv_silu_fwd_kernel<false, CONV_CD, float>), dim3(n), dim3(256), 0, s, p); }
    rc = (int)hipGetLastError();
    if (rc) ss2d_set_error("dwconv3x3_silu_fwd: launch failed: %s", hipGetErrorString((hipError_t)rc));
    return rc;
}

extern "C" int ss2d_dwconv3x3_silu_bwd(int dtype, const void *x, const float *w, const float *bias, const void *dy,
                                       void *dx, float *dw, float *dbias, int B, int D, int H, int W,
                                       int channel_last_in, int ldx, int lddx, void *stream) {
    ConvParams p{};
    if (!x || !w || !dy || !dx || !dw) { ss2d_set_error("dwconv3x3_silu_bwd: null pointer"); return -2; }
    int rc = setup(p, dtype, B, D, H, W, channel_last_in, ldx, "dwconv3x3_silu_bwd");
    if (rc) return rc;
    if (channel_last_in && lddx < D) { ss2d_set_error("dwconv3x3_silu_bwd: lddx %d < D %d", lddx, D); return -4; }
    p.x = x; p.w = w; p.bias = bias; p.dy = dy; p.dx = dx; p.dw = dw;
    p.dbias = bias ? dbias : nullptr; p.lddx = lddx;
    const unsigned n = (unsigned)((long long)B * p.tiles_d * p.tiles_h * p.tiles_w);
    hipStream_t s = (hipStream_t)stream;
    const bool h = dtype == SS2D_DTYPE_BF16;
    if (channel_last_in) { if (h) hipLaunchKernelGGL((dwconv_silu_bwd_kernel<true, CONV_CD, bf16>), dim3(n), dim3(256), 0, s, p); else hipLaunchKernelGGL((dwconv_silu_bwd_kernel<true, CONV_CD, float>), dim3(n), dim3(256), 0, s, p); }
    else                 { if (h) hipLaunchKernelGGL((dwconv_silu_bwd_kernel<false, CONV_CD, bf16>), dim3(n), dim3(256), 0, s, p); else hipLaunchKernelGGL((dwconv_silu_bwd_kernel<false, CONV_CD, float>), dim3(n), dim3(256), 0, s, p); }
    rc = (int)hipGetLastError();
    if (rc) ss2d_set_error("dwconv3x3_silu_bwd: launch failed: %s", hipGetErrorString((hipError_t)rc));
    return rc;
}
