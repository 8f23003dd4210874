// Depthwise 3x3 convolution (pad 1, bias) + SiLU for gfx950, forward and backward
// (/root/reference/code/networks/mamba_sys.py:299-308 `conv2d`, :558-559 `act(conv2d(x))`).
//
// The reference permutes the channel-last in_proj output to (B,D,H,W) with a full copy (:558) before
// the conv; here the kernel reads the channel-last tensor directly (row stride ldx lets it read the x
// half of xz in place) and writes (B,D,H,W), so the permute pass disappears.
// A workgroup owns CD channels x (8 rows x TW cols) of output -- 16 x 32 columns for wide images, 32 x 16
// columns once W <= 16 (stages 2/3: 14x14 and 7x7 would leave a 32-wide tile 60-80 % empty).  The input tile
// with its halo sits in LDS as [position][channel] (odd stride): channel-last global reads run along d,
// the compute phase and the (B,D,H,W) accesses run along w.  Each lane walks down one column of one channel
// with a 3x3 register window, so a tap costs one LDS read per three uses.
#include <cstdlib>
#include "ss2d_common.h"
#include "../../include/ss2d.h"

namespace {

constexpr int TH = 8;  // output rows per tile; TW columns and CD channels are template parameters

struct ConvParams {
    const void *x, *dy;          // IO type (fp32 or bf16)
    const float *w, *bias;
    void *y, *dx;                // IO type
    float *dw, *dbias;           // fp32 accumulators (atomics)
    int B, D, H, W, ldx, lddx, tiles_h, tiles_w, tiles_d;
    int reps, groups_h;          // backward: a workgroup walks `reps` (image, row tile) units; groups_h groups per tile column
};

// v or 0.  The mask goes through an empty asm so the compiler cannot see it is all-ones / zero: otherwise it
// rewrites `load & mask` as `keep ? load : 0`, sinks the load under the branch and waits for each one there.
__device__ __forceinline__ float mask_f(float v, bool keep) {
    unsigned m = keep ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(m));
    return __builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & m);
}

// element `off` of a workgroup-uniform base: the address is a scalar base plus a 32-bit lane offset (one VGPR
// per pending load instead of two; the host checks that a batch slab stays below 4 GB)
template <typename IO>
__device__ __forceinline__ float ld1o(const IO *base, unsigned off) {
    return ld1(reinterpret_cast<const IO *>(reinterpret_cast<const char *>(base) + off * (unsigned)sizeof(IO)));
}

template <typename IO>
__device__ __forceinline__ void st1o(IO *base, unsigned off, float v) {
    st1(reinterpret_cast<IO *>(reinterpret_cast<char *>(base) + off * (unsigned)sizeof(IO)), v);
}

// loads x on [h0-HALO, h0+TH+HALO) x [w0-HALO, w0+TW+HALO) for CD channels into s[pos*CS + d], zero outside.
// A lane issues a whole batch of loads unconditionally (addresses clamped into the tensor, padding zeroed with a
// bit mask afterwards) and only then stores to LDS: a load under `if (inside)` is sunk into its branch with a
// full wait per load, i.e. two dozen dependent round trips to HBM per workgroup.  NB batches bound the
// registers the pending loads hold.
// Channel-last: a lane keeps its channel and column slots (q, q+NQ, ..) and walks the rows, so a load is a
// scalar row base plus one of NJ precomputed lane offsets -- no per-load index arithmetic.
template <bool CL, int HALO, int CD, int TW, int NB, typename IO>
__device__ __forceinline__ void load_x_tile(const ConvParams &p, float *s, int b, int d0, int h0, int w0) {
    constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO, CS = CD + 1;
    const int tid = threadIdx.x, dmax = p.D - 1 - d0;
    if (CL) {
        constexpr int NQ = 256 / CD, NJ = (PW + NQ - 1) / NQ, RB = (PH + NB - 1) / NB;
        const IO *__restrict__ x = static_cast<const IO *>(p.x) + (size_t)b * p.H * p.W * p.ldx + d0;
        const int d = tid % CD, q = tid / CD;
        unsigned coff[NJ];
        bool cok[NJ];
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int ww = q + NQ * j, w = w0 - HALO + ww;
            cok[j] = ww < PW && w >= 0 && w < p.W && d <= dmax;
            coff[j] = (unsigned)min(max(w, 0), p.W - 1) * (unsigned)p.ldx + min(d, dmax);
        }
#pragma unroll 1
        for (int r0 = 0; r0 < NB * RB; r0 += RB) {
            float v[RB][NJ];
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int h = h0 - HALO + r0 + r, hc = min(max(h, 0), p.H - 1);
                const IO *row = x + (size_t)hc * p.W * p.ldx;     // workgroup-uniform
                const bool hok = h >= 0 && h < p.H;
#pragma unroll
                for (int j = 0; j < NJ; ++j) v[r][j] = mask_f(ld1o(row, coff[j]), hok && cok[j]);
            }
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
                    if ((NB * RB == PH || r0 + r < PH) && (NQ * (j + 1) <= PW || q + NQ * j < PW))
                        s[((r0 + r) * PW + q + NQ * j) * CS + d] = v[r][j];
        }
    } else {
        constexpr int TOT = PH * PW * CD, NIT = (TOT + 255) / 256, UB = (NIT + NB - 1) / NB;
        const IO *__restrict__ x = static_cast<const IO *>(p.x) + ((size_t)b * p.D + d0) * p.H * p.W;
#pragma unroll 1
        for (int it0 = 0; it0 < NB * UB; it0 += UB) {
            float v[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int i = tid + (it0 + u) * 256;
                const int d = i / (PH * PW), pos = i % (PH * PW), ww = pos % PW, hh = pos / PW;
                const int h = h0 - HALO + hh, w = w0 - HALO + ww;
                const bool ok = d <= dmax && h >= 0 && h < p.H && w >= 0 && w < p.W;
                const int hc = min(max(h, 0), p.H - 1), wc = min(max(w, 0), p.W - 1), dc = min(d, dmax);
                v[u] = mask_f(ld1o(x, (unsigned)((dc * p.H + hc) * p.W + wc)), ok);
            }
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int i = tid + (it0 + u) * 256;
                const int d = i / (PH * PW), pos = i % (PH * PW);
                if (i < TOT) s[pos * CS + d] = v[u];
            }
        }
    }
}

template <bool CL, int CD, int TW, typename IO>
__global__ __launch_bounds__(256) void dwconv_silu_fwd_kernel(const ConvParams p) {
    constexpr int PW = TW + 2, CS = CD + 1, NDL = 256 / TW, NI = CD / NDL;
    __shared__ float s[(TH + 2) * PW * CS];
    int blk = xcd_remap(blockIdx.x, gridDim.x);   // neighbours in the tile order share cache lines / operands: one XCD's L2
    // channel tiles fastest: on a channel-last side the CD channels of a pixel are a 32- or 64-byte piece of its row, and
    // the workgroups that hold the other pieces of the same cache lines should run at the same time on the same XCD
    const int d0 = (blk % p.tiles_d) * CD; blk /= p.tiles_d;
    const int w0 = (blk % p.tiles_w) * TW; blk /= p.tiles_w;
    const int h0 = (blk % p.tiles_h) * TH, b = blk / p.tiles_h;
    load_x_tile<CL, 1, CD, TW, 1, IO>(p, s, b, d0, h0, w0);
    __syncthreads();
    const int wl = threadIdx.x % TW, dl = threadIdx.x / TW;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int d = dl + NDL * i, dc = min(d0 + d, p.D - 1);
        const bool cok = d0 + d < p.D && w0 + wl < p.W;
        float k[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) k[j] = p.w[(size_t)dc * 9 + j];
        const float bias = p.bias ? p.bias[dc] : 0.0f;
        float xr[3][3];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int c = 0; c < 3; ++c) xr[a][c] = s[(a * PW + wl + c) * CS + d];
#pragma unroll
        for (int h = 0; h < TH; ++h) {
#pragma unroll
            for (int c = 0; c < 3; ++c) xr[2][c] = s[((h + 2) * PW + wl + c) * CS + d];
            float o = bias;
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) o = fmaf(k[a * 3 + c], xr[a][c], o);
            if (cok && h0 + h < p.H)
                st1(static_cast<IO *>(p.y) + (((size_t)b * p.D + d0 + d) * p.H + h0 + h) * p.W + w0 + wl, o * sigmoidf_fast(o));
#pragma unroll
            for (int c = 0; c < 3; ++c) { xr[0][c] = xr[1][c]; xr[1][c] = xr[2][c]; }
        }
    }
}

// Backward.  Pass 1 rebuilds ds = dy * silu'(conv(x)+b) on the output tile plus a 1-wide halo (x with a 2-wide
// halo) and, with the same 3x3 register window of x, accumulates dw / dbias for the interior rows.  Pass 2 is
// dx = full correlation of ds with the flipped kernel, again with a sliding window.  A workgroup walks p.reps
// row tiles of its column so the per-channel dw / dbias sums leave as one set of atomics per workgroup.
template <bool CL, int CD, int TW, typename IO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(TW == 32 ? 3 : 2)))   // what the LDS tiles allow per CU
void dwconv_silu_bwd_kernel(const ConvParams p) {
    constexpr int XW = TW + 4, XH = TH + 4, GW = TW + 2, GH = TH + 2, CS = CD + 1, NDL = 256 / TW, NI = CD / NDL;
    constexpr int HTOT = 2 * GH * CD, HIT = (HTOT + 255) / 256;   // the two halo columns of ds
    __shared__ float sx[XH * XW * CS];
    __shared__ float sg[GH * GW * CS];       // ds
    __shared__ float swt[CD * 10];           // 9 taps + bias per channel
    int blk = xcd_remap(blockIdx.x, gridDim.x);   // neighbours in the tile order share cache lines / operands: one XCD's L2
    const int d0 = (blk % p.tiles_d) * CD; blk /= p.tiles_d;     // channel tiles fastest (see the forward)
    const int w0 = (blk % p.tiles_w) * TW; blk /= p.tiles_w;
    const int ug = blk;
    const int tid = threadIdx.x, wl = tid % TW, dl = tid / TW;
    for (int i = tid; i < CD * 10; i += 256) {
        const int d = i / 10, j = i - d * 10;
        float v = 0.0f;
        if (d0 + d < p.D) v = j < 9 ? p.w[(size_t)(d0 + d) * 9 + j] : (p.bias ? p.bias[d0 + d] : 0.0f);
        swt[i] = v;
    }
    float accw[NI][9], accb[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        accb[i] = 0.0f;
#pragma unroll
        for (int j = 0; j < 9; ++j) accw[i][j] = 0.0f;
    }
    for (int rep = 0; rep < p.reps; ++rep) {
        const int unit = ug * p.reps + rep;      // (image, row tile) pairs of this column of tiles
        if (unit >= p.B * p.tiles_h) break;
        const int b = unit / p.tiles_h, h0 = (unit % p.tiles_h) * TH;
        const IO *__restrict__ dy = static_cast<const IO *>(p.dy) + ((size_t)b * p.D + d0) * p.H * p.W;  // this workgroup's CD channels
        // dy on the haloed tile, issued before the x tile so both sets of loads are in flight together
        // (unconditional, clamped, masked -- see load_x_tile).  Zero outside the tensor makes ds zero there.
        float dyv[NI][GH], dyh[HIT];
        const int dmax = p.D - 1 - d0;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int d = dl + NDL * i, w = w0 + wl;
            const int dc = min(d, dmax), wc = min(w, p.W - 1);
#pragma unroll
            for (int hh = 0; hh < GH; ++hh) {
                const int h = h0 - 1 + hh, hc = min(max(h, 0), p.H - 1);
                dyv[i][hh] = mask_f(ld1o(dy, (unsigned)((dc * p.H + hc) * p.W + wc)), d <= dmax && w < p.W && h >= 0 && h < p.H);
            }
        }
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const int idx = tid + it * 256, side = idx & 1, r = idx >> 1, hh = r % GH, d = r / GH;
            const int h = h0 - 1 + hh, w = side ? w0 + TW : w0 - 1;
            const int dc = min(d, dmax), hc = min(max(h, 0), p.H - 1), wc = min(max(w, 0), p.W - 1);
            dyh[it] = mask_f(ld1o(dy, (unsigned)((dc * p.H + hc) * p.W + wc)),
                             idx < HTOT && d <= dmax && w >= 0 && w < p.W && h >= 0 && h < p.H);
        }
        load_x_tile<CL, 2, CD, TW, 3, IO>(p, sx, b, d0, h0, w0);
        __syncthreads();
        // pass 1, interior columns: lane (wl, d) walks down ds column wl + 1
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            __builtin_amdgcn_sched_barrier(0);  // one channel at a time: interleaving them doubles the live windows
            const int d = dl + NDL * i;
            float k[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) k[j] = swt[d * 10 + j];
            const float bias = swt[d * 10 + 9];
            float xr[3][3];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) xr[a][c] = sx[(a * XW + wl + 1 + c) * CS + d];
#pragma unroll
            for (int hh = 0; hh < GH; ++hh) {
#pragma unroll
                for (int c = 0; c < 3; ++c) xr[2][c] = sx[((hh + 2) * XW + wl + 1 + c) * CS + d];
                float o = bias;
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int c = 0; c < 3; ++c) o = fmaf(k[a * 3 + c], xr[a][c], o);
                const float sgm = sigmoidf_fast(o);
                const float g = dyv[i][hh] * (sgm * (1.0f + o * (1.0f - sgm)));
                sg[(hh * GW + wl + 1) * CS + d] = g;
                if (hh >= 1 && hh <= TH) {   // rows of this tile: dw[a][c] += ds[h][w] * x[h+a-1][w+c-1], the window in hand
                    accb[i] += g;
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#pragma unroll
                        for (int c = 0; c < 3; ++c) accw[i][a * 3 + c] = fmaf(g, xr[a][c], accw[i][a * 3 + c]);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) { xr[0][c] = xr[1][c]; xr[1][c] = xr[2][c]; }
            }
        }
        // pass 1, the two halo columns of ds
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const int idx = tid + it * 256, side = idx & 1, r = idx >> 1, hh = r % GH, d = r / GH;
            if (HTOT % 256 != 0 && idx >= HTOT) break;
            const int ww = side ? GW - 1 : 0;
            float o = swt[d * 10 + 9];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) o = fmaf(swt[d * 10 + a * 3 + c], sx[((hh + a) * XW + ww + c) * CS + d], o);
            const float sgm = sigmoidf_fast(o);
            sg[(hh * GW + ww) * CS + d] = dyh[it] * (sgm * (1.0f + o * (1.0f - sgm)));
        }
        __syncthreads();
        // pass 2: dx[h][w] = sum_{a,c} k[a][c] * ds[h - a + 1][w - c + 1]
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            __builtin_amdgcn_sched_barrier(0);
            const int d = dl + NDL * i;
            const bool cok = d0 + d < p.D && w0 + wl < p.W;
            float k[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) k[j] = swt[d * 10 + j];
            float gr[3][3];
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < 3; ++c) gr[a][c] = sg[(a * GW + wl + c) * CS + d];
#pragma unroll
            for (int h = 0; h < TH; ++h) {
#pragma unroll
                for (int c = 0; c < 3; ++c) gr[2][c] = sg[((h + 2) * GW + wl + c) * CS + d];
                float dxv = 0.0f;
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int c = 0; c < 3; ++c) dxv = fmaf(k[a * 3 + c], gr[2 - a][2 - c], dxv);
                if (!CL) {
                    if (cok && h0 + h < p.H) st1(static_cast<IO *>(p.dx) + (((size_t)b * p.D + d0 + d) * p.H + h0 + h) * p.W + w0 + wl, dxv);
                } else {
                    sx[(h * TW + wl) * CS + d] = dxv;  // x tile is dead since the barrier above: staging tile for the channel-last write, which runs along d
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) { gr[0][c] = gr[1][c]; gr[1][c] = gr[2][c]; }
            }
        }
        if (CL) {
            __syncthreads();
            // lane = (channel d, column slot q): consecutive lanes write consecutive channels of one pixel
            constexpr int NQ = 256 / CD, WJ = TW / NQ;
            IO *dxb = static_cast<IO *>(p.dx) + (size_t)b * p.H * p.W * p.lddx + d0;
            const int d = tid % CD, q = tid / CD;
#pragma unroll
            for (int it = 0; it < TH * WJ; ++it) {
                const int w = q + NQ * (it % WJ), h = it / WJ;
                if (d0 + d < p.D && h0 + h < p.H && w0 + w < p.W)
                    st1o(dxb, (unsigned)((h0 + h) * p.W + w0 + w) * (unsigned)p.lddx + d, sx[(h * TW + w) * CS + d]);
            }
            __syncthreads();  // before the next trip refills sx
        }
        // sg: the next trip writes it only after its post-load barrier, this trip's last reads are before that
    }
    // dw / dbias: reduce over the TW lanes that share a channel, one atomic per channel and tap
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int d = d0 + dl + NDL * i;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            float v = j < 9 ? accw[i][j] : accb[i];
#pragma unroll
            for (int o = TW / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, TW);
            if (wl == 0 && d < p.D) {
                if (j < 9) atomicAdd(p.dw + (size_t)d * 9 + j, v);
                else if (p.dbias) atomicAdd(p.dbias + d, v);
            }
        }
    }
}

int setup(ConvParams &p, int dtype, int B, int D, int H, int W, int cl, int ldx, int CD, int TW, const char *name) {
    if (dtype != SS2D_DTYPE_F32 && dtype != SS2D_DTYPE_BF16) { ss2d_set_error("%s: unknown dtype %d", name, dtype); return -1; }
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) { ss2d_set_error("%s: non-positive size", name); return -3; }
    if (cl && ldx < D) { ss2d_set_error("%s: ldx %d < D %d", name, ldx, D); return -4; }
    p.B = B; p.D = D; p.H = H; p.W = W; p.ldx = ldx;
    p.tiles_h = (H + TH - 1) / TH; p.tiles_w = (W + TW - 1) / TW; p.tiles_d = (D + CD - 1) / CD;
    p.reps = 1; p.groups_h = p.tiles_h;
    const long long n = (long long)B * p.tiles_d * p.tiles_h * p.tiles_w;
    if (n > 0x7fffffffLL) { ss2d_set_error("%s: grid too large", name); return -6; }
    // the kernels address one batch slab with 32-bit byte offsets
    const long long slab = cl ? (long long)H * W * (ldx > D ? ldx : D) * 2 : (long long)CD * H * W;   // 2x: lddx is checked by the caller against the same bound
    if (slab * 4 >= 0xffffffffLL || (long long)CD * H * W * 4 >= 0xffffffffLL) { ss2d_set_error("%s: image slab of 4 GB or more", name); return -6; }
    return 0;
}

template <bool FWD, int CD, int TW>
void launch(const ConvParams &p, bool cl, bool h, unsigned n, hipStream_t s) {
#define SS2D_CONV_LAUNCH(CLV, T)                                                                                   \
    do {                                                                                                           \
        if (FWD) ss2d_launch((dwconv_silu_fwd_kernel<CLV, CD, TW, T>), dim3(n), dim3(256), 0, s, p);        \
        else ss2d_launch((dwconv_silu_bwd_kernel<CLV, CD, TW, T>), dim3(n), dim3(256), 0, s, p);            \
    } while (0)
    if (cl) { if (h) SS2D_CONV_LAUNCH(true, bf16); else SS2D_CONV_LAUNCH(true, float); }
    else    { if (h) SS2D_CONV_LAUNCH(false, bf16); else SS2D_CONV_LAUNCH(false, float); }
#undef SS2D_CONV_LAUNCH
}

}  // namespace

extern "C" int ss2d_dwconv3x3_silu_fwd(int dtype, const void *x, const float *w, const float *bias, void *y, int B,
                                       int D, int H, int W, int channel_last_in, int ldx, void *stream) {
    ConvParams p{};
    if (!x || !w || !y) { ss2d_set_error("dwconv3x3_silu_fwd: null pointer"); return -2; }
    const bool narrow = W <= 16;
    int rc = setup(p, dtype, B, D, H, W, channel_last_in, ldx, narrow ? 32 : 16, narrow ? 16 : 32, "dwconv3x3_silu_fwd");
    if (rc) return rc;
    p.x = x; p.w = w; p.bias = bias; p.y = y;
    const unsigned n = (unsigned)((long long)B * p.tiles_d * p.tiles_h * p.tiles_w);
    hipStream_t s = (hipStream_t)stream;
    const bool h = dtype == SS2D_DTYPE_BF16;
    if (narrow) launch<true, 32, 16>(p, channel_last_in != 0, h, n, s);
    else launch<true, 16, 32>(p, channel_last_in != 0, h, n, s);
    rc = ss2d_take_launch_rc();
    if (rc) ss2d_set_error("dwconv3x3_silu_fwd: launch failed: %s", hipGetErrorString((hipError_t)rc));
    return rc;
}

extern "C" int ss2d_dwconv3x3_silu_bwd(int dtype, const void *x, const float *w, const float *bias, const void *dy,
                                       void *dx, float *dw, float *dbias, int B, int D, int H, int W,
                                       int channel_last_in, int ldx, int lddx, void *stream) {
    ConvParams p{};
    if (!x || !w || !dy || !dx || !dw) { ss2d_set_error("dwconv3x3_silu_bwd: null pointer"); return -2; }
    const bool narrow = W <= 16;
    int rc = setup(p, dtype, B, D, H, W, channel_last_in, ldx, narrow ? 32 : 16, narrow ? 16 : 32, "dwconv3x3_silu_bwd");
    if (rc) return rc;
    if (channel_last_in && lddx < D) { ss2d_set_error("dwconv3x3_silu_bwd: lddx %d < D %d", lddx, D); return -4; }
    p.x = x; p.w = w; p.bias = bias; p.dy = dy; p.dx = dx; p.dw = dw;
    p.dbias = bias ? dbias : nullptr; p.lddx = lddx;
    // A workgroup walks `reps` (image, row tile) units of its tile column, so the dw/dbias sums of those units
    // leave as one set of atomics: the atomics execute at the memory side (~5 G/s measured) and are the largest
    // single cost of this kernel when every tile sends its own.  Fastest measured: one resident round of
    // workgroups (256 CUs x 3 of the wide / 2 of the narrow variant), each walking as many units as that takes.
    const long long units = (long long)B * p.tiles_h, cols = (long long)p.tiles_d * p.tiles_w;
    const long long slots = narrow ? 512 : 768;
    long long groups = slots / cols;
    if (groups < 1) groups = 1;
    long long reps = (units + groups - 1) / groups;
    static const long long env_reps = ss2d_env_ll("SS2D_DWCONV_REPS", 0);   // tuning knob, read once
    if (env_reps > 0) reps = env_reps;
    if (reps > units) reps = units;
    if (reps < 1) reps = 1;
    p.groups_h = (int)((units + reps - 1) / reps);
    p.reps = (int)((units + p.groups_h - 1) / p.groups_h);            // balanced groups
    const unsigned n = (unsigned)(cols * p.groups_h);
    hipStream_t s = (hipStream_t)stream;
    const bool h = dtype == SS2D_DTYPE_BF16;
    if (narrow) launch<false, 32, 16>(p, channel_last_in != 0, h, n, s);
    else launch<false, 16, 32>(p, channel_last_in != 0, h, n, s);
    rc = ss2d_take_launch_rc();
    if (rc) ss2d_set_error("dwconv3x3_silu_bwd: launch failed: %s", hipGetErrorString((hipError_t)rc));
    return rc;
}
