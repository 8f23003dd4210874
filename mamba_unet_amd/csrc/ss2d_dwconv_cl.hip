// Depthwise 3x3 convolution (pad 1, bias) + SiLU with channel-last tensors on BOTH sides, forward and backward
// (/root/reference/code/networks/mamba_sys.py:299-308 `conv2d`, :558-559 `act(conv2d(x))`), for gfx950.
//
// ss2d_dwconv.hip reads a channel-last x and writes (B, D, H, W): its tile is transposed through LDS and every global
// access moves one element, which makes it instruction-bound (~110 instructions per output element in the backward).
// Here a lane owns FOUR consecutive channels of a pixel column and walks down the rows with a 3 x 3 window of float4 in
// registers: every global access is 8 (bf16) or 16 (fp32) bytes along the channel axis, the arithmetic is packed
// (v_pk_fma_f32 on channel pairs), nothing is transposed.  The caller keeps the convolution output channel-last
// (ss2d_dirs_scatter reads either layout) and hands the backward a channel-last dy (ss2d_dirs_gather6_cl_ld).
//   forward : lanes are dealt over the flattened (column, channel quad) axis of an image row -- consecutive lanes,
//             consecutive addresses, no tile edges -- and a workgroup walks a band of rows.  No LDS, no barrier.
//   backward: ds = dy * silu'(conv(x) + b) is needed on a 3 x 3 neighbourhood of every output, i.e. by the neighbouring
//             lanes: a workgroup owns TW columns x NQ channel quads (TW * NQ = 256 lanes) x 8 rows, pass 1 computes ds for
//             all TW columns and 10 rows into LDS (and dw / dbias for the interior in registers), pass 2 correlates it with
//             the flipped taps for the TW - 2 interior columns.  Tiles advance by TW - 2 columns.
// Rows, columns and channels outside the tensor are never branched around: loads go through buffer descriptors and get an
// out-of-range offset (they return 0), stores likewise (they are dropped).
#include "ss2d_common.h"
#include "../../include/ss2d.h"

namespace {

constexpr int TH = 8;              // backward: output rows per unit
constexpr unsigned OOR = (unsigned)SS2D_BUF_OOR;

struct ClConvParams {
    const void *x, *dy;          // IO type (fp32 or bf16), channel-last: pixel pitch ldx / lddy elements
    const float *w, *bias;       // (D, 9), (D) or null
    void *y, *dx;                // IO type, channel-last: pixel pitch ldy / lddx
    float *dw, *dbias;           // fp32 accumulators (atomics)
    int B, D, H, W, ldx, ldy, lddy, lddx;
    int chunks, bands, rows_per_band;   // forward: 256-lane chunks of a row's W * D/4 quads; row bands per image
    int tiles_w, tiles_d, tiles_h;      // backward
    int reps, groups;                   // backward: a workgroup walks `reps` (image, row band) units
};

template <typename IO> struct RawOf;
template <> struct RawOf<float> { typedef u32x4 type; };
template <> struct RawOf<bf16> { typedef u32x2 type; };

template <typename IO>
__device__ __forceinline__ typename RawOf<IO>::type raw_load4(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff);
template <> __device__ __forceinline__ u32x4 raw_load4<float>(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
}
template <> __device__ __forceinline__ u32x2 raw_load4<bf16>(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff) {
    return __builtin_amdgcn_raw_buffer_load_b64(rs, voff, soff, 0);
}
__device__ __forceinline__ f4 cvt4(u32x4 r) { return __builtin_bit_cast(f4, r); }
__device__ __forceinline__ f4 cvt4(u32x2 r) {
    const f2 a = bf16pair_to_f2(r.x), b = bf16pair_to_f2(r.y);
    return f4{a.x, a.y, b.x, b.y};
}
template <typename IO> __device__ __forceinline__ void raw_store4(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, f4 v);
template <> __device__ __forceinline__ void raw_store4<float>(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, f4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, voff, soff, 0);
}
template <> __device__ __forceinline__ void raw_store4<bf16>(__amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned soff, f4 v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, __builtin_convertvector(v, bf16x4)), rs, voff, soff, 0);
}

__device__ __forceinline__ f4 fma4(f4 a, f4 b, f4 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f4 splat4(float v) { return f4{v, v, v, v}; }
__device__ __forceinline__ f4 sigmoid4(f4 o) {
    const f4 t = o * splat4(-SS2D_LOG2E);
    const f4 e = f4{hw_exp2(t.x), hw_exp2(t.y), hw_exp2(t.z), hw_exp2(t.w)} + splat4(1.0f);
    return f4{hw_rcp(e.x), hw_rcp(e.y), hw_rcp(e.z), hw_rcp(e.w)};
}

// the 9 taps + bias of the lane's four channels d .. d+3 (d % 4 == 0): 36 consecutive floats of w, regrouped per tap
__device__ __forceinline__ void load_taps(const float *__restrict__ w, const float *__restrict__ bias, int d, bool ok, f4 (&k)[9], f4 &bv) {
    float flat[36];
    const f4 *src = reinterpret_cast<const f4 *>(w + (size_t)(ok ? d : 0) * 9);     // 144-byte steps: 16-byte aligned
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const f4 v = src[i];
        flat[4 * i] = v.x; flat[4 * i + 1] = v.y; flat[4 * i + 2] = v.z; flat[4 * i + 3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) k[j] = ok ? f4{flat[j], flat[9 + j], flat[18 + j], flat[27 + j]} : f4{0, 0, 0, 0};
    bv = (ok && bias) ? *reinterpret_cast<const f4 *>(bias + d) : f4{0, 0, 0, 0};
}

// conv at one pixel from the window rows (top, mid, bot), each three columns wide
__device__ __forceinline__ f4 conv9(const f4 (&k)[9], f4 bv, const f4 (&t)[3], const f4 (&m)[3], const f4 (&b)[3]) {
    f4 o = bv;
#pragma unroll
    for (int c = 0; c < 3; ++c) o = fma4(k[c], t[c], o);
#pragma unroll
    for (int c = 0; c < 3; ++c) o = fma4(k[3 + c], m[c], o);
#pragma unroll
    for (int c = 0; c < 3; ++c) o = fma4(k[6 + c], b[c], o);
    return o;
}

template <typename IO>
__global__ __launch_bounds__(256) void dwconv_cl_fwd_kernel(const ClConvParams p) {
    typedef typename RawOf<IO>::type Raw;
    constexpr unsigned ES = sizeof(IO);
    int blk = xcd_remap(blockIdx.x, gridDim.x);      // the chunks of a band share the rows they read: one XCD's L2
    const int chunk = blk % p.chunks; blk /= p.chunks;
    const int band = blk % p.bands, b = blk / p.bands;
    const int DQ = p.D >> 2, idx = chunk * 256 + threadIdx.x;
    const bool lane_ok = idx < p.W * DQ;
    const int w = lane_ok ? idx / DQ : 0, d = lane_ok ? (idx - w * DQ) * 4 : 0;
    const int h_begin = band * p.rows_per_band, h_end = min(h_begin + p.rows_per_band, p.H);
    const size_t npix = (size_t)p.H * p.W;
    const unsigned xbytes = (unsigned)(((npix - 1) * p.ldx + p.D) * ES), ybytes = (unsigned)(((npix - 1) * p.ldy + p.D) * ES);
    const IO *xb = static_cast<const IO *>(p.x) + (size_t)b * npix * p.ldx;
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(static_cast<IO *>(p.y) + (size_t)b * npix * p.ldy, ybytes);
    unsigned xoff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int wc = w - 1 + j;
        xoff[j] = (lane_ok && wc >= 0 && wc < p.W) ? (unsigned)(wc * p.ldx + d) * ES : OOR;
    }
    const unsigned yoff = lane_ok ? (unsigned)(w * p.ldy + d) * ES : OOR;
    const unsigned xrow = (unsigned)(p.W * p.ldx) * ES, yrow = (unsigned)(p.W * p.ldy) * ES;
    f4 k[9], bv;
    load_taps(p.w, p.bias, d, lane_ok, k, bv);

    // three rows in flight: slot s holds row h with (h - h_begin + 1) % 3 == s, raw until it enters the window
    Raw raw[3][3];
    f4 win[3][3];
    auto request = [&](int h, Raw (&r)[3]) {
        const bool ok = h >= 0 && h < p.H && h <= h_end;         // workgroup-uniform: a row outside the image reads zeros; rows past the band's halo are not fetched
        const __amdgpu_buffer_rsrc_t rs = make_rsrc(xb, ok ? xbytes : 0u);
        const unsigned so = ok ? (unsigned)h * xrow : 0u;
#pragma unroll
        for (int j = 0; j < 3; ++j) r[j] = raw_load4<IO>(rs, xoff[j], so);
    };
    auto enter = [&](const Raw (&r)[3], f4 (&wr)[3]) {
#pragma unroll
        for (int j = 0; j < 3; ++j) wr[j] = cvt4(r[j]);
    };
    request(h_begin - 1, raw[0]);
    request(h_begin, raw[1]);
    request(h_begin + 1, raw[2]);
    enter(raw[0], win[0]);
    request(h_begin + 2, raw[0]);
    enter(raw[1], win[1]);
    request(h_begin + 3, raw[1]);
    // output row h: window rows (h-1, h, h+1) in slots (A, B, C); row h+1 enters slot C, whose raw registers then take row h+4
#define SS2D_CL_FWD_STEP(A, B_, C, hh)                                                        \
    do {                                                                                      \
        enter(raw[C], win[C]);                                                                \
        request((hh) + 4, raw[C]);                                                            \
        const f4 o = conv9(k, bv, win[A], win[B_], win[C]);                                   \
        raw_store4<IO>(rs_y, yoff, (unsigned)(hh) * yrow, o * sigmoid4(o));                   \
        __builtin_amdgcn_sched_barrier(0);                                                    \
    } while (0)
    for (int h = h_begin; h < h_end; h += 3) {
        SS2D_CL_FWD_STEP(0, 1, 2, h);
        if (h + 1 >= h_end) break;
        SS2D_CL_FWD_STEP(1, 2, 0, h + 1);
        if (h + 2 >= h_end) break;
        SS2D_CL_FWD_STEP(2, 0, 1, h + 2);
    }
#undef SS2D_CL_FWD_STEP
}

// Backward.  Lane (c, q): column w = w0 - 1 + c of the tile (c = 0 and c = TW - 1 are halo columns whose ds the
// neighbours need), channels d0 + 4q .. +3.
template <int TW, int NQ, typename IO>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))
void dwconv_cl_bwd_kernel(const ClConvParams p) {
    static_assert(TW * NQ == 256, "one lane per (column, channel quad)");
    typedef typename RawOf<IO>::type Raw;
    constexpr unsigned ES = sizeof(IO);
    constexpr int GH = TH + 2;
    __shared__ f4 sg[GH * 256];                      // ds: [row][c * NQ + q]; reused for the dw / dbias reduction at the end
    int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int d0 = (blk % p.tiles_d) * (NQ * 4); blk /= p.tiles_d;     // channel tiles fastest: the pieces of a pixel's cache lines together
    const int w0 = (blk % p.tiles_w) * (TW - 2); blk /= p.tiles_w;
    const int ug = blk;
    const int tid = threadIdx.x, q = tid % NQ, c = tid / NQ;
    const int d = d0 + 4 * q, w = w0 - 1 + c;
    const bool ch_ok = d < p.D;
    const bool interior = c >= 1 && c <= TW - 2 && w < p.W && ch_ok;   // columns this workgroup owns (dw / dbias / dx)
    const size_t npix = (size_t)p.H * p.W;
    const unsigned xbytes = (unsigned)(((npix - 1) * p.ldx + p.D) * ES), gbytes = (unsigned)(((npix - 1) * p.lddy + p.D) * ES),
                   obytes = (unsigned)(((npix - 1) * p.lddx + p.D) * ES);
    unsigned xoff[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int wc = w - 1 + j;
        xoff[j] = (ch_ok && wc >= 0 && wc < p.W) ? (unsigned)(wc * p.ldx + d) * ES : OOR;
    }
    const unsigned goff = (ch_ok && w >= 0 && w < p.W) ? (unsigned)(w * p.lddy + d) * ES : OOR;
    const unsigned ooff = interior ? (unsigned)(w * p.lddx + d) * ES : OOR;
    const unsigned xrow = (unsigned)(p.W * p.ldx) * ES, grow = (unsigned)(p.W * p.lddy) * ES, orow = (unsigned)(p.W * p.lddx) * ES;
    f4 k[9], bv;
    load_taps(p.w, p.bias, d, ch_ok, k, bv);
    const f4 own = splat4(interior ? 1.0f : 0.0f);
    f4 accw[9], accb = f4{0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 9; ++j) accw[j] = f4{0, 0, 0, 0};
    // LDS neighbours of pass 2 (clamped: the halo lanes compute a value that is never stored)
    const int cl = c >= 1 ? tid - NQ : tid, cr = c <= TW - 2 ? tid + NQ : tid;

    const int nunits = p.B * p.tiles_h, ubeg = ug * p.reps, uend = min(ubeg + p.reps, nunits);
    for (int unit = ubeg; unit < uend; ++unit) {
        const int b = unit / p.tiles_h, h0 = (unit % p.tiles_h) * TH;
        const IO *xb = static_cast<const IO *>(p.x) + (size_t)b * npix * p.ldx;
        const IO *gb = static_cast<const IO *>(p.dy) + (size_t)b * npix * p.lddy;
        const __amdgpu_buffer_rsrc_t rs_o = make_rsrc(static_cast<IO *>(p.dx) + (size_t)b * npix * p.lddx, obytes);
        Raw raw[3][3], rawg[3];
        f4 win[3][3];
        auto request = [&](int h, Raw (&r)[3]) {
            const bool ok = h >= 0 && h < p.H;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(xb, ok ? xbytes : 0u);
            const unsigned so = ok ? (unsigned)h * xrow : 0u;
#pragma unroll
            for (int j = 0; j < 3; ++j) r[j] = raw_load4<IO>(rs, xoff[j], so);
        };
        auto request_g = [&](int h, Raw &r) {
            const bool ok = h >= 0 && h < p.H;
            const __amdgpu_buffer_rsrc_t rs = make_rsrc(gb, ok ? gbytes : 0u);
            r = raw_load4<IO>(rs, goff, ok ? (unsigned)h * grow : 0u);
        };
        auto enter = [&](const Raw (&r)[3], f4 (&wr)[3]) {
#pragma unroll
            for (int j = 0; j < 3; ++j) wr[j] = cvt4(r[j]);
        };
        // pass 1: ds rows hh = 0 .. GH-1 are image rows h0-1 .. h0+TH; ds row hh needs x rows h0-2+hh .. h0+hh.
        // x row h0-2+i lives in slot i % 3, dy row h0-1+hh in slot hh % 3; three rows of each are in flight.
        request(h0 - 2, raw[0]);
        request(h0 - 1, raw[1]);
        request(h0, raw[2]);
        request_g(h0 - 1, rawg[0]);
        request_g(h0, rawg[1]);
        request_g(h0 + 1, rawg[2]);
        enter(raw[0], win[0]);
        request(h0 + 1, raw[0]);
        enter(raw[1], win[1]);
        request(h0 + 2, raw[1]);
#pragma unroll
        for (int hh = 0; hh < GH; ++hh) {
            const int A = hh % 3, B_ = (hh + 1) % 3, C = (hh + 2) % 3;      // slots of x rows h-1, h, h+1 (compile-time after unrolling)
            enter(raw[C], win[C]);
            if (hh + 5 < GH + 2) request(h0 - 2 + hh + 5, raw[C]);
            const f4 dyv = cvt4(rawg[A]);
            if (hh + 3 < GH) request_g(h0 - 1 + hh + 3, rawg[A]);
            const f4 o = conv9(k, bv, win[A], win[B_], win[C]);
            const f4 sgm = sigmoid4(o);
            const f4 g = dyv * (sgm * fma4(o, splat4(1.0f) - sgm, splat4(1.0f)));
            sg[hh * 256 + tid] = g;
            if (hh >= 1 && hh <= TH) {      // rows of this unit: dw[a][cc] += ds[h][w] * x[h+a-1][w+cc-1], the window in hand
                const f4 gm = g * own;
                accb += gm;
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    accw[cc] = fma4(gm, win[A][cc], accw[cc]);
                    accw[3 + cc] = fma4(gm, win[B_][cc], accw[3 + cc]);
                    accw[6 + cc] = fma4(gm, win[C][cc], accw[6 + cc]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        // pass 2: dx[h][w] = sum_{a,cc} k[a][cc] * ds[h - a + 1][w - cc + 1]; ds row of image row h0 + r + a' is sg row r + a' + 1 - 1
        {
            f4 gr[3][3];
#pragma unroll
            for (int a = 0; a < 2; ++a) { gr[a][0] = sg[a * 256 + cl]; gr[a][1] = sg[a * 256 + tid]; gr[a][2] = sg[a * 256 + cr]; }
#pragma unroll
            for (int r = 0; r < TH; ++r) {
                const int S0 = r % 3, S1 = (r + 1) % 3, S2 = (r + 2) % 3;    // sg rows r, r+1, r+2 = image rows h-1, h, h+1
                gr[S2][0] = sg[(r + 2) * 256 + cl]; gr[S2][1] = sg[(r + 2) * 256 + tid]; gr[S2][2] = sg[(r + 2) * 256 + cr];
                f4 dxv = f4{0, 0, 0, 0};
#pragma unroll
                for (int cc = 0; cc < 3; ++cc) {
                    dxv = fma4(k[cc], gr[S2][2 - cc], dxv);          // a = 0: ds row h+1
                    dxv = fma4(k[3 + cc], gr[S1][2 - cc], dxv);      // a = 1: ds row h
                    dxv = fma4(k[6 + cc], gr[S0][2 - cc], dxv);      // a = 2: ds row h-1
                }
                const int h = h0 + r;
                raw_store4<IO>(rs_o, h < p.H ? ooff : OOR, (unsigned)min(h, p.H - 1) * orow, dxv);
            }
        }
        __syncthreads();      // before the next unit overwrites sg
    }
    // dw / dbias: the TW lanes that share a channel quad add up through LDS, one atomic per channel and tap
#pragma unroll
    for (int j = 0; j < 9; ++j) sg[j * 256 + tid] = accw[j];
    sg[9 * 256 + tid] = accb;
    __syncthreads();
    for (int i = tid; i < 10 * NQ; i += 256) {
        const int j = i / NQ, qq = i - j * NQ;
        f4 s = f4{0, 0, 0, 0};
#pragma unroll 4
        for (int cc = 0; cc < TW; ++cc) s += sg[j * 256 + cc * NQ + qq];
        const int dd = d0 + 4 * qq;
        if (dd < p.D) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float v = s[e];
                if (j < 9) atomicAdd(p.dw + (size_t)(dd + e) * 9 + j, v);
                else if (p.dbias) atomicAdd(p.dbias + dd + e, v);
            }
        }
    }
}

int check_common(int dtype, const void *x, const float *w, int B, int D, int H, int W, long long ld_in, const char *name) {
    if (dtype != SS2D_DTYPE_F32 && dtype != SS2D_DTYPE_BF16) { ss2d_set_error("%s: unknown dtype %d", name, dtype); return -1; }
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) { ss2d_set_error("%s: non-positive size", name); return -3; }
    if (D % 4) { ss2d_set_error("%s: D %d must be a multiple of 4 (use ss2d_dwconv3x3_silu_fwd/_bwd)", name, D); return -4; }
    if (((uintptr_t)w & 15) != 0) { ss2d_set_error("%s: weight must be 16-byte aligned", name); return -5; }
    (void)x; (void)ld_in;
    return 0;
}

// pixel pitch `ld` (elements) of a channel-last tensor: >= D, vector accesses aligned, one image below 2^31 bytes
int check_pitch(const void *ptr, long long ld, int D, int H, int W, int es, const char *what, const char *name) {
    if (ld < D || ld % 4) { ss2d_set_error("%s: pixel pitch of %s (%lld) must be >= D and a multiple of 4", name, what, ld); return -4; }
    if (((uintptr_t)ptr & (uintptr_t)(4 * es - 1)) != 0) { ss2d_set_error("%s: %s must be aligned to 4 elements", name, what); return -5; }
    if ((long long)H * W * ld * es >= 0x7fff0000LL) { ss2d_set_error("%s: one image of %s is 2 GB or more", name, what); return -6; }
    return 0;
}

}  // namespace

extern "C" int ss2d_dwconv3x3_silu_cl_fwd(int dtype, const void *x, const float *w, const float *bias, void *y, int B, int D,
                                          int H, int W, int ldx, int ldy, void *stream) {
    const char *name = "dwconv3x3_silu_cl_fwd";
    if (!x || !w || !y) { ss2d_set_error("%s: null pointer", name); return -2; }
    int rc = check_common(dtype, x, w, B, D, H, W, ldx, name);
    if (rc) return rc;
    const int es = dtype == SS2D_DTYPE_BF16 ? 2 : 4;
    if ((rc = check_pitch(x, ldx, D, H, W, es, "x", name))) return rc;
    if ((rc = check_pitch(y, ldy, D, H, W, es, "y", name))) return rc;
    if (bias && ((uintptr_t)bias & 15)) { ss2d_set_error("%s: bias must be 16-byte aligned", name); return -5; }
    ClConvParams p{};
    p.x = x; p.w = w; p.bias = bias; p.y = y; p.B = B; p.D = D; p.H = H; p.W = W; p.ldx = ldx; p.ldy = ldy;
    p.chunks = (int)(((long long)W * (D / 4) + 255) / 256);
    // row bands: enough workgroups for four rounds of the chip, at least 2 rows each (a band re-reads two halo rows)
    long long bands = (1024 + (long long)B * p.chunks - 1) / ((long long)B * p.chunks);
    if (bands > (H + 1) / 2) bands = (H + 1) / 2;
    if (bands < 1) bands = 1;
    p.rows_per_band = (int)((H + bands - 1) / bands);
    p.bands = (H + p.rows_per_band - 1) / p.rows_per_band;
    const long long n = (long long)B * p.bands * p.chunks;
    if (n > 0x7fffffffLL) { ss2d_set_error("%s: grid too large", name); return -6; }
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SS2D_DTYPE_BF16) ss2d_launch((dwconv_cl_fwd_kernel<bf16>), dim3((unsigned)n), dim3(256), 0, s, p);
    else ss2d_launch((dwconv_cl_fwd_kernel<float>), dim3((unsigned)n), dim3(256), 0, s, p);
    rc = ss2d_take_launch_rc();
    if (rc) ss2d_set_error("%s: launch failed: %s", name, hipGetErrorString((hipError_t)rc));
    return rc;
}

extern "C" int ss2d_dwconv3x3_silu_cl_bwd(int dtype, const void *x, const float *w, const float *bias, const void *dy, void *dx,
                                          float *dw, float *dbias, int B, int D, int H, int W, int ldx, int lddy, int lddx,
                                          void *stream) {
    const char *name = "dwconv3x3_silu_cl_bwd";
    if (!x || !w || !dy || !dx || !dw) { ss2d_set_error("%s: null pointer", name); return -2; }
    int rc = check_common(dtype, x, w, B, D, H, W, ldx, name);
    if (rc) return rc;
    const int es = dtype == SS2D_DTYPE_BF16 ? 2 : 4;
    if ((rc = check_pitch(x, ldx, D, H, W, es, "x", name))) return rc;
    if ((rc = check_pitch(dy, lddy, D, H, W, es, "dy", name))) return rc;
    if ((rc = check_pitch(dx, lddx, D, H, W, es, "dx", name))) return rc;
    if (bias && ((uintptr_t)bias & 15)) { ss2d_set_error("%s: bias must be 16-byte aligned", name); return -5; }
    ClConvParams p{};
    p.x = x; p.w = w; p.bias = bias; p.dy = dy; p.dx = dx; p.dw = dw; p.dbias = bias ? dbias : nullptr;
    p.B = B; p.D = D; p.H = H; p.W = W; p.ldx = ldx; p.lddy = lddy; p.lddx = lddx;
    const bool narrow = W <= 14;                       // 16 columns (14 owned) x 16 quads, else 32 columns (30 owned) x 8 quads
    const int tw = narrow ? 16 : 32, cd = narrow ? 64 : 32;
    p.tiles_w = (W + tw - 3) / (tw - 2); p.tiles_d = (D + cd - 1) / cd; p.tiles_h = (H + TH - 1) / TH;
    // one resident round of workgroups (two per CU), each walking as many (image, row band) units as that takes: the dw /
    // dbias sums of those units leave as one set of atomics
    const long long units = (long long)B * p.tiles_h, cols = (long long)p.tiles_d * p.tiles_w;
    long long groups = 512 / cols;
    if (groups < 1) groups = 1;
    long long reps = (units + groups - 1) / groups;
    static const long long env_reps = ss2d_env_ll("SS2D_DWCONV_REPS", 0);   // tuning / test knob, read once
    if (env_reps > 0) reps = env_reps;
    if (reps > units) reps = units;
    if (reps < 1) reps = 1;
    p.groups = (int)((units + reps - 1) / reps);
    p.reps = (int)((units + p.groups - 1) / p.groups);
    const long long n = cols * p.groups;
    if (n > 0x7fffffffLL) { ss2d_set_error("%s: grid too large", name); return -6; }
    hipStream_t s = (hipStream_t)stream;
    const bool h = dtype == SS2D_DTYPE_BF16;
    if (narrow) { if (h) ss2d_launch((dwconv_cl_bwd_kernel<16, 16, bf16>), dim3((unsigned)n), dim3(256), 0, s, p);
                  else ss2d_launch((dwconv_cl_bwd_kernel<16, 16, float>), dim3((unsigned)n), dim3(256), 0, s, p); }
    else { if (h) ss2d_launch((dwconv_cl_bwd_kernel<32, 8, bf16>), dim3((unsigned)n), dim3(256), 0, s, p);
           else ss2d_launch((dwconv_cl_bwd_kernel<32, 8, float>), dim3((unsigned)n), dim3(256), 0, s, p); }
    rc = ss2d_take_launch_rc();
    if (rc) ss2d_set_error("%s: launch failed: %s", name, hipGetErrorString((hipError_t)rc));
    return rc;
}
