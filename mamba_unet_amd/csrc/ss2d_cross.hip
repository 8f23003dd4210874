// CrossScan / CrossMerge for gfx950 (/root/reference/code/networks/mamba_sys.py:403-404 and :445-449).
//
// With p = h*W + w (row-major) and q = w*H + h (column-major):
//   scan : xs[b,0,d,p] = xs[b,1,d,q] = xs[b,2,d,L-1-p] = xs[b,3,d,L-1-q] = x[b,d,h,w]
//   merge: y[b,d,h,w] = ys[b,0,d,p] + ys[b,1,d,q] + ys[b,2,d,L-1-p] + ys[b,3,d,L-1-q]
// The two are adjoint, so cross_scan_bwd = merge and cross_merge_bwd = scan.
// A workgroup owns a 3-D tile (32 channels x 16 rows x 16 cols) held in LDS: row-major directions are
// touched with w on the fast lanes, column-major ones with h on the fast lanes (64-byte runs either way),
// and a channel-last side (B,H,W,D) with d on the fast lanes (128-byte runs).
#include "ss2d_common.h"
#include "../../include/ss2d.h"

namespace {

constexpr int TD = 32, TH = 16, TW = 16;
constexpr int SW = TW + 1;              // padded w stride
constexpr int SD = TH * SW + 1;         // padded channel stride (odd -> conflict-free channel-fast access)

struct CrossParams {
    const void *in;    // scan: x ; merge: ys   (IO type: fp32 or bf16)
    void *out;         // scan: xs; merge: y
    int B, D, H, W, tiles_h, tiles_w, tiles_d;
    // slot layout of the direction axis: row-major pair (slot 0, slot rmB), column-major pair (cmA, cmB); revB = the
    // second slot of each pair is stored reversed (the reference's layout: rmB 2, cmA 1, cmB 3, revB 1).  The
    // mirror-free layouts used with ss2d_selective_scan_dir_*: 4 slots (0,1 | 2,3), revB 0, or 2 slots (0 | 1).
    int nslots, rmB, cmA, cmB, revB;
};

__device__ __forceinline__ void tile_coords(const CrossParams &p, int &b, int &d0, int &h0, int &w0) {
    int blk = blockIdx.x;
    const int tw = blk % p.tiles_w; blk /= p.tiles_w;
    const int th = blk % p.tiles_h; blk /= p.tiles_h;
    const int td = blk % p.tiles_d; b = blk / p.tiles_d;
    d0 = td * TD; h0 = th * TH; w0 = tw * TW;
}

// x side is channel-last (B,H,W,D) when CL, else (B,D,H,W)
template <bool CL, typename IO>
__global__ __launch_bounds__(256) void cross_scan_kernel(const CrossParams p) {
    __shared__ float s[TD * SD];
    const IO *__restrict__ in = static_cast<const IO *>(p.in);
    IO *__restrict__ out = static_cast<IO *>(p.out);
    int b, d0, h0, w0;
    tile_coords(p, b, d0, h0, w0);
    const int tid = threadIdx.x, H = p.H, W = p.W, D = p.D;
    const size_t L = (size_t)H * W;
    if (CL) {
        for (int i = tid; i < TD * TH * TW; i += 256) {
            const int d = i & (TD - 1), pos = i >> 5, w = pos & (TW - 1), h = pos >> 4;
            float v = 0.0f;
            if (d0 + d < D && h0 + h < H && w0 + w < W) v = ld1(in + (((size_t)b * H + h0 + h) * W + w0 + w) * D + d0 + d);
            s[d * SD + h * SW + w] = v;
        }
    } else {
        for (int i = tid; i < TD * TH * TW; i += 256) {
            const int w = i & (TW - 1), h = (i >> 4) & (TH - 1), d = i >> 8;
            float v = 0.0f;
            if (d0 + d < D && h0 + h < H && w0 + w < W) v = ld1(in + (((size_t)b * D + d0 + d) * H + h0 + h) * W + w0 + w);
            s[d * SD + h * SW + w] = v;
        }
    }
    __syncthreads();
    // row-major directions 0 and 2: w on the fast lanes
    for (int i = tid; i < TD * TH * TW; i += 256) {
        const int w = i & (TW - 1), h = (i >> 4) & (TH - 1), d = i >> 8;
        if (d0 + d < D && h0 + h < H && w0 + w < W) {
            const float v = s[d * SD + h * SW + w];
            const size_t pp = (size_t)(h0 + h) * W + w0 + w;
            IO *o0 = out + (((size_t)b * p.nslots + 0) * D + d0 + d) * L;
            st1(o0 + pp, v);
            if (p.nslots == 4) st1(out + (((size_t)b * 4 + p.rmB) * D + d0 + d) * L + (p.revB ? L - 1 - pp : pp), v);
        }
    }
    // column-major directions 1 and 3: h on the fast lanes
    for (int i = tid; i < TD * TH * TW; i += 256) {
        const int h = i & (TH - 1), w = (i >> 4) & (TW - 1), d = i >> 8;
        if (d0 + d < D && h0 + h < H && w0 + w < W) {
            const float v = s[d * SD + h * SW + w];
            const size_t q = (size_t)(w0 + w) * H + h0 + h;
            IO *o1 = out + (((size_t)b * p.nslots + p.cmA) * D + d0 + d) * L;
            st1(o1 + q, v);
            if (p.nslots == 4) st1(out + (((size_t)b * 4 + p.cmB) * D + d0 + d) * L + (p.revB ? L - 1 - q : q), v);
        }
    }
}

// sums in fp32 (LDS), one rounding at the end for bf16 output
template <bool CL, typename IO>
__global__ __launch_bounds__(256) void cross_merge_kernel(const CrossParams p) {
    __shared__ float s[TD * SD];
    const IO *__restrict__ in = static_cast<const IO *>(p.in);
    IO *__restrict__ out = static_cast<IO *>(p.out);
    int b, d0, h0, w0;
    tile_coords(p, b, d0, h0, w0);
    const int tid = threadIdx.x, H = p.H, W = p.W, D = p.D;
    const size_t L = (size_t)H * W;
    for (int i = tid; i < TD * TH * TW; i += 256) {
        const int w = i & (TW - 1), h = (i >> 4) & (TH - 1), d = i >> 8;
        float v = 0.0f;
        if (d0 + d < D && h0 + h < H && w0 + w < W) {
            const size_t pp = (size_t)(h0 + h) * W + w0 + w;
            const IO *i0 = in + (((size_t)b * p.nslots + 0) * D + d0 + d) * L;
            v = ld1(i0 + pp);
            if (p.nslots == 4) v += ld1(in + (((size_t)b * 4 + p.rmB) * D + d0 + d) * L + (p.revB ? L - 1 - pp : pp));
        }
        s[d * SD + h * SW + w] = v;
    }
    __syncthreads();
    for (int i = tid; i < TD * TH * TW; i += 256) {
        const int h = i & (TH - 1), w = (i >> 4) & (TW - 1), d = i >> 8;
        if (d0 + d < D && h0 + h < H && w0 + w < W) {
            const size_t q = (size_t)(w0 + w) * H + h0 + h;
            const IO *i1 = in + (((size_t)b * p.nslots + p.cmA) * D + d0 + d) * L;
            // the reference adds in the order k0 + k2 + k1 + k3 (mamba_sys.py:448)
            float acc = s[d * SD + h * SW + w] + ld1(i1 + q);
            if (p.nslots == 4) acc += ld1(in + (((size_t)b * 4 + p.cmB) * D + d0 + d) * L + (p.revB ? L - 1 - q : q));
            s[d * SD + h * SW + w] = acc;
        }
    }
    __syncthreads();
    if (CL) {
        for (int i = tid; i < TD * TH * TW; i += 256) {
            const int d = i & (TD - 1), pos = i >> 5, w = pos & (TW - 1), h = pos >> 4;
            if (d0 + d < D && h0 + h < H && w0 + w < W)
                st1(out + (((size_t)b * H + h0 + h) * W + w0 + w) * D + d0 + d, s[d * SD + h * SW + w]);
        }
    } else {
        for (int i = tid; i < TD * TH * TW; i += 256) {
            const int w = i & (TW - 1), h = (i >> 4) & (TH - 1), d = i >> 8;
            if (d0 + d < D && h0 + h < H && w0 + w < W)
                st1(out + (((size_t)b * D + d0 + d) * H + h0 + h) * W + w0 + w, s[d * SD + h * SW + w]);
        }
    }
}

// ---- vector variants: every lane moves 4 consecutive elements along the fast axis of the phase (w, h or d),
// i.e. 16-byte (fp32) / 8-byte (bf16) global accesses; the LDS tile and its odd strides are those of the scalar
// kernels above.  Require H % 4 == 0, W % 4 == 0 (and D % 4 == 0 on a channel-last side) and aligned pointers;
// the host falls back to the scalar kernels otherwise (14x14 and 7x7 stages).
__device__ __forceinline__ f4 rev4(f4 v) { return f4{v.w, v.z, v.y, v.x}; }

template <bool CL, typename IO>
__global__ __launch_bounds__(256) void cross_scan_vec_kernel(const CrossParams p) {
    __shared__ float s[TD * SD];
    const IO *__restrict__ in = static_cast<const IO *>(p.in);
    IO *__restrict__ out = static_cast<IO *>(p.out);
    int b, d0, h0, w0;
    tile_coords(p, b, d0, h0, w0);
    const int tid = threadIdx.x, H = p.H, W = p.W, D = p.D;
    const size_t L = (size_t)H * W;
    if (CL) {  // 4 consecutive channels per lane
        for (int i = tid; i < TD * TH * TW / 4; i += 256) {
            const int d = (i & 7) * 4, pos = i >> 3, w = pos & (TW - 1), h = pos >> 4;
            f4 v = f4{0, 0, 0, 0};
            if (d0 + d < D && h0 + h < H && w0 + w < W) v = ld4(in + (((size_t)b * H + h0 + h) * W + w0 + w) * D + d0 + d);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[(d + e) * SD + h * SW + w] = v[e];
        }
    } else {   // 4 consecutive columns per lane
        for (int i = tid; i < TD * TH * TW / 4; i += 256) {
            const int w = (i & 3) * 4, h = (i >> 2) & (TH - 1), d = i >> 6;
            f4 v = f4{0, 0, 0, 0};
            if (d0 + d < D && h0 + h < H && w0 + w < W) v = ld4(in + (((size_t)b * D + d0 + d) * H + h0 + h) * W + w0 + w);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[d * SD + h * SW + w + e] = v[e];
        }
    }
    __syncthreads();
    for (int i = tid; i < TD * TH * TW / 4; i += 256) {   // directions 0 and 2: runs along w
        const int w = (i & 3) * 4, h = (i >> 2) & (TH - 1), d = i >> 6;
        if (d0 + d < D && h0 + h < H && w0 + w < W) {
            f4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = s[d * SD + h * SW + w + e];
            const size_t pp = (size_t)(h0 + h) * W + w0 + w;
            st4(out + (((size_t)b * p.nslots + 0) * D + d0 + d) * L + pp, v);
            if (p.nslots == 4) {
                IO *o2 = out + (((size_t)b * 4 + p.rmB) * D + d0 + d) * L;
                if (p.revB) st4(o2 + (L - 4 - pp), rev4(v)); else st4(o2 + pp, v);
            }
        }
    }
    for (int i = tid; i < TD * TH * TW / 4; i += 256) {   // directions 1 and 3: runs along h
        const int h = (i & 3) * 4, w = (i >> 2) & (TW - 1), d = i >> 6;
        if (d0 + d < D && h0 + h < H && w0 + w < W) {
            f4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = s[d * SD + (h + e) * SW + w];
            const size_t q = (size_t)(w0 + w) * H + h0 + h;
            st4(out + (((size_t)b * p.nslots + p.cmA) * D + d0 + d) * L + q, v);
            if (p.nslots == 4) {
                IO *o3 = out + (((size_t)b * 4 + p.cmB) * D + d0 + d) * L;
                if (p.revB) st4(o3 + (L - 4 - q), rev4(v)); else st4(o3 + q, v);
            }
        }
    }
}

template <bool CL, typename IO>
__global__ __launch_bounds__(256) void cross_merge_vec_kernel(const CrossParams p) {
    __shared__ float s[TD * SD];
    const IO *__restrict__ in = static_cast<const IO *>(p.in);
    IO *__restrict__ out = static_cast<IO *>(p.out);
    int b, d0, h0, w0;
    tile_coords(p, b, d0, h0, w0);
    const int tid = threadIdx.x, H = p.H, W = p.W, D = p.D;
    const size_t L = (size_t)H * W;
    for (int i = tid; i < TD * TH * TW / 4; i += 256) {
        const int w = (i & 3) * 4, h = (i >> 2) & (TH - 1), d = i >> 6;
        f4 v = f4{0, 0, 0, 0};
        if (d0 + d < D && h0 + h < H && w0 + w < W) {
            const size_t pp = (size_t)(h0 + h) * W + w0 + w;
            v = ld4(in + (((size_t)b * p.nslots + 0) * D + d0 + d) * L + pp);
            if (p.nslots == 4) {
                const IO *i2 = in + (((size_t)b * 4 + p.rmB) * D + d0 + d) * L;
                v += p.revB ? rev4(ld4(i2 + (L - 4 - pp))) : ld4(i2 + pp);
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) s[d * SD + h * SW + w + e] = v[e];
    }
    __syncthreads();
    for (int i = tid; i < TD * TH * TW / 4; i += 256) {
        const int h = (i & 3) * 4, w = (i >> 2) & (TW - 1), d = i >> 6;
        if (d0 + d < D && h0 + h < H && w0 + w < W) {
            const size_t q = (size_t)(w0 + w) * H + h0 + h;
            const f4 v1 = ld4(in + (((size_t)b * p.nslots + p.cmA) * D + d0 + d) * L + q);
            f4 v3 = f4{0, 0, 0, 0};
            if (p.nslots == 4) {
                const IO *i3 = in + (((size_t)b * 4 + p.cmB) * D + d0 + d) * L;
                v3 = p.revB ? rev4(ld4(i3 + (L - 4 - q))) : ld4(i3 + q);
            }
            // the reference adds in the order k0 + k2 + k1 + k3 (mamba_sys.py:448)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float acc = s[d * SD + (h + e) * SW + w] + v1[e];
                s[d * SD + (h + e) * SW + w] = p.nslots == 4 ? acc + v3[e] : acc;
            }
        }
    }
    __syncthreads();
    if (CL) {
        for (int i = tid; i < TD * TH * TW / 4; i += 256) {
            const int d = (i & 7) * 4, pos = i >> 3, w = pos & (TW - 1), h = pos >> 4;
            if (d0 + d < D && h0 + h < H && w0 + w < W) {
                f4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = s[(d + e) * SD + h * SW + w];
                st4(out + (((size_t)b * H + h0 + h) * W + w0 + w) * D + d0 + d, v);
            }
        }
    } else {
        for (int i = tid; i < TD * TH * TW / 4; i += 256) {
            const int w = (i & 3) * 4, h = (i >> 2) & (TH - 1), d = i >> 6;
            if (d0 + d < D && h0 + h < H && w0 + w < W) {
                f4 v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = s[d * SD + h * SW + w + e];
                st4(out + (((size_t)b * D + d0 + d) * H + h0 + h) * W + w0 + w, v);
            }
        }
    }
}

int run(bool merge, bool cl, int layout, int dtype, const void *in, void *out, int B, int D, int H, int W, void *stream, const char *name) {
    if (dtype != SS2D_DTYPE_F32 && dtype != SS2D_DTYPE_BF16) { ss2d_set_error("%s: unknown dtype %d", name, dtype); return -1; }
    if (!in || !out) { ss2d_set_error("%s: null pointer", name); return -2; }
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) { ss2d_set_error("%s: non-positive size", name); return -3; }
    CrossParams p;
    p.in = in; p.out = out; p.B = B; p.D = D; p.H = H; p.W = W;
    if (layout == SS2D_DIRS_REFERENCE) { p.nslots = 4; p.rmB = 2; p.cmA = 1; p.cmB = 3; p.revB = 1; }
    else if (layout == SS2D_DIRS_MIRROR_FREE_4) { p.nslots = 4; p.rmB = 1; p.cmA = 2; p.cmB = 3; p.revB = 0; }
    else if (layout == SS2D_DIRS_MIRROR_FREE_2) { p.nslots = 2; p.rmB = 0; p.cmA = 1; p.cmB = 1; p.revB = 0; }
    else { ss2d_set_error("%s: unknown direction layout %d", name, layout); return -7; }
    p.tiles_h = (H + TH - 1) / TH; p.tiles_w = (W + TW - 1) / TW; p.tiles_d = (D + TD - 1) / TD;
    const long long n = (long long)B * p.tiles_d * p.tiles_h * p.tiles_w;
    if (n > 0x7fffffffLL) { ss2d_set_error("%s: grid too large", name); return -6; }
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)n), block(256);
    const bool h = dtype == SS2D_DTYPE_BF16;
    const uintptr_t amask = h ? 7 : 15;   // 4 elements
    const bool vec = (H % 4 == 0) && (W % 4 == 0) && (!cl || D % 4 == 0) && ((((uintptr_t)in | (uintptr_t)out) & amask) == 0);
    if (vec) {
        if (merge) {
            if (cl) { if (h) hipLaunchKernelGGL((cross_merge_vec_kernel<true, bf16>), grid, block, 0, s, p); else hipLaunchKernelGGL((cross_merge_vec_kernel<true, float>), grid, block, 0, s, p); }
            else    { if (h) hipLaunchKernelGGL((cross_merge_vec_kernel<false, bf16>), grid, block, 0, s, p); else hipLaunchKernelGGL((cross_merge_vec_kernel<false, float>), grid, block, 0, s, p); }
        } else {
            if (cl) { if (h) hipLaunchKernelGGL((cross_scan_vec_kernel<true, bf16>), grid, block, 0, s, p); else hipLaunchKernelGGL((cross_scan_vec_kernel<true, float>), grid, block, 0, s, p); }
            else    { if (h) hipLaunchKernelGGL((cross_scan_vec_kernel<false, bf16>), grid, block, 0, s, p); else hipLaunchKernelGGL((cross_scan_vec_kernel<false, float>), grid, block, 0, s, p); }
        }
    } else if (merge) {
        if (cl) { if (h) hipLaunchKernelGGL((cross_merge_kernel<true, bf16>), grid, block, 0, s, p); else hipLaunchKernelGGL((cross_merge_kernel<true, float>), grid, block, 0, s, p); }
        else    { if (h) hipLaunchKernelGGL((cross_merge_kernel<false, bf16>), grid, block, 0, s, p); else hipLaunchKernelGGL((cross_merge_kernel<false, float>), grid, block, 0, s, p); }
    } else {
        if (cl) { if (h) hipLaunchKernelGGL((cross_scan_kernel<true, bf16>), grid, block, 0, s, p); else hipLaunchKernelGGL((cross_scan_kernel<true, float>), grid, block, 0, s, p); }
        else    { if (h) hipLaunchKernelGGL((cross_scan_kernel<false, bf16>), grid, block, 0, s, p); else hipLaunchKernelGGL((cross_scan_kernel<false, float>), grid, block, 0, s, p); }
    }
    const int rc = (int)hipGetLastError();
    if (rc) ss2d_set_error("%s: launch failed: %s", name, hipGetErrorString((hipError_t)rc));
    return rc;
}

}  // namespace

extern "C" int ss2d_cross_scan_fwd(int dtype, const void *x, void *xs, int B, int D, int H, int W, void *stream) {
    return run(false, false, SS2D_DIRS_REFERENCE, dtype, x, xs, B, D, H, W, stream, "cross_scan_fwd");
}
extern "C" int ss2d_cross_scan_bwd(int dtype, const void *dxs, void *dx, int B, int D, int H, int W, void *stream) {
    return run(true, false, SS2D_DIRS_REFERENCE, dtype, dxs, dx, B, D, H, W, stream, "cross_scan_bwd");
}
extern "C" int ss2d_cross_merge_fwd(int dtype, const void *ys, void *y, int B, int D, int H, int W, int channel_last, void *stream) {
    return run(true, channel_last != 0, SS2D_DIRS_REFERENCE, dtype, ys, y, B, D, H, W, stream, "cross_merge_fwd");
}
extern "C" int ss2d_cross_merge_bwd(int dtype, const void *dy, void *dys, int B, int D, int H, int W, int channel_last, void *stream) {
    return run(false, channel_last != 0, SS2D_DIRS_REFERENCE, dtype, dy, dys, B, D, H, W, stream, "cross_merge_bwd");
}
// generic forms: `layout` picks the slot layout of the direction axis (include/ss2d.h)
extern "C" int ss2d_dirs_scatter(int dtype, int layout, const void *x, void *xs, int B, int D, int H, int W, int x_channel_last, void *stream) {
    return run(false, x_channel_last != 0, layout, dtype, x, xs, B, D, H, W, stream, "dirs_scatter");
}
extern "C" int ss2d_dirs_gather(int dtype, int layout, const void *ys, void *y, int B, int D, int H, int W, int y_channel_last, void *stream) {
    return run(true, y_channel_last != 0, layout, dtype, ys, y, B, D, H, W, stream, "dirs_gather");
}
