// CrossScan / CrossMerge for gfx950 (/root/reference/code/networks/mamba_sys.py:403-404 and :445-449).
//
// With p = h*W + w (row-major) and q = w*H + h (column-major):
//   scan : xs[b,0,d,p] = xs[b,1,d,q] = xs[b,2,d,L-1-p] = xs[b,3,d,L-1-q] = x[b,d,h,w]
//   merge: y[b,d,h,w] = ys[b,0,d,p] + ys[b,1,d,q] + ys[b,2,d,L-1-p] + ys[b,3,d,L-1-q]
// The two are adjoint, so cross_scan_bwd = merge and cross_merge_bwd = scan.
// A workgroup owns a 3-D tile (32 channels x 16 rows x 16 cols; 128 x 8 x 8 for images up to 8x8) held in LDS: row-major directions are
// touched with w on the fast lanes, column-major ones with h on the fast lanes (64-byte runs either way),
// and a channel-last side (B,H,W,D) with d on the fast lanes (128-byte runs).
#include "ss2d_common.h"
#include "../../include/ss2d.h"

namespace {

// Tile held in LDS: TD channels x TH rows x TW cols, padded w stride and an odd channel stride (conflict-free
// channel-fast access).  32x16x16 in general; 128x8x8 for images of at most 8x8 (the 7x7 stage), where a 16x16
// tile would be 80 % padding.
template <int TD_, int TH_, int TW_>
struct Tile {
    static constexpr int TD = TD_, TH = TH_, TW = TW_, SW = TW_ + 1, SD = TH_ * SW + 1, N = TD_ * TH_ * TW_;
};
using TileBig = Tile<32, 16, 16>;
using TileSmall = Tile<128, 8, 8>;

struct CrossParams {
    const void *in;    // scan: x ; merge: ys   (IO type: fp32 or bf16)
    const void *in2;   // merge, EXTRA kernels: a second, 2-slot input (row-major | column-major, un-reversed) added to the sum
    void *out;         // scan: xs; merge: y
    int B, D, H, W, tiles_h, tiles_w, tiles_d;
    // slot layout of the direction axis: row-major pair (slot 0, slot rmB), column-major pair (cmA, cmB); revB = the
    // second slot of each pair is stored reversed (the reference's layout: rmB 2, cmA 1, cmB 3, revB 1).  The
    // mirror-free layouts used with ss2d_selective_scan_dir_*: 4 slots (0,1 | 2,3), revB 0, or 2 slots (0 | 1).
    int nslots, rmB, cmA, cmB, revB;
    long long lrow;    // row pitch (elements) of the direction-side tensor: H*W, or more when the caller pads its rows
};

template <typename T>
__device__ __forceinline__ void tile_coords(const CrossParams &p, int &b, int &d0, int &h0, int &w0) {
    // neighbouring tiles share cache lines (a tile row is a 32-byte piece of a 112-byte image row): keep them on one XCD's
    // L2 instead of dealing them round-robin over the eight
    int blk = xcd_remap(blockIdx.x, gridDim.x);
    const int tw = blk % p.tiles_w; blk /= p.tiles_w;
    const int th = blk % p.tiles_h; blk /= p.tiles_h;
    const int td = blk % p.tiles_d; b = blk / p.tiles_d;
    d0 = td * T::TD; h0 = th * T::TH; w0 = tw * T::TW;
}

// Index decoders of the three access phases (E elements per lane along the fast axis):
//   row phase: w fast, then h, then d      col phase: h fast, then w, then d      channel phase: d fast, then w, then h
template <typename T, int E> __device__ __forceinline__ void dec_row(int i, int &d, int &h, int &w) {
    w = (i % (T::TW / E)) * E; h = (i / (T::TW / E)) % T::TH; d = i / (T::TW / E * T::TH);
}
template <typename T, int E> __device__ __forceinline__ void dec_col(int i, int &d, int &h, int &w) {
    h = (i % (T::TH / E)) * E; w = (i / (T::TH / E)) % T::TW; d = i / (T::TH / E * T::TW);
}
template <typename T, int E> __device__ __forceinline__ void dec_chn(int i, int &d, int &h, int &w) {
    d = (i % (T::TD / E)) * E; const int pos = i / (T::TD / E); w = pos % T::TW; h = pos / T::TW;
}

// All global reads below are unconditional, with coordinates clamped into the tensor, and are issued a batch
// (UB) at a time before the LDS stores that consume them: a read under `if (inside)` is sunk into its branch
// with a full wait, i.e. one HBM round trip per loop trip.  Tile padding then holds duplicates of edge
// elements; it is never written out (every global write keeps its bounds check).
constexpr int UB = 8;

// x side is channel-last (B,H,W,D) when CL, else (B,D,H,W)
template <bool CL, typename IO, typename T>
__global__ __launch_bounds__(256) void cross_scan_kernel(const CrossParams p) {
    __shared__ float s[T::TD * T::SD];
    const IO *__restrict__ in = static_cast<const IO *>(p.in);
    IO *__restrict__ out = static_cast<IO *>(p.out);
    int b, d0, h0, w0;
    tile_coords<T>(p, b, d0, h0, w0);
    const int tid = threadIdx.x, H = p.H, W = p.W, D = p.D;
    const size_t L = (size_t)H * W, LR = (size_t)p.lrow;   // positions of a row; row pitch of the direction side (>= L)
    constexpr int NIT = T::N / 256;
#pragma unroll 1
    for (int it0 = 0; it0 < NIT; it0 += UB) {
        float v[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            int d, h, w;
            if (CL) dec_chn<T, 1>(tid + (it0 + u) * 256, d, h, w); else dec_row<T, 1>(tid + (it0 + u) * 256, d, h, w);
            const int dc = min(d0 + d, D - 1), hc = min(h0 + h, H - 1), wc = min(w0 + w, W - 1);
            v[u] = CL ? ld1(in + (((size_t)b * H + hc) * W + wc) * D + dc) : ld1(in + (((size_t)b * D + dc) * H + hc) * W + wc);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            int d, h, w;
            if (CL) dec_chn<T, 1>(tid + (it0 + u) * 256, d, h, w); else dec_row<T, 1>(tid + (it0 + u) * 256, d, h, w);
            s[d * T::SD + h * T::SW + w] = v[u];
        }
    }
    __syncthreads();
    // row-major directions 0 and 2: w on the fast lanes
    for (int i = tid; i < T::N; i += 256) {
        int d, h, w;
        dec_row<T, 1>(i, d, h, w);
        if (d0 + d < D && h0 + h < H && w0 + w < W) {
            const float v = s[d * T::SD + h * T::SW + w];
            const size_t pp = (size_t)(h0 + h) * W + w0 + w;
            IO *o0 = out + (((size_t)b * p.nslots + 0) * D + d0 + d) * LR;
            st1(o0 + pp, v);
            if (p.nslots == 4) st1(out + (((size_t)b * 4 + p.rmB) * D + d0 + d) * LR + (p.revB ? L - 1 - pp : pp), v);
        }
    }
    // column-major directions 1 and 3: h on the fast lanes
    for (int i = tid; i < T::N; i += 256) {
        int d, h, w;
        dec_col<T, 1>(i, d, h, w);
        if (d0 + d < D && h0 + h < H && w0 + w < W) {
            const float v = s[d * T::SD + h * T::SW + w];
            const size_t q = (size_t)(w0 + w) * H + h0 + h;
            IO *o1 = out + (((size_t)b * p.nslots + p.cmA) * D + d0 + d) * LR;
            st1(o1 + q, v);
            if (p.nslots == 4) st1(out + (((size_t)b * 4 + p.cmB) * D + d0 + d) * LR + (p.revB ? L - 1 - q : q), v);
        }
    }
}

// sums in fp32 (LDS), one rounding at the end for bf16 output
template <bool CL, typename IO, typename T, bool EXTRA = false>
__global__ __launch_bounds__(256) void cross_merge_kernel(const CrossParams p) {
    __shared__ float s[T::TD * T::SD];
    const IO *__restrict__ in = static_cast<const IO *>(p.in);
    const IO *__restrict__ in2 = static_cast<const IO *>(p.in2);
    IO *__restrict__ out = static_cast<IO *>(p.out);
    int b, d0, h0, w0;
    tile_coords<T>(p, b, d0, h0, w0);
    const int tid = threadIdx.x, H = p.H, W = p.W, D = p.D;
    const size_t L = (size_t)H * W, LR = (size_t)p.lrow;   // positions of a row; row pitch of the direction side (>= L)
    constexpr int NIT = T::N / 256;
    const bool four = p.nslots == 4;
    const int sB = four ? 4 : p.nslots;
#pragma unroll 1
    for (int it0 = 0; it0 < NIT; it0 += UB) {
        float v[UB], v2[UB], v4[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            int d, h, w;
            dec_row<T, 1>(tid + (it0 + u) * 256, d, h, w);
            const int dc = min(d0 + d, D - 1), hc = min(h0 + h, H - 1), wc = min(w0 + w, W - 1);
            const size_t pp = (size_t)hc * W + wc;
            v[u] = ld1(in + (((size_t)b * p.nslots + 0) * D + dc) * LR + pp);
            // with two slots rmB is 0: the same element again, not added
            v2[u] = ld1(in + (((size_t)b * sB + p.rmB) * D + dc) * LR + (p.revB ? L - 1 - pp : pp));
            if (EXTRA) v4[u] = ld1(in2 + (((size_t)b * 2 + 0) * D + dc) * LR + pp);
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            int d, h, w;
            dec_row<T, 1>(tid + (it0 + u) * 256, d, h, w);
            const float r = four ? v[u] + v2[u] : v[u];
            s[d * T::SD + h * T::SW + w] = EXTRA ? r + v4[u] : r;
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int it0 = 0; it0 < NIT; it0 += UB) {
        float v1[UB], v3[UB], v5[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            int d, h, w;
            dec_col<T, 1>(tid + (it0 + u) * 256, d, h, w);
            const int dc = min(d0 + d, D - 1), hc = min(h0 + h, H - 1), wc = min(w0 + w, W - 1);
            const size_t q = (size_t)wc * H + hc;
            v1[u] = ld1(in + (((size_t)b * p.nslots + p.cmA) * D + dc) * LR + q);
            v3[u] = ld1(in + (((size_t)b * sB + p.cmB) * D + dc) * LR + (p.revB ? L - 1 - q : q));
            if (EXTRA) v5[u] = ld1(in2 + (((size_t)b * 2 + 1) * D + dc) * LR + q);
        }
        // the reference adds in the order k0 + k2 + k1 + k3 (mamba_sys.py:448)
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            int d, h, w;
            dec_col<T, 1>(tid + (it0 + u) * 256, d, h, w);
            float acc = s[d * T::SD + h * T::SW + w] + v1[u];
            acc = four ? acc + v3[u] : acc;
            s[d * T::SD + h * T::SW + w] = EXTRA ? acc + v5[u] : acc;
        }
    }
    __syncthreads();
    for (int i = tid; i < T::N; i += 256) {
        int d, h, w;
        if (CL) dec_chn<T, 1>(i, d, h, w); else dec_row<T, 1>(i, d, h, w);
        if (d0 + d < D && h0 + h < H && w0 + w < W) {
            const float v = s[d * T::SD + h * T::SW + w];
            if (CL) st1(out + (((size_t)b * H + h0 + h) * W + w0 + w) * D + d0 + d, v);
            else st1(out + (((size_t)b * D + d0 + d) * H + h0 + h) * W + w0 + w, v);
        }
    }
}

// ---- vector variants: every lane moves E = 4 (or 2, for the 14 x 14 stage) consecutive elements along the fast axis of
// the phase (w, h or d), i.e. 16- / 8-byte (fp32) or 8- / 4-byte (bf16) global accesses; the LDS tile and its odd strides are
// those of the scalar kernels above.  Require H % E == 0, W % E == 0, a row pitch % E == 0 (and D % E == 0 on a channel-last
// side) and aligned pointers; the host falls back to the scalar kernels otherwise (the 7 x 7 stage).
typedef float f2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
template <int E> struct VecOf;
template <> struct VecOf<4> { typedef f4 type; };
template <> struct VecOf<2> { typedef f2v type; };
__device__ __forceinline__ f4 rev4(f4 v) { return f4{v.w, v.z, v.y, v.x}; }
template <int E> __device__ __forceinline__ typename VecOf<E>::type revv(typename VecOf<E>::type v);
template <> __device__ __forceinline__ f4 revv<4>(f4 v) { return rev4(v); }
template <> __device__ __forceinline__ f2v revv<2>(f2v v) { return f2v{v.y, v.x}; }
template <int E> __device__ __forceinline__ typename VecOf<E>::type ldv(const float *p) { return *reinterpret_cast<const typename VecOf<E>::type *>(p); }
template <int E> __device__ __forceinline__ typename VecOf<E>::type ldv(const bf16 *p);
template <> __device__ __forceinline__ f4 ldv<4>(const bf16 *p) { return ld4(p); }
template <> __device__ __forceinline__ f2v ldv<2>(const bf16 *p) { return __builtin_convertvector(*reinterpret_cast<const bf16x2v *>(p), f2v); }
template <int E> __device__ __forceinline__ void stv(float *p, typename VecOf<E>::type v) { *reinterpret_cast<typename VecOf<E>::type *>(p) = v; }
__device__ __forceinline__ void stv4b(bf16 *p, f4 v) { st4(p, v); }
template <int E> __device__ __forceinline__ void stv(bf16 *p, typename VecOf<E>::type v);
template <> __device__ __forceinline__ void stv<4>(bf16 *p, f4 v) { st4(p, v); }
template <> __device__ __forceinline__ void stv<2>(bf16 *p, f2v v) { *reinterpret_cast<bf16x2v *>(p) = __builtin_convertvector(v, bf16x2v); }

template <bool CL, typename IO, typename T, int E = 4>
__global__ __launch_bounds__(256) void cross_scan_vec_kernel(const CrossParams p) {
    __shared__ float s[T::TD * T::SD];
    const IO *__restrict__ in = static_cast<const IO *>(p.in);
    IO *__restrict__ out = static_cast<IO *>(p.out);
    int b, d0, h0, w0;
    tile_coords<T>(p, b, d0, h0, w0);
    const int tid = threadIdx.x, H = p.H, W = p.W, D = p.D;
    const size_t L = (size_t)H * W, LR = (size_t)p.lrow;   // positions of a row; row pitch of the direction side (>= L)
    typedef typename VecOf<E>::type fv;
    constexpr int NIT = T::N / E / 256;
    {
        fv v[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int d, h, w;
            if (CL) dec_chn<T, E>(tid + it * 256, d, h, w); else dec_row<T, E>(tid + it * 256, d, h, w);
            // E consecutive channels (CL) or columns per lane; sizes are multiples of E, so a group is inside or outside as a whole
            const int dc = min(d0 + d, CL ? D - E : D - 1), hc = min(h0 + h, H - 1), wc = min(w0 + w, CL ? W - 1 : W - E);
            v[it] = CL ? ldv<E>(in + (((size_t)b * H + hc) * W + wc) * D + dc) : ldv<E>(in + (((size_t)b * D + dc) * H + hc) * W + wc);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int d, h, w;
            if (CL) dec_chn<T, E>(tid + it * 256, d, h, w); else dec_row<T, E>(tid + it * 256, d, h, w);
#pragma unroll
            for (int e = 0; e < E; ++e) s[(CL ? d + e : d) * T::SD + h * T::SW + (CL ? w : w + e)] = v[it][e];
        }
    }
    __syncthreads();
    for (int i = tid; i < T::N / E; i += 256) {   // directions 0 and 2: runs along w
        int d, h, w;
        dec_row<T, E>(i, d, h, w);
        if (d0 + d < D && h0 + h < H && w0 + w < W) {
            fv v;
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = s[d * T::SD + h * T::SW + w + e];
            const size_t pp = (size_t)(h0 + h) * W + w0 + w;
            stv<E>(out + (((size_t)b * p.nslots + 0) * D + d0 + d) * LR + pp, v);
            if (p.nslots == 4) {
                IO *o2 = out + (((size_t)b * 4 + p.rmB) * D + d0 + d) * LR;
                if (p.revB) stv<E>(o2 + (L - E - pp), revv<E>(v)); else stv<E>(o2 + pp, v);
            }
        }
    }
    for (int i = tid; i < T::N / E; i += 256) {   // directions 1 and 3: runs along h
        int d, h, w;
        dec_col<T, E>(i, d, h, w);
        if (d0 + d < D && h0 + h < H && w0 + w < W) {
            fv v;
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = s[d * T::SD + (h + e) * T::SW + w];
            const size_t q = (size_t)(w0 + w) * H + h0 + h;
            stv<E>(out + (((size_t)b * p.nslots + p.cmA) * D + d0 + d) * LR + q, v);
            if (p.nslots == 4) {
                IO *o3 = out + (((size_t)b * 4 + p.cmB) * D + d0 + d) * LR;
                if (p.revB) stv<E>(o3 + (L - E - q), revv<E>(v)); else stv<E>(o3 + q, v);
            }
        }
    }
}

template <bool CL, typename IO, typename T, bool EXTRA = false, int E = 4>
__global__ __launch_bounds__(256) void cross_merge_vec_kernel(const CrossParams p) {
    __shared__ float s[T::TD * T::SD];
    const IO *__restrict__ in = static_cast<const IO *>(p.in);
    const IO *__restrict__ in2 = static_cast<const IO *>(p.in2);
    IO *__restrict__ out = static_cast<IO *>(p.out);
    int b, d0, h0, w0;
    tile_coords<T>(p, b, d0, h0, w0);
    const int tid = threadIdx.x, H = p.H, W = p.W, D = p.D;
    const size_t L = (size_t)H * W, LR = (size_t)p.lrow;   // positions of a row; row pitch of the direction side (>= L)
    typedef typename VecOf<E>::type fv;
    constexpr int NIT = T::N / E / 256;
    const bool four = p.nslots == 4;
    const int sB = four ? 4 : p.nslots;
    {
        fv v[NIT], v2[NIT], v4[EXTRA ? NIT : 1];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int d, h, w;
            dec_row<T, E>(tid + it * 256, d, h, w);
            const int dc = min(d0 + d, D - 1), hc = min(h0 + h, H - 1), wc = min(w0 + w, W - E);
            const size_t pp = (size_t)hc * W + wc;
            v[it] = ldv<E>(in + (((size_t)b * p.nslots + 0) * D + dc) * LR + pp);
            const IO *i2 = in + (((size_t)b * sB + p.rmB) * D + dc) * LR;
            v2[it] = ldv<E>(i2 + (p.revB ? L - E - pp : pp));
            if (EXTRA) v4[it] = ldv<E>(in2 + (((size_t)b * 2 + 0) * D + dc) * LR + pp);
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int d, h, w;
            dec_row<T, E>(tid + it * 256, d, h, w);
            const fv r = p.revB ? revv<E>(v2[it]) : v2[it];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float a = four ? v[it][e] + r[e] : v[it][e];
                s[d * T::SD + h * T::SW + w + e] = EXTRA ? a + v4[it][e] : a;
            }
        }
    }
    __syncthreads();
    {
        fv v1[NIT], v3[NIT], v5[EXTRA ? NIT : 1];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int d, h, w;
            dec_col<T, E>(tid + it * 256, d, h, w);
            const int dc = min(d0 + d, D - 1), hc = min(h0 + h, H - E), wc = min(w0 + w, W - 1);
            const size_t q = (size_t)wc * H + hc;
            v1[it] = ldv<E>(in + (((size_t)b * p.nslots + p.cmA) * D + dc) * LR + q);
            const IO *i3 = in + (((size_t)b * sB + p.cmB) * D + dc) * LR;
            v3[it] = ldv<E>(i3 + (p.revB ? L - E - q : q));
            if (EXTRA) v5[it] = ldv<E>(in2 + (((size_t)b * 2 + 1) * D + dc) * LR + q);
        }
        // the reference adds in the order k0 + k2 + k1 + k3 (mamba_sys.py:448)
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int d, h, w;
            dec_col<T, E>(tid + it * 256, d, h, w);
            const fv r = p.revB ? revv<E>(v3[it]) : v3[it];
#pragma unroll
            for (int e = 0; e < E; ++e) {
                float acc = s[d * T::SD + (h + e) * T::SW + w] + v1[it][e];
                acc = four ? acc + r[e] : acc;
                s[d * T::SD + (h + e) * T::SW + w] = EXTRA ? acc + v5[it][e] : acc;
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < T::N / E; i += 256) {
        int d, h, w;
        if (CL) dec_chn<T, E>(i, d, h, w); else dec_row<T, E>(i, d, h, w);
        if (d0 + d < D && h0 + h < H && w0 + w < W) {
            fv v;
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] = s[(CL ? d + e : d) * T::SD + h * T::SW + (CL ? w : w + e)];
            if (CL) stv<E>(out + (((size_t)b * H + h0 + h) * W + w0 + w) * D + d0 + d, v);
            else stv<E>(out + (((size_t)b * D + d0 + d) * H + h0 + h) * W + w0 + w, v);
        }
    }
}

// E = 0: the scalar kernels; 4 / 2: the vector kernels with that many elements per lane
template <typename T, int E>
void launch(bool merge, bool cl, bool h, dim3 grid, hipStream_t s, const CrossParams &p) {
    const dim3 block(256);
#define SS2D_CROSS_LAUNCH(K, ...)                                                                                    \
    do {                                                                                                             \
        if (cl) { if (h) ss2d_launch((K<true, bf16, T, ##__VA_ARGS__>), grid, block, 0, s, p); else ss2d_launch((K<true, float, T, ##__VA_ARGS__>), grid, block, 0, s, p); } \
        else    { if (h) ss2d_launch((K<false, bf16, T, ##__VA_ARGS__>), grid, block, 0, s, p); else ss2d_launch((K<false, float, T, ##__VA_ARGS__>), grid, block, 0, s, p); } \
    } while (0)
    if constexpr (E == 0) {
        if (merge && p.in2) {   // 4 + 2 slots -> image (the backward of an SS2D call)
            SS2D_CROSS_LAUNCH(cross_merge_kernel, true);
        } else if (merge) SS2D_CROSS_LAUNCH(cross_merge_kernel);
        else SS2D_CROSS_LAUNCH(cross_scan_kernel);
    } else {
        if (merge && p.in2) {
            SS2D_CROSS_LAUNCH(cross_merge_vec_kernel, true, E);
        } else if (merge) SS2D_CROSS_LAUNCH(cross_merge_vec_kernel, false, E);
        else SS2D_CROSS_LAUNCH(cross_scan_vec_kernel, E);
    }
#undef SS2D_CROSS_LAUNCH
}

int run(bool merge, bool cl, int layout, int dtype, const void *in, void *out, int B, int D, int H, int W, void *stream, const char *name,
        const void *in2 = nullptr, long long lrow = 0) {
    if (dtype != SS2D_DTYPE_F32 && dtype != SS2D_DTYPE_BF16) { ss2d_set_error("%s: unknown dtype %d", name, dtype); return -1; }
    if (!in || !out) { ss2d_set_error("%s: null pointer", name); return -2; }
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0) { ss2d_set_error("%s: non-positive size", name); return -3; }
    if (lrow != 0 && lrow < (long long)H * W) { ss2d_set_error("%s: row pitch %lld smaller than H*W", name, lrow); return -3; }
    CrossParams p;
    p.in = in; p.in2 = in2; p.out = out; p.B = B; p.D = D; p.H = H; p.W = W;
    p.lrow = lrow ? lrow : (long long)H * W;
    if (layout == SS2D_DIRS_REFERENCE) { p.nslots = 4; p.rmB = 2; p.cmA = 1; p.cmB = 3; p.revB = 1; }
    else if (layout == SS2D_DIRS_MIRROR_FREE_4) { p.nslots = 4; p.rmB = 1; p.cmA = 2; p.cmB = 3; p.revB = 0; }
    else if (layout == SS2D_DIRS_MIRROR_FREE_2) { p.nslots = 2; p.rmB = 0; p.cmA = 1; p.cmB = 1; p.revB = 0; }
    else { ss2d_set_error("%s: unknown direction layout %d", name, layout); return -7; }
    const bool small = H <= TileSmall::TH && W <= TileSmall::TW;
    const int td = small ? TileSmall::TD : TileBig::TD, th = small ? TileSmall::TH : TileBig::TH, tw = small ? TileSmall::TW : TileBig::TW;
    p.tiles_h = (H + th - 1) / th; p.tiles_w = (W + tw - 1) / tw; p.tiles_d = (D + td - 1) / td;
    const long long n = (long long)B * p.tiles_d * p.tiles_h * p.tiles_w;
    if (n > 0x7fffffffLL) { ss2d_set_error("%s: grid too large", name); return -6; }
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)n);
    const bool h = dtype == SS2D_DTYPE_BF16;
    const uintptr_t ptrs = (uintptr_t)in | (uintptr_t)in2 | (uintptr_t)out;
    auto fits = [&](int e) {   // e elements per lane: sizes and pitch multiples of e, pointers aligned to e elements
        return H % e == 0 && W % e == 0 && p.lrow % e == 0 && (!cl || D % e == 0) && (ptrs & (uintptr_t)(e * (h ? 2 : 4) - 1)) == 0;
    };
    const int e = fits(4) ? 4 : fits(2) ? 2 : 0;
    if (small) { if (e == 4) launch<TileSmall, 4>(merge, cl, h, grid, s, p); else if (e == 2) launch<TileSmall, 2>(merge, cl, h, grid, s, p); else launch<TileSmall, 0>(merge, cl, h, grid, s, p); }
    else { if (e == 4) launch<TileBig, 4>(merge, cl, h, grid, s, p); else if (e == 2) launch<TileBig, 2>(merge, cl, h, grid, s, p); else launch<TileBig, 0>(merge, cl, h, grid, s, p); }
    const int rc = ss2d_take_launch_rc();
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
// y (B, D, H, W) = the four direction slots of ys4 (layout SS2D_DIRS_MIRROR_FREE_4) + the two orientation slots of ys2
// (SS2D_DIRS_MIRROR_FREE_2), one pass: the backward of an SS2D call, where du of the scan (4 slots) and the input gradient
// of x_proj (2 slots) meet in dx
extern "C" int ss2d_dirs_gather6(int dtype, const void *ys4, const void *ys2, void *y, int B, int D, int H, int W, void *stream) {
    if (!ys2) { ss2d_set_error("dirs_gather6: null pointer"); return -2; }
    return run(true, false, SS2D_DIRS_MIRROR_FREE_4, dtype, ys4, y, B, D, H, W, stream, "dirs_gather6", ys2);
}
// The same three with a row pitch: the direction-side rows (H*W positions) lie `row_pitch` elements apart.  A caller whose
// H*W is odd (the 7 x 7 stage) pads its rows to a multiple of 8, zero-fills the pad once, and runs x_proj / dt_proj / the
// scan on the padded length with their aligned fast paths (zero u and zero dy in the pad leave every result unchanged).
extern "C" int ss2d_dirs_scatter_ld(int dtype, int layout, const void *x, void *xs, int B, int D, int H, int W, int x_channel_last,
                                    long long row_pitch, void *stream) {
    return run(false, x_channel_last != 0, layout, dtype, x, xs, B, D, H, W, stream, "dirs_scatter_ld", nullptr, row_pitch);
}
extern "C" int ss2d_dirs_gather_ld(int dtype, int layout, const void *ys, void *y, int B, int D, int H, int W, int y_channel_last,
                                   long long row_pitch, void *stream) {
    return run(true, y_channel_last != 0, layout, dtype, ys, y, B, D, H, W, stream, "dirs_gather_ld", nullptr, row_pitch);
}
extern "C" int ss2d_dirs_gather6_ld(int dtype, const void *ys4, const void *ys2, void *y, int B, int D, int H, int W, long long row_pitch,
                                    void *stream) {
    if (!ys2) { ss2d_set_error("dirs_gather6_ld: null pointer"); return -2; }
    return run(true, false, SS2D_DIRS_MIRROR_FREE_4, dtype, ys4, y, B, D, H, W, stream, "dirs_gather6_ld", ys2, row_pitch);
}
// ... with y channel-last (B, H, W, D): the gradient of a channel-last convolution output (ss2d_dwconv3x3_silu_cl_bwd)
extern "C" int ss2d_dirs_gather6_cl_ld(int dtype, const void *ys4, const void *ys2, void *y, int B, int D, int H, int W, long long row_pitch,
                                       void *stream) {
    if (!ys2) { ss2d_set_error("dirs_gather6_cl_ld: null pointer"); return -2; }
    return run(true, true, SS2D_DIRS_MIRROR_FREE_4, dtype, ys4, y, B, D, H, W, stream, "dirs_gather6_cl_ld", ys2, row_pitch);
}
