// dt_proj forward for gfx950:  dts[b,g,d,l] = sum_r W[g,d,r] * x[b,g,r,l]        (rank-R expansion, R = dt_rank)
// (/root/reference/code/networks/mamba_sys.py:420, einsum "b k r l, k d r -> b k d l"; weight dt_projs_weight :319-327).
// A K = 6 (or 12) contraction that writes a tensor 32x (16x) larger than it reads: as a batched GEMM it runs at 1 TB/s
// (116 us for 115 MB of bf16 at stage 0).  Here a lane owns 4 consecutive positions, keeps their R inputs in registers
// and walks the D output channels; the weights of a channel are wave-uniform (scalar loads).  Pure streaming store.
// x is read in place from the first R rows of every (b, g) block of x_dbl (block stride `xs_bg`).
#include "ss2d_common.h"
#include "../../include/ss2d.h"

namespace {

struct DtProjParams {
    const void *x;
    const float *w;
    void *out;
    int G, D, L, ltiles, dsplit, dchunk;
    long long xs_bg;   // elements between consecutive (b, g) blocks of x
};

template <int R, typename IO>
__global__ __launch_bounds__(256) void dt_proj_fwd_kernel(const DtProjParams p) {
    int blk = xcd_remap(blockIdx.x, gridDim.x);   // neighbours in the tile order share cache lines / operands: one XCD's L2
    const int ds = blk % p.dsplit; blk /= p.dsplit;
    const int lt = blk % p.ltiles, bg = blk / p.ltiles, g = bg % p.G;
    const int dlo = ds * p.dchunk, dhi = min(p.D, dlo + p.dchunk);
    const int l0 = (lt * 256 + threadIdx.x) * 4;
    if (l0 >= p.L) return;
    const IO *__restrict__ x = static_cast<const IO *>(p.x) + (size_t)bg * p.xs_bg + l0;
    IO *__restrict__ out = static_cast<IO *>(p.out) + (size_t)bg * p.D * p.L + l0;
    const float *__restrict__ w = p.w + (size_t)g * p.D * R;
    f4 xv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) xv[r] = ld4(x + (size_t)r * p.L);
#pragma unroll 4
    for (int d = dlo; d < dhi; ++d) {
        f4 acc = f4{0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < R; ++r) acc += xv[r] * w[d * R + r];
        st4(out + (size_t)d * p.L, acc);
    }
}

}  // namespace

extern "C" int ss2d_dt_proj_fwd(int dtype, const void *x, const float *w, void *out, int batch, int ngroups, int D,
                                int R, int L, long long x_bg_stride, void *stream) {
    if (dtype != SS2D_DTYPE_F32 && dtype != SS2D_DTYPE_BF16) { ss2d_set_error("dt_proj_fwd: unknown dtype %d", dtype); return -1; }
    if (!x || !w || !out) { ss2d_set_error("dt_proj_fwd: null pointer"); return -2; }
    if (batch <= 0 || ngroups <= 0 || D <= 0 || L <= 0) { ss2d_set_error("dt_proj_fwd: non-positive size"); return -3; }
    if (R != 6 && R != 12) { ss2d_set_error("dt_proj_fwd: dt_rank %d not supported (6 or 12)", R); return -4; }
    const int es = dtype == SS2D_DTYPE_BF16 ? 2 : 4;
    if (L % 4 || x_bg_stride % 4 || ((uintptr_t)x | (uintptr_t)out) % (4 * es)) {
        ss2d_set_error("dt_proj_fwd: L and the block stride must be multiples of 4 and the pointers aligned to 4 elements"); return -4; }
    if (x_bg_stride < (long long)R * L) { ss2d_set_error("dt_proj_fwd: block stride smaller than R*L"); return -4; }
    DtProjParams p;
    p.x = x; p.w = w; p.out = out; p.G = ngroups; p.D = D; p.L = L; p.xs_bg = x_bg_stride;
    p.ltiles = (L / 4 + 255) / 256;
    // enough workgroups to fill the chip: split the output channels when (batch * groups * L-tiles) is small
    p.dchunk = D;
    while (p.dchunk > 16 && (long long)batch * ngroups * p.ltiles * ((D + p.dchunk - 1) / p.dchunk) < 2048) p.dchunk = (p.dchunk + 1) / 2;
    p.dsplit = (D + p.dchunk - 1) / p.dchunk;
    const long long nwg = (long long)batch * ngroups * p.ltiles * p.dsplit;
    if (nwg > 0x7fffffffLL) { ss2d_set_error("dt_proj_fwd: grid too large"); return -6; }
    const dim3 grid((unsigned)nwg), block(256);
    hipStream_t s = (hipStream_t)stream;
    const bool h = dtype == SS2D_DTYPE_BF16;
    if (R == 6) { if (h) ss2d_launch((dt_proj_fwd_kernel<6, bf16>), grid, block, 0, s, p); else ss2d_launch((dt_proj_fwd_kernel<6, float>), grid, block, 0, s, p); }
    else        { if (h) ss2d_launch((dt_proj_fwd_kernel<12, bf16>), grid, block, 0, s, p); else ss2d_launch((dt_proj_fwd_kernel<12, float>), grid, block, 0, s, p); }
    const int rc = ss2d_take_launch_rc();
    if (rc) ss2d_set_error("dt_proj_fwd: launch failed: %s", hipGetErrorString((hipError_t)rc));
    return rc;
}
