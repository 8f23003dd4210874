// Fused statistics of the segmentation losses of the reference's training scripts, for gfx950:
// softmax over the class axis + the three per-class sums of DiceLoss (/root/reference/code/utils/losses.py:179-215:
// intersect, y_sum, z_sum over the WHOLE batch) + the summed cross entropy of nn.CrossEntropyLoss
// (train_fully_supervised_2D_VIM.py:153-157, train_Semi_Mamba_UNet.py:424-446) in ONE pass over the logits, and the
// matching backward in one pass.  In torch this is softmax, one-hot, three products, three multi-block reductions,
// log_softmax + nll and their autograd twins over (B, C, H, W) tensors: ~3 ms of a 27 ms training step.
//
//   stats[c]        = sum_{b,i} p[b,c,i] * [y[b,i] == c]        (intersect)
//   stats[C + c]    = sum_{b,i} [y[b,i] == c]                   (y_sum)
//   stats[2C + c]   = sum_{b,i} p[b,c,i]^2                      (z_sum)
//   stats[3C]       = sum_{b,i} -log p[b,y[b,i],i]              (cross entropy, summed: the caller divides)
//
// HBM-bound streaming kernels: 4 pixels per lane (16-byte loads per class), per-lane accumulators, wave sum, one
// partial row per workgroup in the workspace and a second one-workgroup launch that adds the rows in a FIXED order
// (deterministic, no atomics, no zero-fill).  The loss formulas themselves (13 numbers) stay in the caller, so the
// batch-global all-reduce of the statistics, class weights or pseudo labels need no other kernel.
#include "ss2d_common.h"
#include "../../include/ss2d.h"

namespace {

constexpr int MAX_WG = 1024;

__device__ __forceinline__ int load_label(const void *labels, int bytes, size_t i) {
    if (bytes == 1) return static_cast<const unsigned char *>(labels)[i];
    if (bytes == 4) return static_cast<const int *>(labels)[i];
    return (int)static_cast<const long long *>(labels)[i];
}

// softmax of one pixel: p[c] for c < C; returns -log p[y] (0 when y is outside [0, C): such a pixel carries no CE term)
template <int CMAX>
__device__ __forceinline__ float pixel_softmax(float (&l)[CMAX], int C, int y) {
    float m = l[0];
#pragma unroll
    for (int c = 1; c < CMAX; ++c) if (c < C) m = fmaxf(m, l[c]);
    float s = 0.0f, ly = m;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) if (c < C) {
        if (c == y) ly = l[c];
        l[c] = hw_exp2((l[c] - m) * SS2D_LOG2E);
        s += l[c];
    }
    const float r = 1.0f / s;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) if (c < C) l[c] *= r;
    return (y >= 0 && y < C) ? (m - ly) + hw_log2(s) * SS2D_LN2 : 0.0f;
}

// PX = 4: HW % 4 == 0 and 16-byte aligned logits; PX = 1: anything else
template <int CMAX, int PX, typename IO>
__global__ __launch_bounds__(256) void seg_loss_stats_kernel(const IO *__restrict__ logits, const void *__restrict__ labels, int label_bytes,
                                                             float *__restrict__ partials, int C, long long HW, long long groups_per_image,
                                                             long long ngroups) {
    __shared__ float red[4][3 * CMAX + 1];
    float inter[CMAX], ysum[CMAX], zsum[CMAX], ce = 0.0f;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) { inter[c] = 0.0f; ysum[c] = 0.0f; zsum[c] = 0.0f; }
    for (long long grp = (long long)blockIdx.x * 256 + threadIdx.x; grp < ngroups; grp += (long long)gridDim.x * 256) {
        const long long b = grp / groups_per_image, i0 = (grp - b * groups_per_image) * PX;
        const IO *lp = logits + (size_t)b * C * HW + i0;
        float l[PX][CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            if (c < C) {
                if (PX == 4) {
                    const f4 v = ld4(lp + (size_t)c * HW);
                    l[0][c] = v.x; l[1 % PX][c] = v.y; l[2 % PX][c] = v.z; l[3 % PX][c] = v.w;
                } else {
                    l[0][c] = ld1(lp + (size_t)c * HW);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            const int y = load_label(labels, label_bytes, (size_t)b * HW + i0 + q);
            ce += pixel_softmax<CMAX>(l[q], C, y);
#pragma unroll
            for (int c = 0; c < CMAX; ++c) if (c < C) {
                const float p = l[q][c], hit = (c == y) ? 1.0f : 0.0f;
                inter[c] = fmaf(p, hit, inter[c]);
                ysum[c] += hit;
                zsum[c] = fmaf(p, p, zsum[c]);
            }
        }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < CMAX; ++c) if (c < C) {
        const float a = lanes_sum<64>(inter[c]), b2 = lanes_sum<64>(ysum[c]), d = lanes_sum<64>(zsum[c]);
        if (lane == 0) { red[wave][c] = a; red[wave][C + c] = b2; red[wave][2 * C + c] = d; }
    }
    const float cs = lanes_sum<64>(ce);
    if (lane == 0) red[wave][3 * C] = cs;
    __syncthreads();
    if (threadIdx.x <= 3 * C)
        partials[(size_t)blockIdx.x * (3 * C + 1) + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// stats[j] = sum over the workgroup rows, in a fixed order.  One 1024-thread workgroup: wave w walks the rows w, w + 16, ...
// with lane j on statistic j (a row is one coalesced run of 3C + 1 <= 64 floats), eight independent partial sums per lane
// so that eight loads are in flight, combined in a fixed tree; the 16 wave sums are added in wave order by wave 0.  (The
// first version gave each statistic 4 lanes walking 256 rows each with one dependent load per trip: 63 us.)
__global__ __launch_bounds__(1024) void seg_loss_finish_kernel(const float *__restrict__ partials, float *__restrict__ stats, int nstat, int nrows) {
    __shared__ float red[16][64];
    const int j = threadIdx.x & 63, w = threadIdx.x >> 6;
    float a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = 0.0f;
    if (j < nstat) {
        for (int r0 = w; r0 < nrows; r0 += 16 * 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = r0 + 16 * u;
                a[u] += partials[(size_t)(r < nrows ? r : 0) * nstat + j] * (r < nrows ? 1.0f : 0.0f);
            }
        }
    }
    red[w][j] = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    __syncthreads();
    if (w == 0 && j < nstat) {
        float s = 0.0f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += red[q][j];
        stats[j] = s;
    }
}

// dlogits[c] = p_c * (q_c - sum_k p_k q_k) + g_ce * (p_c - [y == c]),  q_c = g_inter[c] * [y == c] + 2 * g_z[c] * p_c
// (chain rule through the softmax; g = dL/dstats: g_inter = g[0..C), g_z = g[2C..3C), g_ce = g[3C]; y_sum has no
// dependence on the logits)
template <int CMAX, int PX, typename IO>
__global__ __launch_bounds__(256) void seg_loss_bwd_kernel(const IO *__restrict__ logits, const void *__restrict__ labels, int label_bytes,
                                                           const float *__restrict__ gstats, IO *__restrict__ dlogits, int C, long long HW,
                                                           long long groups_per_image, long long ngroups) {
    float gi[CMAX], gz[CMAX];
#pragma unroll
    for (int c = 0; c < CMAX; ++c) { gi[c] = c < C ? gstats[c] : 0.0f; gz[c] = c < C ? 2.0f * gstats[2 * C + c] : 0.0f; }
    const float gce = gstats[3 * C];
    for (long long grp = (long long)blockIdx.x * 256 + threadIdx.x; grp < ngroups; grp += (long long)gridDim.x * 256) {
        const long long b = grp / groups_per_image, i0 = (grp - b * groups_per_image) * PX;
        const size_t off = (size_t)b * C * HW + i0;
        float l[PX][CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            if (c < C) {
                if (PX == 4) {
                    const f4 v = ld4(logits + off + (size_t)c * HW);
                    l[0][c] = v.x; l[1 % PX][c] = v.y; l[2 % PX][c] = v.z; l[3 % PX][c] = v.w;
                } else {
                    l[0][c] = ld1(logits + off + (size_t)c * HW);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            const int y = load_label(labels, label_bytes, (size_t)b * HW + i0 + q);
            pixel_softmax<CMAX>(l[q], C, y);
            const bool yok = y >= 0 && y < C;
            float dot = 0.0f, qv[CMAX];
#pragma unroll
            for (int c = 0; c < CMAX; ++c) if (c < C) {
                qv[c] = fmaf(gz[c], l[q][c], (c == y) ? gi[c] : 0.0f);
                dot = fmaf(l[q][c], qv[c], dot);
            }
#pragma unroll
            for (int c = 0; c < CMAX; ++c) if (c < C) {
                const float p = l[q][c];
                l[q][c] = p * (qv[c] - dot) + (yok ? gce * (p - ((c == y) ? 1.0f : 0.0f)) : 0.0f);
            }
        }
#pragma unroll
        for (int c = 0; c < CMAX; ++c) {
            if (c < C) {
                if (PX == 4) st4(dlogits + off + (size_t)c * HW, f4{l[0][c], l[1 % PX][c], l[2 % PX][c], l[3 % PX][c]});
                else st1(dlogits + off + (size_t)c * HW, l[0][c]);
            }
        }
    }
}

int check_args(const char *what, int dtype, const void *logits, const void *labels, int label_bytes, int B, int C, long long HW) {
    if (dtype != SS2D_DTYPE_F32 && dtype != SS2D_DTYPE_BF16) { ss2d_set_error("%s: unknown dtype %d", what, dtype); return -1; }
    if (!logits || !labels) { ss2d_set_error("%s: null required pointer", what); return -2; }
    if (B <= 0 || C <= 0 || HW <= 0) { ss2d_set_error("%s: non-positive size", what); return -3; }
    if (C > SS2D_SEG_LOSS_MAX_CLASSES) { ss2d_set_error("%s: %d classes > %d", what, C, SS2D_SEG_LOSS_MAX_CLASSES); return -4; }
    if (label_bytes != 1 && label_bytes != 4 && label_bytes != 8) { ss2d_set_error("%s: labels must be 1, 4 or 8 bytes wide (got %d)", what, label_bytes); return -5; }
    return 0;
}

struct Geometry { int px; long long gpi, ngroups; unsigned nwg; };
Geometry geometry(int dtype, const void *logits, const void *other, long long HW, int B) {
    const uintptr_t amask = dtype == SS2D_DTYPE_BF16 ? 7 : 15;
    Geometry g;
    g.px = (HW % 4 == 0 && (((uintptr_t)logits | (uintptr_t)other) & amask) == 0) ? 4 : 1;
    g.gpi = HW / g.px;
    g.ngroups = g.gpi * B;
    const long long want = (g.ngroups + 255) / 256;
    g.nwg = (unsigned)(want < MAX_WG ? want : MAX_WG);
    return g;
}

}  // namespace

extern "C" size_t ss2d_seg_loss_workspace_floats(int C) { return C > 0 ? (size_t)MAX_WG * (3 * (size_t)C + 1) : 0; }

extern "C" int ss2d_seg_loss_stats(int dtype, const void *logits, const void *labels, int label_bytes, float *stats, float *workspace,
                                   int B, int C, long long HW, void *stream) {
    int rc = check_args("seg_loss_stats", dtype, logits, labels, label_bytes, B, C, HW);
    if (rc) return rc;
    if (!stats || !workspace) { ss2d_set_error("seg_loss_stats: null stats / workspace"); return -2; }
    const Geometry g = geometry(dtype, logits, nullptr, HW, B);
    hipStream_t s = (hipStream_t)stream;
    const bool h = dtype == SS2D_DTYPE_BF16;
#define SS2D_SEG_STATS(CM, PX, IO) ss2d_launch(seg_loss_stats_kernel<CM, PX, IO>, dim3(g.nwg), dim3(256), 0, s, static_cast<const IO *>(logits), \
                                               labels, label_bytes, workspace, C, HW, g.gpi, g.ngroups)
    if (C <= 4) {
        if (g.px == 4) { if (h) SS2D_SEG_STATS(4, 4, bf16); else SS2D_SEG_STATS(4, 4, float); }
        else { if (h) SS2D_SEG_STATS(4, 1, bf16); else SS2D_SEG_STATS(4, 1, float); }
    } else {
        if (g.px == 4) { if (h) SS2D_SEG_STATS(16, 4, bf16); else SS2D_SEG_STATS(16, 4, float); }
        else { if (h) SS2D_SEG_STATS(16, 1, bf16); else SS2D_SEG_STATS(16, 1, float); }
    }
#undef SS2D_SEG_STATS
    ss2d_launch(seg_loss_finish_kernel, dim3(1), dim3(1024), 0, s, (const float *)workspace, stats, 3 * C + 1, (int)g.nwg);
    rc = ss2d_take_launch_rc();
    if (rc) ss2d_set_error("seg_loss_stats: launch failed: %s", hipGetErrorString((hipError_t)rc));
    return rc;
}

extern "C" int ss2d_seg_loss_bwd(int dtype, const void *logits, const void *labels, int label_bytes, const float *gstats, void *dlogits,
                                 int B, int C, long long HW, void *stream) {
    int rc = check_args("seg_loss_bwd", dtype, logits, labels, label_bytes, B, C, HW);
    if (rc) return rc;
    if (!gstats || !dlogits) { ss2d_set_error("seg_loss_bwd: null gstats / dlogits"); return -2; }
    const Geometry g = geometry(dtype, logits, dlogits, HW, B);
    hipStream_t s = (hipStream_t)stream;
    const bool h = dtype == SS2D_DTYPE_BF16;
    const unsigned nwg = (unsigned)((g.ngroups + 255) / 256 < 8192 ? (g.ngroups + 255) / 256 : 8192);
#define SS2D_SEG_BWD(CM, PX, IO) ss2d_launch(seg_loss_bwd_kernel<CM, PX, IO>, dim3(nwg), dim3(256), 0, s, static_cast<const IO *>(logits), labels, \
                                             label_bytes, gstats, static_cast<IO *>(dlogits), C, HW, g.gpi, g.ngroups)
    if (C <= 4) {
        if (g.px == 4) { if (h) SS2D_SEG_BWD(4, 4, bf16); else SS2D_SEG_BWD(4, 4, float); }
        else { if (h) SS2D_SEG_BWD(4, 1, bf16); else SS2D_SEG_BWD(4, 1, float); }
    } else {
        if (g.px == 4) { if (h) SS2D_SEG_BWD(16, 4, bf16); else SS2D_SEG_BWD(16, 4, float); }
        else { if (h) SS2D_SEG_BWD(16, 1, bf16); else SS2D_SEG_BWD(16, 1, float); }
    }
#undef SS2D_SEG_BWD
    rc = ss2d_take_launch_rc();
    if (rc) ss2d_set_error("seg_loss_bwd: launch failed: %s", hipGetErrorString((hipError_t)rc));
    return rc;
}
