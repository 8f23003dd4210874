// Column sums of a (rows, M) matrix -> (M) fp32: the bias gradient of the Linears that have one (the skip-concat
// projections `concat_back_dim`, /root/reference/code/networks/mamba_sys.py:774-778, applied :835).  torch's own
// multi-block reduction does this job in eager mode, but returns wrong sums from the second replay of a captured HIP
// graph on this stack (tools/graph_dice_repro.py); this kernel has no such state: per-workgroup partial rows in a
// caller-provided workspace, added in a fixed order by a second launch (deterministic, no atomics, no zero-fill).
#include "ss2d_common.h"
#include "../../include/ss2d.h"

namespace {

constexpr int MAX_WG = 512;

// a workgroup = 256 threads = CQ column quads x (256 / CQ) row lanes over a slab of rows
template <typename IO>
__global__ __launch_bounds__(256) void colsum_kernel(const IO *__restrict__ x, float *__restrict__ partials, long long rows, int M, long long ldx,
                                                     int cq, long long slab, int vec) {
    __shared__ f4 red[256];
    const int tid = threadIdx.x, c = tid % cq, rl = tid / cq, nrl = 256 / cq;
    const long long r0 = (long long)blockIdx.x * slab, r1 = r0 + slab < rows ? r0 + slab : rows;
    const int nq = (M + 3) / 4;
    for (int q0 = 0; q0 < nq; q0 += cq) {
        const int q = q0 + c;
        f4 acc = f4{0.0f, 0.0f, 0.0f, 0.0f};
        if (q < nq && rl < nrl) {
            const bool full = vec && 4 * q + 3 < M;
            for (long long r = r0 + rl; r < r1; r += nrl) {
                const IO *p = x + (size_t)r * ldx + 4 * q;
                if (full) acc += ld4(p);
                else
                    for (int e = 0; e < 4; ++e) if (4 * q + e < M) acc[e] += ld1(p + e);
            }
        }
        red[tid] = acc;
        __syncthreads();
        if (rl == 0 && q < nq) {
            f4 s = red[c];
            for (int k = 1; k < nrl; ++k) s += red[k * cq + c];          // fixed order
            float *dst = partials + (size_t)blockIdx.x * M + 4 * q;
            for (int e = 0; e < 4; ++e) if (4 * q + e < M) dst[e] = s[e];
        }
        __syncthreads();
    }
}

// 64 columns per workgroup; the partial rows are split over 4 row lanes (8 independent loads in flight each), added in a
// fixed order
__global__ __launch_bounds__(256) void colsum_finish_kernel(const float *__restrict__ partials, float *__restrict__ out, int M, int nrows) {
    __shared__ float red[256];
    const int c = threadIdx.x & 63, part = threadIdx.x >> 6, j = blockIdx.x * 64 + c;
    float s = 0.0f;
    if (j < M) {
        int r = part;
        for (; r + 28 < nrows; r += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partials[(size_t)(r + 4 * u) * M + j];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; r < nrows; r += 4) s += partials[(size_t)r * M + j];
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (part == 0 && j < M) out[j] = (red[c] + red[64 + c]) + (red[128 + c] + red[192 + c]);
}

}  // namespace

extern "C" size_t ss2d_colsum_workspace_floats(int M) { return M > 0 ? (size_t)MAX_WG * M : 0; }

extern "C" int ss2d_colsum(int dtype, const void *x, float *out, float *workspace, long long rows, int M, long long ldx, void *stream) {
    if (dtype != SS2D_DTYPE_F32 && dtype != SS2D_DTYPE_BF16) { ss2d_set_error("colsum: unknown dtype %d", dtype); return -1; }
    if (!x || !out || !workspace) { ss2d_set_error("colsum: null required pointer"); return -2; }
    if (rows <= 0 || M <= 0 || ldx < M) { ss2d_set_error("colsum: bad sizes rows %lld M %d ldx %lld", rows, M, ldx); return -3; }
    const uintptr_t amask = dtype == SS2D_DTYPE_BF16 ? 7 : 15;
    const int vec = (((uintptr_t)x & amask) == 0) && ldx % 4 == 0;          // 4-element reads need aligned rows
    const int nq = (M + 3) / 4;
    int cq = 1;
    while (cq < nq && cq < 64) cq *= 2;                                     // column quads per pass: a power of two <= 64
    long long nwg = (rows + 127) / 128;      // >= 128 rows per workgroup: a few dozen reads per thread, enough workgroups for a 4704-row input
    if (nwg > MAX_WG) nwg = MAX_WG;
    const long long slab = (rows + nwg - 1) / nwg;
    nwg = (rows + slab - 1) / slab;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SS2D_DTYPE_BF16) ss2d_launch(colsum_kernel<bf16>, dim3((unsigned)nwg), dim3(256), 0, s, static_cast<const bf16 *>(x), workspace, rows, M, ldx, cq, slab, vec);
    else ss2d_launch(colsum_kernel<float>, dim3((unsigned)nwg), dim3(256), 0, s, static_cast<const float *>(x), workspace, rows, M, ldx, cq, slab, vec);
    ss2d_launch(colsum_finish_kernel, dim3((unsigned)((M + 63) / 64)), dim3(256), 0, s, (const float *)workspace, out, M, (int)nwg);
    const int rc = ss2d_take_launch_rc();
    if (rc) ss2d_set_error("colsum: launch failed: %s", hipGetErrorString((hipError_t)rc));
    return rc;
}
