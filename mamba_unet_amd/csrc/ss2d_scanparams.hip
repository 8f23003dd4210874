// Parameter transforms in front of the selective scan, forward and backward, for gfx950
// (/root/reference/code/networks/mamba_sys.py:428-431: As = -exp(A_logs.float()).view(K*D, N), Ds.view(K*D),
// dt_projs_bias.view(K*D); the dt_projs_weight (K, D, R) of :420 rides along).
// In the reference these are a handful of tiny elementwise ops per SS2D call, times their autograd twins; with the
// fused-direction scan every one of them is followed by a regrouping copy (reference direction k = 2*rev +
// orientation -> scan group 2*orientation + rev).  Here the lot is one launch each way: ~20 kernels of a few
// hundred KB per block become 2.  `swap` regroups (k <-> g swaps the two index bits; an involution), 0 keeps k.
#include "ss2d_common.h"
#include "../../include/ss2d.h"

namespace {

struct ScanParamsArgs {
    const float *a, *d, *b, *w;     // fwd: A_logs, Ds, dt_bias, dt_w        bwd: dA, dD, dbias, dw (scan-group order; nullable)
    float *oa, *od, *ob, *ow;       // fwd: A, D, bias, w (scan-group order)  bwd: dA_logs, dDs, ddt_bias, ddt_w (reference order)
    const float *A;                 // bwd: A = -exp(A_logs) as produced by fwd
    int K, Dm, N, R, swap;
};

__device__ __forceinline__ int src_group(int g, int swap) { return swap ? (((g & 1) << 1) | (g >> 1)) : g; }

template <bool BWD>
__global__ __launch_bounds__(256) void scan_params_kernel(const ScanParamsArgs p) {
    const int per_row = p.N + 2 + p.R;                    // A row, D, bias, w row of one (group, channel)
    const long long total = (long long)p.K * p.Dm * per_row;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int e = (int)(i % per_row);
        const long long row = i / per_row;                // output row in fwd (scan-group order), input row in bwd
        const int g = (int)(row / p.Dm), d = (int)(row % p.Dm);
        const long long other = (long long)src_group(g, p.swap) * p.Dm + d;   // the row in reference order
        if (e < p.N) {
            if (!BWD) p.oa[row * p.N + e] = -expf(p.a[other * p.N + e]);
            else p.oa[other * p.N + e] = p.a ? p.a[row * p.N + e] * p.A[row * p.N + e] : 0.0f;   // d(-exp x) = -exp x
        } else if (e == p.N) {
            if (!BWD) p.od[row] = p.d[other]; else p.od[other] = p.d ? p.d[row] : 0.0f;
        } else if (e == p.N + 1) {
            if (!BWD) p.ob[row] = p.b[other]; else p.ob[other] = p.b ? p.b[row] : 0.0f;
        } else {
            const int r = e - p.N - 2;
            if (!BWD) p.ow[row * p.R + r] = p.w[other * p.R + r]; else p.ow[other * p.R + r] = p.w ? p.w[row * p.R + r] : 0.0f;
        }
    }
}

int run(bool bwd, ScanParamsArgs &p, void *stream, const char *name) {
    if (p.K <= 0 || p.Dm <= 0 || p.N <= 0 || p.R <= 0) { ss2d_set_error("%s: non-positive size", name); return -3; }
    if (p.swap && p.K != 4) { ss2d_set_error("%s: regrouping needs K = 4 directions, got %d", name, p.K); return -4; }
    const long long total = (long long)p.K * p.Dm * (p.N + 2 + p.R);
    long long nwg = (total + 255) / 256;
    if (nwg > 2048) nwg = 2048;
    hipStream_t s = (hipStream_t)stream;
    if (bwd) ss2d_launch(scan_params_kernel<true>, dim3((unsigned)nwg), dim3(256), 0, s, p);
    else ss2d_launch(scan_params_kernel<false>, dim3((unsigned)nwg), dim3(256), 0, s, p);
    const int rc = ss2d_take_launch_rc();
    if (rc) ss2d_set_error("%s: launch failed: %s", name, hipGetErrorString((hipError_t)rc));
    return rc;
}

}  // namespace

extern "C" int ss2d_scan_params_fwd(const float *A_logs, const float *Ds, const float *dt_bias, const float *dt_w,
                                    float *A, float *D, float *bias, float *w, int K, int Dm, int N, int R,
                                    int regroup, void *stream) {
    if (!A_logs || !Ds || !dt_bias || !dt_w || !A || !D || !bias || !w) { ss2d_set_error("scan_params_fwd: null pointer"); return -2; }
    ScanParamsArgs p{};
    p.a = A_logs; p.d = Ds; p.b = dt_bias; p.w = dt_w; p.oa = A; p.od = D; p.ob = bias; p.ow = w;
    p.K = K; p.Dm = Dm; p.N = N; p.R = R; p.swap = regroup != 0;
    return run(false, p, stream, "scan_params_fwd");
}

extern "C" int ss2d_scan_params_bwd(const float *A, const float *dA, const float *dD, const float *dbias, const float *dw,
                                    float *dA_logs, float *dDs, float *ddt_bias, float *ddt_w, int K, int Dm, int N,
                                    int R, int regroup, void *stream) {
    if (!A || !dA_logs || !dDs || !ddt_bias || !ddt_w) { ss2d_set_error("scan_params_bwd: null pointer"); return -2; }
    ScanParamsArgs p{};
    p.A = A; p.a = dA; p.d = dD; p.b = dbias; p.w = dw; p.oa = dA_logs; p.od = dDs; p.ob = ddt_bias; p.ow = ddt_w;
    p.K = K; p.Dm = Dm; p.N = N; p.R = R; p.swap = regroup != 0;
    return run(true, p, stream, "scan_params_bwd");
}

// ---- every parameter-derived operand of one fused-direction SS2D call in ONE launch each way (mamba_sys.py:407, :420,
// :428-431).  Besides the four outputs of scan_params (regrouped) the call needs the x_proj weight with the two directions
// of an orientation stacked, that matrix transposed (input gradient of x_proj), the dt_proj weight transposed (d_dts) and,
// for the ranks whose delta is a GEMM, the dt_proj weight itself -- all four in the activations' storage type with rows
// padded with zeros to 16 bytes, the operand layout of ss2d_chan_gemm.  As torch ops these were a regrouping copy, a
// cast, a transposing copy and a pad per operand and block, each a 4-5 us launch: ~90 launches per training step.
namespace {

struct BlockParamsArgs {
    ScanParamsArgs sp;              // the scan_params part (K = 4, swap = 1)
    const float *xw_src;            // fwd: x_proj_weight (4, Cx, Dm) fp32, reference order   bwd: d_xw (2, 2Cx, Dm) fp32
    void *xw, *xwT, *dtwT, *dtwio;  // fwd outputs (IO type): (2, 2Cx, ldD), (2, Dm, ldC), (4, R, ldD), (4, Dm, ldR; nullable)
    float *dxw;                     // bwd output: d x_proj_weight (4, Cx, Dm) fp32, reference order
    int Cx, ldD, ldC, ldR;
};

// (32-bit index arithmetic throughout: the host checks the element count; 64-bit divisions made this 6.5 us per launch)
template <typename IO, bool BWD>
__global__ __launch_bounds__(256) void block_params_kernel(const BlockParamsArgs q) {
    const ScanParamsArgs &p = q.sp;
    const unsigned Dm = p.Dm, R = p.R, N = p.N, Cx = q.Cx, ldD = q.ldD, ldC = q.ldC, ldR = q.ldR;
    const unsigned n0 = 4u * Dm * (N + 2 + R);
    const unsigned n1 = BWD ? 4u * Cx * Dm : 4u * Cx * ldD;             // xw (bwd: d x_proj_weight)
    const unsigned n2 = BWD ? 0u : 2u * Dm * ldC;                       // xwT
    const unsigned n3 = BWD ? 0u : 4u * R * ldD;                        // dtwT
    const unsigned n4 = (BWD || !q.dtwio) ? 0u : 4u * Dm * ldR;         // dt weight in the IO type
    const unsigned total = n0 + n1 + n2 + n3 + n4;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        if (i < n0) {   // as scan_params_kernel
            const unsigned per_row = N + 2 + R;
            const unsigned row = i / per_row, e = i - row * per_row;
            const unsigned g = row / Dm, d = row - g * Dm;
            const unsigned other = (unsigned)src_group((int)g, 1) * Dm + d;
            if (e < N) {
                if (!BWD) p.oa[row * N + e] = -expf(p.a[other * N + e]);
                else p.oa[other * N + e] = p.a ? p.a[row * N + e] * p.A[row * N + e] : 0.0f;
            } else if (e == N) {
                if (!BWD) p.od[row] = p.d[other]; else p.od[other] = p.d ? p.d[row] : 0.0f;
            } else if (e == N + 1) {
                if (!BWD) p.ob[row] = p.b[other]; else p.ob[other] = p.b ? p.b[row] : 0.0f;
            } else {
                const unsigned r = e - N - 2;
                if (!BWD) p.ow[row * R + r] = p.w[other * R + r]; else p.ow[other * R + r] = p.w ? p.w[row * R + r] : 0.0f;
            }
            continue;
        }
        unsigned j = i - n0;
        if (BWD) {      // d x_proj_weight[k = 2 rev + o][c][d] = d_xw[o][rev Cx + c][d]
            const unsigned rowk = j / Dm, d = j - rowk * Dm;
            const unsigned k = rowk / Cx, c = rowk - k * Cx, o = k & 1, rev = k >> 1;
            q.dxw[j] = q.xw_src ? q.xw_src[((o * 2 + rev) * Cx + c) * Dm + d] : 0.0f;
            continue;
        }
        if (j < n1) {   // xw[o][m = rev Cx + c][d], zero for d >= Dm
            const unsigned rowm = j / ldD, d = j - rowm * ldD;
            const unsigned o = rowm / (2 * Cx), m = rowm - o * 2 * Cx, rev = m >= Cx ? 1u : 0u, c = m - rev * Cx;
            st1(static_cast<IO *>(q.xw) + j, d < Dm ? q.xw_src[((2 * rev + o) * Cx + c) * Dm + d] : 0.0f);
            continue;
        }
        j -= n1;
        if (j < n2) {   // xwT[o][d][m], zero for m >= 2 Cx
            const unsigned rowd = j / ldC, m = j - rowd * ldC;
            const unsigned o = rowd / Dm, d = rowd - o * Dm, rev = m >= Cx ? 1u : 0u, c = m - rev * Cx;
            st1(static_cast<IO *>(q.xwT) + j, m < 2 * Cx ? q.xw_src[((2 * rev + o) * Cx + c) * Dm + d] : 0.0f);
            continue;
        }
        j -= n2;
        if (j < n3) {   // dtwT[g][r][d], zero for d >= Dm
            const unsigned rowr = j / ldD, d = j - rowr * ldD;
            const unsigned g = rowr / R, r = rowr - g * R;
            st1(static_cast<IO *>(q.dtwT) + j, d < Dm ? p.w[((unsigned)src_group((int)g, 1) * Dm + d) * R + r] : 0.0f);
            continue;
        }
        j -= n3;
        {               // dt weight [g][d][r] in the IO type, zero for r >= R
            const unsigned row = j / ldR, r = j - row * ldR;
            const unsigned g = row / Dm, d = row - g * Dm;
            st1(static_cast<IO *>(q.dtwio) + j, r < R ? p.w[((unsigned)src_group((int)g, 1) * Dm + d) * R + r] : 0.0f);
        }
    }
}

}  // namespace

extern "C" int ss2d_block_params_fwd(int dtype, const float *A_logs, const float *Ds, const float *dt_bias, const float *dt_w,
                                     const float *x_w, float *A, float *D, float *bias, float *w, void *xw, void *xwT, void *dtwT,
                                     void *dtw_io, int Dm, int N, int R, int ldD, int ldC, int ldR, void *stream) {
    if (dtype != SS2D_DTYPE_F32 && dtype != SS2D_DTYPE_BF16) { ss2d_set_error("block_params_fwd: unknown dtype %d", dtype); return -1; }
    if (!A_logs || !Ds || !dt_bias || !dt_w || !x_w || !A || !D || !bias || !w || !xw || !xwT || !dtwT) {
        ss2d_set_error("block_params_fwd: null pointer"); return -2; }
    if (Dm <= 0 || N <= 0 || R <= 0) { ss2d_set_error("block_params_fwd: non-positive size"); return -3; }
    const int Cx = R + 2 * N;
    if (ldD < Dm || ldC < 2 * Cx || (dtw_io && ldR < R)) { ss2d_set_error("block_params_fwd: leading dimension smaller than the row"); return -4; }
    BlockParamsArgs q{};
    q.sp.a = A_logs; q.sp.d = Ds; q.sp.b = dt_bias; q.sp.w = dt_w; q.sp.oa = A; q.sp.od = D; q.sp.ob = bias; q.sp.ow = w;
    q.sp.K = 4; q.sp.Dm = Dm; q.sp.N = N; q.sp.R = R; q.sp.swap = 1;
    q.xw_src = x_w; q.xw = xw; q.xwT = xwT; q.dtwT = dtwT; q.dtwio = dtw_io; q.Cx = Cx; q.ldD = ldD; q.ldC = ldC; q.ldR = ldR;
    const long long total = (long long)4 * Dm * (N + 2 + R) + (long long)4 * Cx * ldD + (long long)2 * Dm * ldC + (long long)4 * R * ldD +
                            (dtw_io ? (long long)4 * Dm * ldR : 0);
    if (total >= (1ll << 31)) { ss2d_set_error("block_params_fwd: more than 2^31 elements"); return -6; }
    long long nwg = (total + 255) / 256;
    if (nwg > 2048) nwg = 2048;
    hipStream_t s = (hipStream_t)stream;
    if (dtype == SS2D_DTYPE_BF16) ss2d_launch((block_params_kernel<bf16, false>), dim3((unsigned)nwg), dim3(256), 0, s, q);
    else ss2d_launch((block_params_kernel<float, false>), dim3((unsigned)nwg), dim3(256), 0, s, q);
    const int rc = ss2d_take_launch_rc();
    if (rc) ss2d_set_error("block_params_fwd: launch failed: %s", hipGetErrorString((hipError_t)rc));
    return rc;
}

extern "C" int ss2d_block_params_bwd(const float *A, const float *dA, const float *dD, const float *dbias, const float *dw,
                                     const float *d_xw, float *dA_logs, float *dDs, float *ddt_bias, float *ddt_w, float *dx_w,
                                     int Dm, int N, int R, void *stream) {
    if (!A || !dA_logs || !dDs || !ddt_bias || !ddt_w || !dx_w) { ss2d_set_error("block_params_bwd: null pointer"); return -2; }
    if (Dm <= 0 || N <= 0 || R <= 0) { ss2d_set_error("block_params_bwd: non-positive size"); return -3; }
    BlockParamsArgs q{};
    q.sp.A = A; q.sp.a = dA; q.sp.d = dD; q.sp.b = dbias; q.sp.w = dw; q.sp.oa = dA_logs; q.sp.od = dDs; q.sp.ob = ddt_bias; q.sp.ow = ddt_w;
    q.sp.K = 4; q.sp.Dm = Dm; q.sp.N = N; q.sp.R = R; q.sp.swap = 1;
    q.xw_src = d_xw; q.dxw = dx_w; q.Cx = R + 2 * N;
    const long long total = (long long)4 * Dm * (N + 2 + R) + (long long)4 * q.Cx * Dm;
    if (total >= (1ll << 31)) { ss2d_set_error("block_params_bwd: more than 2^31 elements"); return -6; }
    long long nwg = (total + 255) / 256;
    if (nwg > 2048) nwg = 2048;
    ss2d_launch((block_params_kernel<float, true>), dim3((unsigned)nwg), dim3(256), 0, (hipStream_t)stream, q);
    const int rc = ss2d_take_launch_rc();
    if (rc) ss2d_set_error("block_params_bwd: launch failed: %s", hipGetErrorString((hipError_t)rc));
    return rc;
}

// ---- gradient of the x_proj output, assembled in one pass (mamba_sys.py:417, the autograd of `torch.split`):
//     out[blk][c][l] = c < R ? d_dts[blk][c][l] : c < R+N ? dB[blk][c-R][l] : dC[blk][c-R-N][l]
// d_dts in the activations' storage type, dB / dC the scan backward's fp32 accumulators: the two fp32 -> bf16 cast
// passes and the concatenation of the eager graph are this one kernel (4 elements per lane).
namespace {

struct PackArgs {
    const void *d_dts;
    const float *dB, *dC;
    void *out;
    long long nblocks;
    int R, N, L;
};

template <typename IO, int E>
__global__ __launch_bounds__(256) void xdbl_grad_pack_kernel(const PackArgs p) {
    const int Cx = p.R + 2 * p.N, lq = p.L / E;                     // E-element groups per row
    const long long total = p.nblocks * Cx * lq;
    const IO *__restrict__ dts = static_cast<const IO *>(p.d_dts);
    IO *__restrict__ out = static_cast<IO *>(p.out);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int l = (int)(i % lq) * E;
        const long long row = i / lq;
        const int c = (int)(row % Cx);
        const long long blk = row / Cx;
        IO *o = out + row * p.L + l;
        if (E == 4) {
            f4 v;
            if (c < p.R) v = ld4(dts + (blk * p.R + c) * p.L + l);
            else if (c < p.R + p.N) v = ld4(p.dB + (blk * p.N + c - p.R) * p.L + l);
            else v = ld4(p.dC + (blk * p.N + c - p.R - p.N) * p.L + l);
            st4(o, v);
        } else {
            float v;
            if (c < p.R) v = ld1(dts + (blk * p.R + c) * p.L + l);
            else if (c < p.R + p.N) v = p.dB[(blk * p.N + c - p.R) * p.L + l];
            else v = p.dC[(blk * p.N + c - p.R - p.N) * p.L + l];
            st1(o, v);
        }
    }
}

}  // namespace

extern "C" int ss2d_xdbl_grad_pack(int dtype, const void *d_dts, const float *dB, const float *dC, void *out,
                                   long long nblocks, int R, int N, int L, void *stream) {
    if (dtype != SS2D_DTYPE_F32 && dtype != SS2D_DTYPE_BF16) { ss2d_set_error("xdbl_grad_pack: unknown dtype %d", dtype); return -1; }
    if (!d_dts || !dB || !dC || !out) { ss2d_set_error("xdbl_grad_pack: null pointer"); return -2; }
    if (nblocks <= 0 || R <= 0 || N <= 0 || L <= 0) { ss2d_set_error("xdbl_grad_pack: non-positive size"); return -3; }
    PackArgs p{d_dts, dB, dC, out, nblocks, R, N, L};
    const bool h = dtype == SS2D_DTYPE_BF16;
    const uintptr_t amask = h ? 7 : 15;
    const bool vec = L % 4 == 0 && ((((uintptr_t)d_dts | (uintptr_t)out) & amask) == 0) && ((((uintptr_t)dB | (uintptr_t)dC) & 15) == 0);
    const long long total = nblocks * (R + 2 * N) * (vec ? L / 4 : L);
    long long nwg = (total + 255) / 256;
    if (nwg > 16384) nwg = 16384;
    hipStream_t s = (hipStream_t)stream;
    if (vec) { if (h) ss2d_launch((xdbl_grad_pack_kernel<bf16, 4>), dim3((unsigned)nwg), dim3(256), 0, s, p); else ss2d_launch((xdbl_grad_pack_kernel<float, 4>), dim3((unsigned)nwg), dim3(256), 0, s, p); }
    else     { if (h) ss2d_launch((xdbl_grad_pack_kernel<bf16, 1>), dim3((unsigned)nwg), dim3(256), 0, s, p); else ss2d_launch((xdbl_grad_pack_kernel<float, 1>), dim3((unsigned)nwg), dim3(256), 0, s, p); }
    const int rc = ss2d_take_launch_rc();
    if (rc) ss2d_set_error("xdbl_grad_pack: launch failed: %s", hipGetErrorString((hipError_t)rc));
    return rc;
}
