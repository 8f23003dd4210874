// Pieces shared by the selective-scan kernels (ss2d_scan_fwd.hip, ss2d_scan_bwd.hip, ss2d_scan_bwd_sr.hip):
// the DPP row-scan / row-sum / cross-row primitives and the backward's parameter block.
#pragma once
#include "ss2d_common.h"

struct ScanBwdParams {
    const void *u, *delta, *B, *C, *z, *dout;   // IO type (fp32 or bf16)
    const float *A, *D, *bias, *ckpt;
    void *du, *ddelta, *dz;                      // IO type
    float *dA, *dB, *dC, *dD, *dbias;            // fp32 accumulators (atomics)
    int batch, dim, L, N, G, dpg, nchunks, softplus, wgs_per_group, nwg, rounds, cb;
    int Gu;              // u and dout have Gu groups per batch; scan group g uses group g / (G / Gu)
    unsigned rev_mask;   // bit g: group g runs over its rows in reverse memory order
    long long bc_stride; // elements between consecutive (batch, group) blocks of B and of C (N*L when dense); dB/dC are dense
    int rows_per_wg;     // state-row kernel: (batch, channel) rows per workgroup
    float *ws;           // state-row kernel: partial dB/dC tiles (one per workgroup and (batch, group) it touches), or null = atomics
    const float *dtw;    // fused-delta form (null otherwise): dt_proj weight (dim, R); `delta` then points at the dts rows (R, L) of
    int R;               // every (batch, group) block (block stride bc_stride) and delta = dtw . dts is recomputed in the kernel
    void *dxdbl;         // fused-delta form, nullable: the gradient of the x_proj output (IO type, x_dbl's layout); the reduction pass
                         // then writes dB | dC as rows [R, R + 32) of every block instead of the fp32 dB / dC tensors
};

// The state-row backward (ss2d_scan_bwd_sr.hip): dstate == 16, no z.  Returns a hipError_t / 0.
// ws (nullable): ss2d_scan_bwd_sr_ws_bytes() bytes; with it dB/dC are written with plain stores + one reduction
// kernel (deterministic, no zero-fill needed), without it they are accumulated with float atomics.
int ss2d_scan_bwd_sr_launch(int dtype, ScanBwdParams p, bool vec4, bool dirs, float *ws, size_t ws_bytes, hipStream_t s);
size_t ss2d_scan_bwd_sr_ws_bytes(int batch, int dim, int L, int ngroups);
// Device address (on the current device) of the SS2D_STATIC_WS_BYTES workspace that lives in the code object.
int ss2d_scan_bwd_sr_static_ws(float **ptr);

// Packed fp32 math with one operand broadcast from HALF of a register pair (op_sel), so that `splat(x) * v` needs no
// v_mov to build the splat: `s` is a pair of neighbouring values (two consecutive positions as they come out of an
// LDS read), _lo uses s.x for both lanes of the packed op, _hi uses s.y.  The operands must not come straight out of
// a transcendental instruction (the compiler's gfx950 hazard pass does not look inside inline asm).
__device__ __forceinline__ f2 pk_mul_lo(f2 s, f2 b) { f2 d; asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(d) : "v"(s), "v"(b)); return d; }
__device__ __forceinline__ f2 pk_mul_hi(f2 s, f2 b) { f2 d; asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1]" : "=v"(d) : "v"(s), "v"(b)); return d; }
__device__ __forceinline__ f2 pk_fma_lo(f2 s, f2 b, f2 c) { f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(s), "v"(b), "v"(c)); return d; }
__device__ __forceinline__ f2 pk_fma_hi(f2 s, f2 b, f2 c) { f2 d; asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(s), "v"(b), "v"(c)); return d; }
template <int E> __device__ __forceinline__ f2 pk_mul_sel(f2 s, f2 b) { return E ? pk_mul_hi(s, b) : pk_mul_lo(s, b); }
template <int E> __device__ __forceinline__ f2 pk_fma_sel(f2 s, f2 b, f2 c) { return E ? pk_fma_hi(s, b, c) : pk_fma_lo(s, b, c); }
// c += splat(s.E) * b, in place
template <int E> __device__ __forceinline__ void pk_fmac_sel(f2 &c, f2 s, f2 b) {
    if (E) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(c) : "v"(s), "v"(b));
    else asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(c) : "v"(s), "v"(b));
}

#define SS2D_FSTEP(n)                                                                  \
    "v_fmac_f32_dpp %0, %0, %2 row_shr:" #n " row_mask:0xf bank_mask:0xf\n\t"         \
    "v_fmac_f32_dpp %1, %1, %3 row_shr:" #n " row_mask:0xf bank_mask:0xf\n\t"         \
    "v_mul_f32_dpp %2, %2, %2 row_shr:" #n " row_mask:0xf bank_mask:0xf\n\t"          \
    "v_mul_f32_dpp %3, %3, %3 row_shr:" #n " row_mask:0xf bank_mask:0xf\n\t"
#define SS2D_RSTEP(n)                                                                  \
    "v_fmac_f32_dpp %0, %0, %2 row_shl:" #n " row_mask:0xf bank_mask:0xf\n\t"         \
    "v_fmac_f32_dpp %1, %1, %3 row_shl:" #n " row_mask:0xf bank_mask:0xf\n\t"         \
    "v_mul_f32_dpp %2, %2, %2 row_shl:" #n " row_mask:0xf bank_mask:0xf\n\t"          \
    "v_mul_f32_dpp %3, %3, %3 row_shl:" #n " row_mask:0xf bank_mask:0xf\n\t"

// inclusive prefix scan of (a, h) along the 16 lanes of a row: h_i <- h_i + a_i * h_{i-s}.  Two independent
// states interleaved so that every DPP read is >= 2 instructions behind the write of its source (hardware hazard).
__device__ __forceinline__ void row_scan_fwd_pair(float &hx, float &hy, float &ax, float &ay) {
    asm volatile("s_nop 1\n\t" SS2D_FSTEP(1) SS2D_FSTEP(2) SS2D_FSTEP(4)
                 "v_fmac_f32_dpp %0, %0, %2 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %1, %1, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 : "+v"(hx), "+v"(hy), "+v"(ax), "+v"(ay));
}
// inclusive suffix scan: r_i <- r_i + a_i * r_{i+s}
__device__ __forceinline__ void row_scan_rev_pair(float &rx, float &ry, float &ax, float &ay) {
    asm volatile("s_nop 1\n\t" SS2D_RSTEP(1) SS2D_RSTEP(2) SS2D_RSTEP(4)
                 "v_fmac_f32_dpp %0, %0, %2 row_shl:8 row_mask:0xf bank_mask:0xf\n\t"
                 "v_fmac_f32_dpp %1, %1, %3 row_shl:8 row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\t"
                 : "+v"(rx), "+v"(ry), "+v"(ax), "+v"(ay));
}
// sum over the 16 lanes of a row; the total lands in lane 15 of the row.  Two values at once so that the DPP
// read-after-write hazard of one chain is covered by the other; the shifted operand reads 0 outside the row
// (bound_ctrl), and the add is fused into the DPP instruction (the compiler emits v_mov_dpp + v_add instead).
__device__ __forceinline__ void row_sum2_to_lane15(float &x, float &y) {
#define SS2D_ASTEP(n)                                                                              \
    "v_add_f32_dpp %0, %0, %0 row_shr:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
    "v_add_f32_dpp %1, %1, %1 row_shr:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
    "s_nop 0\n\t"
    asm volatile("s_nop 1\n\t" SS2D_ASTEP(1) SS2D_ASTEP(2) SS2D_ASTEP(4) SS2D_ASTEP(8) : "+v"(x), "+v"(y));
#undef SS2D_ASTEP
}

// six row sums at once: the chains cover each other's DPP hazard, no s_nop between the steps
__device__ __forceinline__ void row_sum6_to_lane15(float &a, float &b, float &c, float &d, float &e, float &f) {
#define SS2D_ASTEP6(n)                                                                             \
    "v_add_f32_dpp %0, %0, %0 row_shr:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
    "v_add_f32_dpp %1, %1, %1 row_shr:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
    "v_add_f32_dpp %2, %2, %2 row_shr:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
    "v_add_f32_dpp %3, %3, %3 row_shr:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
    "v_add_f32_dpp %4, %4, %4 row_shr:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
    "v_add_f32_dpp %5, %5, %5 row_shr:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm volatile("s_nop 1\n\t" SS2D_ASTEP6(1) SS2D_ASTEP6(2) SS2D_ASTEP6(4) SS2D_ASTEP6(8)
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
#undef SS2D_ASTEP6
}

// the same over the two halves of a row: lane 7 gets the sum of lanes 0-7, lane 15 that of lanes 8-15 (the packed
// backward: two channels per row)
__device__ __forceinline__ void row_sum6_to_lanes7_15(float &a, float &b, float &c, float &d, float &e, float &f) {
#define SS2D_ASTEP6(n)                                                                             \
    "v_add_f32_dpp %0, %0, %0 row_shr:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
    "v_add_f32_dpp %1, %1, %1 row_shr:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
    "v_add_f32_dpp %2, %2, %2 row_shr:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
    "v_add_f32_dpp %3, %3, %3 row_shr:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
    "v_add_f32_dpp %4, %4, %4 row_shr:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"          \
    "v_add_f32_dpp %5, %5, %5 row_shr:" #n " row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
    asm volatile("s_nop 1\n\t" SS2D_ASTEP6(1) SS2D_ASTEP6(2) SS2D_ASTEP6(4)
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
#undef SS2D_ASTEP6
}

// Sum four per-lane values over the 4 DPP rows of the wave: on return row r of the result holds the
// total (over rows) of x_r.  v_permlane16_swap exchanges the odd rows of its first operand with the
// even rows of its second, v_permlane32_swap the upper half of the first with the lower half of the
// second; each swap is followed by one add (1.5 VALU ops per value instead of an LDS float atomic,
// which runs ~3 cycles PER LANE on gfx950).  Inline asm: the clang builtin returns a wrong second result.
__device__ __forceinline__ void rows_reduce4x2(float &x0, float x1, float x2, float x3, float &y0, float y1, float y2, float y3) {
    // two independent 4-value reductions in one block: fewer hazard nops (x0 / y0 receive the results)
    asm volatile("s_nop 1\n\t"
                 "v_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\t"
                 "v_permlane16_swap_b32 %4, %5\n\tv_permlane16_swap_b32 %6, %7\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32 %0, %0, %1\n\tv_add_f32 %2, %2, %3\n\tv_add_f32 %4, %4, %5\n\tv_add_f32 %6, %6, %7\n\t"
                 "s_nop 1\n\t"
                 "v_permlane32_swap_b32 %0, %2\n\tv_permlane32_swap_b32 %4, %6\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32 %0, %0, %2\n\tv_add_f32 %4, %4, %6\n\t"
                 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3));
}
