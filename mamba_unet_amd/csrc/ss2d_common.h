// Shared device helpers for the SS2D kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>
#include <tuple>
#include <utility>

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

#define SS2D_MAX_DEVICES 64   // per-device host-side caches (function attributes, symbol addresses)
#define SS2D_LOG2E 1.4426950408889634f
#define SS2D_LN2 0.6931471805599453f

__device__ __forceinline__ float hw_exp2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32
__device__ __forceinline__ float hw_log2(float x) { return __builtin_amdgcn_logf(x); }   // v_log_f32
__device__ __forceinline__ float hw_rcp(float x) { return __builtin_amdgcn_rcpf(x); }    // v_rcp_f32

__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat2(float x) { return f2{x, x}; }
__device__ __forceinline__ f2 exp2_2(f2 x) { return f2{hw_exp2(x.x), hw_exp2(x.y)}; }

// softplus(x) with torch's threshold 20; also returns sigmoid(x) (= d softplus / dx) when asked.
// log1p(e) is evaluated as log(w) + (e - (w - 1)), w = 1 + e: w - 1 is exact, so e - (w - 1) is the part of e that
// rounding 1 + e dropped, and d log(w) / dw = 1 / w is 1 to within e where that part matters (e << 1) and the part is
// below half an ulp of w where it is not.  Full relative accuracy for the small e (delta ~ 1e-3) this model lives on,
// two transcendentals and no division (the form log(w) * e / (w - 1) costs a v_rcp_f32 and a select more).
template <bool WANT_SIG>
__device__ __forceinline__ float softplus_sig(float x, float &sig) {
    const float e = hw_exp2(x * SS2D_LOG2E);
    const float w = 1.0f + e;
    const float r = fmaf(hw_log2(w), SS2D_LN2, e - (w - 1.0f));
    if (WANT_SIG) sig = (x > 20.0f) ? 1.0f : e * hw_rcp(w);
    return (x > 20.0f) ? x : r;
}

// x * sigmoid(x) and its derivative pieces
__device__ __forceinline__ float sigmoidf_fast(float x) { return hw_rcp(1.0f + hw_exp2(-x * SS2D_LOG2E)); }

template <int CTRL, int ROW_MASK, bool BOUND_ZERO>
__device__ __forceinline__ float dpp_mov(float old, float src) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, src),
                                                                 CTRL, ROW_MASK, 0xf, BOUND_ZERO));
}
#define DPP_ROW_SHR(n) (0x110 + (n))
#define DPP_ROW_SHL(n) (0x100 + (n))
#define DPP_ROW_ROR(n) (0x120 + (n))

// XCD-aware bijective remap: hardware deals consecutive workgroup ids round-robin over the 8 XCDs,
// so logical ids that should share an L2 (same B/C tile) are made consecutive within one XCD.
__device__ __forceinline__ int xcd_remap(int orig, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = orig & 7;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (orig >> 3);
}

// ---- element I/O in the two storage types of the ABI (fp32, bf16); arithmetic is always fp32.
// bf16 -> fp32 is a shift, fp32 -> bf16 is v_cvt_pk_bf16_f32 (round to nearest even) on gfx950.
typedef __bf16 bf16;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float ld1(const float *p) { return *p; }
__device__ __forceinline__ float ld1(const bf16 *p) { return (float)*p; }
__device__ __forceinline__ void st1(float *p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16 *p, float v) { *p = (bf16)v; }
__device__ __forceinline__ f4 ld4(const float *p) { return *reinterpret_cast<const f4 *>(p); }
__device__ __forceinline__ f4 ld4(const bf16 *p) { return __builtin_convertvector(*reinterpret_cast<const bf16x4 *>(p), f4); }
__device__ __forceinline__ void st4(float *p, f4 v) { *reinterpret_cast<f4 *>(p) = v; }
__device__ __forceinline__ void st4(bf16 *p, f4 v) { *reinterpret_cast<bf16x4 *>(p) = __builtin_convertvector(v, bf16x4); }

// 8 consecutive elements per lane, zero-filled past `L`.  VEC4 needs L % 4 == 0 and a base pointer aligned to
// 4 elements (16 bytes for fp32, 8 bytes for bf16).
template <bool VEC4, typename IO>
__device__ __forceinline__ void load8(const IO *__restrict__ row, int p0, int L, bool ok, float (&v)[8]) {
    if (VEC4) {
        f4 a = f4{0, 0, 0, 0}, b = f4{0, 0, 0, 0};
        if (ok && p0 < L) a = ld4(row + p0);
        if (ok && p0 + 4 < L) b = ld4(row + p0 + 4);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = (ok && p0 + t < L) ? ld1(row + p0 + t) : 0.0f;
    }
}
template <bool VEC4, typename IO>
__device__ __forceinline__ void store8(IO *__restrict__ row, int p0, int L, bool ok, const float (&v)[8]) {
    if (VEC4) {
        if (ok && p0 < L) st4(row + p0, f4{v[0], v[1], v[2], v[3]});
        if (ok && p0 + 4 < L) st4(row + p0 + 4, f4{v[4], v[5], v[6], v[7]});
    } else {
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (ok && p0 + t < L) st1(row + p0 + t, v[t]);
    }
}

// ---- mirrored variants: `rev` (wave-uniform) means scan position l lives at memory index L-1-l, which is how a
// reversed scan direction reads and writes the un-reversed tensors in place (no flipped copy in HBM).
template <bool VEC4, typename IO>
__device__ __forceinline__ void load8m(const IO *__restrict__ row, int p0, int L, bool ok, bool rev, float (&v)[8]) {
    if (!rev) { load8<VEC4>(row, p0, L, ok, v); return; }
    if (VEC4) {  // L % 4 == 0 and p0 % 4 == 0: each group of 4 is entirely inside or entirely outside
        f4 a = f4{0, 0, 0, 0}, b = f4{0, 0, 0, 0};
        if (ok && p0 < L) a = ld4(row + (L - 4 - p0));
        if (ok && p0 + 4 < L) b = ld4(row + (L - 8 - p0));
        v[0] = a.w; v[1] = a.z; v[2] = a.y; v[3] = a.x; v[4] = b.w; v[5] = b.z; v[6] = b.y; v[7] = b.x;
    } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = (ok && p0 + t < L) ? ld1(row + (L - 1 - p0 - t)) : 0.0f;
    }
}
template <bool VEC4, typename IO>
__device__ __forceinline__ void store8m(IO *__restrict__ row, int p0, int L, bool ok, bool rev, const float (&v)[8]) {
    if (!rev) { store8<VEC4>(row, p0, L, ok, v); return; }
    if (VEC4) {
        if (ok && p0 < L) st4(row + (L - 4 - p0), f4{v[3], v[2], v[1], v[0]});
        if (ok && p0 + 4 < L) st4(row + (L - 8 - p0), f4{v[7], v[6], v[5], v[4]});
    } else {
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (ok && p0 + t < L) st1(row + (L - 1 - p0 - t), v[t]);
    }
}
// compile-time switch: DIRS = false is exactly load8 / store8
template <bool VEC4, bool DIRS, typename IO>
__device__ __forceinline__ void load8d(const IO *__restrict__ row, int p0, int L, bool ok, bool rev, float (&v)[8]) {
    if (DIRS) load8m<VEC4>(row, p0, L, ok, rev, v); else load8<VEC4>(row, p0, L, ok, v);
}
template <bool VEC4, bool DIRS, typename IO>
__device__ __forceinline__ void store8d(IO *__restrict__ row, int p0, int L, bool ok, bool rev, const float (&v)[8]) {
    if (DIRS) store8m<VEC4>(row, p0, L, ok, rev, v); else store8<VEC4>(row, p0, L, ok, v);
}
// 4 consecutive scan positions pos..pos+3 of one row (zero past L)
template <bool VEC4, typename IO>
__device__ __forceinline__ f4 load4m(const IO *__restrict__ row, int pos, int L, bool rev) {
    f4 r = f4{0, 0, 0, 0};
    if (VEC4) {
        if (pos < L) {
            if (!rev) r = ld4(row + pos);
            else { const f4 t = ld4(row + (L - 4 - pos)); r = f4{t.w, t.z, t.y, t.x}; }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (pos + e < L) r[e] = ld1(row + (rev ? L - 1 - pos - e : pos + e));
    }
    return r;
}

// 2 consecutive scan positions pos, pos+1 (pos even).  The address is clamped instead of the load being guarded (a
// guarded load is sunk by the compiler into its branch with a full wait, which defeats prefetching); `ok` says
// whether the value counts.  VEC needs L % 2 == 0 and a base pointer aligned to 2 elements.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 ld2(const float *p) { return *reinterpret_cast<const f2 *>(p); }
__device__ __forceinline__ f2 ld2(const bf16 *p) { return __builtin_convertvector(*reinterpret_cast<const bf16x2 *>(p), f2); }
__device__ __forceinline__ void st2(float *p, f2 v) { *reinterpret_cast<f2 *>(p) = v; }
__device__ __forceinline__ void st2(bf16 *p, f2 v) { *reinterpret_cast<bf16x2 *>(p) = __builtin_convertvector(v, bf16x2); }
template <bool VEC, typename IO>
__device__ __forceinline__ f2 load2m_nomask(const IO *__restrict__ row, int pos, int L, bool rev) {
    if (VEC) {
        const int ps = pos < L ? pos : 0;
        const f2 t = ld2(row + (unsigned)(rev ? L - 2 - ps : ps));
        return rev ? f2{t.y, t.x} : t;
    }
    f2 r;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int ps = pos + e < L ? pos + e : 0;
        r[e] = ld1(row + (unsigned)(rev ? L - 1 - ps : ps));
    }
    return r;
}
// A wave-uniform element offset pinned into scalar registers: `base + uniform_off(x) + lane_offset` then selects the
// saddr form of global_load / global_store (SGPR base + 32-bit lane offset) instead of a 64-bit per-lane address that
// the compiler keeps alive across a loop.  (Pinning the POINTER through integers would lose the address space and
// turn every access into a flat_ one.)
__device__ __forceinline__ size_t uniform_off(size_t v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((size_t)hi << 32) | lo;
}
template <bool VEC, typename IO>
__device__ __forceinline__ void store2m(IO *__restrict__ row, int pos, int L, bool rev, f2 v) {
    if (VEC) {
        if (pos < L) st2(row + (unsigned)(rev ? L - 2 - pos : pos), rev ? f2{v.y, v.x} : v);
    } else {
#pragma unroll
        for (int e = 0; e < 2; ++e)
            if (pos + e < L) st1(row + (unsigned)(rev ? L - 1 - pos - e : pos + e), v[e]);
    }
}

// thread-local error string shared by the C-ABI entry points
void ss2d_set_error(const char *fmt, ...);

// Kernel launches report through hipLaunchKernel's own return value, kept in a thread-local until the entry point
// collects it: hipGetLastError() would also return -- and clear -- an error that another library (torch) left on
// this thread, misattributing it to this library and hiding it from its owner.
inline int &ss2d_launch_rc_ref() { static thread_local int rc = 0; return rc; }
inline int ss2d_take_launch_rc() { int &r = ss2d_launch_rc_ref(); const int v = r; r = 0; return v; }   // first failure since the last call
template <typename... KArgs, size_t... I>
inline hipError_t ss2d_launch_impl(void (*kern)(KArgs...), dim3 g, dim3 b, size_t shm, hipStream_t s, std::tuple<KArgs...> &t,
                                   std::index_sequence<I...>) {
    void *ptrs[] = {static_cast<void *>(&std::get<I>(t))...};
    return hipLaunchKernel(reinterpret_cast<const void *>(kern), g, b, ptrs, shm, s);
}
template <typename... KArgs, typename... Args>
inline void ss2d_launch(void (*kern)(KArgs...), dim3 grid, dim3 block, size_t shm, hipStream_t s, Args &&...args) {
    std::tuple<KArgs...> t{static_cast<KArgs>(args)...};
    const hipError_t e = ss2d_launch_impl(kern, grid, block, shm, s, t, std::index_sequence_for<KArgs...>{});
    if (e != hipSuccess && ss2d_launch_rc_ref() == 0) ss2d_launch_rc_ref() = (int)e;
}
// tuning knobs from the environment are read once per process
inline long long ss2d_env_ll(const char *name, long long dflt) { const char *e = getenv(name); return e ? atoll(e) : dflt; }

// Sum over the 64 lanes of the wave (LW = 64) or over this lane's 32-lane half (LW = 32), result in every lane:
// four rotate-and-add DPP steps inside each 16-lane row (VALU), then the row totals through v_readlane.  A
// __shfl_xor butterfly is five or six ds_bpermute (LDS-pipe) instructions per sum, and that pipe rather than HBM
// bounded the row-per-wave norm kernels.
template <int LW>
__device__ __forceinline__ float lanes_sum(float v) {
    v += dpp_mov<DPP_ROW_ROR(8), 0xf, false>(0.0f, v);
    v += dpp_mov<DPP_ROW_ROR(4), 0xf, false>(0.0f, v);
    v += dpp_mov<DPP_ROW_ROR(2), 0xf, false>(0.0f, v);
    v += dpp_mov<DPP_ROW_ROR(1), 0xf, false>(0.0f, v);
    const int iv = __builtin_bit_cast(int, v);
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float c = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    if (LW == 64) return a + c;
    return (threadIdx.x & 32) ? c : a;
}

// Workgroup column sums for the dgamma / dbeta of the norm kernels: every lane holds K four-element partial sums of
// two quantities for column chunk (lane % LW) + LW*k.  Half-waves that share columns (LW = 32) are folded with a
// lane exchange, every wave stores its partials in its own LDS slot, and after one barrier the threads add the NWV
// slots of a column and send ONE global atomic per column and workgroup.  (LDS float atomics instead of the slots
// cost 7 us per launch at 1024 threads.)  red: NWV * 2 * K * 256 floats.
template <int K, int NWV, int LW>
__device__ __forceinline__ void wg_colsum2(float *red, f4 (&a)[K], f4 (&b)[K], int D, float *out_a, float *out_b) {
    const int lane64 = threadIdx.x & 63, wave = threadIdx.x >> 6, nchunk = D >> 2;
    if (LW == 32) {
#pragma unroll
        for (int k = 0; k < K; ++k)
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[k][e] += __shfl_xor(a[k][e], 32, 64); b[k][e] += __shfl_xor(b[k][e], 32, 64); }
    }
    if (lane64 < LW) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int c = lane64 + LW * k;
            if (c < nchunk) {
                *reinterpret_cast<f4 *>(red + (wave * 2 + 0) * K * 256 + 4 * c) = a[k];
                *reinterpret_cast<f4 *>(red + (wave * 2 + 1) * K * 256 + 4 * c) = b[k];
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * D; i += NWV * 64) {
        const int which = i >= D, col = which ? i - D : i;
        float s = 0.0f;
#pragma unroll
        for (int w = 0; w < NWV; ++w) s += red[(w * 2 + which) * K * 256 + col];
        atomicAdd((which ? out_b : out_a) + col, s);
    }
}

// ---- raw buffer access (SGPR resource + 32-bit lane offset): the address of a wave-uniform row costs no VGPR, and the
// hardware range check (offset >= num_records reads 0 / drops the store) replaces the tail masks of a ragged row.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, bytes, 0x00020000);   // raw, untyped dwords
}
__device__ __forceinline__ f2 bf16pair_to_f2(unsigned v) { return f2{__builtin_bit_cast(float, v << 16), __builtin_bit_cast(float, v & 0xffff0000u)}; }
// 2 consecutive scan positions pos, pos+1 (pos even) of the row behind `rs` (num_records = row bytes); positions past
// the row read 0.  VEC: L even, row base aligned to 2 elements.
template <bool VEC, typename IO>
__device__ __forceinline__ f2 buf_load2m(__amdgpu_buffer_rsrc_t rs, int pos, int L, bool rev) {
    constexpr bool F32 = sizeof(IO) == 4;
    if (VEC) {
        const int idx = rev ? L - 2 - pos : pos;   // negative -> huge unsigned offset -> out of range -> 0
        f2 t;
        if (F32) t = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rs, idx * 4, 0, 0));
        else t = bf16pair_to_f2(__builtin_amdgcn_raw_buffer_load_b32(rs, idx * 2, 0, 0));
        return rev ? f2{t.y, t.x} : t;
    }
    f2 r;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int idx = rev ? L - 1 - pos - e : pos + e;
        if (F32) r[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, idx * 4, 0, 0));
        else r[e] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, idx * 2, 0, 0) << 16);
    }
    return r;
}
template <bool VEC, typename IO>
__device__ __forceinline__ void buf_store2m(__amdgpu_buffer_rsrc_t rs, int pos, int L, bool rev, f2 v) {
    constexpr bool F32 = sizeof(IO) == 4;
    if (VEC) {
        const int idx = rev ? L - 2 - pos : pos;
        const f2 w = rev ? f2{v.y, v.x} : v;
        if (F32) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, w), rs, idx * 4, 0, 0);
        else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, __builtin_convertvector(w, bf16x2)), rs, idx * 2, 0, 0);
    } else {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int idx = rev ? L - 1 - pos - e : pos + e;
            const float ve = v[e];   // a copy: __builtin_bit_cast applied to the vector ELEMENT v[e] reads element 0 (hipcc 7.2)
            if (F32) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ve), rs, idx * 4, 0, 0);
            else __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (bf16)ve), rs, idx * 2, 0, 0);
        }
    }
}
// The same two accesses for a block of SEVERAL rows behind one descriptor (num_records = the whole block, base = its first
// row) with the row chosen by a scalar byte offset `soff`: gfx950 includes the scalar offset in the range check (an access
// is dropped when voffset >= num_records - soffset: tools/ubench/soffset_check.hip), so the descriptor cannot be one row
// long, and positions past the row end are pushed out of range explicitly.  The block must be shorter than 2^31 bytes.
constexpr int SS2D_BUF_OOR = 0x7ffffff0;   // + any soff < 2^31 stays a 32-bit offset beyond every such block
template <bool VEC, typename IO>
__device__ __forceinline__ f2 buf_load2s(__amdgpu_buffer_rsrc_t rs, int pos, int L, bool rev, int soff, int vextra = 0) {
    // vextra (VEC only): a per-lane byte offset on top of the scalar one (lanes of one wave on different rows)
    constexpr bool F32 = sizeof(IO) == 4;
    if (VEC) {
        const int off = pos < L ? (rev ? L - 2 - pos : pos) * (int)sizeof(IO) + vextra : SS2D_BUF_OOR;
        f2 t;
        if (F32) t = __builtin_bit_cast(f2, __builtin_amdgcn_raw_buffer_load_b64(rs, off, soff, 0));
        else t = bf16pair_to_f2(__builtin_amdgcn_raw_buffer_load_b32(rs, off, soff, 0));
        return rev ? f2{t.y, t.x} : t;
    }
    f2 r;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const int off = pos + e < L ? (rev ? L - 1 - pos - e : pos + e) * (int)sizeof(IO) : SS2D_BUF_OOR;
        if (F32) r[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, soff, 0));
        else r[e] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, off, soff, 0) << 16);
    }
    return r;
}
template <bool VEC, typename IO>
__device__ __forceinline__ void buf_store2s(__amdgpu_buffer_rsrc_t rs, int pos, int L, bool rev, f2 v, int soff, int vextra = 0) {
    constexpr bool F32 = sizeof(IO) == 4;
    if (VEC) {
        const int off = pos < L ? (rev ? L - 2 - pos : pos) * (int)sizeof(IO) + vextra : SS2D_BUF_OOR;
        const f2 w = rev ? f2{v.y, v.x} : v;
        if (F32) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, w), rs, off, soff, 0);
        else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, __builtin_convertvector(w, bf16x2)), rs, off, soff, 0);
    } else {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int off = pos + e < L ? (rev ? L - 1 - pos - e : pos + e) * (int)sizeof(IO) : SS2D_BUF_OOR;
            const float ve = v[e];   // a copy (see buf_store2m)
            if (F32) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, ve), rs, off, soff, 0);
            else __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (bf16)ve), rs, off, soff, 0);
        }
    }
}
// 4 consecutive scan positions pos..pos+3 (pos % 4 == 0) of row `row_elems`-th element on inside the block behind `rs`
template <bool VEC, typename IO>
__device__ __forceinline__ f4 buf_load4m(__amdgpu_buffer_rsrc_t rs, unsigned row_off, int pos, int L, bool rev) {
    constexpr bool F32 = sizeof(IO) == 4;
    f4 r;
    if (VEC) {
        const int idx = rev ? L - 4 - pos : pos;
        const unsigned off = (pos < L) ? row_off + (unsigned)idx : 0xffffffffu / 8;   // past the row: out of range -> 0
        if (F32) r = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(rs, off * 4, 0, 0));
        else {
            const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs, off * 2, 0, 0);
            const f2 a = bf16pair_to_f2(t.x), b = bf16pair_to_f2(t.y);
            r = f4{a.x, a.y, b.x, b.y};
        }
        return rev ? f4{r.w, r.z, r.y, r.x} : r;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int idx = rev ? L - 1 - pos - e : pos + e;
        const unsigned off = (pos + e < L) ? row_off + (unsigned)idx : 0xffffffffu / 8;
        if (F32) r[e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off * 4, 0, 0));
        else r[e] = __builtin_bit_cast(float, (unsigned)__builtin_amdgcn_raw_buffer_load_b16(rs, off * 2, 0, 0) << 16);
    }
    return r;
}

// ---- fp32 operands on the bf16 MFMA.  An fp32 value is the EXACT sum of three bf16 values -- its top 8 significant bits,
// the next 8 and the last 8, taken by truncation: v = h1 + h2 + h3 with |h2| < 2^-7 |v|, |h3| < 2^-15 |v|.  A product of two
// such sums keeps six terms (h1h1, h1h2, h2h1, h1h3, h3h1, h2h2); the three dropped ones are below 2^-21 of the product,
// the rounding level of a long fp32 accumulation.  Six v_mfma_f32_32x32x16_bf16 (exact bf16 products, fp32 sums) then do the
// work of eight v_mfma_f32_32x32x2_f32 in 3/8 of the MFMA cycles: the fp32 GEMMs of this library are built on this.
// (Inf / NaN: inf - inf in the remainder turns an Inf operand into NaN products.)
struct Split4 { u32x2 h1, h2, h3; };          // four fp32 values -> four bf16 values of each rank
__device__ __forceinline__ unsigned bf16_hi_pair(unsigned lo, unsigned hi) {   // the upper halves of two registers in one
    return __builtin_amdgcn_perm(hi, lo, 0x07060302u);
}
__device__ __forceinline__ Split4 split3_bf16(const u32x4 v) {
    const unsigned u0 = v.x, u1 = v.y, u2 = v.z, u3 = v.w;       // (components by name: see buf_store2m)
    const float r0 = __builtin_bit_cast(float, u0) - __builtin_bit_cast(float, u0 & 0xffff0000u);
    const float r1 = __builtin_bit_cast(float, u1) - __builtin_bit_cast(float, u1 & 0xffff0000u);
    const float r2 = __builtin_bit_cast(float, u2) - __builtin_bit_cast(float, u2 & 0xffff0000u);
    const float r3 = __builtin_bit_cast(float, u3) - __builtin_bit_cast(float, u3 & 0xffff0000u);
    const unsigned s0 = __builtin_bit_cast(unsigned, r0), s1 = __builtin_bit_cast(unsigned, r1);
    const unsigned s2 = __builtin_bit_cast(unsigned, r2), s3 = __builtin_bit_cast(unsigned, r3);
    const float q0 = r0 - __builtin_bit_cast(float, s0 & 0xffff0000u), q1 = r1 - __builtin_bit_cast(float, s1 & 0xffff0000u);
    const float q2 = r2 - __builtin_bit_cast(float, s2 & 0xffff0000u), q3 = r3 - __builtin_bit_cast(float, s3 & 0xffff0000u);
    Split4 o;
    o.h1 = u32x2{bf16_hi_pair(u0, u1), bf16_hi_pair(u2, u3)};
    o.h2 = u32x2{bf16_hi_pair(s0, s1), bf16_hi_pair(s2, s3)};
    o.h3 = u32x2{bf16_hi_pair(__builtin_bit_cast(unsigned, q0), __builtin_bit_cast(unsigned, q1)),
                 bf16_hi_pair(__builtin_bit_cast(unsigned, q2), __builtin_bit_cast(unsigned, q3))};
    return o;
}
