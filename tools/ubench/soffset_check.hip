// Is the scalar offset of a raw buffer access part of the range check on gfx950?  (One descriptor per part + a scalar row
// offset per channel in ss2d_scan_bwd_sr.hip relies on the answer.)  Build: hipcc --offload-arch=gfx950 -O2 -o soffset_check soffset_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(const float *base, float *out, int soff) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(base), 0, 64, 0x00020000);
    const int lane = threadIdx.x;   // 32 lanes: voffset 0 .. 124 bytes, records = 64 bytes
    out[lane] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane * 4, soff, 0));
    float *ob = out + 64;
    const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(ob, 0, 64, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, 7.0f + lane), ws, lane * 4, soff, 0);
}
int main() {
    float *d, *o, h[1024], r[1024];
    for (int i = 0; i < 1024; ++i) h[i] = 1000.0f + i;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(r));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipMemset(o, 0, sizeof(r));
    hipLaunchKernelGGL(probe, dim3(1), dim3(32), 0, 0, d, o, 1024);
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("loads with soffset 1024 B, records 64 B:");
    for (int i = 0; i < 32; ++i) printf(" %g", r[i]);
    printf("\nstores (region at soffset):");
    for (int i = 0; i < 32; ++i) printf(" %g", r[64 + 256 + i]);
    printf("\n");
    return 0;
}
