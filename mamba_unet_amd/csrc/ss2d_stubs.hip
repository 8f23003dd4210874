// TEMPORARY: entry points declared in include/ss2d.h whose kernels have not landed yet.
#include "ss2d_common.h"
#include "../../include/ss2d.h"
#define NOTIMPL(name) { ss2d_set_error(name ": not implemented yet"); return -100; }
extern "C" int ss2d_cross_scan_fwd(int, const void *, void *, int, int, int, int, void *) NOTIMPL("ss2d_cross_scan_fwd")
extern "C" int ss2d_cross_scan_bwd(int, const void *, void *, int, int, int, int, void *) NOTIMPL("ss2d_cross_scan_bwd")
extern "C" int ss2d_cross_merge_fwd(int, const void *, void *, int, int, int, int, int, void *) NOTIMPL("ss2d_cross_merge_fwd")
extern "C" int ss2d_cross_merge_bwd(int, const void *, void *, int, int, int, int, int, void *) NOTIMPL("ss2d_cross_merge_bwd")
extern "C" int ss2d_dwconv3x3_silu_fwd(int, const void *, const float *, const float *, void *, int, int, int, int, int, int, void *) NOTIMPL("ss2d_dwconv3x3_silu_fwd")
extern "C" int ss2d_dwconv3x3_silu_bwd(int, const void *, const float *, const float *, const void *, void *, float *, float *, int, int, int, int, int, int, int, void *) NOTIMPL("ss2d_dwconv3x3_silu_bwd")
