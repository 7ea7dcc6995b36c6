/* crs_common.h — pieces shared by the CRS2_LM gather kernels (crs_kernels.hip: the resumable gather-sum of a pass;
 * crs_chain.hip: the same sum with the accept/reject chain resolved inside the launch). */
#ifndef NLA_CRS_COMMON_H
#define NLA_CRS_COMMON_H
#include "dev_common.h"

/* vector width helpers for the gather-sum */
template <int VEC> struct VecT;
template <> struct VecT<1> { typedef double T; };
template <> struct VecT<2> { typedef double2 T; };

template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::T ldv(const double *p);
template <> __device__ __forceinline__ double ldv<1>(const double *p) { return *p; }
template <> __device__ __forceinline__ double2 ldv<2>(const double *p) { return *reinterpret_cast<const double2 *>(p); }

__device__ __forceinline__ void add_row(double &a, double v) { a = a + v; }
__device__ __forceinline__ void add_row(double2 &a, double2 v) { a.x = a.x + v.x; a.y = a.y + v.y; }
__device__ __forceinline__ void acc_row(double &a, double v, double m) { a = a + v * m; }
__device__ __forceinline__ void acc_row(double2 &a, double2 v, double m) { a.x = a.x + v.x * m; a.y = a.y + v.y * m; }

/* LDS-only barrier: orders this wavefront's LDS traffic, leaves its global loads in flight
 * (__syncthreads() carries a workgroup release fence that drains vmcnt). */
__device__ __forceinline__ void nla_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#define NLA_ADV_RCAP 8192               /* picks staged in LDS per segment (32 KB) */


#endif
