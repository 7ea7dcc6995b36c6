/* nlopt_amd_device.h — how a user puts an objective on the GPU for libnlopt_amd (HIP, gfx950).
 *
 * The reference's objective is a host callback (src/api/nlopt.h:60-62), which a GPU cannot call; SURVEY.md §8b lists "an
 * additive setter" for device objectives as the required extension.  This header is the device half of that extension:
 * the user describes the objective by its per-coordinate terms, the macro below turns the description into the one kernel
 * libnlopt_amd launches, the user compiles it into a code object
 *
 *     hipcc --offload-arch=gfx950 -O3 -ffp-contract=off --genco -I<libnlopt_amd>/include my_objective.hip -o my_objective.hsaco
 *
 * and binds it with nlopt_amd_set_min_device_objective(opt, "my_objective.hsaco", "myobj", host_twin, data)
 * (include/nlopt_amd.h).  No other file of the library is needed to compile it.
 *
 * Contract (the same as the compiled-in objectives, nlopt_amd/csrc/hip/dev_common.h): ONE WAVEFRONT (64 lanes) evaluates
 * one candidate.  An objective is described as
 *
 *     f(x) = finish(n, A, B, x)   with   A = sum_i a_i(x),  B = sum_i b_i(x)     (B a product if B_IS_PRODUCT)
 *     df/dx_i = grad(n, i, x, A, B)
 *
 *   struct MyObj {
 *       static constexpr bool B_IS_PRODUCT = false;
 *       __device__ static void terms(int n, int i, const double *x, double *a, double *b);    // a_i, b_i  (i < n)
 *       __device__ static double finish(int n, double A, double B, const double *x);
 *       __device__ static double grad(int n, int i, const double *x, double A, double B);
 *   };
 *   NLOPT_AMD_DEVICE_OBJECTIVE(myobj, MyObj)
 *
 * Lane l accumulates the terms of coordinates l, l+64, ... in that order; the 64 partial pairs are combined by a
 * xor-butterfly.  Only the association order of that reduction differs from a sequential host loop over the same terms, so
 * f agrees with a host twin written from the same formulas to rounding (1e-10 relative is the library's parity bar).
 * Compile with -ffp-contract=off if the host twin is.
 *
 * Kernel ABI (what the library launches; <name>_evalgrad, 64 x 4 threads per workgroup, one wavefront per candidate):
 *   (int n, int ld, long count, const int *list, const double *X, double *F, double *G, double sign)
 *   list == NULL: candidates are rows 0 .. count-1 of X (ld doubles apart): F[c] = sign f(row c); if G: row c of G =
 *   sign grad.  list != NULL: candidate c is row i with e = list[c], i = e >= 0 ? e : -(e+1); its gradient is wanted iff
 *   e >= 0; results go to F[i] and row i of G.
 */
#ifndef NLOPT_AMD_DEVICE_H
#define NLOPT_AMD_DEVICE_H

#include <hip/hip_runtime.h>

#define NLOPT_AMD_DEVICE_ABI 1

namespace nlopt_amd_device {

template <class OBJ>
__device__ __forceinline__ void evalgrad_wave(int n, const double *x, double *f_out, double *g, double sign)
{
    const int lane = threadIdx.x & 63;
    double A = 0, B = OBJ::B_IS_PRODUCT ? 1 : 0;
    for (int i = lane; i < n; i += 64) {
        double a, b;
        OBJ::terms(n, i, x, &a, &b);
        A += a;
        if (OBJ::B_IS_PRODUCT) B *= b; else B += b;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const double oa = __shfl_xor(A, m, 64), ob = __shfl_xor(B, m, 64);
        A += oa;
        if (OBJ::B_IS_PRODUCT) B *= ob; else B += ob;
    }
    const double f = OBJ::finish(n, A, B, x);
    if (lane == 0) *f_out = sign * f;
    if (g) for (int i = lane; i < n; i += 64) g[i] = sign * OBJ::grad(n, i, x, A, B);
}

template <class OBJ>
__device__ __forceinline__ void evalgrad_kernel_body(int n, int ld, long count, const int *list, const double *X, double *F, double *G,
                                                     double sign)
{
    const long c = (long) blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (c >= count) return;
    long row = c;
    bool wantg = G != nullptr;
    if (list) { const int e = list[c]; row = e >= 0 ? e : -(long) e - 1; wantg = wantg && e >= 0; }
    evalgrad_wave<OBJ>(n, X + row * (long) ld, F + row, wantg ? G + row * (long) ld : nullptr, sign);
}

}  // namespace nlopt_amd_device

#define NLOPT_AMD_DEVICE_OBJECTIVE(name, OBJ)                                                                                  \
    extern "C" __global__ __launch_bounds__(256) void name##_evalgrad(int n, int ld, long count, const int *list,              \
                                                                      const double *X, double *F, double *G, double sign)     \
    { nlopt_amd_device::evalgrad_kernel_body<OBJ>(n, ld, count, list, X, F, G, sign); }                                       \
    extern "C" __global__ void name##_abi(int *out) { *out = NLOPT_AMD_DEVICE_ABI; }

#endif /* NLOPT_AMD_DEVICE_H */
