/* zoo_extra.hip — OUT-OF-TREE device objectives, written against include/nlopt_amd_device.h only (the way a user of the
 * library would): the two n-general functions of the reference's zoo that are not compiled into libnlopt_amd —
 * convexcosh (test/testfuncs.c:288-299) and Shubert (:323-339) — plus a re-statement of Rastrigin, which the tests compare
 * with the compiled-in one.  Build:
 *     hipcc --offload-arch=gfx950 -O3 -ffp-contract=off --genco -I include tests/userobj/zoo_extra.hip -o tests/userobj/zoo_extra.hsaco
 * Bind:  nlopt_amd_set_min_device_objective(opt, "tests/userobj/zoo_extra.hsaco", "convexcosh", host_twin_or_NULL, data) */
#include <nlopt_amd_device.h>

/* f = prod_i cosh((x_i - i)(i + 1)),  df/dx_i = f tanh((x_i - i)(i + 1)) (i + 1) */
struct ConvexCosh {
    static constexpr bool B_IS_PRODUCT = true;
    __device__ static void terms(int n, int i, const double *x, double *a, double *b) { *a = 0; *b = cosh((x[i] - i) * (i + 1)); }
    __device__ static double finish(int n, double A, double B, const double *x) { return B; }
    __device__ static double grad(int n, int i, const double *x, double A, double B) { return B * tanh((x[i] - i) * (i + 1)) * (i + 1); }
};
NLOPT_AMD_DEVICE_OBJECTIVE(convexcosh, ConvexCosh)

/* f = - sum_j sum_i j sin((j+1) x_i + j), j = 1..5 */
struct Shubert {
    static constexpr bool B_IS_PRODUCT = false;
    __device__ static void terms(int n, int i, const double *x, double *a, double *b)
    {
        double s = 0;
        for (int j = 1; j <= 5; ++j) s -= j * sin((j + 1) * x[i] + j);
        *a = s; *b = 0;
    }
    __device__ static double finish(int n, double A, double B, const double *x) { return A; }
    __device__ static double grad(int n, int i, const double *x, double A, double B)
    {
        double g = 0;
        for (int j = 1; j <= 5; ++j) g -= j * (j + 1) * cos((j + 1) * x[i] + j);
        return g;
    }
};
NLOPT_AMD_DEVICE_OBJECTIVE(shubert, Shubert)

/* f = 10 n + sum (x_i^2 - 10 cos(2 pi x_i)) */
struct MyRastrigin {
    static constexpr bool B_IS_PRODUCT = false;
    __device__ static void terms(int n, int i, const double *x, double *a, double *b) { *a = x[i] * x[i] - 10.0 * cos(6.283185307179586 * x[i]); *b = 0; }
    __device__ static double finish(int n, double A, double B, const double *x) { return 10.0 * n + A; }
    __device__ static double grad(int n, int i, const double *x, double A, double B) { return 2 * x[i] + 10.0 * 6.283185307179586 * sin(6.283185307179586 * x[i]); }
};
NLOPT_AMD_DEVICE_OBJECTIVE(myrastrigin, MyRastrigin)
