/* isres_scan_fast.h — the lane walk of the ISRES evolve scan (isres_evolve2.hip, ev2_scan_kernel: isres.c:236-251,266-277) with the
 * fp64 exp taken OFF the lane's serial chain (ev2_scan_fast_kernel, launch flag NLA_EVOLVE_FAST_SCAN, "amd_isres_fast_scan").
 *
 * What the scan computes is a COUNT: how many deviates an individual consumes from candidate start d, i.e. how often a draw
 * x_a + sigma'_a z leaves [lb_a, ub_a] and is redrawn.  The exact kernel evaluates sigma'_a = sigma_a exp(taup z_k + tau z) for each of the
 * 256 lanes x na coordinates of a workgroup — 65 k exps of ~45 instructions each on a lone wavefront per SIMD, the bulk of the kernel's
 * 70 us — although z comes out of a window of only ~4 na staged deviates and z_k is one value per lane.  Here the workgroup computes
 * exp(tau z) ONCE per staged deviate (≈ 4 per thread instead of 256), the lane exp(taup z_k) once, and the walk uses the product
 *       s^ = (sigma_a e_k) e_z        (two multiplications; equal to sigma' to ~1e-15 relative: three exps of <= 1 ulp, two roundings,
 *                                      the rounding of the exact argument's sum)
 * for a DECISION WITH A MARGIN: with m = min(x^ - lb, ub - x^) and tol_a = 1e-12 (|lb| + |ub| + |x_a|),
 *       m >  tol_a   the draw is inside the box        m < -tol_a   it is outside        otherwise   not decided here.
 * Why the margin is safe: a draw within tol of a bound has |s z| <= |ub - lb| + |x_a| + tol, so its x^ differs from the exact x by
 * <= 1e-14 (|lb| + |ub| + |x_a|) — a hundredth of tol; a draw with a larger |s z| lies farther from both bounds than its own error.
 * An undecided draw (never seen in practice: a band of relative width 1e-12 around each bound; also every NaN) falls through to the
 * EXACT expressions of the exact kernel for the rest of its coordinate, so the count is the exact kernel's in every case, whatever the
 * device's exp returns.  The write kernel recomputes the resolved individuals with the exact expressions as before; nothing but
 * the scan changes, and E / T are bit-identical to the exact scan's.
 *
 * Compiled twice: by hipcc into ev2_scan_fast_kernel and by g++ into tools/scan_fast_check.cpp (the fast walk against the exact walk on
 * drawn individuals, with the fast path's exps perturbed by a few ulp and bounds planted a few ulp from the draws). */
#ifndef NLA_ISRES_SCAN_FAST_H
#define NLA_ISRES_SCAN_FAST_H

#ifndef SF_PRIMITIVES_DEFINED
#define SF_DEV __device__ __forceinline__
#define SF_EXP_FAST(x) exp(x)            /* the staged factors (any function within a few ulp of exp) */
#define SF_EXP_EXACT(x) exp(x)           /* the undecided band: the exact kernel's own expression */
/* all of a coordinate's LDS reads issued here (the compiler would sink some below the first use) */
#define SF_ISSUE8(a, b, c, d, e, f, g, h) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h))
#endif

#define SF_TOL_REL 1e-12

SF_DEV double sf_tol(double l, double h, double xa) { return SF_TOL_REL * (__builtin_fabs(l) + __builtin_fabs(h) + __builtin_fabs(xa)); }
SF_DEV double sf_stage_e(double tau, double z) { return SF_EXP_FAST(tau * z); }

/* One lane = candidate start d of one individual.  zw: the staged deviates (zw[q] = z[base + q], q < zwlen), ezw[q] = sf_stage_e(tau, zw[q]);
 * xi / sg / lo / hi / smax / tol: the individual's na mutated coordinates; Tcol = T + d with stride tstride (redraw count in front of every
 * coordinate chunk); `undecided` counts the draws that took the exact path (statistics only).  Returns the E entry: deviates consumed from
 * start d, -1 window exceeded, -2 deviates ran out.  Control flow, T and the result are ev2_scan_kernel's, statement for statement. */
SF_DEV int ev2_walk_fast(int na, int d, int zwlen, bool zw_cut, bool before_stream, double taup, double tau, const double *zw,
                         const double *ezw, const double *xi, const double *sg, const double *lo, const double *hi, const double *smax,
                         const double *tol, short *Tcol, long tstride, unsigned *undecided)
{
    const int chunk = (na + 63) >> 6;
    int res = 0;
    if (before_stream || d >= zwlen) res = zw_cut && !before_stream ? -2 : -1;
    const double taup_rand = res == 0 ? taup * zw[d] : 0.0;
    const double eg = SF_EXP_FAST(taup_rand);
    int cur = d + 1, red = 0, cnext = 0, c = 0;
    for (int a = 0; a < na; ++a) {
        if (a == cnext) { if (c < 64) Tcol[(long) c * tstride] = (short) red; ++c; cnext += chunk; }
        if (res != 0) continue;
        if (cur + 1 >= zwlen) { res = zw_cut ? -2 : -1; continue; }
        double ea = ezw[cur], z1 = zw[cur + 1], sa = sg[a], sm_ = smax[a], xa = xi[a], l = lo[a], h = hi[a], tl = tol[a];
        SF_ISSUE8(ea, z1, sa, sm_, xa, l, h, tl);
        double s2 = (sa * eg) * ea;
        if (s2 > sm_) s2 = sm_;
        int t = 1;
        double zz = z1;
        for (;;) {
            const double xn = xa + s2 * zz;
            const double m = __builtin_fmin(xn - l, h - xn);
            if (m > tl) break;                                   /* inside, whatever the last digits of sigma' are */
            if (!(m < -tl)) {
                /* undecided: the exact kernel's expressions for the rest of this coordinate (the draws before this one were outside) */
                double sx = sa * SF_EXP_EXACT(taup_rand + tau * zw[cur]);
                if (sx > sm_) sx = sm_;
                if (undecided) ++*undecided;
                double xe = xa + sx * zz;
                while (xe < l || xe > h) {
                    ++t;
                    if (cur + t >= zwlen) { res = zw_cut ? -2 : -1; break; }
                    xe = xa + sx * zw[cur + t];
                }
                break;
            }
            ++t;                                                 /* outside: redraw (isres.c:245-248) */
            if (cur + t >= zwlen) { res = zw_cut ? -2 : -1; break; }
            zz = zw[cur + t];
        }
        cur += 1 + t; red += t - 1;
    }
    return res != 0 ? res : 1 + 2 * na + red;
}

#endif
