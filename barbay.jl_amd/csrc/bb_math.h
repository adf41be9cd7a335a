// bb_math.h -- fp64 elementary functions for the ADVI kernels, written for the argument ranges the
// step actually produces and ~1-2 ulp accuracy (parity tolerances are 1e-9 relative on gradients).
//
// Why not the ocml versions: the sampling and update sweeps are VALU-bound on fp64 transcendentals
// (profiles/r01*: 59 % of k_sample in "draw"); the library forms carry full special-case handling
// and IEEE-exact divide / sqrt expansions (div_scale / div_fmas / div_fixup).  These use the hardware
// seeds v_rcp_f64 / v_rsq_f64 with Newton steps and short fma polynomials.  Compiles for the host
// too (emulation build, accuracy tests in tests/test_bb_math.py).
#pragma once
#include <math.h>

#ifndef BB_DEV
#define BB_DEV static inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define BB_RCP_SEED(x) __builtin_amdgcn_rcp(x)
#define BB_RSQ_SEED(x) __builtin_amdgcn_rsq(x)
#else
#define BB_RCP_SEED(x) (1.0 / (x))
#define BB_RSQ_SEED(x) (1.0 / sqrt(x))
#endif

// Polynomial coefficients live in constant memory and reach the FMAs through SCALAR registers: a handful of wide s_load
// instructions per function call (the compiler merges the adjacent table reads) instead of two moves per coefficient.  A wave
// issues one instruction -- of any kind -- every ~4-5 cycles (tools/probe/fp64_rate.hip: a dependent v_fma_f64 chain runs at
// 8 cycles per step, the same step behind two s_mov_b32 at 19.5), so every move in front of an FMA costs as much as the FMA.
// The table pointer is laundered once per call: the loads then cannot be hoisted out of the resident launch's step loop, where
// ~45 coefficient pairs would stay alive in (spilled) registers for the whole launch.
#if defined(__HIP_DEVICE_COMPILE__)
#define BB_TABLE __device__ __constant__ static const double
typedef __attribute__((address_space(4))) const double bb_cdouble;     // constant address space: uniform reads become s_load
BB_DEV bb_cdouble* bb_tab(const double* t) { bb_cdouble* p = (bb_cdouble*)t; asm volatile("" : "+s"(p)); return p; }
#else
#define BB_TABLE static const double
typedef const double bb_cdouble;
BB_DEV bb_cdouble* bb_tab(const double* t) { return t; }
#endif
// Horner chains p = p * x + c[i] as ONE block of three-address v_fma_f64 with the coefficients in scalar register pairs.
// Left alone hipcc prefers the two-address v_fmac_f64 and first copies every scalar coefficient into a VGPR accumulator (two
// v_mov_b32 per step); single-instruction asm statements get an s_nop each from the hazard recogniser (it cannot see that
// the unknown instruction is a plain VALU one) -- either way an issue slot or two per step, as costly as the FMA itself.
BB_DEV double bb_horner6(double p, double x, bb_cdouble* c) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_f64 %0, %0, %1, %2\n\t" "v_fma_f64 %0, %0, %1, %3\n\t" "v_fma_f64 %0, %0, %1, %4\n\t" "v_fma_f64 %0, %0, %1, %5\n\t" "v_fma_f64 %0, %0, %1, %6\n\t" "v_fma_f64 %0, %0, %1, %7\n\t" : "+v"(p) : "v"(x), "s"(c[0]), "s"(c[1]), "s"(c[2]), "s"(c[3]), "s"(c[4]), "s"(c[5]));
    return p;
#else
    for (int i = 0; i < 6; ++i) p = fma(p, x, c[i]);
    return p;
#endif
}
BB_DEV double bb_horner11(double p, double x, bb_cdouble* c) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_fma_f64 %0, %0, %1, %2\n\t" "v_fma_f64 %0, %0, %1, %3\n\t" "v_fma_f64 %0, %0, %1, %4\n\t" "v_fma_f64 %0, %0, %1, %5\n\t" "v_fma_f64 %0, %0, %1, %6\n\t" "v_fma_f64 %0, %0, %1, %7\n\t" "v_fma_f64 %0, %0, %1, %8\n\t" "v_fma_f64 %0, %0, %1, %9\n\t" "v_fma_f64 %0, %0, %1, %10\n\t" "v_fma_f64 %0, %0, %1, %11\n\t" "v_fma_f64 %0, %0, %1, %12\n\t" : "+v"(p) : "v"(x), "s"(c[0]), "s"(c[1]), "s"(c[2]), "s"(c[3]), "s"(c[4]), "s"(c[5]), "s"(c[6]), "s"(c[7]), "s"(c[8]), "s"(c[9]), "s"(c[10]));
    return p;
#else
    for (int i = 0; i < 11; ++i) p = fma(p, x, c[i]);
    return p;
#endif
}

BB_TABLE bb_c_exp[16] = {1.0 / 6227020800.0, 1.0 / 479001600.0, 1.0 / 39916800.0, 1.0 / 3628800.0, 1.0 / 362880.0, 1.0 / 40320.0,
                         1.0 / 5040.0, 1.0 / 720.0, 1.0 / 120.0, 1.0 / 24.0, 1.0 / 6.0, 0.5,
                         1.4426950408889634074, 6.93147180369123816490e-01, 1.90821492927058770002e-10, 0.0};
BB_TABLE bb_c_log[16] = {1.0 / 25.0, 1.0 / 23.0, 1.0 / 21.0, 1.0 / 19.0, 1.0 / 17.0, 1.0 / 15.0, 1.0 / 13.0, 1.0 / 11.0, 1.0 / 9.0,
                         1.0 / 7.0, 1.0 / 5.0, 1.0 / 3.0, 6.93147180369123816490e-01, 1.90821492927058770002e-10, 0.70710678118654752440, 0.0};
BB_TABLE bb_c_sc[16] = {-1.0 / 1307674368000.0, 1.0 / 6227020800.0, -1.0 / 39916800.0, 1.0 / 362880.0, -1.0 / 5040.0, 1.0 / 120.0, -1.0 / 6.0,
                        1.0 / 20922789888000.0, -1.0 / 87178291200.0, 1.0 / 479001600.0, -1.0 / 3628800.0, 1.0 / 40320.0, -1.0 / 720.0,
                        1.0 / 24.0, 3.14159265358979323846, 0.0};

// 1/x for finite, non-zero, normal x: seed + 2 Newton steps.
BB_DEV double bb_rcp(double x) {
    double r = BB_RCP_SEED(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// a / b with one residual correction (<= 1 ulp).  The reciprocal behind it needs one Newton step only: the correction squares
// its error (gfx950: seed 2^-24.4, one step 2.2e-15 -- tools/probe/seed_accuracy.hip).
BB_DEV double bb_div(double a, double b) {
    double r = BB_RCP_SEED(b);
    r = fma(fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}

// sqrt(x), x >= 0 (0 -> 0): rsq seed, ONE coupled Newton (Goldschmidt) iteration + final residual correction -- measured on
// gfx950 (tools/probe/seed_accuracy.hip): the seed is good to 2^-24.2, the iteration squares that and the correction leaves 1.1e-16,
// the same as with a second iteration.
BB_DEV double bb_sqrt(double x) {
    if (!(x > 0.0)) return x == 0.0 ? 0.0 : sqrt(x);
    const double y = BB_RSQ_SEED(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    return fma(fma(-g, g, x), h, g);
}

// exp(x): k = rint(x log2 e), r = x - k ln2 (two-piece), degree-13 Taylor on |r| <= 0.347, ldexp.
BB_DEV double bb_exp(double x) {
    bb_cdouble* c = bb_tab(bb_c_exp);
    x = fmin(fmax(x, -746.0), 710.0);
    const double k = rint(x * c[12]);
    double r = fma(-k, c[13], x);
    r = fma(-k, c[14], r);
    double p = bb_horner11(c[0], r, c + 1);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}

// log(x), x > 0 finite: x = m 2^e with m in [sqrt(1/2), sqrt(2)); log m = 2 atanh(s), s = (m-1)/(m+1).
BB_DEV double bb_log(double x) {
    bb_cdouble* c = bb_tab(bb_c_log);
    int e;
    double m = frexp(x, &e);
    if (m < c[14]) { m *= 2.0; e -= 1; }
    const double s = bb_div(m - 1.0, m + 1.0);
    const double z = s * s;
    double p = bb_horner11(c[0], z, c + 1);
    const double lm = fma(2.0 * s * z, p, 2.0 * s);     // 2s + 2s z P(z)
    const double ef = (double)e;
    return fma(ef, c[12], fma(ef, c[13], lm));
}

// softplus / sigmoid of omega sharing one exp, one reciprocal and one log:
//   e = exp(-|w|) in (0, 1], u = 1 + e, log1p(e) = log(u) + (e - (u - 1)) / u  (exact-sum correction)
// (exp of a non-positive argument: no upper clamp; log of u in (1, 2]: the exponent is 0 or 1 -- the same bits as bb_exp / bb_log)
BB_DEV double bb_exp_nonpos(double x) {
    bb_cdouble* c = bb_tab(bb_c_exp);
    x = fmax(x, -746.0);
    const double k = rint(x * c[12]);
    double r = fma(-k, c[13], x);
    r = fma(-k, c[14], r);
    double p = bb_horner11(c[0], r, c + 1);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}
BB_DEV double bb_log_1to2(double u) {
    bb_cdouble* c = bb_tab(bb_c_log);
    const double hu = 0.5 * u;
    const bool up = !(hu < c[14]);                      // u / 2 in [sqrt(1/2), 1]: m = u / 2, exponent 1; else m = u, exponent 0
    const double m = up ? hu : u;
    const double s = bb_div(m - 1.0, m + 1.0);
    const double z = s * s;
    double p = bb_horner11(c[0], z, c + 1);
    const double lm = fma(2.0 * s * z, p, 2.0 * s);
    const double ef = up ? 1.0 : 0.0;
    return fma(ef, c[12], fma(ef, c[13], lm));
}
BB_DEV void bb_softplus_sigmoid_fast(double om, double* sp, double* sig) {
    const double e = bb_exp_nonpos(-fabs(om));
    const double u = 1.0 + e;
    const double inv = bb_rcp(u);
    const double l1p = e < 0x1.0p-54 ? e : fma(e - (u - 1.0), inv, bb_log_1to2(u));
    *sp = fmax(om, 0.0) + l1p;
    *sig = om >= 0.0 ? inv : e * inv;
}

// sin(pi x), cos(pi x) for x in [0, 2): quadrant n = rint(2x), r = x - n/2 in [-1/4, 1/4], Taylor in y = pi r.
BB_DEV void bb_sincospi_02(double x, double* s, double* c) {
    bb_cdouble* t = bb_tab(bb_c_sc);
    const double n = rint(2.0 * x);
    const double y = fma(-0.5, n, x) * t[14];
    const double z = y * y;
    const double ps = bb_horner6(t[0], z, t + 1);      // -1/15!, 1/13!, ...
    const double sy = fma(y * z, ps, y);
    double pc = bb_horner6(t[7], z, t + 8);             //  1/16!, -1/14!, ...
    pc = fma(pc, z, -0.5);
    const double cy = fma(z, pc, 1.0);
    const int q = (int)n & 3;
    const double ss = (q & 1) ? cy : sy, cc = (q & 1) ? sy : cy;
    *s = (q & 2) ? -ss : ss;
    *c = (q == 1 || q == 2) ? -cc : cc;
}
