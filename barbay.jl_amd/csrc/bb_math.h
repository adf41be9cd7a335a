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

// BB_FMAK(a, b, k) = a * b + k for a compile-time constant k.  On the device the constant is moved into a FIXED scalar register
// pair right in front of the v_fma_f64 that reads it (one asm block, s[92:93] clobbered).  Left to hipcc a Horner step becomes
// the two-address v_fmac_f64 whose accumulator is a VGPR pair initialised with the constant (two VALU moves per coefficient),
// and the ~60 coefficients of a sampling pass are either hoisted out of the resident launch's step loop (dozens of VGPRs alive
// for the whole launch) or, as register-allocated scalars, push ~60 live scalars of the kernel into VGPR lanes and back
// (v_writelane / v_readlane) every step.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(BB_NO_ASM_FMA)
#define BB_FMAK(a, b, k)                                                                                                   \
    ({                                                                                                                     \
        double bb_r_;                                                                                                      \
        asm("s_mov_b32 s92, %3\n\ts_mov_b32 s93, %4\n\tv_fma_f64 %0, %1, %2, s[92:93]"                                  \
            : "=v"(bb_r_)                                                                                                  \
            : "v"((double)(a)), "v"((double)(b)), "i"((int)(__builtin_bit_cast(unsigned long long, (double)(k)) & 0xffffffffull)), \
              "i"((int)(__builtin_bit_cast(unsigned long long, (double)(k)) >> 32))                                        \
            : "s92", "s93");                                                                                             \
        bb_r_;                                                                                                             \
    })
#else
#define BB_FMAK(a, b, k) fma((double)(a), (double)(b), (double)(k))
#endif

// 1/x for finite, non-zero, normal x: seed + 2 Newton steps.
BB_DEV double bb_rcp(double x) {
    double r = BB_RCP_SEED(x);
    r = fma(fma(-x, r, 1.0), r, r);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}

// a / b with one residual correction (<= 1 ulp).
BB_DEV double bb_div(double a, double b) {
    const double r = bb_rcp(b);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}

// sqrt(x), x >= 0 (0 -> 0): rsq seed, coupled Newton (Goldschmidt) + final residual correction.
BB_DEV double bb_sqrt(double x) {
    if (!(x > 0.0)) return x == 0.0 ? 0.0 : sqrt(x);
    const double y = BB_RSQ_SEED(x);
    double g = x * y, h = 0.5 * y;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    return fma(fma(-g, g, x), h, g);
}

// exp(x): k = rint(x log2 e), r = x - k ln2 (two-piece), degree-13 Taylor on |r| <= 0.347, ldexp.
BB_DEV double bb_exp(double x) {
    x = fmin(fmax(x, -746.0), 710.0);
    const double k = rint(x * 1.4426950408889634074);
    double r = fma(-k, 6.93147180369123816490e-01, x);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = BB_FMAK(p, r, 1.0 / 479001600.0);
    p = BB_FMAK(p, r, 1.0 / 39916800.0);
    p = BB_FMAK(p, r, 1.0 / 3628800.0);
    p = BB_FMAK(p, r, 1.0 / 362880.0);
    p = BB_FMAK(p, r, 1.0 / 40320.0);
    p = BB_FMAK(p, r, 1.0 / 5040.0);
    p = BB_FMAK(p, r, 1.0 / 720.0);
    p = BB_FMAK(p, r, 1.0 / 120.0);
    p = BB_FMAK(p, r, 1.0 / 24.0);
    p = BB_FMAK(p, r, 1.0 / 6.0);
    p = BB_FMAK(p, r, 0.5);
    p = BB_FMAK(p, r, 1.0);
    p = BB_FMAK(p, r, 1.0);
    return ldexp(p, (int)k);
}

// log(x), x > 0 finite: x = m 2^e with m in [sqrt(1/2), sqrt(2)); log m = 2 atanh(s), s = (m-1)/(m+1).
BB_DEV double bb_log(double x) {
    int e;
    double m = frexp(x, &e);
    if (m < 0.70710678118654752440) { m *= 2.0; e -= 1; }
    const double s = bb_div(m - 1.0, m + 1.0);
    const double z = s * s;
    double p = 1.0 / 25.0;
    p = BB_FMAK(p, z, 1.0 / 23.0);
    p = BB_FMAK(p, z, 1.0 / 21.0);
    p = BB_FMAK(p, z, 1.0 / 19.0);
    p = BB_FMAK(p, z, 1.0 / 17.0);
    p = BB_FMAK(p, z, 1.0 / 15.0);
    p = BB_FMAK(p, z, 1.0 / 13.0);
    p = BB_FMAK(p, z, 1.0 / 11.0);
    p = BB_FMAK(p, z, 1.0 / 9.0);
    p = BB_FMAK(p, z, 1.0 / 7.0);
    p = BB_FMAK(p, z, 1.0 / 5.0);
    p = BB_FMAK(p, z, 1.0 / 3.0);
    const double lm = fma(2.0 * s * z, p, 2.0 * s);     // 2s + 2s z P(z)
    const double ef = (double)e;
    return fma(ef, 6.93147180369123816490e-01, fma(ef, 1.90821492927058770002e-10, lm));
}

// softplus / sigmoid of omega sharing one exp, one reciprocal and one log:
//   e = exp(-|w|) in (0, 1], u = 1 + e, log1p(e) = log(u) + (e - (u - 1)) / u  (exact-sum correction)
BB_DEV void bb_softplus_sigmoid_fast(double om, double* sp, double* sig) {
    const double e = bb_exp(-fabs(om));
    const double u = 1.0 + e;
    const double inv = bb_rcp(u);
    const double l1p = e < 0x1.0p-54 ? e : fma(e - (u - 1.0), inv, bb_log(u));
    *sp = fmax(om, 0.0) + l1p;
    *sig = om >= 0.0 ? inv : e * inv;
}

// sin(pi x), cos(pi x) for x in [0, 2): quadrant n = rint(2x), r = x - n/2 in [-1/4, 1/4], Taylor in y = pi r.
BB_DEV void bb_sincospi_02(double x, double* s, double* c) {
    const double n = rint(2.0 * x);
    const double y = fma(-0.5, n, x) * 3.14159265358979323846;
    const double z = y * y;
    double ps = -1.0 / 1307674368000.0;            // -1/15!
    ps = BB_FMAK(ps, z, 1.0 / 6227020800.0);           //  1/13!
    ps = BB_FMAK(ps, z, -1.0 / 39916800.0);
    ps = BB_FMAK(ps, z, 1.0 / 362880.0);
    ps = BB_FMAK(ps, z, -1.0 / 5040.0);
    ps = BB_FMAK(ps, z, 1.0 / 120.0);
    ps = BB_FMAK(ps, z, -1.0 / 6.0);
    const double sy = fma(y * z, ps, y);
    double pc = 1.0 / 20922789888000.0;            //  1/16!
    pc = BB_FMAK(pc, z, -1.0 / 87178291200.0);         // -1/14!
    pc = BB_FMAK(pc, z, 1.0 / 479001600.0);
    pc = BB_FMAK(pc, z, -1.0 / 3628800.0);
    pc = BB_FMAK(pc, z, 1.0 / 40320.0);
    pc = BB_FMAK(pc, z, -1.0 / 720.0);
    pc = BB_FMAK(pc, z, 1.0 / 24.0);
    pc = BB_FMAK(pc, z, -0.5);
    const double cy = fma(z, pc, 1.0);
    const int q = (int)n & 3;
    const double ss = (q & 1) ? cy : sy, cc = (q & 1) ? sy : cy;
    *s = (q & 2) ? -ss : ss;
    *c = (q == 1 || q == 2) ? -cc : cc;
}
