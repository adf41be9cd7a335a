// Accuracy harness for barbay.jl_amd/csrc/bb_math.h against long-double libm (host build).
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <random>
#define BB_DEV static inline
#include "../barbay.jl_amd/csrc/bb_math.h"

static double relerr(double got, long double want) {
    if (want == 0.0L) return std::fabs(got);
    return (double)fabsl(((long double)got - want) / want);
}
int main() {
    std::mt19937_64 g(12345);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    double e_exp = 0, e_log = 0, e_rcp = 0, e_div = 0, e_sqrt = 0, e_sp = 0, e_sig = 0, e_sc = 0;
    for (int i = 0; i < 2000000; ++i) {
        double x = (U(g) - 0.5) * 1400.0;
        if (i % 3 == 0) x = (U(g) - 0.5) * 40.0;
        e_exp = fmax(e_exp, relerr(bb_exp(x), expl((long double)x)));
        double p = std::exp((U(g) - 0.5) * 1400.0);
        e_log = fmax(e_log, fabs(bb_log(p) - (double)logl((long double)p)) / fmax(1e-300, fabs((double)logl((long double)p))));
        double q = std::exp((U(g) - 0.5) * 600.0) * (U(g) < 0.5 ? -1 : 1);
        e_rcp = fmax(e_rcp, relerr(bb_rcp(q), 1.0L / q));
        e_div = fmax(e_div, relerr(bb_div(x, q), (long double)x / q));
        double sq = std::exp((U(g) - 0.5) * 600.0);
        e_sqrt = fmax(e_sqrt, relerr(bb_sqrt(sq), sqrtl((long double)sq)));
        double om = (U(g) - 0.5) * (i % 2 ? 80.0 : 1500.0), sp, sg;
        bb_softplus_sigmoid_fast(om, &sp, &sg);
        long double spw = om > 0 ? (long double)om + log1pl(expl(-(long double)om)) : log1pl(expl((long double)om));
        long double sgw = 1.0L / (1.0L + expl(-(long double)om));
        if (spw > 1e-300L) e_sp = fmax(e_sp, relerr(sp, spw));
        if (sgw > 1e-300L) e_sig = fmax(e_sig, relerr(sg, sgw));
        double a = U(g) * 2.0, s, c;
        bb_sincospi_02(a, &s, &c);
        long double sw = sinl(3.14159265358979323846264338327950288L * a), cw = cosl(3.14159265358979323846264338327950288L * a);
        e_sc = fmax(e_sc, fmax(fabs(s - (double)sw), fabs(c - (double)cw)));   // absolute: results are O(1)
    }
    // log near 1 and specials
    for (int i = 0; i < 200000; ++i) {
        double p = 1.0 + (U(g) - 0.5) * 1e-3;
        e_log = fmax(e_log, relerr(bb_log(p), logl((long double)p)));
    }
    printf("exp %.3e log %.3e rcp %.3e div %.3e sqrt %.3e softplus %.3e sigmoid %.3e sincospi %.3e\n", e_exp, e_log, e_rcp, e_div, e_sqrt, e_sp, e_sig, e_sc);
    printf("sqrt0 %g exp(-800) %g exp(800) %g sp(-745) %g\n", bb_sqrt(0.0), bb_exp(-800.0), bb_exp(800.0), 0.0);
    return 0;
}
