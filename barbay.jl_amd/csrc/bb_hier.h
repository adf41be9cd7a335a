// bb_hier.h -- device-side `process_hierarchical_samples!` (src/utils.jl:1284-1343).
//
// For every unit u of the theta_tilde block (mutant x replicate, or mutant for the genotype model) the reference
// draws n_samples Normal samples of theta, logtau and theta_tilde from the mean-field posterior, forms
//     s = theta[idx(u)] + exp(logtau_u) * theta_tilde_u
// and reports median(s) (under the column name `mean`) and std(s) (corrected, n - 1).  The theta draws
// are shared by all units that use the same theta (the reference's `hcat(repeat([theta_mat], n_rep)...)`).
// One workgroup per unit: draws (Philox4x32-10, keyed by (theta index | unit, sample)) into LDS, two-pass mean /
// variance, bitonic sort in LDS (+inf padding to a power of two), Julia's median (mean of the two middle order
// statistics for even n).  Written as barrier-separated passes like the step programs (host emulation in tests).
#pragma once
#include "bb_block.h"

#define BB_STREAM_HIER_THETA 0xFFFFFFF0u
#define BB_STREAM_HIER_UNIT 0xFFFFFFF1u

struct HierArgs {
    const double* mean;       // [D] posterior mean / sigma (device)
    const double* sigma;
    double* median_out;       // [n_units]
    double* std_out;
    long long n_units;
    long long lo_theta, lo_tt, lo_lt;   // flat offsets of the theta / theta_tilde / logtau blocks
    long long theta_mod;      // unit u uses theta[u % theta_mod]  (0: genotype model, theta[geno_idx[u]])
    const int* geno_idx;
    int n_samples, n_pad;
    unsigned long long seed;
};

BB_DEV void bb_block_hier(BBCtx& cx, const HierArgs& H, int nblocks) {
    double* smp = cx.lds;                  // n_pad doubles
    double* red = cx.lds + H.n_pad;        // nthr doubles
    for (long long u = cx.block; u < H.n_units; u += nblocks) {
        const long long ith = H.theta_mod > 0 ? u % H.theta_mod : H.geno_idx[u];
        const double m_th = H.mean[H.lo_theta + ith], s_th = H.sigma[H.lo_theta + ith];
        const double m_tt = H.mean[H.lo_tt + u], s_tt = H.sigma[H.lo_tt + u];
        const double m_lt = H.mean[H.lo_lt + u], s_lt = H.sigma[H.lo_lt + u];
        // pass 1: draws, partial sums
        BB_PASS(cx, tid) {
            double acc = 0.0;
            for (int j = tid; j < H.n_pad; j += cx.nthr) {
                double v = INFINITY;
                if (j < H.n_samples) {
                    double n_th, dummy, n_lt, n_tt;
                    // theta draw j of theta index ith: pair (j >> 1), branch j & 1 -> shared by every unit using ith
                    bb_normal_pair(H.seed, ((unsigned long long)ith << 20) | (unsigned long long)(j >> 1), (unsigned)(j & 1), BB_STREAM_HIER_THETA, &n_th, &dummy);
                    if (j & 1) n_th = dummy;
                    bb_normal_pair(H.seed, (unsigned long long)u, (unsigned)j, BB_STREAM_HIER_UNIT, &n_lt, &n_tt);
                    v = fma(s_th, n_th, m_th) + bb_exp(fma(s_lt, n_lt, m_lt)) * fma(s_tt, n_tt, m_tt);
                    acc += v;
                }
                smp[j] = v;
            }
            red[tid] = acc;
        }
        BB_SYNC(cx);
        BB_PASS(cx, tid) {
            if (tid == 0) { double s = 0.0; for (int i = 0; i < cx.nthr; ++i) s += red[i]; red[cx.nthr] = s / (double)H.n_samples; }
        }
        BB_SYNC(cx);
        BB_PASS(cx, tid) {
            const double mu = red[cx.nthr];
            double acc = 0.0;
            for (int j = tid; j < H.n_samples; j += cx.nthr) { const double d = smp[j] - mu; acc += d * d; }
            red[tid] = acc;
        }
        BB_SYNC(cx);
        BB_PASS(cx, tid) {
            if (tid == 0) {
                double s = 0.0;
                for (int i = 0; i < cx.nthr; ++i) s += red[i];
                H.std_out[u] = bb_sqrt(s / (double)(H.n_samples - 1));     // StatsBase.std: corrected
            }
        }
        BB_SYNC(cx);
        // bitonic sort of smp[0 .. n_pad)
        for (int k = 2; k <= H.n_pad; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                BB_PASS(cx, tid) {
                    for (int i = tid; i < H.n_pad; i += cx.nthr) {
                        const int l = i ^ j;
                        if (l > i) {
                            const double a = smp[i], b = smp[l];
                            const bool up = (i & k) == 0;
                            if ((a > b) == up) { smp[i] = b; smp[l] = a; }
                        }
                    }
                }
                BB_SYNC(cx);
            }
        }
        BB_PASS(cx, tid) {
            if (tid == 0) {
                const int n = H.n_samples;
                H.median_out[u] = (n & 1) ? smp[n >> 1] : 0.5 * (smp[(n >> 1) - 1] + smp[n >> 1]);   // Statistics.median
            }
        }
        BB_SYNC(cx);
    }
}
