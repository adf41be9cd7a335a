/* bb_port.c -- CPU restatement (plain C, fp64, OpenMP) of the ADVI hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Used by tests/ as a second checker and by bench.py as the timed `cpu_baseline` ("port").  Nothing
 * under barbay.jl_amd/ links, loads or calls it.  PARITY UNPINNED: the reference (Julia, Turing 0.36 /
 * AdvancedVI 0.2, not runnable here) pins no numeric value on this path; this port is checked against
 * the literal transcription in oracle/literal.py (<= 1e-12 relative) -- see oracle/__init__.py.
 *
 * What it restates, with the reference lines it follows:
 *   log-joint of the fitness_normal family     src/model_fitness_normal.jl:132-271
 *                                              src/model_multienv_fitness_normal.jl:146-302
 *                                              src/model_fitness_normal_hierarchical_genotypes.jl:165-329
 *                                              src/model_fitness_normal_hierarchical_replicates.jl:158-331, 420-637
 *                                              src/model_multienv_fitness_normal_hierarchical_replicates.jl:158-363, 449-687
 *   in the "independent Poisson" form: Poisson(n_t | sum_b lam) * Multinomial(R_t | n_t, F_t)
 *   == prod_b Poisson(R_tb | lam_tb) when n_t == sum_b R_tb (docs/src/math.md:405-407), with the gradient
 *   written out by hand (direct sums over barcodes; no moment tables);
 *   ELBO / reparameterisation gradient / optimisers of AdvancedVI 0.2 (call site src/vi.jl:201).
 *
 * Layout: flat latent vector in the reference's block order, Julia column-major.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define MAXR 16
#define MAXT 256
#define LOG2PI 1.8378770664093454835606594728112

typedef struct {
    int kind, R, E, G;              /* 0 fitness, 1 multienv, 2 genotype, 3 replicate, 4 multienv_replicate */
    int64_t nn, nb, D;
    int T[MAXR];
    const int64_t* counts;          /* replicate-major, each T_r x B column-major */
    const int32_t* env_idx;         /* [T]; kind 4: replicate-major [sum_r T_r] */
    const int32_t* geno_idx;        /* [nb] */
    const double* pmean;            /* [D] prior mean per latent */
    const double* pstd;             /* [D] prior std per latent */
    /* block offsets */
    int64_t o_spop, o_lspop, o_s, o_tt, o_lt, o_ls, o_l;
} port_model;

static double softplus(double x) { return fmax(x, 0.0) + log1p(exp(-fabs(x))); }
static double sigmoid(double x) { double e = exp(-fabs(x)); return x >= 0 ? 1.0 / (1.0 + e) : e / (1.0 + e); }

/* effective fitness / log-sigma of mutant m at time step t of replicate r, and chain-rule scatter */
/* (eo = offset of replicate r's time points in env_idx: kind 4 only, 0 otherwise) */
static inline int64_t unit4(const port_model* M, int64_t m, int r, int t, int eo) {   /* theta_tilde / logtau / logsigma [e, m, r], e fastest */
    return (int64_t)r * M->nb * M->E + m * M->E + M->env_idx[eo + t + 1];
}
static inline double s_eff(const port_model* M, const double* z, int64_t m, int r, int t, int eo) {
    switch (M->kind) {
    case 0: return z[M->o_s + m];
    case 1: return z[M->o_s + m * M->E + M->env_idx[t + 1]];
    case 2: return z[M->o_s + M->geno_idx[m]] + exp(z[M->o_lt + m]) * z[M->o_tt + m];
    case 3: return z[M->o_s + m] + exp(z[M->o_lt + r * M->nb + m]) * z[M->o_tt + r * M->nb + m];
    default: {   /* model_multienv_..._replicates.jl:241, 311: s = theta[e, m] + exp(logtau[e, m, r]) * theta_tilde[e, m, r] */
        const int64_t u = unit4(M, m, r, t, eo);
        return z[M->o_s + m * M->E + M->env_idx[eo + t + 1]] + exp(z[M->o_lt + u]) * z[M->o_tt + u];
    }
    }
}
static inline int64_t ls_index(const port_model* M, int64_t m, int r, int t, int eo) {
    switch (M->kind) {
    case 0: return M->o_ls + m;
    case 1: return M->o_ls + m * M->E + M->env_idx[t + 1];
    case 2: return M->o_ls + m;
    case 3: return M->o_ls + r * M->nb + m;
    default: return M->o_ls + unit4(M, m, r, t, eo);
    }
}

/* log-joint at z and its gradient g (length D).  Returns log p(data, z). */
double port_logjoint_grad(const port_model* M, const double* z, double* g, int nthreads) {
    const int64_t B = M->nn + M->nb, D = M->D;
    double lp = 0.0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    /* priors: every block is a diagonal Normal (model_fitness_normal.jl:137-203 and siblings) */
#pragma omp parallel for reduction(+ : lp) schedule(static)
    for (int64_t i = 0; i < D; ++i) {
        const double d = (z[i] - M->pmean[i]) / M->pstd[i];
        lp += -0.5 * d * d - log(M->pstd[i]) - 0.5 * LOG2PI;
        g[i] = -d / M->pstd[i];
    }
    int64_t lo = M->o_l, co = 0;
    int to = 0, eo = 0;
    for (int r = 0; r < M->R; ++r) {
        const int T = M->T[r];
        const double* l = z + lo;
        const int64_t* cnt = M->counts + co;
        double S[MAXT], L[MAXT], Dt[MAXT], Gt[MAXT];
        memset(S, 0, sizeof S);
        /* Lambda = exp.(logLambda); row sums (model_fitness_normal.jl:209-212) and the Poisson form of
           :224-244: sum_tb R*l - exp(l) - lgamma(R+1) */
        double lobs = 0.0;
#pragma omp parallel for reduction(+ : S[:MAXT], lobs) schedule(static)
        for (int64_t b = 0; b < B; ++b)
            for (int t = 0; t < T; ++t) {
                const double lam = exp(l[b * T + t]);
                const double Rc = (double)cnt[b * T + t];
                S[t] += lam;
                lobs += Rc * l[b * T + t] - lam - lgamma(Rc + 1.0);
            }
        lp += lobs;
        for (int t = 0; t < T; ++t) L[t] = log(S[t]);
        /* Normal likelihood of the log frequency ratios (model_fitness_normal.jl:215, 251-270):
           gamma_tb = (l[t+1,b]-l[t,b]) - (L[t+1]-L[t]); neutrals ~ N(-sbar_t, e^{lsbar_t}); mutants ~ N(s_eff - sbar_t, e^{ls_eff}) */
        double lnorm = 0.0;
        memset(Dt, 0, sizeof Dt);
        double gs[MAXT], gls[MAXT];
        memset(gs, 0, sizeof gs);
        memset(gls, 0, sizeof gls);
#pragma omp parallel for reduction(+ : Dt[:MAXT], gs[:MAXT], gls[:MAXT], lnorm) schedule(static)
        for (int64_t b = 0; b < B; ++b) {
            for (int t = 0; t < T - 1; ++t) {
                const double gam = (l[b * T + t + 1] - l[b * T + t]) - (L[t + 1] - L[t]);
                const double sbar = z[M->o_spop + to + t];
                double mean, lsv;
                if (b < M->nn) { mean = -sbar; lsv = z[M->o_lspop + to + t]; }
                else { mean = s_eff(M, z, b - M->nn, r, t, eo) - sbar; lsv = z[ls_index(M, b - M->nn, r, t, eo)]; }
                const double w = exp(-2.0 * lsv), res = gam - mean;
                lnorm += -0.5 * w * res * res - lsv - 0.5 * LOG2PI;
                const double dres = -w * res;           /* d/d res */
                /* res depends on l[t+1] (+), l[t] (-), L[t+1] (-), L[t] (+), sbar (+), s_eff (-) */
                g[lo + b * T + t + 1] += dres;
                g[lo + b * T + t] -= dres;
                Dt[t] += -dres;                         /* sum_b w res: d/d(L[t+1]-L[t]) collects +w res */
                gs[t] += dres;                          /* d/d sbar_t */
                if (b < M->nn) gls[t] += w * res * res - 1.0;
                else {
                    const int64_t m = b - M->nn;
                    const double ds = -dres;            /* d/d s_eff */
                    g[ls_index(M, m, r, t, eo)] += w * res * res - 1.0;
                    switch (M->kind) {
                    case 0: g[M->o_s + m] += ds; break;
                    case 1: g[M->o_s + m * M->E + M->env_idx[t + 1]] += ds; break;
                    case 2: {
                        const double et = exp(z[M->o_lt + m]);
                        g[M->o_tt + m] += ds * et;
                        g[M->o_lt + m] += ds * et * z[M->o_tt + m];
                        /* theta[geno]: several mutants share it -> accumulated below, serially */
                        break;
                    }
                    case 3: {
                        const double et = exp(z[M->o_lt + r * M->nb + m]);
                        g[M->o_s + m] += ds;            /* one barcode per thread iteration: no race */
                        g[M->o_tt + r * M->nb + m] += ds * et;
                        g[M->o_lt + r * M->nb + m] += ds * et * z[M->o_tt + r * M->nb + m];
                        break;
                    }
                    default: {
                        const int64_t u = unit4(M, m, r, t, eo);
                        const double et = exp(z[M->o_lt + u]);
                        g[M->o_s + m * M->E + M->env_idx[eo + t + 1]] += ds;   /* this barcode's entries only: no race */
                        g[M->o_tt + u] += ds * et;
                        g[M->o_lt + u] += ds * et * z[M->o_tt + u];
                    }
                    }
                }
            }
        }
        lp += lnorm;
        if (M->kind == 2) {   /* d/d theta_g = sum over the genotype's mutants of d/d s_eff (serial: shared targets) */
            for (int64_t m = 0; m < M->nb; ++m)
                for (int t = 0; t < T - 1; ++t) {
                    const int64_t b = M->nn + m;
                    const double gam = (l[b * T + t + 1] - l[b * T + t]) - (L[t + 1] - L[t]);
                    const double res = gam - (s_eff(M, z, m, r, t, eo) - z[M->o_spop + to + t]);
                    g[M->o_s + M->geno_idx[m]] += exp(-2.0 * z[M->o_ls + m]) * res;
                }
        }
        for (int t = 0; t < T - 1; ++t) {
            g[M->o_spop + to + t] += gs[t];
            g[M->o_lspop + to + t] += gls[t];
        }
        /* through the normalisers: dL_t/dl_tb = lam_tb / S_t; res_t carries -(L[t+1]-L[t]) */
        for (int t = 0; t < T; ++t) Gt[t] = (t > 0 ? Dt[t - 1] : 0.0) - (t < T - 1 ? Dt[t] : 0.0);
#pragma omp parallel for schedule(static)
        for (int64_t b = 0; b < B; ++b)
            for (int t = 0; t < T; ++t) {
                const double lam = exp(l[b * T + t]);
                g[lo + b * T + t] += (double)cnt[b * T + t] - lam + lam / S[t] * Gt[t];
            }
        lo += (int64_t)T * B;
        co += (int64_t)T * B;
        to += T - 1;
        if (M->kind == 4) eo += T;
    }
    return lp;
}

/* ELBO = (1/S) sum_s logjoint(mu + softplus(omega) eps_s) + H(q) and its gradient w.r.t. (mu, omega). */
double port_elbo_grad(const port_model* M, const double* mu, const double* omega, const double* eps, int S,
                      double* gmu, double* gom, double* work /* 2 D */, int nthreads) {
    const int64_t D = M->D;
    double* z = work;
    double* g = work + D;
    double acc = 0.0;
    for (int64_t i = 0; i < D; ++i) { gmu[i] = 0.0; gom[i] = 0.0; }
    for (int s = 0; s < S; ++s) {
        const double* e = eps + (int64_t)s * D;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < D; ++i) z[i] = mu[i] + softplus(omega[i]) * e[i];
        acc += port_logjoint_grad(M, z, g, nthreads) / S;
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < D; ++i) {
            gmu[i] += g[i] / S;
            gom[i] += g[i] * e[i] * sigmoid(omega[i]) / S;
        }
    }
    double H = 0.5 * (double)D * (1.0 + LOG2PI);
#pragma omp parallel for reduction(+ : H) schedule(static)
    for (int64_t i = 0; i < D; ++i) {
        const double sp = softplus(omega[i]);
        H += log(sp);
        gom[i] += sigmoid(omega[i]) / sp;
    }
    return acc + H;
}

/* ---- the engine's normal stream (oracle/rng.py), restated in C ------------------------------------ */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
void port_normals(uint64_t seed, uint32_t step, uint32_t stream, int64_t D, double* out) {
#pragma omp parallel for schedule(static)
    for (int64_t q = 0; q < (D + 1) / 2; ++q) {
        uint32_t c[4] = {(uint32_t)q, (uint32_t)((uint64_t)q >> 32), step, stream};
        philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
        const uint64_t a = ((uint64_t)c[1] << 32) | c[0], b = ((uint64_t)c[3] << 32) | c[2];
        const double u1 = ((double)(a >> 11) + 1.0) * 0x1.0p-53, u2 = (double)(b >> 11) * 0x1.0p-53;
        const double rr = sqrt(-2.0 * log(u1)), ang = 2.0 * M_PI * u2;
        out[2 * q] = rr * cos(ang);
        if (2 * q + 1 < D) out[2 * q + 1] = rr * sin(ang);
    }
}

/* AdvancedVI.optimize!: n_steps of grad(-ELBO) -> optimiser -> theta -= delta, theta = [mu; omega] in place.
 * opt 0: TruncatedADAGrad(eta, tau, window): window_exact != 0 re-adds the whole window every step (the
 *        reference's `sum(g2)`), else a running sum re-added exactly once per window for ten windows, then once per ten.
 * opt 1: DecayedADAGrad(eta, pre, post).
 * state: caller-allocated, zero-initialised: opt 0 -> (window + 1) * 2D doubles, opt 1 -> 2D doubles set to 1e-8. */
static void port_two_sum(double a, double b, double* s, double* e) {   /* a + b = s + e exactly */
    const double t = a + b, bb = t - a;
    *s = t;
    *e = (a - (t - bb)) + (b - bb);
}

int port_run(const port_model* M, double* mu, double* omega, int64_t first_step, int64_t n_steps, int S, int opt,
             double eta, double tau, int window, int window_exact, double pre, double post, uint64_t seed,
             double* state, double* elbo_trace /* n_steps or NULL */, int nthreads) {
    const int64_t D = M->D;
    double* eps = (double*)malloc(sizeof(double) * (size_t)S * D);
    double* work = (double*)malloc(sizeof(double) * 2 * D);
    double* gmu = (double*)malloc(sizeof(double) * 2 * D);
    if (!eps || !work || !gmu) return -1;
    double* gom = gmu + D;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
    for (int64_t it = first_step; it < first_step + n_steps; ++it) {
        for (int s = 0; s < S; ++s) port_normals(seed, (uint32_t)it, (uint32_t)s, D, eps + (int64_t)s * D);
        const double el = port_elbo_grad(M, mu, omega, eps, S, gmu, gom, work, nthreads);
        if (elbo_trace) elbo_trace[it - first_step] = el;
        const int slot = (int)(it % window);
        /* the engine's schedule (bb_slot_of): exact re-add every step (window_exact) or never -- the running sum is compensated */
        const int resum = window_exact;
#pragma omp parallel for schedule(static)
        for (int64_t j = 0; j < 2 * D; ++j) {
            double* p = j < D ? &mu[j] : &omega[j - D];
            const double d = -gmu[j];               /* gradient of -ELBO; gmu/gom are contiguous */
            double upd;
            if (opt == 0) {
                double* hist = state;               /* [window][2D], then acc [2D], then their low-order parts [2D] */
                double* acc = state + (int64_t)window * 2 * D;
                double* lo = acc + 2 * D;             /* low-order parts (the engine keeps them as floats) */
                const double n2 = d * d, old = hist[(int64_t)slot * 2 * D + j];
                hist[(int64_t)slot * 2 * D + j] = n2;
                /* compensated running sum (engine: bb_opt_apply, same arithmetic): two error-free sums, the rounding errors
                   collect in a float */
                double sacc;
                if (resum) {
                    sacc = 0.0;
                    for (int k = 0; k < window; ++k) sacc += hist[(int64_t)k * 2 * D + j];
                    lo[j] = 0.0;
                } else {
                    double t, e1;
                    port_two_sum(acc[j], n2, &t, &e1);
                    const double u = t - old, e2 = (t - u) - old;      /* fast two-sum: t >= old */
                    const double l = lo[j] + (e1 + e2);
                    sacc = u + l;
                    lo[j] = (double)(float)(l - (sacc - u));
                    sacc = sacc > 0.0 ? sacc : 0.0;
                }
                acc[j] = sacc;
                upd = d * (eta / (tau + sqrt(sacc)));
            } else {
                const double a = post * state[j] + pre * d * d;
                state[j] = a;
                upd = d * (eta / (sqrt(a) + 1e-8));
            }
            *p -= upd;
        }
    }
    free(eps); free(work); free(gmu);
    return 0;
}

int port_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
