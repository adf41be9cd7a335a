// bb_block.h -- block programs of the ADVI step for the fitness_normal model family.
//
// One workgroup owns a tile of NB consecutive barcodes (all time points, all replicates, and
// every per-mutant latent of those barcodes).  A step of one MC sample is two sweeps with the
// only cross-barcode coupling -- the per-(replicate, time) normaliser sum_b exp(loglambda)
// (src/model_fitness_normal.jl:212) and the global latents -- exchanged between them as K moment
// rows:
//
//   block_sample : z = mu + softplus(omega) * eps for the tile's latents (Philox4x32-10 + Box-Muller,
//                  keyed by the latent's index in the reference's flat vector), staged in LDS;
//                  per-(r,t) partial moments S, M0, M1, M2, N1, N2 of the tile.
//   block_update : sums the moment rows, finishes the global quantities (log-normalisers, c_t,
//                  G_t, global-latent gradients), recomputes the tile's residuals in LDS, gathers
//                  each latent's log-joint gradient, adds prior + entropy terms and applies the
//                  optimiser (TruncatedADAGrad / DecayedADAGrad) in place.
//
// The programs are written as barrier-separated passes over "thread ids" and communicate between
// passes only through LDS / global memory, so the same source also compiles as a sequential host
// emulation (BB_EMU, used by tests/ to debug indexing without a GPU; never shipped).
#pragma once
#include "bb_types.h"
#include <math.h>

#ifdef BB_EMU
#define BB_DEV static inline
struct BBCtx { int nthr; int block; double* lds; };
#define BB_PASS(cx, tid) for (int tid = 0; tid < (cx).nthr; ++tid)
#define BB_SYNC(cx) ((void)0)
BB_DEV unsigned bb_umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
BB_DEV void bb_sincospi(double x, double* s, double* c) { *s = sin(M_PI * x); *c = cos(M_PI * x); }
#else
#include <hip/hip_runtime.h>
#define BB_DEV __device__ __forceinline__
struct BBCtx { int nthr; int block; double* lds; };
#define BB_PASS(cx, tid) for (int tid = threadIdx.x, _once = 1; _once; _once = 0)
#define BB_SYNC(cx) __syncthreads()
BB_DEV unsigned bb_umulhi(unsigned a, unsigned b) { return __umulhi(a, b); }
BB_DEV void bb_sincospi(double x, double* s, double* c) { sincospi(x, s, c); }
#endif

#define BB_STREAM_INIT_MU 0xFFFFFFFFu
#define BB_STREAM_INIT_OMEGA 0xFFFFFFFEu
#define BB_LOG2PI 1.8378770664093454835606594728112

struct alignas(16) bb_d2 { double x, y; };

// ------------------------------------------------------------------------------------------------
// RNG: Philox4x32-10, counter = (q_lo, q_hi, step, stream), key = seed; Box-Muller on two 53-bit
// uniforms.  Latent 2q takes the cosine branch, 2q+1 the sine branch.
// ------------------------------------------------------------------------------------------------
BB_DEV void bb_philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                             unsigned* o) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        unsigned long long p0 = 0xD2511F53ull * c0;
        unsigned long long p1 = 0xCD9E8D57ull * c2;
        unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        unsigned n1 = (unsigned)p1;
        unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        unsigned n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

BB_DEV void bb_normal_pair(unsigned long long seed, unsigned long long q, unsigned step, unsigned stream,
                           double* n0, double* n1) {
    unsigned o[4];
    bb_philox4x32_10((unsigned)q, (unsigned)(q >> 32), step, stream, (unsigned)seed, (unsigned)(seed >> 32), o);
    unsigned long long a = ((unsigned long long)o[1] << 32) | o[0];
    unsigned long long b = ((unsigned long long)o[3] << 32) | o[2];
    double u1 = ((double)(a >> 11) + 1.0) * 0x1.0p-53;   // (0, 1]
    double u2 = (double)(b >> 11) * 0x1.0p-53;           // [0, 1)
    double r = sqrt(-2.0 * log(u1));
    double s, c;
    bb_sincospi(2.0 * u2, &s, &c);
    *n0 = r * c;
    *n1 = r * s;
}

BB_DEV void bb_softplus_sigmoid(double om, double* sp, double* sig) {
    double e = exp(-fabs(om));
    double inv = 1.0 / (1.0 + e);
    *sp = fmax(om, 0.0) + log1p(e);
    *sig = om >= 0.0 ? inv : e * inv;
}

BB_DEV void bb_prior_of(const DevModel& M, int blk, long long j, double* mean, double* inv_var) {
    const DevPrior& p = M.pri[blk];
    if (p.mean_e) { *mean = p.mean_e[j]; *inv_var = p.inv_var_e[j]; }
    else { *mean = p.mean; *inv_var = p.inv_var; }
}

// ------------------------------------------------------------------------------------------------
// LDS carve-up (offsets in doubles) for a tile of NB barcodes worked by nthr threads.
// ------------------------------------------------------------------------------------------------
struct BBLds {
    int zl, zs0, zs1, zs2, zs3, seff, weff, res, As, Qs, acc, wk, Lt, invS, cc, GG, wbar, gglob, misc, part;
    int total;
};

BB_DEV int bb_xdim(const DevModel& M) { return M.kind == 1 ? M.E : M.R; }

static inline
#ifndef BB_EMU
__host__ __device__
#endif
BBLds bb_lds_layout(int R, int E, int kind, int Ttot, int nt1, int K, int NB, int nthr) {
    BBLds L;
    int X = (kind == 1) ? E : R;
    int o = 0;
    L.zl = o;   o += NB * Ttot;
    L.zs0 = o;  o += NB * X;
    L.zs1 = o;  o += NB * X;
    L.zs2 = o;  o += NB * X;
    L.zs3 = o;  o += NB;
    L.seff = o; o += NB * X;
    L.weff = o; o += NB * X;
    L.res = o;  o += NB * (Ttot - R);
    L.As = o;   o += NB * X;
    L.Qs = o;   o += NB * X;
    L.acc = o;  o += (BB_NQ + 1) * nthr;
    L.wk = o;   o += K;
    L.Lt = o;   o += Ttot;
    L.invS = o; o += Ttot;
    L.cc = o;   o += Ttot;
    L.GG = o;   o += Ttot;
    L.wbar = o; o += Ttot;
    L.gglob = o; o += 2 * nt1;
    L.misc = o; o += 16 + BB_MAX_REP;
    L.part = o; o += 32;
    L.total = (o + 1) & ~1;
    return L;
}

struct BBTile {
    long long b0;      // first barcode of the tile
    int nbt;           // barcodes in the tile
    int nshift;        // neutrals in the tile (they come first)
    long long m0;      // first mutant (0-based among mutants) of the tile
    int nmt;           // mutants in the tile
    int NB;
};

BB_DEV BBTile bb_tile(const DevModel& M, const RunArgs& A, int block, int NB) {
    BBTile t;
    t.NB = NB;
    t.b0 = A.b_lo + (long long)block * NB;
    long long b1 = t.b0 + NB < A.b_hi ? t.b0 + NB : A.b_hi;
    t.nbt = (int)(b1 > t.b0 ? b1 - t.b0 : 0);
    long long ns = M.nn - t.b0;
    t.nshift = (int)(ns < 0 ? 0 : (ns > t.nbt ? t.nbt : ns));
    t.m0 = t.b0 + t.nshift - M.nn;
    t.nmt = t.nbt - t.nshift;
    return t;
}

// which per-unit slot time step t of replicate r uses (environment of t+1 / replicate / 0)
BB_DEV int bb_xof(const DevModel& M, int r, int t) {
    return M.kind == 1 ? M.env_idx[t + 1] : (M.kind == 3 ? r : 0);
}

// ------------------------------------------------------------------------------------------------
// per-latent work over a flat index range [lo, hi), pairs (2q, 2q+1) share one Philox call
// ------------------------------------------------------------------------------------------------
// Draw, transform, save (eps, softplus, sigmoid) and hand z to dst(j = i - lo, z).
// Returns this thread's ELBO terms (prior quadratic + log sigma) when asked.
template <class Dst>
BB_DEV double bb_sample_seg(const DevModel& M, const DevState& S, const RunArgs& A, unsigned step, int blk,
                            long long lo, long long hi, long long t0, long long tstride, bool want_elbo, Dst dst) {
    double el = 0.0;
    if (hi <= lo) return el;
    const long long blo = M.blk_lo[blk];
    const long long qhi = (hi - 1) >> 1;
    for (long long q = (lo >> 1) + t0; q <= qhi; q += tstride) {
        const long long i0 = 2 * q, i1 = i0 + 1;
        const bool a0 = i0 >= lo, a1 = i1 < hi;
        double e0, e1;
        if (S.eps_in) {
            e0 = a0 ? S.eps_in[(long long)A.sample * M.D + i0] : 0.0;
            e1 = a1 ? S.eps_in[(long long)A.sample * M.D + i1] : 0.0;
        } else {
            bb_normal_pair(A.seed, (unsigned long long)q, step, (unsigned)A.sample, &e0, &e1);
        }
        double mu0 = 0, mu1 = 0, om0 = 0, om1 = 0;
        if (a0 && a1) {
            bb_d2 m = *(const bb_d2*)(S.mu + i0), o = *(const bb_d2*)(S.om + i0);
            mu0 = m.x; mu1 = m.y; om0 = o.x; om1 = o.y;
        } else if (a0) { mu0 = S.mu[i0]; om0 = S.om[i0]; }
        else { mu1 = S.mu[i1]; om1 = S.om[i1]; }
        double sp0, sg0, sp1, sg1;
        bb_softplus_sigmoid(om0, &sp0, &sg0);
        bb_softplus_sigmoid(om1, &sp1, &sg1);
        const double z0 = fma(sp0, e0, mu0), z1 = fma(sp1, e1, mu1);
        if (a0 && a1) {
            *(bb_d2*)(S.eps + i0) = bb_d2{e0, e1};
            *(bb_d2*)(S.sp + i0) = bb_d2{sp0, sp1};
            *(bb_d2*)(S.sig + i0) = bb_d2{sg0, sg1};
        } else if (a0) { S.eps[i0] = e0; S.sp[i0] = sp0; S.sig[i0] = sg0; }
        else { S.eps[i1] = e1; S.sp[i1] = sp1; S.sig[i1] = sg1; }
        if (a0) dst(i0 - lo, z0);
        if (a1) dst(i1 - lo, z1);
        if (want_elbo) {
            double pm, iv;
            if (a0) { bb_prior_of(M, blk, i0 - blo, &pm, &iv); el += -0.5 * (z0 - pm) * (z0 - pm) * iv + log(sp0); }
            if (a1) { bb_prior_of(M, blk, i1 - blo, &pm, &iv); el += -0.5 * (z1 - pm) * (z1 - pm) * iv + log(sp1); }
        }
    }
    return el;
}

// Rebuild z = mu + softplus(omega) * eps of the saved draw.
template <class Dst>
BB_DEV void bb_loadz_seg(const DevState& S, long long lo, long long hi, long long t0, long long tstride, Dst dst) {
    for (long long i = lo + t0; i < hi; i += tstride) dst(i - lo, fma(S.sp[i], S.eps[i], S.mu[i]));
}

// One optimiser update of parameter *p with gradient-of-(-ELBO) d.  which: 0 = mu, 1 = omega.
BB_DEV void bb_opt_apply(const DevModel& M, const DevState& S, const RunArgs& A, unsigned long long step,
                         int which, long long i, double d, double* p, double* acc) {
    double upd;
    if (A.opt == 0) {   // TruncatedADAGrad: g2[mod(i-1,n)+1] = d^2; s = sum(g2); d *= eta / (tau + sqrt(s))
        const int slot = (int)(step % (unsigned long long)A.W);
        double* hs = S.hist + ((long long)slot * 2 + which) * M.D + i;
        const double n2 = d * d;
        const double old = *hs;
        *hs = n2;
        double s;
        const bool resum = A.resum_every == 1 || (step > 0 && step % (unsigned long long)A.resum_every == 0);
        if (resum) {
            s = 0.0;
            for (int j = 0; j < A.W; ++j) s += (j == slot) ? n2 : S.hist[((long long)j * 2 + which) * M.D + i];
        } else {
            s = fmax(*acc + n2 - old, 0.0);
        }
        *acc = s;
        upd = d * (A.eta / (A.tau + sqrt(s)));
    } else {            // DecayedADAGrad: acc = post*acc + pre*d^2; d *= eta / (sqrt(acc) + 1e-8)
        const double a = A.post * (*acc) + A.pre * d * d;
        *acc = a;
        upd = d * (A.eta / (sqrt(a) + 1e-8));
    }
    *p -= upd;
}

// Finish one latent: total gradient of the log-joint (likelihood part from glik + prior), the
// reparameterisation gradient w.r.t. (mu, omega), S-sample averaging, entropy term, optimiser.
template <class Grad>
BB_DEV void bb_update_seg(const DevModel& M, const DevState& S, const RunArgs& A, unsigned long long step, int blk,
                          long long lo, long long hi, long long t0, long long tstride, Grad glik) {
    const long long blo = M.blk_lo[blk];
    const double invS = 1.0 / (double)A.S;
    for (long long i = lo + t0; i < hi; i += tstride) {
        const double mu = S.mu[i], om = S.om[i], e = S.eps[i], sp = S.sp[i], sg = S.sig[i];
        const double z = fma(sp, e, mu);
        double pm, iv;
        bb_prior_of(M, blk, i - blo, &pm, &iv);
        const double g = glik(i - lo, z) - (z - pm) * iv;   // d logjoint / d z_i
        double gm = g, go = g * e * sg;
        if (A.S > 1) {
            if (!A.first_sample) { gm += S.gacc_mu[i]; go += S.gacc_om[i]; }
            if (!A.last_sample) { S.gacc_mu[i] = gm; S.gacc_om[i] = go; continue; }
            gm *= invS; go *= invS;
        }
        go += sg / sp;                                        // d H / d omega
        if (!A.apply) { S.gacc_mu[i] = gm; S.gacc_om[i] = go; continue; }   // export d ELBO / d theta
        double pmu = mu, pom = om, am = S.acc_mu[i], ao = S.acc_om[i];
        bb_opt_apply(M, S, A, step, 0, i, -gm, &pmu, &am);
        bb_opt_apply(M, S, A, step, 1, i, -go, &pom, &ao);
        S.mu[i] = pmu; S.om[i] = pom; S.acc_mu[i] = am; S.acc_om[i] = ao;
    }
}

// Deterministic sum of one LDS row of n entries: 16 partial sums, then one thread.
BB_DEV void bb_row_sum(BBCtx& cx, const double* row, int n, double* part16, double* out) {
    BB_PASS(cx, tid) {
        if (tid < 16) {
            const int chunk = (n + 15) / 16;
            double s = 0.0;
            for (int i = tid * chunk; i < (tid + 1) * chunk && i < n; ++i) s += row[i];
            part16[tid] = s;
        }
    }
    BB_SYNC(cx);
    BB_PASS(cx, tid) {
        if (tid == 0) {
            double s = 0.0;
            for (int i = 0; i < 16; ++i) s += part16[i];
            *out = s;
        }
    }
    BB_SYNC(cx);
}

// Per-unit effective fitness and precision tables for the tile (slot = bl * X + x):
//   fitness   s_eff = s_bc                          w = exp(-2 logsigma_bc)      model_fitness_normal.jl:262-270
//   multienv  s_eff[e] = s_bc[e, m]                 w[e] likewise                model_multienv_fitness_normal.jl:293-301
//   genotype  s_eff = theta[geno] + exp(logtau)*tt                               ..._genotypes.jl:230
//   replicate s_eff[r] = theta + exp(logtau_r)*tt_r                              ..._replicates.jl:216
// Returns the thread's ELBO term  - sum_units logsigma_eff * (#time steps using the slot).
BB_DEV double bb_effective_tables(BBCtx& cx, int tid, const DevModel& M, const DevState& S, const BBLds& L,
                                  const BBTile& t, bool want_elbo) {
    double* lds = cx.lds;
    const int X = bb_xdim(M);
    double el = 0.0;
    for (int u = tid; u < t.nbt * X; u += cx.nthr) {
        const int bl = u / X, x = u - bl * X;
        double se = 0.0, we = 0.0;
        if (bl >= t.nshift) {
            const int ml = bl - t.nshift;
            double ls;
            if (M.kind == 0) { se = lds[L.zs0 + ml]; ls = lds[L.zs1 + ml]; }
            else if (M.kind == 1) { se = lds[L.zs0 + ml * X + x]; ls = lds[L.zs1 + ml * X + x]; }
            else if (M.kind == 2) {
                se = S.ztheta[M.geno_idx[t.m0 + ml]] + exp(lds[L.zs1 + ml]) * lds[L.zs0 + ml];
                ls = lds[L.zs2 + ml];
            } else {
                se = lds[L.zs3 + ml] + exp(lds[L.zs1 + x * t.NB + ml]) * lds[L.zs0 + x * t.NB + ml];
                ls = lds[L.zs2 + x * t.NB + ml];
            }
            we = exp(-2.0 * ls);
            if (want_elbo) {
                int cnt;
                if (M.kind == 1) { cnt = 0; for (int tt = 0; tt < M.T[0] - 1; ++tt) cnt += (M.env_idx[tt + 1] == x); }
                else cnt = M.T[M.kind == 3 ? x : 0] - 1;
                el -= ls * cnt;
            }
        }
        lds[L.seff + u] = se;
        lds[L.weff + u] = we;
    }
    return el;
}

// Stage the tile's latent samples into LDS.  SAMPLE = draw them; otherwise rebuild the saved draw.
template <bool SAMPLE>
BB_DEV double bb_stage_tile(BBCtx& cx, int tid, const DevModel& M, const DevState& S, const RunArgs& A,
                            const BBLds& L, const BBTile& t, unsigned step, bool want_elbo) {
    double* lds = cx.lds;
    double el = 0.0;
    auto seg = [&](int blk, long long lo, long long n, int ldsoff) {
        auto dst = [&](long long j, double z) { lds[ldsoff + j] = z; };
        if (SAMPLE) el += bb_sample_seg(M, S, A, step, blk, lo, lo + n, tid, cx.nthr, want_elbo, dst);
        else bb_loadz_seg(S, lo, lo + n, tid, cx.nthr, dst);
    };
    for (int r = 0; r < M.R; ++r)
        seg(BK_L, M.off_l[r] + t.b0 * M.T[r], (long long)t.nbt * M.T[r], L.zl + t.NB * M.tcum[r]);
    if (t.nmt > 0) {
        if (M.kind == 0) {
            seg(BK_S, M.blk_lo[BK_S] + t.m0, t.nmt, L.zs0);
            seg(BK_LS, M.blk_lo[BK_LS] + t.m0, t.nmt, L.zs1);
        } else if (M.kind == 1) {
            seg(BK_S, M.blk_lo[BK_S] + t.m0 * M.E, (long long)t.nmt * M.E, L.zs0);
            seg(BK_LS, M.blk_lo[BK_LS] + t.m0 * M.E, (long long)t.nmt * M.E, L.zs1);
        } else if (M.kind == 2) {
            seg(BK_TT, M.blk_lo[BK_TT] + t.m0, t.nmt, L.zs0);
            seg(BK_LT, M.blk_lo[BK_LT] + t.m0, t.nmt, L.zs1);
            seg(BK_LS, M.blk_lo[BK_LS] + t.m0, t.nmt, L.zs2);
        } else {
            seg(BK_S, M.blk_lo[BK_S] + t.m0, t.nmt, L.zs3);
            for (int r = 0; r < M.R; ++r) {
                seg(BK_TT, M.blk_lo[BK_TT] + r * M.nb + t.m0, t.nmt, L.zs0 + r * t.NB);
                seg(BK_LT, M.blk_lo[BK_LT] + r * M.nb + t.m0, t.nmt, L.zs1 + r * t.NB);
                seg(BK_LS, M.blk_lo[BK_LS] + r * M.nb + t.m0, t.nmt, L.zs2 + r * t.NB);
            }
        }
    }
    return el;
}

// ================================================================================================
// block_sample: sampling sweep + partial moments of one tile
// ================================================================================================
BB_DEV void bb_block_sample(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, int NB) {
    const BBLds L = bb_lds_layout(M.R, M.E, M.kind, M.Ttot, M.nt1, M.K, NB, cx.nthr);
    double* lds = cx.lds;
    const BBTile t = bb_tile(M, A, cx.block, NB);
    const unsigned step = (unsigned)S.ctr[A.par];
    const int X = bb_xdim(M);
    const bool we = A.with_elbo != 0;

    // pass S: draw every latent of the tile (block 0 also draws the replicated global latents)
    BB_PASS(cx, tid) {
        double el = bb_stage_tile<true>(cx, tid, M, S, A, L, t, step, we);
        if (cx.block == 0) {
            double eg = 0.0;
            eg += bb_sample_seg(M, S, A, step, BK_SPOP, M.blk_lo[BK_SPOP], M.blk_hi[BK_SPOP], tid, cx.nthr, we,
                                [&](long long j, double z) { S.zg[j] = z; });
            eg += bb_sample_seg(M, S, A, step, BK_LSPOP, M.blk_lo[BK_LSPOP], M.blk_hi[BK_LSPOP], tid, cx.nthr, we,
                                [&](long long j, double z) { S.zg[M.nt1 + j] = z; });
            if (A.count_globals) el += eg;
        }
        lds[L.acc + BB_NQ * cx.nthr + tid] = el;
        for (int k = tid; k < M.K; k += cx.nthr) lds[L.wk + k] = 0.0;
    }
    BB_SYNC(cx);

    // pass E: effective fitness / precision per unit
    BB_PASS(cx, tid) {
        double el = bb_effective_tables(cx, tid, M, S, L, t, we);
        lds[L.acc + BB_NQ * cx.nthr + tid] += el;
    }
    BB_SYNC(cx);

    // pass M: per replicate, every (barcode, time) element contributes to the moments of its time
    for (int r = 0; r < M.R; ++r) {
        const int T = M.T[r];
        const int bstride = cx.nthr / T, nact = bstride * T;
        const double* zl = lds + L.zl + NB * M.tcum[r];
        BB_PASS(cx, tid) {
            double aS = 0, aM0 = 0, aM1 = 0, aM2 = 0, aN1 = 0, aN2 = 0, el = 0;
            if (tid < nact) {
                const int blq = (int)bb_umulhi((unsigned)tid, M.Tmagic[r]);
                const int tt = tid - blq * T;
                const int x = (tt < T - 1) ? bb_xof(M, r, tt) : 0;
                for (int bl = blq; bl < t.nbt; bl += bstride) {
                    const double z = zl[bl * T + tt];
                    const double lam = exp(z);
                    aS += lam;
                    if (we) el += (double)M.counts[M.cnt_off[r] + (t.b0 + bl) * T + tt] * z - lam;
                    if (tt < T - 1) {
                        const double d = zl[bl * T + tt + 1] - z;
                        if (bl >= t.nshift) {
                            const double a = d - lds[L.seff + bl * X + x];
                            const double w = lds[L.weff + bl * X + x];
                            aM0 += w; aM1 += w * a; aM2 += w * a * a;
                        } else {
                            aN1 += d; aN2 += d * d;
                        }
                    }
                }
            }
            double* acc = lds + L.acc;
            acc[0 * cx.nthr + tid] = aS;  acc[1 * cx.nthr + tid] = aM0; acc[2 * cx.nthr + tid] = aM1;
            acc[3 * cx.nthr + tid] = aM2; acc[4 * cx.nthr + tid] = aN1; acc[5 * cx.nthr + tid] = aN2;
            acc[BB_NQ * cx.nthr + tid] += el;
        }
        BB_SYNC(cx);
        BB_PASS(cx, tid) {
            for (int j = tid; j < BB_NQ * T; j += cx.nthr) {
                const int q = (int)bb_umulhi((unsigned)j, M.Tmagic[r]);
                const int tt = j - q * T;
                if (q == 0 || tt < T - 1) {
                    const double* row = lds + L.acc + q * cx.nthr;
                    double s = 0.0;
                    for (int i = 0; i < bstride; ++i) s += row[tt + T * i];
                    lds[L.wk + (q == 0 ? M.kq[r] + tt : M.kq[r] + T + 5 * tt + (q - 1))] = s;
                }
            }
        }
        BB_SYNC(cx);
    }
    bb_row_sum(cx, lds + L.acc + BB_NQ * cx.nthr, cx.nthr, lds + L.part, lds + L.wk + (M.K - 2));
    BB_PASS(cx, tid) {
        for (int k = tid; k < M.K; k += cx.nthr) S.partials[(long long)k * A.nblk + cx.block] = lds[L.wk + k];
    }
    BB_SYNC(cx);
}

// Sum the moment rows and finish everything that depends on them (per replicate, tiny).
BB_DEV void bb_finalize(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BBLds& L) {
    double* lds = cx.lds;
    BB_PASS(cx, tid) {
        for (int k = tid; k < M.K; k += cx.nthr) {
            double s = 0.0;
            for (int j = 0; j < A.nred; ++j) s += A.red[(long long)k * A.nred + j];
            lds[L.wk + k] = s;
        }
    }
    BB_SYNC(cx);
    BB_PASS(cx, tid) {
        if (tid < M.R) {
            const int r = tid, T = M.T[r], tc = M.tcum[r], kq = M.kq[r];
            const double nn = (double)M.nn;
            for (int tt = 0; tt < T; ++tt) {
                const double St = lds[L.wk + kq + tt];
                lds[L.invS + tc + tt] = 1.0 / St;
                lds[L.Lt + tc + tt] = log(St);
            }
            double Dprev = 0.0, elb = 0.0;
            for (int tt = 0; tt < T - 1; ++tt) {
                const double* mm = lds + L.wk + kq + T + 5 * tt;
                const double M0 = mm[0], M1 = mm[1], M2 = mm[2], N1 = mm[3], N2 = mm[4];
                const double sbar = S.zg[M.off_t[r] + tt], ls = S.zg[M.nt1 + M.off_t[r] + tt];
                const double wb = exp(-2.0 * ls);
                const double c = lds[L.Lt + tc + tt + 1] - lds[L.Lt + tc + tt] - sbar;
                const double quadN = N2 - 2.0 * c * N1 + nn * c * c;
                const double quadM = M2 - 2.0 * c * M1 + c * c * M0;
                const double Dt = (M1 - c * M0) + wb * (N1 - c * nn);
                lds[L.gglob + M.off_t[r] + tt] = -Dt;
                lds[L.gglob + M.nt1 + M.off_t[r] + tt] = wb * quadN - nn;
                lds[L.GG + tc + tt] = Dprev - Dt;
                Dprev = Dt;
                lds[L.cc + tc + tt] = c;
                lds[L.wbar + tc + tt] = wb;
                elb += -0.5 * (quadM + wb * quadN) - nn * ls;
            }
            lds[L.GG + tc + T - 1] = Dprev;
            lds[L.misc + 16 + r] = elb;
        }
    }
    BB_SYNC(cx);
}

// ================================================================================================
// block_update: gradient of the log-joint for the tile's latents + optimiser step
// ================================================================================================
BB_DEV void bb_block_update(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, int NB) {
    const BBLds L = bb_lds_layout(M.R, M.E, M.kind, M.Ttot, M.nt1, M.K, NB, cx.nthr);
    double* lds = cx.lds;
    const BBTile t = bb_tile(M, A, cx.block, NB);
    const unsigned long long step = S.ctr[A.par];
    const int X = bb_xdim(M);

    bb_finalize(cx, M, S, A, L);

    BB_PASS(cx, tid) { bb_stage_tile<false>(cx, tid, M, S, A, L, t, (unsigned)step, false); }
    BB_SYNC(cx);
    BB_PASS(cx, tid) { bb_effective_tables(cx, tid, M, S, L, t, false); }
    BB_SYNC(cx);

    // pass R: residuals r = (l[t+1] - l[t]) - s_eff - c_t of every (barcode, time step)
    for (int r = 0; r < M.R; ++r) {
        const int T = M.T[r], T1 = T - 1, tc = M.tcum[r];
        const double* zl = lds + L.zl + NB * tc;
        double* res = lds + L.res + NB * (tc - r);
        const unsigned magic1 = M.Tmagic1[r];
        BB_PASS(cx, tid) {
            for (int j = tid; j < t.nbt * T1; j += cx.nthr) {
                const int bl = T1 == 1 ? j : (int)bb_umulhi((unsigned)j, magic1), tt = j - bl * T1;
                double a = zl[bl * T + tt + 1] - zl[bl * T + tt];
                if (bl >= t.nshift) a -= lds[L.seff + bl * X + bb_xof(M, r, tt)];
                res[j] = a - lds[L.cc + tc + tt];
            }
        }
    }
    BB_SYNC(cx);

    // pass U: per-unit sums  As = sum_t w r  (= dlogp/ds_eff),  Qs = sum_t (w r^2 - 1)  (= dlogp/dlogsigma_eff)
    BB_PASS(cx, tid) {
        for (int u = tid; u < t.nbt * X; u += cx.nthr) {
            const int bl = u / X, x = u - bl * X;
            double as = 0.0, qs = 0.0;
            if (bl >= t.nshift) {
                const double w = lds[L.weff + u];
                const int r = M.kind == 3 ? x : 0;
                const int T1 = M.T[r] - 1;
                const double* res = lds + L.res + NB * (M.tcum[r] - r) + bl * T1;
                for (int tt = 0; tt < T1; ++tt) {
                    if (M.kind == 1 && M.env_idx[tt + 1] != x) continue;
                    const double rr = res[tt];
                    as += w * rr;
                    qs += w * rr * rr - 1.0;
                }
                if (M.kind == 2) S.ds[t.m0 + (bl - t.nshift)] = as;
            }
            lds[L.As + u] = as;
            lds[L.Qs + u] = qs;
        }
    }
    BB_SYNC(cx);

    // pass G: gather each latent's gradient and update it
    const bool do_update = true;
    BB_PASS(cx, tid) {
        if (do_update) {
            for (int r = 0; r < M.R; ++r) {
                const int T = M.T[r], T1 = T - 1, tc = M.tcum[r];
                const double* res = lds + L.res + NB * (tc - r);
                const long long lo = M.off_l[r] + t.b0 * T;
                const unsigned* cnt = M.counts + M.cnt_off[r] + t.b0 * T;
                bb_update_seg(M, S, A, step, BK_L, lo, lo + (long long)t.nbt * T, tid, cx.nthr,
                              [&](long long j, double z) {
                                  const int bl = (int)bb_umulhi((unsigned)j, M.Tmagic[r]), tt = (int)j - bl * T;
                                  const bool mut = bl >= t.nshift;
                                  const double lam = exp(z);
                                  double g = (double)cnt[j] - lam + lam * lds[L.invS + tc + tt] * lds[L.GG + tc + tt];
                                  if (tt < T1) {
                                      const double w = mut ? lds[L.weff + bl * X + bb_xof(M, r, tt)] : lds[L.wbar + tc + tt];
                                      g += w * res[bl * T1 + tt];
                                  }
                                  if (tt > 0) {
                                      const double w = mut ? lds[L.weff + bl * X + bb_xof(M, r, tt - 1)] : lds[L.wbar + tc + tt - 1];
                                      g -= w * res[bl * T1 + tt - 1];
                                  }
                                  return g;
                              });
            }
            if (t.nmt > 0) {
                const int ns = t.nshift;
                if (M.kind == 0 || M.kind == 1) {
                    const int E = M.kind == 1 ? M.E : 1;
                    bb_update_seg(M, S, A, step, BK_S, M.blk_lo[BK_S] + t.m0 * E, M.blk_lo[BK_S] + (t.m0 + t.nmt) * E,
                                  tid, cx.nthr, [&](long long j, double) { return lds[L.As + ns * E + j]; });
                    bb_update_seg(M, S, A, step, BK_LS, M.blk_lo[BK_LS] + t.m0 * E, M.blk_lo[BK_LS] + (t.m0 + t.nmt) * E,
                                  tid, cx.nthr, [&](long long j, double) { return lds[L.Qs + ns * E + j]; });
                } else if (M.kind == 2) {
                    bb_update_seg(M, S, A, step, BK_TT, M.blk_lo[BK_TT] + t.m0, M.blk_lo[BK_TT] + t.m0 + t.nmt, tid, cx.nthr,
                                  [&](long long j, double) { return lds[L.As + ns + j] * exp(lds[L.zs1 + j]); });
                    bb_update_seg(M, S, A, step, BK_LT, M.blk_lo[BK_LT] + t.m0, M.blk_lo[BK_LT] + t.m0 + t.nmt, tid, cx.nthr,
                                  [&](long long j, double z) { return lds[L.As + ns + j] * exp(z) * lds[L.zs0 + j]; });
                    bb_update_seg(M, S, A, step, BK_LS, M.blk_lo[BK_LS] + t.m0, M.blk_lo[BK_LS] + t.m0 + t.nmt, tid, cx.nthr,
                                  [&](long long j, double) { return lds[L.Qs + ns + j]; });
                } else {
                    const int R = M.R;
                    bb_update_seg(M, S, A, step, BK_S, M.blk_lo[BK_S] + t.m0, M.blk_lo[BK_S] + t.m0 + t.nmt, tid, cx.nthr,
                                  [&](long long j, double) {
                                      double s = 0.0;
                                      for (int r = 0; r < R; ++r) s += lds[L.As + (ns + j) * R + r];
                                      return s;
                                  });
                    for (int r = 0; r < R; ++r) {
                        const long long o = (long long)r * M.nb + t.m0;
                        bb_update_seg(M, S, A, step, BK_TT, M.blk_lo[BK_TT] + o, M.blk_lo[BK_TT] + o + t.nmt, tid, cx.nthr,
                                      [&](long long j, double) { return lds[L.As + (ns + j) * R + r] * exp(lds[L.zs1 + r * NB + j]); });
                        bb_update_seg(M, S, A, step, BK_LT, M.blk_lo[BK_LT] + o, M.blk_lo[BK_LT] + o + t.nmt, tid, cx.nthr,
                                      [&](long long j, double z) { return lds[L.As + (ns + j) * R + r] * exp(z) * lds[L.zs0 + r * NB + j]; });
                        bb_update_seg(M, S, A, step, BK_LS, M.blk_lo[BK_LS] + o, M.blk_lo[BK_LS] + o + t.nmt, tid, cx.nthr,
                                      [&](long long j, double) { return lds[L.Qs + (ns + j) * R + r]; });
                    }
                }
            }
            if (cx.block == 0) {   // replicated global latents: every rank applies the identical update
                bb_update_seg(M, S, A, step, BK_SPOP, M.blk_lo[BK_SPOP], M.blk_hi[BK_SPOP], tid, cx.nthr,
                              [&](long long j, double) { return lds[L.gglob + j]; });
                bb_update_seg(M, S, A, step, BK_LSPOP, M.blk_lo[BK_LSPOP], M.blk_hi[BK_LSPOP], tid, cx.nthr,
                              [&](long long j, double) { return lds[L.gglob + M.nt1 + j]; });
            }
        }
    }
    BB_SYNC(cx);

    BB_PASS(cx, tid) {
        if (cx.block == 0 && tid == 0) {
            if (A.with_elbo) {
                double v = lds[L.wk + M.K - 2] + lds[L.wk + M.K - 1] + A.elbo_const;
                for (int r = 0; r < M.R; ++r) v += lds[L.misc + 16 + r];
                S.elbo_sample[A.sample] = v;
                if (A.apply && A.elbo_every > 0) {
                    double* slot = S.elbo_ring + (step / (unsigned long long)A.elbo_every) % BB_ELBO_RING;
                    *slot = (A.first_sample ? 0.0 : *slot) + v / (double)A.S;
                }
            }
            if (A.last_sample && A.apply) S.ctr[1 - A.par] = step + 1;
        }
    }
    BB_SYNC(cx);
}

// ================================================================================================
// genotype model: the per-genotype theta block (sampled / updated by a grid over genotypes)
// ================================================================================================
// update (optional) then sample theta; gsum[g] = sum of ds over the genotype's mutants.
BB_DEV void bb_block_geno(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, int nblocks,
                          int do_update, int do_sample, int upd_par) {
    const long long gt0 = (long long)cx.block * cx.nthr, gstride = (long long)nblocks * cx.nthr;
    double* lds = cx.lds;
    if (do_update) {
        const unsigned long long step = S.ctr[upd_par];
        RunArgs Au = A;
        BB_PASS(cx, tid) {
            bb_update_seg(M, S, Au, step, BK_S, M.blk_lo[BK_S], M.blk_hi[BK_S], gt0 + tid, gstride,
                          [&](long long j, double) { return S.gsum[j]; });
        }
        BB_SYNC(cx);
    }
    if (do_sample) {
        const unsigned step = (unsigned)S.ctr[A.par];
        BB_PASS(cx, tid) {
            double el = bb_sample_seg(M, S, A, step, BK_S, M.blk_lo[BK_S], M.blk_hi[BK_S], gt0 + tid, gstride,
                                      A.with_elbo != 0, [&](long long j, double z) { S.ztheta[j] = z; });
            lds[tid] = el;
        }
        BB_SYNC(cx);
        bb_row_sum(cx, lds, cx.nthr, lds + cx.nthr, lds + cx.nthr + 16);
        BB_PASS(cx, tid) { if (tid == 0) S.geno_el[cx.block] = A.count_globals ? lds[cx.nthr + 16] : 0.0; }
        BB_SYNC(cx);
    }
}

// gsum[g] = sum over the genotype's mutants (CSR order) of ds: deterministic segmented sum.
BB_DEV void bb_block_geno_sum(BBCtx& cx, const DevModel& M, const DevState& S, int nblocks, long long m_lo, long long m_hi) {
    BB_PASS(cx, tid) {
        for (long long g = (long long)cx.block * cx.nthr + tid; g < M.G; g += (long long)nblocks * cx.nthr) {
            double s = 0.0;
            for (int k = M.geno_ptr[g]; k < M.geno_ptr[g + 1]; ++k) {
                const int m = M.geno_mem[k];
                if (m >= m_lo && m < m_hi) s += S.ds[m];
            }
            S.gsum[g] = s;
        }
    }
    BB_SYNC(cx);
}

// partials [K][nblk] -> totals [K] (one block); row K-1 additionally takes the theta-block ELBO partials.
BB_DEV void bb_block_reduce(BBCtx& cx, const DevModel& M, const DevState& S, int nblk, int ngeno_blocks) {
    BB_PASS(cx, tid) {
        for (int k = tid; k < M.K; k += cx.nthr) {
            double s = 0.0;
            for (int j = 0; j < nblk; ++j) s += S.partials[(long long)k * nblk + j];
            if (k == M.K - 1) for (int j = 0; j < ngeno_blocks; ++j) s += S.geno_el[j];
            S.totals[k] = s;
        }
    }
    BB_SYNC(cx);
}

// Turing.meanfield: mu0 = randn(D), omega0 = randn(D) from the init streams.
BB_DEV void bb_block_init(BBCtx& cx, const DevModel& M, const DevState& S, unsigned long long seed, int nblocks) {
    BB_PASS(cx, tid) {
        const long long npair = (M.D + 1) / 2;
        for (long long q = (long long)cx.block * cx.nthr + tid; q < npair; q += (long long)nblocks * cx.nthr) {
            double a, b;
            bb_normal_pair(seed, (unsigned long long)q, 0u, BB_STREAM_INIT_MU, &a, &b);
            S.mu[2 * q] = a;
            if (2 * q + 1 < M.D) S.mu[2 * q + 1] = b;
            bb_normal_pair(seed, (unsigned long long)q, 0u, BB_STREAM_INIT_OMEGA, &a, &b);
            S.om[2 * q] = a;
            if (2 * q + 1 < M.D) S.om[2 * q + 1] = b;
        }
    }
    BB_SYNC(cx);
}

BB_DEV void bb_block_normals(BBCtx& cx, unsigned long long seed, unsigned step, unsigned stream, long long lo,
                             long long hi, double* out, int nblocks) {
    BB_PASS(cx, tid) {
        for (long long q = (lo >> 1) + (long long)cx.block * cx.nthr + tid; q <= (hi - 1) >> 1;
             q += (long long)nblocks * cx.nthr) {
            double a, b;
            bb_normal_pair(seed, (unsigned long long)q, step, stream, &a, &b);
            if (2 * q >= lo) out[2 * q - lo] = a;
            if (2 * q + 1 < hi) out[2 * q + 1 - lo] = b;
        }
    }
    BB_SYNC(cx);
}
