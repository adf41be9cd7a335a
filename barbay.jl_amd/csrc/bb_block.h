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
#include <string.h>

#ifdef BB_EMU
#define BB_DEV static inline
#define BB_HD static inline
struct BBCtx { int nthr; int block; double* lds; const void* lay = nullptr; /* resident launch: its BBLds, precomputed */ };
#define BB_PASS(cx, tid) for (int tid = 0; tid < (cx).nthr; ++tid)
#define BB_SYNC(cx) ((void)0)
BB_DEV unsigned bb_umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
#else
#include <hip/hip_runtime.h>
#define BB_DEV __device__ __forceinline__
#define BB_HD __host__ __device__ __forceinline__      /* also called by the host engine (the tiles' tables are built there) */
struct BBCtx { int nthr; int block; double* lds; const void* lay = nullptr; /* resident launch: its BBLds, precomputed */ };
#define BB_PASS(cx, tid) for (int tid = threadIdx.x, _once = 1; _once; _once = 0)
#define BB_SYNC(cx) __syncthreads()
BB_DEV unsigned bb_umulhi(unsigned a, unsigned b) { return __umulhi(a, b); }
#endif

// In-kernel stamps (diagnostic build only, -DBB_STAMPS): s_memtime at pass boundaries of every block,
// written to a buffer nothing else reads.  The shipped build compiles them out.
#if defined(BB_STAMPS) && !defined(BB_EMU)
#define BB_STAMP(cx, S, i) do { if (threadIdx.x == 0) (S).stamps[(long long)(cx).block * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BB_STAMP(cx, S, i) ((void)0)
#endif
// wall-clock stamp (s_memrealtime: one 100 MHz counter for the whole device -- s_memtime counters are not comparable across CUs)
#if defined(BB_STAMPS) && !defined(BB_EMU)
#define BB_STAMP_RT(cx, S, i) do { if (threadIdx.x == 0) (S).stamps[(long long)(cx).block * 32 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define BB_STAMP_RT(cx, S, i) ((void)0)
#endif
// wave-0 timeline stamp after draining this wave's outstanding memory operations
#if defined(BB_STAMPS) && !defined(BB_EMU)
#define BB_STAMP_W(cx, S, i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); if (threadIdx.x == 0) (S).stamps[(long long)(cx).block * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BB_STAMP_W(cx, S, i) ((void)0)
#endif

// per-wave stamps (diagnostic build only): event ev in 0..3 of every wave, behind the [tiles + 8][32] block stamps
#if defined(BB_STAMPS) && !defined(BB_EMU)
#define BB_STAMP_WAVE(cx, S, A, ev) do { if ((threadIdx.x & 63) == 0) (S).stamps[((long long)(A).nblk_alloc + 8) * 32 + (long long)(cx).block * 64 + (ev) * 16 + (threadIdx.x >> 6)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BB_STAMP_WAVE(cx, S, A, ev) ((void)0)
#endif

#include "bb_math.h"

#define BB_STREAM_INIT_MU 0xFFFFFFFFu
#define BB_STREAM_INIT_OMEGA 0xFFFFFFFEu
#define BB_LOG2PI 1.8378770664093454835606594728112

struct alignas(16) bb_d2 { double x, y; };

// Loads / stores of words another workgroup of the SAME launch writes / reads (persistent kernel):
// agent-scope relaxed atomics = global_load/store ... sc1 (write-through, L1-bypassing); plain otherwise.
template <bool COH> BB_DEV double bb_ld(const double* p) {
#if !defined(BB_EMU)
    if (COH) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    return *p;
}
template <bool COH> BB_DEV void bb_st(double* p, double v) {
#if !defined(BB_EMU)
    if (COH) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); return; }
#endif
    *p = v;
}
#define BB_MAX_SEG (4 + 4 * BB_MAX_REP)

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
    double r = bb_sqrt(-2.0 * bb_log(u1));
    double s, c;
    bb_sincospi_02(2.0 * u2, &s, &c);
    *n0 = r * c;
    *n1 = r * s;
}

BB_DEV void bb_softplus_sigmoid(double om, double* sp, double* sig) { bb_softplus_sigmoid_fast(om, sp, sig); }

BB_DEV void bb_prior_of(const DevModel& M, int blk, long long j, double* mean, double* inv_var) {
    const DevPrior& p = M.pri[blk];
    if (p.mean_e) { *mean = p.mean_e[j]; *inv_var = p.inv_var_e[j]; }
    else { *mean = p.mean; *inv_var = p.inv_var; }
}

// ------------------------------------------------------------------------------------------------
// LDS carve-up (offsets in doubles) for a tile of NB barcodes worked by nthr threads.
// ------------------------------------------------------------------------------------------------
struct BBLds {
    int zl, zs0, zs1, zs2, zs3, seff, weff, res, As, Qs, acc, wk, Lt, invS, cc, GG, wbar, gglob, misc, part, red, Dt, elbt, seg, zgl, lam;
    int total;
    int acc_cap;    // doubles of staging at `acc` the cross-GPU consume may use (bbp_consume<true>)
    int tmap;       // k_res: [K] ints, the time point (index into Lt / invS) whose normaliser S_t row entry k is, or -1 (0: no table, bb_put_total searches)
};

template <int KIND> BB_DEV int bb_xdim(const DevModel& M) { return KIND == 1 ? M.E : (KIND == 4 ? M.E * M.R : M.R); }

static inline
#ifndef BB_EMU
__host__ __device__
#endif
BBLds bb_lds_layout(int R, int E, int kind, int Ttot, int nt1, int K, int NB, int nthr, int with_lam = 0) {
    BBLds L;
    int X = (kind == 1) ? E : (kind == 4 ? E * R : R);
    int o = 0;
    L.zl = o;   o += NB * Ttot;
    L.zs0 = o;  o += NB * X;
    L.zs1 = o;  o += NB * X;
    L.zs2 = o;  o += NB * X;
    L.zs3 = o;  o += NB * (kind == 4 ? E : 1);
    L.seff = o; o += NB * X;
    L.weff = o; o += NB * X;
    L.res = 0;            // (set below: the a_tb table of the resident launch)
    L.As = o;   o += NB * X;
    L.Qs = o;   o += NB * X;
    L.acc = o;  o += (BB_NQ + 1) * nthr;
    L.wk = o;   o += K + 2 * nt1;
    L.Lt = o;   o += Ttot;
    L.invS = o; o += Ttot;
    L.cc = o;   o += Ttot;
    L.GG = o;   o += Ttot;
    L.wbar = o; o += Ttot;
    L.gglob = o; o += 2 * nt1;
    L.misc = o; o += 16 + BB_MAX_REP;
    L.part = o; o += 32;
    L.red = o;  o += 16 * K + 8 * BB_NQ * Ttot;   // partial sums of the two-level reductions
    L.Dt = o;   o += Ttot;
    L.elbt = o; o += Ttot;
    L.seg = o;  o += 5 * (BB_MAX_SEG + 1);
    L.zgl = o;  o += 2 * nt1;
    L.lam = o;  o += with_lam ? NB * Ttot : 0;   // exp(loglambda sample) of the tile (resident launch only)
    L.res = o;  o += with_lam ? NB * Ttot : 0;   // a_tb = (l[t+1] - l[t]) - s_eff, filled in the exchange's shadow (resident launch only)
    L.total = (o + 1) & ~1;
    L.acc_cap = (BB_NQ + 1) * nthr;
    L.tmap = 0;
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

BB_HD BBTile bb_tile(const DevModel& M, const RunArgs& A, int block, int NB) {
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

// Ragged replicate method (model_fitness_normal_hierarchical_replicates.jl:596-610): the neutral data vector is
// time-fastest (:549) but its mean / variance vectors are `repeat(s_t[range], inner=n_neutral)` (:599-605), so
// element e = t + (T-1) b is paired with population index j = e div n_neutral.
BB_DEV int bb_qj(const DevModel& M, int T1, int t, long long b) { return (int)(((long long)t + (long long)T1 * b) / M.nn); }
// neutrals b in [lo, hi) pair (t, j):  j nn <= t + T1 b < (j + 1) nn
BB_DEV void bb_qrange(const DevModel& M, int T1, int t, int j, long long* lo, long long* hi) {
    const long long a = (long long)j * M.nn - t, b = (long long)(j + 1) * M.nn - t;
    long long l = a <= 0 ? 0 : (a + T1 - 1) / T1, h = b <= 0 ? 0 : (b + T1 - 1) / T1;
    *lo = l > M.nn ? M.nn : l;
    *hi = h > M.nn ? M.nn : h;
}

// which per-unit slot time step t of replicate r uses (environment of t+1 / replicate / 0)
template <int KIND> BB_DEV int bb_xof(const DevModel& M, int r, int t) {
    return KIND == 1 ? M.env_idx[t + 1] : (KIND == 3 ? r : (KIND == 4 ? r * M.E + M.env_idx[M.tcum[r] + t + 1] : 0));
}

// ------------------------------------------------------------------------------------------------
// Segment table: the latents of a tile are a handful of contiguous ranges of the flat vector
// (loglambda slab per replicate, per-mutant blocks, the replicated global blocks on block 0).
// All per-latent work runs as ONE flat loop over "pairs" (2q, 2q+1) of those ranges -- a pair shares
// one Philox call and moves as 16-byte accesses -- so that every thread gets the same share.
// ------------------------------------------------------------------------------------------------
enum BBSegKind {
    SK_L = 0,      // loglambda of replicate r             -> zl
    SK_S,          // s_bc (fitness / multienv)            -> zs0
    SK_LS_E,       // logsigma_bc (fitness / multienv)     -> zs1
    SK_TT_G, SK_LT_G, SK_LS_G,            // genotype: theta_tilde -> zs0, logtau -> zs1, logsigma_bc -> zs2
    SK_TH_R, SK_TT_R, SK_LT_R, SK_LS_R,   // replicate: theta -> zs3, per-replicate tt / lt / ls -> zs0/1/2 + r NB
    SK_GS, SK_GLS  // global s_pop / logsigma_pop          -> zg (global memory)
};
struct BBSeg { long long lo, hi; int pbeg, blk, kind, ldsoff, r, pad; };   // 40 bytes = 5 doubles; pad = bb_hdelta of the segment
// a segment's latents in a row of the TruncatedADAGrad window: entry = flat index - delta (DevModel.Dh)
BB_HD long long bb_hdelta(const DevModel& M, int blk, int r) { return blk == BK_L ? M.hdl[r] : M.hd0[blk] + (long long)r * M.hd1[blk]; }

BB_HD int bb_seg_pairs(long long lo, long long hi) { return hi > lo ? (int)(((hi - 1) >> 1) - (lo >> 1) + 1) : 0; }

// Built by one thread; returns the number of segments, sg[n].pbeg = total pairs.
template <int KIND>
BB_DEV int bb_build_segs(BBSeg* sg, const DevModel& M, const BBLds& L, const BBTile& t, bool globals) {
    int n = 0, p = 0;
    auto add = [&](int blk, int kind, long long lo, long long cnt, int ldsoff, int r) {
        if (cnt <= 0) return;
        BBSeg s; s.lo = lo; s.hi = lo + cnt; s.pbeg = p; s.blk = blk; s.kind = kind; s.ldsoff = ldsoff; s.r = r; s.pad = (int)bb_hdelta(M, blk, r);
        sg[n++] = s;
        p += bb_seg_pairs(lo, lo + cnt);
    };
    for (int r = 0; r < M.R; ++r)
        add(BK_L, SK_L, M.off_l[r] + t.b0 * M.T[r], (long long)t.nbt * M.T[r], L.zl + t.NB * M.tcum[r], r);
    if (t.nmt > 0) {
        if (KIND == 0 || KIND == 1) {
            const int E = KIND == 1 ? M.E : 1;
            add(BK_S, SK_S, M.blk_lo[BK_S] + t.m0 * E, (long long)t.nmt * E, L.zs0, 0);
            add(BK_LS, SK_LS_E, M.blk_lo[BK_LS] + t.m0 * E, (long long)t.nmt * E, L.zs1, 0);
        } else if (KIND == 2) {
            add(BK_TT, SK_TT_G, M.blk_lo[BK_TT] + t.m0, t.nmt, L.zs0, 0);
            add(BK_LT, SK_LT_G, M.blk_lo[BK_LT] + t.m0, t.nmt, L.zs1, 0);
            add(BK_LS, SK_LS_G, M.blk_lo[BK_LS] + t.m0, t.nmt, L.zs2, 0);
        } else {   // replicate (E_ = 1) and multienv_replicate: theta[e, m]; per replicate tt / lt / ls [e, m, r], env fastest
            const int E_ = KIND == 4 ? M.E : 1;
            add(BK_S, SK_TH_R, M.blk_lo[BK_S] + t.m0 * E_, (long long)t.nmt * E_, L.zs3, 0);
            for (int r = 0; r < M.R; ++r) {
                const long long o = ((long long)r * M.nb + t.m0) * E_;
                add(BK_TT, SK_TT_R, M.blk_lo[BK_TT] + o, (long long)t.nmt * E_, L.zs0 + r * t.NB * E_, r);
                add(BK_LT, SK_LT_R, M.blk_lo[BK_LT] + o, (long long)t.nmt * E_, L.zs1 + r * t.NB * E_, r);
                add(BK_LS, SK_LS_R, M.blk_lo[BK_LS] + o, (long long)t.nmt * E_, L.zs2 + r * t.NB * E_, r);
            }
        }
    }
    if (globals) {
        add(BK_SPOP, SK_GS, M.blk_lo[BK_SPOP], M.blk_hi[BK_SPOP] - M.blk_lo[BK_SPOP], 0, 0);
        add(BK_LSPOP, SK_GLS, M.blk_lo[BK_LSPOP], M.blk_hi[BK_LSPOP] - M.blk_lo[BK_LSPOP], 0, 0);
    }
    sg[n].pbeg = p;
    return n;
}

// f(seg, i0, a0, a1) for every pair of the tile owned by this thread.
template <class F>
BB_DEV void bb_for_pairs(const BBCtx& cx, int tid, const BBSeg* sg, int nseg, F f) {
    const int ptotal = sg[nseg].pbeg;
    int si = 0;
    for (int p = tid; p < ptotal; p += cx.nthr) {
        while (si + 1 < nseg && p >= sg[si + 1].pbeg) ++si;
        const BBSeg s = sg[si];
        const long long i0 = 2 * ((s.lo >> 1) + (p - s.pbeg));
        f(s, i0, i0 >= s.lo, i0 + 1 < s.hi);
    }
}

// Draw one pair: eps from Philox (or the caller's buffer), z = mu + softplus(omega) eps; saves what the
// update sweep needs -- z, a = eps * sigmoid(omega) (= dz/domega) and h = sigmoid/softplus (= dH/domega).
// Returns the pair's ELBO terms (prior quadratic + log sigma) when asked.
BB_DEV double bb_sample_pair(const DevModel& M, const DevState& S, const RunArgs& A, unsigned step, int blk,
                             long long i0, bool a0, bool a1, bool want_elbo, double* z0, double* z1) {
    const long long i1 = i0 + 1;
    double e0, e1;
    if (S.eps_in) {
        e0 = a0 ? S.eps_in[(long long)A.sample * M.D + i0] : 0.0;
        e1 = a1 ? S.eps_in[(long long)A.sample * M.D + i1] : 0.0;
    } else {
        bb_normal_pair(A.seed, (unsigned long long)(i0 >> 1), step, (unsigned)A.sample, &e0, &e1);
    }
    double mu0 = 0, mu1 = 0, om0 = 0, om1 = 0;
    if (a0 && a1) {
        const bb_d2 m = *(const bb_d2*)(S.mu + i0), o = *(const bb_d2*)(S.om + i0);
        mu0 = m.x; mu1 = m.y; om0 = o.x; om1 = o.y;
    } else if (a0) { mu0 = S.mu[i0]; om0 = S.om[i0]; }
    else { mu1 = S.mu[i1]; om1 = S.om[i1]; }
    double sp0, sg0, sp1, sg1;
    bb_softplus_sigmoid(om0, &sp0, &sg0);
    bb_softplus_sigmoid(om1, &sp1, &sg1);
    *z0 = fma(sp0, e0, mu0);
    *z1 = fma(sp1, e1, mu1);
    const double h0 = sg0 * bb_rcp(sp0), h1 = sg1 * bb_rcp(sp1);
    if (a0 && a1) {
        *(bb_d2*)(S.zsv + i0) = bb_d2{*z0, *z1};
        *(bb_d2*)(S.asv + i0) = bb_d2{e0 * sg0, e1 * sg1};
        *(bb_d2*)(S.hsv + i0) = bb_d2{h0, h1};
    } else if (a0) { S.zsv[i0] = *z0; S.asv[i0] = e0 * sg0; S.hsv[i0] = h0; }
    else { S.zsv[i1] = *z1; S.asv[i1] = e1 * sg1; S.hsv[i1] = h1; }
    double el = 0.0;
    if (want_elbo) {
        const long long blo = M.blk_lo[blk];
        double pm, iv;
        if (a0) { bb_prior_of(M, blk, i0 - blo, &pm, &iv); el += -0.5 * (*z0 - pm) * (*z0 - pm) * iv + bb_log(sp0); }
        if (a1) { bb_prior_of(M, blk, i1 - blo, &pm, &iv); el += -0.5 * (*z1 - pm) * (*z1 - pm) * iv + bb_log(sp1); }
    }
    return el;
}

// Where a step lands in the TruncatedADAGrad window: computed ONCE per sweep (64-bit modulo on the CU's one
// scalar unit is expensive when every wave repeats it per parameter).
struct BBSlot { int slot; bool resum; };
BB_DEV BBSlot bb_slot_of(const RunArgs& A, unsigned long long step) {
    BBSlot w;
    w.slot = A.opt == 0 ? (int)(step % (unsigned long long)A.W) : 0;
    // exact re-add of the window: every step (resum_every == 1: the reference's arithmetic), every resum_every steps, or
    // or (resum_every == 0, the default) never: the running sum is a compensated one (bb_opt_apply) and needs no re-adds -- each
    // reads the whole window (32 D W bytes; C2 with W = 100: 0.8 GB, ~30 steps' worth of time)
    w.resum = A.opt == 0 && (A.resum_every == 1 || (A.resum_every > 1 && step > 0 && step % (unsigned long long)A.resum_every == 0));
    return w;
}

// The resident launches carry the window position from step to step instead of dividing a 64-bit step number three times per
// step in every wave (the software division was ~60 vector and ~300 scalar instructions of each step: llvm-objdump of the C2
// instance, round 3): one modulo per launch, then increments -- on scalar registers (the step number is uniform).
struct BBSlotCtr { int slot, rs; bool past0; };
BB_DEV int bb_uniform(int v) {
#if defined(BB_EMU)
    return v;
#else
    return __builtin_amdgcn_readfirstlane(v);
#endif
}
BB_DEV BBSlotCtr bb_slot_init(const RunArgs& A, unsigned long long step) {
    BBSlotCtr c;
    c.slot = A.opt == 0 ? bb_uniform((int)(step % (unsigned long long)A.W)) : 0;
    c.rs = (A.opt == 0 && A.resum_every > 1) ? bb_uniform((int)(step % (unsigned long long)A.resum_every)) : 0;
    c.past0 = bb_uniform(step > 0 ? 1 : 0) != 0;
    return c;
}
BB_DEV BBSlot bb_slot_now(const RunArgs& A, const BBSlotCtr& c) {
    BBSlot w;
    w.slot = c.slot;
    w.resum = A.opt == 0 && (A.resum_every == 1 || (A.resum_every > 1 && c.past0 && c.rs == 0));
    return w;
}
BB_DEV void bb_slot_next(const RunArgs& A, BBSlotCtr& c) {
    if (A.opt == 0) { c.slot = c.slot + 1 == A.W ? 0 : c.slot + 1; if (A.resum_every > 1) c.rs = c.rs + 1 == A.resum_every ? 0 : c.rs + 1; }
    c.past0 = true;
}

// One optimiser update of parameter *p with gradient-of-(-ELBO) d.  which: 0 = mu, 1 = omega.
// (*acc, *lo): the element's compensated running window sum.
// a pair's four low-order parts: [mu 0, omega 0, mu 1, omega 1]
struct bb_f4 { float x, y, z, w; };
BB_DEV bb_f4 bb_load_lo(const DevState& S, long long i0, bool a0, bool a1) {
    bb_f4 v{0.f, 0.f, 0.f, 0.f};
    if (a0) { v.x = S.accl[2 * i0]; v.y = S.accl[2 * i0 + 1]; }
    if (a1) { v.z = S.accl[2 * i0 + 2]; v.w = S.accl[2 * i0 + 3]; }
    return v;
}
BB_DEV void bb_store_lo(const DevState& S, long long i0, bool a0, bool a1, const bb_f4& v) {
    if (a0) { S.accl[2 * i0] = v.x; S.accl[2 * i0 + 1] = v.y; }
    if (a1) { S.accl[2 * i0 + 2] = v.z; S.accl[2 * i0 + 3] = v.w; }
}
// error-free sum: a + b = s + e exactly
BB_DEV void bb_two_sum(double a, double b, double* s, double* e) {
    const double t = a + b, bb = t - a;
    *s = t;
    *e = (a - (t - bb)) + (b - bb);
}
BB_DEV void bb_opt_apply(const DevModel& M, const DevState& S, const RunArgs& A, const BBSlot w,
                         int which, long long ih /* the latent's entry in a window row */, double d, double old_slot, double* new_slot, double* p, double* acc, float* lo) {
    double upd;
    if (A.opt == 0) {   // TruncatedADAGrad: g2[mod(i-1,n)+1] = d^2; s = sum(g2); d *= eta / (tau + sqrt(s))
        const double n2 = d * d;
        *new_slot = n2;
        // Running sum acc + d^2 - (slot leaving the window).  A plain double leaves up to half an ulp of the CURRENT sum behind at
        // every step, and that residue stays while the sum itself falls by orders of magnitude (the first gradients are ~1e6
        // times the later ones): measured on C2, 1500 steps, 0.2 in mu against the exact window with re-adds every ten windows,
        // 5e-5 with one per window early on.  (A per-element "re-add when the sum has fallen 2^16 below its largest value"
        // guard fixes the accuracy -- 8e-9 -- but some wave of the grid trips it at almost every step and the whole lock-stepped
        // grid waits for its 100-slot re-add: 15.3 -> 19.8 us per step.)  So the sum is kept as an unevaluated pair (acc, lo):
        // both updates go through an error-free two-sum, the rounding errors collect in lo (a float: 53 + 24 bits together).
        // What is still lost is lo's own rounding, 2^-77 of the sum per step -- an absolute residue again, 2^24 times smaller
        // than the plain double's: while the sum falls by 1e13 from its first peak over thousands of steps, the worst latent's sum
        // is 3e-11 (relative) off the correctly rounded window sum at 1 000 steps, 7e-8 at 5 000, 3e-8 at 10 000, and the step
        // size eta / (tau + sqrt(s)) moves by at most 1e-8 of itself (tools/window_sum_accuracy.py,
        // profiles/window_sum_accuracy_10k.json).  A double lo would remove that (+4 VGPRs per pair slot of k_res, +16 B per latent
        // and step in k_stream); resum_every = k bounds it.  No re-adds, no divergence, no window traffic beyond the one slot.
        double s;
        if (w.resum) {
            s = 0.0;
            for (int j = 0; j < A.W; ++j) s += (j == w.slot) ? n2 : S.hist[((long long)j * 2 + which) * M.Dh + ih];
            *lo = 0.f;
        } else {
            double t, e1;
            bb_two_sum(*acc, n2, &t, &e1);
            // t >= old_slot (the sum holds that slot's value among its non-negative terms): the fast form of the error-free
            // sum is exact here -- three additions instead of six
            const double u = t - old_slot, e2 = (t - u) - old_slot;
            const double l = (double)*lo + (e1 + e2);
            s = u + l;
            *lo = (float)(l - (s - u));
            s = s > 0.0 ? s : 0.0;            // (an all-zero window can come out as -1e-60)
        }
        *acc = s;
        bb_cdouble* oc = bb_tab(S.optc);
        upd = d * (oc[0] * bb_rcp(oc[1] + bb_sqrt(s)));
    } else {            // DecayedADAGrad: acc = post*acc + pre*d^2; d *= eta / (sqrt(acc) + 1e-8)
        bb_cdouble* oc = bb_tab(S.optc);
        const double a = oc[3] * (*acc) + oc[2] * d * d;
        *acc = a;
        upd = d * (oc[0] * bb_rcp(bb_sqrt(a) + 1e-8));
    }
    *p -= upd;
}

// Finish one pair given the likelihood part of d logjoint / d z for its two latents: prior term,
// reparameterisation gradient w.r.t. (mu, omega), S-sample averaging, entropy term, optimiser.
BB_DEV void bb_update_pair(const DevModel& M, const DevState& S, const RunArgs& A, const BBSlot w, int blk, long long hd,
                           long long i0, bool a0, bool a1, double z0, double z1, double gl0, double gl1) {
    const long long i1 = i0 + 1, blo = M.blk_lo[blk];
    const bool both = a0 && a1;
    double pm, iv;
    double g[2] = {0.0, 0.0};
    if (a0) { bb_prior_of(M, blk, i0 - blo, &pm, &iv); g[0] = gl0 - (z0 - pm) * iv; }
    if (a1) { bb_prior_of(M, blk, i1 - blo, &pm, &iv); g[1] = gl1 - (z1 - pm) * iv; }
    bb_d2 av, hv;
    if (both) { av = *(const bb_d2*)(S.asv + i0); hv = *(const bb_d2*)(S.hsv + i0); }
    else if (a0) { av = bb_d2{S.asv[i0], 0.0}; hv = bb_d2{S.hsv[i0], 0.0}; }
    else { av = bb_d2{0.0, S.asv[i1]}; hv = bb_d2{0.0, S.hsv[i1]}; }
    double gm[2] = {g[0], g[1]}, go[2] = {g[0] * av.x, g[1] * av.y};
    if (A.S > 1) {
        const double invS = 1.0 / (double)A.S;
        for (int k = 0; k < 2; ++k) {
            if (!(k ? a1 : a0)) continue;
            const long long i = i0 + k;
            if (!A.first_sample) { gm[k] += S.gacc_mu[i]; go[k] += S.gacc_om[i]; }
            if (!A.last_sample) { S.gacc_mu[i] = gm[k]; S.gacc_om[i] = go[k]; }
            gm[k] *= invS; go[k] *= invS;
        }
        if (!A.last_sample) return;
    }
    go[0] += hv.x;                                   // d H / d omega
    go[1] += hv.y;
    if (!A.apply) {                                  // export d ELBO / d theta
        if (a0) { S.gacc_mu[i0] = gm[0]; S.gacc_om[i0] = go[0]; }
        if (a1) { S.gacc_mu[i1] = gm[1]; S.gacc_om[i1] = go[1]; }
        return;
    }
    bb_d2 mu, om, am, ao, hm = bb_d2{0, 0}, ho = bb_d2{0, 0};
    double* hs_m = nullptr;
    double* hs_o = nullptr;
    if (A.opt == 0) {
        hs_m = S.hist + ((long long)w.slot * 2 + 0) * M.Dh - hd;          // (hd: where the segment's latents sit in a window row, bb_hdelta)
        hs_o = S.hist + ((long long)w.slot * 2 + 1) * M.Dh - hd;
    }
    if (both) {
        mu = *(const bb_d2*)(S.mu + i0); om = *(const bb_d2*)(S.om + i0);
        am = *(const bb_d2*)(S.acc_mu + i0); ao = *(const bb_d2*)(S.acc_om + i0);
        if (hs_m) { hm = *(const bb_d2*)(hs_m + i0); ho = *(const bb_d2*)(hs_o + i0); }
    } else {
        const long long i = a0 ? i0 : i1;
        const double m_ = S.mu[i], o_ = S.om[i], am_ = S.acc_mu[i], ao_ = S.acc_om[i];
        const double hm_ = hs_m ? hs_m[i] : 0.0, ho_ = hs_o ? hs_o[i] : 0.0;
        mu = a0 ? bb_d2{m_, 0} : bb_d2{0, m_}; om = a0 ? bb_d2{o_, 0} : bb_d2{0, o_};
        am = a0 ? bb_d2{am_, 0} : bb_d2{0, am_}; ao = a0 ? bb_d2{ao_, 0} : bb_d2{0, ao_};
        hm = a0 ? bb_d2{hm_, 0} : bb_d2{0, hm_}; ho = a0 ? bb_d2{ho_, 0} : bb_d2{0, ho_};
    }
    bb_d2 nhm = hm, nho = ho;
    bb_f4 lo = bb_load_lo(S, i0, a0, a1);
    if (a0) {
        bb_opt_apply(M, S, A, w, 0, i0 - hd, -gm[0], hm.x, &nhm.x, &mu.x, &am.x, &lo.x);
        bb_opt_apply(M, S, A, w, 1, i0 - hd, -go[0], ho.x, &nho.x, &om.x, &ao.x, &lo.y);
    }
    if (a1) {
        bb_opt_apply(M, S, A, w, 0, i1 - hd, -gm[1], hm.y, &nhm.y, &mu.y, &am.y, &lo.z);
        bb_opt_apply(M, S, A, w, 1, i1 - hd, -go[1], ho.y, &nho.y, &om.y, &ao.y, &lo.w);
    }
    bb_store_lo(S, i0, a0, a1, lo);
    if (both) {
        *(bb_d2*)(S.mu + i0) = mu; *(bb_d2*)(S.om + i0) = om;
        *(bb_d2*)(S.acc_mu + i0) = am; *(bb_d2*)(S.acc_om + i0) = ao;
        if (hs_m) { *(bb_d2*)(hs_m + i0) = nhm; *(bb_d2*)(hs_o + i0) = nho; }
    } else {
        const long long i = a0 ? i0 : i1;
        S.mu[i] = a0 ? mu.x : mu.y; S.om[i] = a0 ? om.x : om.y;
        S.acc_mu[i] = a0 ? am.x : am.y; S.acc_om[i] = a0 ? ao.x : ao.y;
        if (hs_m) { hs_m[i] = a0 ? nhm.x : nhm.y; hs_o[i] = a0 ? nho.x : nho.y; }
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
//   fitness   s_eff = s_bc                          w = bb_exp(-2 logsigma_bc)      model_fitness_normal.jl:262-270
//   multienv  s_eff[e] = s_bc[e, m]                 w[e] likewise                model_multienv_fitness_normal.jl:293-301
//   genotype  s_eff = theta[geno] + bb_exp(logtau)*tt                               ..._genotypes.jl:230
//   replicate s_eff[r] = theta + bb_exp(logtau_r)*tt_r                              ..._replicates.jl:216
// Returns the thread's ELBO term  - sum_units logsigma_eff * (#time steps using the slot).
template <int KIND>
BB_DEV double bb_effective_tables(BBCtx& cx, int tid, const DevModel& M, const DevState& S, const BBLds& L,
                                  const BBTile& t, bool want_elbo) {
    double* lds = cx.lds;
    const int X = bb_xdim<KIND>(M);
    double el = 0.0;
    for (int u = tid; u < t.nbt * X; u += cx.nthr) {
        const int bl = u / X, x = u - bl * X;
        double se = 0.0, we = 0.0;
        if (bl >= t.nshift) {
            const int ml = bl - t.nshift;
            double ls;
            if (KIND == 0) { se = lds[L.zs0 + ml]; ls = lds[L.zs1 + ml]; }
            else if (KIND == 1) { se = lds[L.zs0 + ml * X + x]; ls = lds[L.zs1 + ml * X + x]; }
            else if (KIND == 2) {
                se = S.ztheta[M.geno_idx[t.m0 + ml]] + bb_exp(lds[L.zs1 + ml]) * lds[L.zs0 + ml];
                ls = lds[L.zs2 + ml];
            } else if (KIND == 3) {
                se = lds[L.zs3 + ml] + bb_exp(lds[L.zs1 + x * t.NB + ml]) * lds[L.zs0 + x * t.NB + ml];
                ls = lds[L.zs2 + x * t.NB + ml];
            } else {   // multienv_replicate: x = r E + e   (model_multienv_..._replicates.jl:241, 311-312)
                const int r = x / M.E, e = x - r * M.E, o = r * t.NB * M.E + ml * M.E + e;
                se = lds[L.zs3 + ml * M.E + e] + bb_exp(lds[L.zs1 + o]) * lds[L.zs0 + o];
                ls = lds[L.zs2 + o];
            }
            we = bb_exp(-2.0 * ls);
            if (want_elbo) {
                int cnt;
                if (KIND == 1) { cnt = 0; for (int tt = 0; tt < M.T[0] - 1; ++tt) cnt += (M.env_idx[tt + 1] == x); }
                else if (KIND == 4) {
                    const int r = x / M.E, e = x - r * M.E;
                    cnt = 0;
                    for (int tt = 0; tt < M.T[r] - 1; ++tt) cnt += (M.env_idx[M.tcum[r] + tt + 1] == e);
                } else cnt = M.T[KIND == 3 ? x : 0] - 1;
                el -= ls * cnt;
            }
        }
        lds[L.seff + u] = se;
        lds[L.weff + u] = we;
    }
    return el;
}

// pass M: per replicate, every (barcode, time) element contributes to the moments of its time; the
// tile's column sums land in lds[L.wk + row].  `we` also accumulates the Poisson ELBO terms.
template <int KIND, bool KEEP_LAM = false>
BB_DEV void bb_pass_moments(BBCtx& cx, const DevModel& M, const DevState& S, const BBLds& L, const BBTile& t, int NB, bool we) {
    double* lds = cx.lds;
    const int X = bb_xdim<KIND>(M);
    for (int r = 0; r < M.R; ++r) {
        const int T = M.T[r];
        const int bstride = cx.nthr / T, nact = bstride * T;
        const double* zl = lds + L.zl + NB * M.tcum[r];
        BB_PASS(cx, tid) {
            double aS = 0, aM0 = 0, aM1 = 0, aM2 = 0, aN1 = 0, aN2 = 0, el = 0;
            if (tid < nact) {
                const int blq = (int)bb_umulhi((unsigned)tid, M.Tmagic[r]);
                const int tt = tid - blq * T;
                const int x = (tt < T - 1) ? bb_xof<KIND>(M, r, tt) : 0;
                for (int bl = blq; bl < t.nbt; bl += bstride) {
                    const double z = zl[bl * T + tt];
                    const double lam = bb_exp(z);
                    if (KEEP_LAM) lds[L.lam + NB * M.tcum[r] + bl * T + tt] = lam;
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
        BB_STAMP(cx, S, 3);
        // column sums per (quantity, time): 8 strided partial sums per column, then one add chain
        BB_PASS(cx, tid) {
            double* tmp = lds + L.red;
            for (int w = tid; w < BB_NQ * T * 8; w += cx.nthr) {
                const int j = w >> 3, c = w & 7;
                const int q = (int)bb_umulhi((unsigned)j, M.Tmagic[r]);
                const int tt = j - q * T;
                const double* row = lds + L.acc + q * cx.nthr + tt;
                double s = 0.0;
                for (int i = c; i < bstride; i += 8) s += row[T * i];
                tmp[w] = s;
            }
        }
        BB_SYNC(cx);
        BB_PASS(cx, tid) {
            const double* tmp = lds + L.red;
            for (int j = tid; j < BB_NQ * T; j += cx.nthr) {
                const int q = (int)bb_umulhi((unsigned)j, M.Tmagic[r]);
                const int tt = j - q * T;
                if (q == 0 || tt < T - 1) {
                    double s = 0.0;
                    for (int c = 0; c < 8; ++c) s += tmp[j * 8 + c];
                    lds[L.wk + (q == 0 ? M.kq[r] + tt : M.kq[r] + T + 5 * tt + (q - 1))] = s;
                }
            }
        }
        BB_SYNC(cx);
    }
    if (KIND == 3 && M.quirk && t.nshift > 0) {
        // ragged-method neutral pairing: sums of d = l[t+1] - l[t] and d^2 over the tile's neutrals of every
        // (t, j) pair, one thread per pair, barcodes in order (deterministic); rows kqa[r] + 2 (t T1 + j) + {0, 1}
        BB_PASS(cx, tid) {
            for (int r = 0; r < M.R; ++r) {
                const int T = M.T[r], T1 = T - 1;
                const double* zl = lds + L.zl + NB * M.tcum[r];
                for (int w = tid; w < T1 * T1; w += cx.nthr) {
                    const int tt = w / T1, j = w - tt * T1;
                    long long lo, hi;
                    bb_qrange(M, T1, tt, j, &lo, &hi);
                    if (lo < t.b0) lo = t.b0;
                    if (hi > t.b0 + t.nshift) hi = t.b0 + t.nshift;
                    double s1 = 0.0, s2 = 0.0;
                    for (long long bb = lo; bb < hi; ++bb) {
                        const int bl = (int)(bb - t.b0);
                        const double d = zl[bl * T + tt + 1] - zl[bl * T + tt];
                        s1 += d; s2 += d * d;
                    }
                    lds[L.wk + M.kqa[r] + 2 * w] = s1;
                    lds[L.wk + M.kqa[r] + 2 * w + 1] = s2;
                }
            }
        }
        BB_SYNC(cx);
    }
}

// ================================================================================================
// block_sample: sampling sweep + partial moments of one tile
// ================================================================================================
template <int KIND>
BB_DEV void bb_block_sample(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, int NB) {
    const BBLds L = bb_lds_layout(M.R, M.E, KIND, M.Ttot, M.nt1, M.K, NB, cx.nthr);
    double* lds = cx.lds;
    const BBTile t = bb_tile(M, A, cx.block, NB);
    const unsigned step = (unsigned)S.ctr[A.par];
    const int X = bb_xdim<KIND>(M);
    const bool we = A.with_elbo != 0;

    BB_STAMP(cx, S, 0);
    BBSeg* sg = (BBSeg*)(lds + L.seg);
    int* li = (int*)(lds + L.misc);
    BB_PASS(cx, tid) {
        if (tid == 0) li[0] = bb_build_segs<KIND>(sg, M, L, t, cx.block == 0);
        for (int k = tid; k < M.K; k += cx.nthr) lds[L.wk + k] = 0.0;
    }
    BB_SYNC(cx);
    // pass S: draw every latent of the tile (block 0 also draws the replicated global latents)
    BB_PASS(cx, tid) {
        double el = 0.0;
        bb_for_pairs(cx, tid, sg, li[0], [&](const BBSeg& s, long long i0, bool a0, bool a1) {
            double z0, z1;
            const double e = bb_sample_pair(M, S, A, step, s.blk, i0, a0, a1, we, &z0, &z1);
            if (s.kind >= SK_GS) {
                double* zg = S.zg + (s.kind == SK_GLS ? M.nt1 : 0);
                if (a0) zg[i0 - s.lo] = z0;
                if (a1) zg[i0 + 1 - s.lo] = z1;
                if (A.count_globals) el += e;
            } else {
                if (a0) lds[s.ldsoff + (i0 - s.lo)] = z0;
                if (a1) lds[s.ldsoff + (i0 + 1 - s.lo)] = z1;
                el += e;
            }
        });
        lds[L.acc + BB_NQ * cx.nthr + tid] = el;
    }
    BB_SYNC(cx);
    BB_STAMP(cx, S, 1);

    // pass E: effective fitness / precision per unit
    BB_PASS(cx, tid) {
        double el = bb_effective_tables<KIND>(cx, tid, M, S, L, t, we);
        lds[L.acc + BB_NQ * cx.nthr + tid] += el;
    }
    BB_SYNC(cx);
    BB_STAMP(cx, S, 2);

    bb_pass_moments<KIND>(cx, M, S, L, t, NB, we);
    BB_STAMP(cx, S, 4);
    if (we) bb_row_sum(cx, lds + L.acc + BB_NQ * cx.nthr, cx.nthr, lds + L.part, lds + L.wk + (M.K - 2));
    BB_STAMP(cx, S, 5);
    BB_PASS(cx, tid) {
        for (int k = tid; k < M.K; k += cx.nthr) S.partials[(long long)k * A.nblk + cx.block] = lds[L.wk + k];
    }
    BB_SYNC(cx);
    BB_STAMP(cx, S, 6);
}

// A finished total: store it; the S_t rows also yield 1/S_t and L_t = log S_t on the spot (saves a pass + barrier).
BB_DEV void bb_put_total(const DevModel& M, const BBLds& L, double* lds, int k, double s) {
    lds[L.wk + k] = s;
    if (L.tmap) {       // (k_res: one LDS read instead of a search through the model record -- a chain of dependent scalar loads per replicate)
        const int j = ((const int*)(lds + L.tmap))[k];
        if (j >= 0) { lds[L.invS + j] = bb_rcp(s); lds[L.Lt + j] = bb_log(s); }
        return;
    }
    for (int r = 0; r < M.R; ++r) {
        const int tt = k - M.kq[r];
        if (tt >= 0 && tt < M.T[r]) {
            lds[L.invS + M.tcum[r] + tt] = bb_rcp(s);
            lds[L.Lt + M.tcum[r] + tt] = bb_log(s);
        }
    }
}

// Fixed-order sum of the moment rows into lds[L.wk]; the sampled global latents into lds[L.zgl].
template <bool COH>
BB_DEV void bb_finalize_sum(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BBLds& L, const double* zg) {
    double* lds = cx.lds;
    // fixed-order sum of each moment row; the sampled global latents come along into LDS (gglob is free until
    // the finish writes the global gradients there)
    if (A.nred > 16) {   // two-level: 16 strided partial sums, then one add chain
        double* tmp = lds + L.red;
        BB_PASS(cx, tid) {
            for (int w = tid; w < M.K * 16; w += cx.nthr) {
                const int k = w >> 4, c = w & 15;
                const double* row = A.red + (long long)k * A.nred;
                double s = 0.0;
                for (int j0 = c; j0 < A.nred; j0 += 128) {   // 8 independent loads in flight, fixed add tree
                    double v[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) v[i] = (j0 + 16 * i < A.nred) ? bb_ld<COH>(row + j0 + 16 * i) : 0.0;
                    s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
                }
                tmp[w] = s;
            }
            for (int j = tid; j < 2 * M.nt1; j += cx.nthr) lds[L.zgl + j] = bb_ld<COH>(zg + j);
        }
        BB_SYNC(cx);
        BB_PASS(cx, tid) {
            for (int k = tid; k < M.K; k += cx.nthr) {
                double s = 0.0;
                for (int c = 0; c < 16; ++c) s += tmp[k * 16 + c];
                bb_put_total(M, L, lds, k, s);
            }
        }
    } else {             // few rows (totals, or the 8 group rows of the persistent launch): one batch per thread
        BB_PASS(cx, tid) {
            for (int k = tid; k < M.K; k += cx.nthr) {
                double v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = i < A.nred ? bb_ld<COH>(A.red + (long long)k * A.nred + i) : 0.0;
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < 16; ++i) s += v[i];
                bb_put_total(M, L, lds, k, s);
            }
            for (int j = tid; j < 2 * M.nt1; j += cx.nthr) lds[L.zgl + j] = bb_ld<COH>(zg + j);
        }
    }
    BB_SYNC(cx);
}

// Everything that depends on the totals in lds[L.wk] and the sampled global latents in lds[L.zgl] (tiny).
template <int KIND>
BB_DEV void bb_finalize_finish(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BBLds& L) {
    double* lds = cx.lds;
    BB_STAMP(cx, S, 9);
    // (1/S_t and L_t were written together with the totals, bb_put_total) one (replicate, time) item per thread
    BB_PASS(cx, tid) {
        for (int j = tid; j < M.Ttot; j += cx.nthr) {
            int r = 0;
            while (r + 1 < M.R && j >= M.tcum[r + 1]) ++r;
            const int tt = j - M.tcum[r], T = M.T[r];
            double Dt = 0.0, elb = 0.0;
            if (tt < T - 1) {
                const double nn = (double)M.nn;
                const double* mm = lds + L.wk + M.kq[r] + T + 5 * tt;
                const double M0 = mm[0], M1 = mm[1], M2 = mm[2], N1 = mm[3], N2 = mm[4];
                const double sbar = lds[L.zgl + M.off_t[r] + tt], ls = lds[L.zgl + M.nt1 + M.off_t[r] + tt];
                const double wb = bb_exp(-2.0 * ls);
                const double c = lds[L.Lt + j + 1] - lds[L.Lt + j] - sbar;
                const double quadM = M2 - 2.0 * c * M1 + c * c * M0;
                lds[L.cc + j] = c;
                lds[L.wbar + j] = wb;
                if (!(KIND == 3 && M.quirk)) {
                    const double quadN = N2 - 2.0 * c * N1 + nn * c * c;
                    Dt = (M1 - c * M0) + wb * (N1 - c * nn);
                    lds[L.gglob + M.off_t[r] + tt] = -Dt;
                    lds[L.gglob + M.nt1 + M.off_t[r] + tt] = wb * quadN - nn;
                    elb = -0.5 * (quadM + wb * quadN) - nn * ls;
                } else {
                    // ragged method: this thread is time step tt for D_t (neutral element (tt, b) pairs index jq) and
                    // population index tt for the global gradients (pairs (t', tt) over all t')
                    const int T1 = T - 1;
                    const double cL = lds[L.Lt + j + 1] - lds[L.Lt + j];
                    double Dn = 0.0, gs = 0.0, gls = 0.0, en = 0.0;
                    for (int q = 0; q < T1; ++q) {
                        long long lo, hi;
                        {   // D_t: neutrals at time tt paired with index q
                            bb_qrange(M, T1, tt, q, &lo, &hi);
                            const double n = (double)(hi - lo);
                            const double* aa = lds + L.wk + M.kqa[r] + 2 * (tt * T1 + q);
                            const double sq = lds[L.zgl + M.off_t[r] + q], wq = bb_exp(-2.0 * lds[L.zgl + M.nt1 + M.off_t[r] + q]);
                            Dn += wq * (aa[0] - n * (cL - sq));
                        }
                        {   // gradients of s_pop[tt], logsigma_pop[tt]: neutrals at time q paired with index tt
                            bb_qrange(M, T1, q, tt, &lo, &hi);
                            const double n = (double)(hi - lo);
                            const double* aa = lds + L.wk + M.kqa[r] + 2 * (q * T1 + tt);
                            const double cq = (lds[L.Lt + M.tcum[r] + q + 1] - lds[L.Lt + M.tcum[r] + q]) - sbar;
                            const double R1 = aa[0] - n * cq, R2 = aa[1] - 2.0 * cq * aa[0] + n * cq * cq;
                            gs -= wb * R1;
                            gls += wb * R2 - n;
                            en += -0.5 * wb * R2 - n * ls;
                        }
                    }
                    Dt = (M1 - c * M0) + Dn;
                    lds[L.gglob + M.off_t[r] + tt] = -(M1 - c * M0) + gs;
                    lds[L.gglob + M.nt1 + M.off_t[r] + tt] = gls;
                    elb = -0.5 * quadM + en;
                }
            }
            lds[L.Dt + j] = Dt;
            lds[L.elbt + j] = elb;
        }
    }
    BB_SYNC(cx);
    // G_t = D_{t-1} - D_t is formed where it is used (bb_glik) from the D table (D == 0 at t == T-1): no third pass
    if (A.with_elbo) {
        BB_PASS(cx, tid) {
            if (tid < M.R) {
                double e = 0.0;
                for (int tt = 0; tt < M.T[tid] - 1; ++tt) e += lds[L.elbt + M.tcum[tid] + tt];
                lds[L.misc + 16 + tid] = e;
            }
        }
        BB_SYNC(cx);
    }
}

// Sum the moment rows and finish everything that depends on them.
template <int KIND, bool COH>
BB_DEV void bb_finalize(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BBLds& L, const double* zg) {
    bb_finalize_sum<COH>(cx, M, S, A, L, zg);
    bb_finalize_finish<KIND>(cx, M, S, A, L);
}

// Residual r = (l[t+1] - l[t]) - s_eff - c_t of (barcode bl, time step tt) of replicate r, formed from the staged
// samples where it is needed (no residual table, no separate pass).
// TABLE (resident launch): the part that does not depend on the totals, a = (l[t+1] - l[t]) - s_eff, was left in
// lds[L.res + NB tcum[r] + bl T + tt] while the moment rows were in flight (bbp_residual_ahead).
template <int KIND, bool TABLE = false>
BB_DEV double bb_residual(const double* lds, const DevModel& M, const BBLds& L, const BBTile& t, int NB, int X, int r, int bl, int tt) {
    const int T = M.T[r], tc = M.tcum[r];
    if (TABLE && !(KIND == 3 && M.quirk)) return lds[L.res + NB * tc + bl * T + tt] - lds[L.cc + tc + tt];
    const double* zl = lds + L.zl + NB * tc + bl * T + tt;
    double a = zl[1] - zl[0];
    if (bl >= t.nshift) a -= lds[L.seff + bl * X + bb_xof<KIND>(M, r, tt)];
    else if (KIND == 3 && M.quirk)   // ragged method: residual against -s_pop[jq] instead of -s_pop[tt]
        a += lds[L.zgl + M.off_t[r] + bb_qj(M, T - 1, tt, t.b0 + bl)] - lds[L.zgl + M.off_t[r] + tt];
    return a - lds[L.cc + tc + tt];
}

// pass U: the per-unit sums As = sum_t w r (= dlogp/ds_eff) and Qs = sum_t (w r^2 - 1) (= dlogp/dlogsigma_eff).
template <int KIND, bool TABLE = false>
BB_DEV void bb_pass_residuals_units(BBCtx& cx, const DevModel& M, const DevState& S, const BBLds& L, const BBTile& t, int NB) {
    double* lds = cx.lds;
    const int X = bb_xdim<KIND>(M);
    BB_STAMP(cx, S, 13);

    // pass U: per-unit sums  As = sum_t w r  (= dlogp/ds_eff),  Qs = sum_t (w r^2 - 1)  (= dlogp/dlogsigma_eff)
    BB_PASS(cx, tid) {
        for (int u = tid; u < t.nbt * X; u += cx.nthr) {
            const int bl = u / X, x = u - bl * X;
            double as = 0.0, qs = 0.0;
            if (bl >= t.nshift) {
                const double w = lds[L.weff + u];
                const int r = KIND == 3 ? x : (KIND == 4 ? x / M.E : 0);
                const int T1 = M.T[r] - 1;
                for (int tt = 0; tt < T1; ++tt) {
                    if (KIND == 1 && M.env_idx[tt + 1] != x) continue;
                    if (KIND == 4 && M.env_idx[M.tcum[r] + tt + 1] != x - r * M.E) continue;
                    const double rr = bb_residual<KIND, TABLE>(lds, M, L, t, NB, X, r, bl, tt);
                    as += w * rr;
                    qs += w * rr * rr - 1.0;
                }
                if (KIND == 2) S.ds[t.m0 + (bl - t.nshift)] = as;
            }
            lds[L.As + u] = as;
            lds[L.Qs + u] = qs;
        }
    }
    BB_SYNC(cx);
    BB_STAMP(cx, S, 14);

}

// Likelihood part of d logjoint / d z for latent j of segment s (z = its sample), gathered from the LDS tables.
template <int KIND, bool LAM_IN_LDS>
BB_DEV double bb_glik(const double* lds, const DevModel& M, const BBLds& L, const BBTile& t, int NB, const BBSeg& s,
                      long long j, double z, double cnt_cached = 0.0) {
    const int ns = t.nshift;
    const int X = bb_xdim<KIND>(M);
    switch (s.kind) {
    case SK_L: {
        const int r = s.r, T = M.T[r], T1 = T - 1, tc = M.tcum[r];
        const int bl = (int)bb_umulhi((unsigned)j, M.Tmagic[r]), tt = (int)j - bl * T;
        const bool mut = bl >= ns;
        const double lam = LAM_IN_LDS ? lds[L.lam + NB * tc + j] : bb_exp(z);   // the resident launch keeps the moments pass's table
        // (the resident launch keeps its pairs' counts in LDS: they never change, and the load's latency sat in the G pass)
        const double cnt = LAM_IN_LDS ? cnt_cached : (double)M.counts[M.cnt_off[r] + t.b0 * T + j];
        const double Gt = (tt > 0 ? lds[L.Dt + tc + tt - 1] : 0.0) - lds[L.Dt + tc + tt];
        double g = cnt - lam + lam * lds[L.invS + tc + tt] * Gt;
        const bool qk = KIND == 3 && M.quirk && !mut;
        if (tt < T1) {
            const double w = mut ? lds[L.weff + bl * X + bb_xof<KIND>(M, r, tt)]
                                 : lds[L.wbar + tc + (qk ? bb_qj(M, T1, tt, t.b0 + bl) : tt)];
            g += w * bb_residual<KIND, LAM_IN_LDS>(lds, M, L, t, NB, X, r, bl, tt);
        }
        if (tt > 0) {
            const double w = mut ? lds[L.weff + bl * X + bb_xof<KIND>(M, r, tt - 1)]
                                 : lds[L.wbar + tc + (qk ? bb_qj(M, T1, tt - 1, t.b0 + bl) : tt - 1)];
            g -= w * bb_residual<KIND, LAM_IN_LDS>(lds, M, L, t, NB, X, r, bl, tt - 1);
        }
        return g;
    }
    case SK_S: return lds[L.As + ns * X + j];
    case SK_LS_E: return lds[L.Qs + ns * X + j];
    case SK_TT_G: return lds[L.As + ns + j] * bb_exp(lds[L.zs1 + j]);
    case SK_LT_G: return lds[L.As + ns + j] * bb_exp(z) * lds[L.zs0 + j];
    case SK_LS_G: return lds[L.Qs + ns + j];
    case SK_TH_R: case SK_TT_R: case SK_LT_R: case SK_LS_R: {
        // j = ml E_ + e inside the segment (E_ = 1 for the replicate model); unit slot (ns + ml) X + r E_ + e
        const int E_ = KIND == 4 ? M.E : 1;
        const int ml = (int)j / E_, e = (int)j - ml * E_;
        if (s.kind == SK_TH_R) {
            double a = 0.0;
            for (int r = 0; r < M.R; ++r) a += lds[L.As + (ns + ml) * X + r * E_ + e];
            return a;
        }
        const int slot = (ns + ml) * X + s.r * E_ + e, o = s.r * NB * E_ + (int)j;
        if (s.kind == SK_TT_R) return lds[L.As + slot] * bb_exp(lds[L.zs1 + o]);
        if (s.kind == SK_LT_R) return lds[L.As + slot] * bb_exp(z) * lds[L.zs0 + o];
        return lds[L.Qs + slot];
    }
    case SK_GS: return lds[L.gglob + j];
    default: return lds[L.gglob + M.nt1 + j];
    }
}

// ================================================================================================
// block_update: gradient of the log-joint for the tile's latents + optimiser step
// ================================================================================================
template <int KIND>
BB_DEV void bb_block_update(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, int NB) {
    const BBLds L = bb_lds_layout(M.R, M.E, KIND, M.Ttot, M.nt1, M.K, NB, cx.nthr);
    double* lds = cx.lds;
    const BBTile t = bb_tile(M, A, cx.block, NB);
    const unsigned long long step = S.ctr[A.par];
    const int X = bb_xdim<KIND>(M);

    BB_STAMP(cx, S, 8);
    bb_finalize<KIND, false>(cx, M, S, A, L, S.zg);
    BB_STAMP(cx, S, 10);

    BBSeg* sg = (BBSeg*)(lds + L.seg);
    int* li = (int*)(lds + L.misc);
    BB_PASS(cx, tid) {
        if (tid == 0) li[0] = bb_build_segs<KIND>(sg, M, L, t, cx.block == 0);
    }
    BB_SYNC(cx);
    // stage: the saved draw of every tile latent back into LDS
    BB_PASS(cx, tid) {
        bb_for_pairs(cx, tid, sg, li[0], [&](const BBSeg& s, long long i0, bool a0, bool a1) {
            if (s.kind >= SK_GS) return;
            if (a0 && a1) {
                const bb_d2 z = *(const bb_d2*)(S.zsv + i0);
                lds[s.ldsoff + (i0 - s.lo)] = z.x;
                lds[s.ldsoff + (i0 + 1 - s.lo)] = z.y;
            } else if (a0) lds[s.ldsoff + (i0 - s.lo)] = S.zsv[i0];
            else lds[s.ldsoff + (i0 + 1 - s.lo)] = S.zsv[i0 + 1];
        });
    }
    BB_SYNC(cx);
    BB_STAMP(cx, S, 11);
    BB_PASS(cx, tid) { bb_effective_tables<KIND>(cx, tid, M, S, L, t, false); }
    BB_SYNC(cx);
    BB_STAMP(cx, S, 12);

    bb_pass_residuals_units<KIND>(cx, M, S, L, t, NB);
    // pass G: gather each latent's likelihood gradient from the LDS tables and update it
    BB_PASS(cx, tid) {
        const BBSlot wslot = bb_slot_of(A, step);
        auto glik = [&](const BBSeg& s, long long j, double z) -> double { return bb_glik<KIND, false>(lds, M, L, t, NB, s, j, z); };
        bb_for_pairs(cx, tid, sg, li[0], [&](const BBSeg& s, long long i0, bool a0, bool a1) {
            double z0 = 0.0, z1 = 0.0;
            if (a0 && a1) { const bb_d2 z = *(const bb_d2*)(S.zsv + i0); z0 = z.x; z1 = z.y; }
            else if (a0) z0 = S.zsv[i0];
            else z1 = S.zsv[i0 + 1];
            const double g0 = a0 ? glik(s, i0 - s.lo, z0) : 0.0;
            const double g1 = a1 ? glik(s, i0 + 1 - s.lo, z1) : 0.0;
            bb_update_pair(M, S, A, wslot, s.blk, s.pad, i0, a0, a1, z0, z1, g0, g1);
        });
    }
    BB_SYNC(cx);
    BB_STAMP(cx, S, 15);

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
            if (A.last_sample && A.apply) {
                S.ctr[1 - A.par] = step + 1;
                // divergence flag: a NaN / Inf anywhere in theta reaches the exchanged moment totals
                double chk = 0.0;
                for (int k = 0; k < M.K - 2; ++k) chk += lds[L.wk + k];
                if (!(chk - chk == 0.0)) S.hstatus[1] = 1u;
            }
        }
    }
    BB_SYNC(cx);
    BB_STAMP(cx, S, 16);
}

// ================================================================================================
// genotype model: the per-genotype theta block (sampled / updated by a grid over genotypes)
// ================================================================================================
// update (optional) then sample theta; gsum[g] = sum of ds over the genotype's mutants.
BB_DEV void bb_block_geno(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, int nblocks,
                          int do_update, int do_sample, int upd_par) {
    double* lds = cx.lds;
    const long long lo = M.blk_lo[BK_S], hi = M.blk_hi[BK_S];
    const int npairs = bb_seg_pairs(lo, hi);
    if (do_update) {
        const unsigned long long step = S.ctr[upd_par];
        const BBSlot wslot = bb_slot_of(A, step);
        BB_PASS(cx, tid) {
            for (long long p = (long long)cx.block * cx.nthr + tid; p < npairs; p += (long long)nblocks * cx.nthr) {
                const long long i0 = 2 * ((lo >> 1) + p);
                const bool a0 = i0 >= lo, a1 = i0 + 1 < hi;
                const double z0 = a0 ? S.zsv[i0] : 0.0, z1 = a1 ? S.zsv[i0 + 1] : 0.0;
                bb_update_pair(M, S, A, wslot, BK_S, bb_hdelta(M, BK_S, 0), i0, a0, a1, z0, z1, a0 ? S.gsum[i0 - lo] : 0.0, a1 ? S.gsum[i0 + 1 - lo] : 0.0);
            }
        }
        BB_SYNC(cx);
    }
    if (do_sample) {
        const unsigned step = (unsigned)S.ctr[A.par];
        BB_PASS(cx, tid) {
            double el = 0.0;
            for (long long p = (long long)cx.block * cx.nthr + tid; p < npairs; p += (long long)nblocks * cx.nthr) {
                const long long i0 = 2 * ((lo >> 1) + p);
                const bool a0 = i0 >= lo, a1 = i0 + 1 < hi;
                double z0, z1;
                el += bb_sample_pair(M, S, A, step, BK_S, i0, a0, a1, A.with_elbo != 0, &z0, &z1);
                if (a0) S.ztheta[i0 - lo] = z0;
                if (a1) S.ztheta[i0 + 1 - lo] = z1;
            }
            lds[tid] = el;
        }
        BB_SYNC(cx);
        bb_row_sum(cx, lds, cx.nthr, lds + cx.nthr, lds + cx.nthr + 16);
        BB_PASS(cx, tid) { if (tid == 0) S.geno_el[cx.block] = A.count_globals ? lds[cx.nthr + 16] : 0.0; }
        BB_SYNC(cx);
    }
}

// theta rows of the genotype model (mu, omega, the two accumulators and their low-order parts, 2 W window rows) <-> a packed buffer [rows][G]; packing
// writes zeros for the genotypes outside [g_lo, g_hi), so that the sum over all shards' buffers is the gather from the owners.
BB_DEV void bb_block_theta_pack(BBCtx& cx, const DevModel& M, const DevState& S, double* buf, int g_lo, int g_hi, int W, int unpack, int nblocks) {
    const long long total = (long long)(6 + 2 * W) * M.G;
    BB_PASS(cx, tid) {
        for (long long i = (long long)cx.block * cx.nthr + tid; i < total; i += (long long)nblocks * cx.nthr) {
            const int a = (int)(i / M.G), g = (int)(i - (long long)a * M.G);
            const bool own = g >= g_lo && g < g_hi;
            if (a == 4 || a == 5) {          // the accumulators' low-order parts (floats)
                float* q = S.accl + 2 * (M.blk_lo[BK_S] + g) + (a - 4);
                if (unpack) *q = (float)buf[i];
                else buf[i] = own ? (double)*q : 0.0;
                continue;
            }
            double* p = a == 0 ? S.mu : (a == 1 ? S.om : (a == 2 ? S.acc_mu : (a == 3 ? S.acc_om : S.hist + (long long)(a - 6) * M.Dh)));
            p += M.blk_lo[BK_S] + g - (a < 6 ? 0 : bb_hdelta(M, BK_S, 0));
            if (unpack) *p = buf[i];
            else buf[i] = own ? *p : 0.0;
        }
    }
}

// gsum[g] = sum over the genotype's mutants (CSR order) of ds: deterministic segmented sum.  Eight lanes share a
// genotype (lane c adds members k = c, c + 8, ...), the eight partial sums are added in lane order.
BB_DEV void bb_block_geno_sum(BBCtx& cx, const DevModel& M, const DevState& S, int nblocks, long long m_lo, long long m_hi) {
    double* lds = cx.lds;                       // nthr doubles
    const int per_block = cx.nthr / 8;
    for (long long g0 = (long long)cx.block * per_block; g0 < M.G; g0 += (long long)nblocks * per_block) {
        BB_PASS(cx, tid) {
            const long long g = g0 + (tid >> 3);
            const int c = tid & 7;
            double s = 0.0;
            if (g < M.G) {
                for (int k = M.geno_ptr[g] + c; k < M.geno_ptr[g + 1]; k += 8) {
                    const int m = M.geno_mem[k];
                    if (m >= m_lo && m < m_hi) s += S.ds[m];
                }
            }
            lds[tid] = s;
        }
        BB_SYNC(cx);
        BB_PASS(cx, tid) {
            const long long g = g0 + tid;
            if (tid < per_block && g < M.G) {
                double s = 0.0;
                for (int c = 0; c < 8; ++c) s += lds[tid * 8 + c];
                S.gsum[g] = s;
            }
        }
        BB_SYNC(cx);
    }
}

// partials [K][nblk] -> totals [K] (one block), two-level fixed order: 16 strided partial sums per row, then one
// add chain; row K-1 additionally takes the theta-block ELBO partials.
BB_DEV void bb_block_reduce(BBCtx& cx, const DevModel& M, const DevState& S, int nblk, int ngeno_blocks) {
    double* lds = cx.lds;                       // 16 K doubles
    BB_PASS(cx, tid) {
        for (int w = tid; w < M.K * 16; w += cx.nthr) {
            const int k = w >> 4, c = w & 15;
            const double* row = S.partials + (long long)k * nblk;
            double s = 0.0;
            for (int j0 = c; j0 < nblk; j0 += 128) {
                double v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (j0 + 16 * i < nblk) ? row[j0 + 16 * i] : 0.0;
                s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
            lds[w] = s;
        }
    }
    BB_SYNC(cx);
    BB_PASS(cx, tid) {
        for (int k = tid; k < M.K; k += cx.nthr) {
            double s = 0.0;
            for (int c = 0; c < 16; ++c) s += lds[k * 16 + c];
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
