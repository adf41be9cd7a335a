// bb_stream.h -- the ADVI step loop as one resident launch for tiles whose state does NOT fit the register file.
//
// k_res (bb_resident.h) keeps a tile's variational parameters, optimiser accumulators and draw in registers for the whole run: 44
// registers per pair of latents, at most two to four pair slots per thread.  BASELINE config 5 on ONE GPU (200 000 barcodes x 8, 5 000
// genotypes: 2.19 M latents, 782 barcodes and ~4 300 pairs per tile) needs twice the chip's register file and used to fall back to
// two kernels per step -- 184 B per latent of traffic, six launches per step, 135.8 us (0.20 of the HBM roofline).
//
// k_stream keeps k_res's tile map, segment table, moment algebra and in-launch exchange, and STREAMS the per-pair state: a thread
// walks its P pair slots (p = tid + k NT) ONCE per step (round 4; round 3 walked them twice -- an S pass that read mu, omega and
// sampled, then, after the exchange, a G pass that read everything again: 80.6 us per step on C5, the memory system idle for the
// 37 % of it that S, M and the exchange took).  A step:
//
//   M    the loglambda pairs' moment contributions from what the previous step's G pass left in LDS -- per pair the two forward
//        differences (dm, dn) of ITS sample (thread-private entries, no barrier needed for them) and the units' staged forms
//        (s, w = e^{-2 logsigma}, e^{logtau}, theta) -- summed per thread over its slots (a thread's pairs share their time-pair
//        class: NT is a multiple of the lanes per barcode), then by class over the 16-lane rows of the wave (DPP), one LDS entry
//        per (row, class, value)
//   --   barrier -- row sums, publish, exchange, F pass: k_res's own code (br_row_publish, bbp_*_tg, br_finish)
//   G-L  every loglambda pair slot, ONE pass over its state: mu, omega, accumulators, window slot in (56 B per latent); the step's
//        draw AGAIN (Philox is a pure function of (seed, latent, step): nothing of the sample was kept but the differences), z = mu +
//        sigma eps, the neighbour pairs' z through DPP (the lanes of a barcode are neighbours in a 16-lane row), lambda = e^z,
//        gradient, optimiser, everything out (56 B per latent).  The lanes of a barcode add their residuals r, r^2 up (DPP) and
//        leave the barcode's unit sums As = sum_t r, Qs = sum_t r^2 in LDS -- the unit threads no longer walk z rows, so no z row
//        is staged at all.  THEN, in the same slot, the NEXT step's sample of the updated pair: draw, z' = mu' + sigma' eps', the
//        neighbour's z' by DPP, (dm', dn') to the thread's private LDS entry, lambda' = e^{z'} onto the thread's running sums.
//   --   barrier (As, Qs visible)
//   G-U  the unit pair slots the same way: state in, the draw again, gradient from (As, Qs) and the staged forms of THIS step
//        (buffer step & 1), optimiser, out; then the next step's sample, staged into buffer (step + 1) & 1.  Genotype model: theta
//        after one more barrier (its gradient sums the w As its mutants' theta_tilde threads left in LDS).
//   --   barrier: the step's tables are complete
//
// The S pass's arithmetic now runs inside the pass that is bound by memory, its second read of mu, omega (16 B per latent) is gone,
// and nothing but M, the exchange and F runs with the memory system idle: 112 B per latent and step + the counts (4 B per loglambda
// latent) against 96 + 16 (the accumulators' low-order floats) algorithmic.  LDS: the z rows ([NB][T + 1] doubles, 56 KB on C5)
// became the private differences ([NB][T]); the unit stage tables are double-buffered by step parity but only the forms somebody
// ELSE reads are staged (an owner recomputes its own sample with its draw): fitness / multienv 2 tables (s, w), hierarchical 4
// (theta_tilde, e^logtau, w, theta) -- C5's tile: 141 -> 154 KB.  The launch starts with the first step's sample alone (bs_sample0).
//
// Shapes: all five model kinds, every replicate with the same EVEN number of time points (instances: T = 4, 6, 8 -- a barcode takes a
// power-of-two number of lanes, T = 6: four with the last one idle, so that its lanes sit inside one 16-lane row), flat-index-aligned
// pairs (no AP), not the ragged-method pairing, one GPU, one MC sample per step, no ELBO recording.  Everything else keeps its launch.
#pragma once
#include "bb_resident.h"

// per-thread state of the passes (GPU: registers of the one thread; emulation: an array over the tile's threads, so that the lanes'
// DPP exchanges can be written as "pass A leaves a value, pass B reads the neighbour's")
struct BSG {
    BRSt<1> st;                       // the slot's pair descriptor (br_desc)
    bb_d2 mu, om, am, ao, hm, ho;     // state of the pair and its window slot
    bb_f4 lo;
    bb_d2 e, z, sg, sp;               // the step's draw, sample, sigmoid / softplus of omega
    bb_d2 a, h;                       // eps sigmoid(omega) = dz / domega, sigmoid / softplus = dH / domega
    double g0, g1, zv0, zv1;          // likelihood gradient and sample of the two latents
    double rm, rn, wm;                // loglambda, mutant: residuals of the pair's two forward differences; precision of the unit of the first
    double xa, xq;                    // ... and what of them goes to the unit sums being formed (one environment at a time)
    double lam0, lam1;                // running sums of lambda over the thread's loglambda slots: the NEXT step's S_t contributions
    double el;                        // MS: the thread's ELBO terms of the sample it formed last (summed over its slots; the M pass reduces them)
    double cv[BR_NCV];                // M pass: the thread's moment contributions
    int ok;                           // slot holds a pair of the kind the pass handles
};

// the step's normals of pair i0 (flat-index-aligned): out of line (as br_draw_call) where its temporaries would crowd a slot loop
#ifdef BB_EMU
static inline
#else
__device__ __attribute__((noinline))
#endif
bb_d2 bs_draw(unsigned long long seed, long long i0, unsigned step, unsigned stream = 0u) {
    double a, b;
    bb_normal_pair(seed, (unsigned long long)(i0 >> 1), step, stream, &a, &b);
    return bb_d2{a, b};
}
BB_DEV bb_d2 bs_draw_inline(unsigned long long seed, long long i0, unsigned step, unsigned stream = 0u) {
    double a, b;
    bb_normal_pair(seed, (unsigned long long)(i0 >> 1), step, stream, &a, &b);
    return bb_d2{a, b};
}
// MS instances (several MC samples per step -- Turing.ADVI(samples_per_step, ..), src/vi.jl:98 -- and / or the ELBO trace): what changes
// from sample to sample, all wave-uniform.  Sample s of a step is exchange xc = step NS + s: its parity picks the stage-table / row
// buffers; its gradient joins running sums in memory (DevState.gacc_*: first sample stores, later ones add, the last one averages, adds
// the entropy term and updates); the sample formed at the end of its G passes is (step, stream s + 1), or (step + 1, stream 0) after the last.
struct BSMs {
    int NS, smp, buf;
    bool last, el_next;          // the sample that updates; the NEXT sample (formed in this G pass) records its ELBO terms
    unsigned nstep, nstream;     // step and stream of the next sample's draw
};
BB_DEV BSMs bs_ms_plain(unsigned step) { return BSMs{1, 0, (int)(step & 1u), true, false, step + 1u, 0u}; }
// (inline in the G pass: the pair's state loads stay in flight behind it -- a call drains them first; C5 93.1 -> 91.4 us, round 3)
#ifndef BS_G_INLINE_DRAW
#define BS_G_INLINE_DRAW 1
#endif
// ... and the NEXT step's draw inside the same slot (the fused S part): 1 = inline, 0 = the out-of-line call
#ifndef BS_S_INLINE_DRAW
#define BS_S_INLINE_DRAW 1          /* (C5 63.9 -> 62.6 us) */
#endif
// BS_NT_HIST = 1: the window slot is read with the non-temporal policy (it is not touched again for a whole window; the state arrays,
// re-read every step, keep the Infinity Cache): C5 90.8 -> 83.9 us (round 3)
#ifndef BS_NT_HIST
#define BS_NT_HIST 1
#endif

// ---- lane exchanges: value of the previous / next lane of the 16-lane row, sum over the LPB lanes of a barcode ------------------------
#ifndef BB_EMU
template <int CTRL> __device__ __forceinline__ double bs_dpp_add(double x) { return x + br_dpp<CTRL>(x); }
// sum over the lanes of a 16-lane row with equal (lane % LPB): rotations by 8, 4, .. down to LPB
template <int LPB> __device__ __forceinline__ double bs_class_sum(double x) {
    if (LPB <= 8) x = bs_dpp_add<0x128>(x);
    if (LPB <= 4) x = bs_dpp_add<0x124>(x);
    if (LPB <= 2) x = bs_dpp_add<0x122>(x);
    if (LPB <= 1) x = bs_dpp_add<0x121>(x);
    return x;
}
// sum over the LPB consecutive lanes of a barcode (aligned groups of LPB lanes; every lane gets it): quad_perm [1,0,3,2], [2,3,0,1],
// row_half_mirror -- a pairwise tree, ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7))
template <int LPB> __device__ __forceinline__ double bs_group_sum(double x) {
    if (LPB >= 2) x = bs_dpp_add<0xB1>(x);
    if (LPB >= 4) x = bs_dpp_add<0x4E>(x);
    if (LPB >= 8) x = bs_dpp_add<0x141>(x);
    return x;
}
#define BS_PREV(gv, tid, F) br_dpp<0x111>(BB_PSTATE(gv, tid).F)       /* row_shr:1 -- lane i gets lane i - 1's (first lane of a row: 0) */
#define BS_NEXT(gv, tid, F) br_dpp<0x101>(BB_PSTATE(gv, tid).F)       /* row_shl:1 -- lane i gets lane i + 1's (last lane of a row: 0)  */
#define BS_GROUP_SUM(LPB, gv, tid, F) bs_group_sum<LPB>(BB_PSTATE(gv, tid).F)
#else
#define BS_PREV(gv, tid, F) (((tid) & 15) ? (gv)[(tid) - 1].F : 0.0)
#define BS_NEXT(gv, tid, F) ((((tid) & 15) != 15) ? (gv)[(tid) + 1].F : 0.0)
template <int LPB> static inline double bs_emu_tree(const double* v) {          // the GPU's pairwise tree, same order
    if (LPB == 1) return v[0];
    double t[8];
    int n = LPB;
    for (int i = 0; i < n; ++i) t[i] = v[i];
    while (n > 1) { for (int i = 0; i < n / 2; ++i) t[i] = t[2 * i] + t[2 * i + 1]; n /= 2; }
    return t[0];
}
#define BS_GROUP_SUM(LPB, gv, tid, F) ([&]() { double v_[8]; for (int i_ = 0; i_ < (LPB); ++i_) v_[i_] = (gv)[((tid) & ~((LPB) - 1)) + i_].F; return bs_emu_tree<LPB>(v_); }())
#endif

// BS_ZKEEP = 1: a loglambda pair's sample z' stays in its LDS entry from the pass that formed it (the G pass of the step before) -- the M pass
// takes its differences from there (the next pair's first sample: the neighbour entry, behind the barrier that ends the G passes), and the
// G pass does NOT draw again: with z, mu and softplus(omega) at hand eps = (z - mu) / softplus(omega), which is what d z / d omega =
// eps sigmoid(omega) needs.  One Philox4x32-10 + Box-Muller less per pair and step (205 VALU instructions, 20 of them quarter-rate 64-bit
// multiplies, of ~1 200) in the pass that is bound by instruction issue.  The recovered eps is not the draw bit for bit: its relative
// error is ulp(z) / |sigma eps| -- 1e-16 at the start of a run, 1e-13 .. 1e-12 for a converged loglambda (mu ~ 10, sigma ~ 1e-3) -- in ONE
// term of d ELBO / d omega; the sample z, lambda = e^z and everything else are the S pass's own numbers.  0: the draw again, bit for bit.
#ifndef BS_ZKEEP
#define BS_ZKEEP 1
#endif

// lanes per barcode (br_lpb_stream, as a constant)
template <int TT> BB_DEV constexpr int bs_lpb() { return TT <= 2 ? 1 : (TT <= 4 ? 2 : (TT <= 8 ? 4 : 8)); }
// end of the loglambda segments in the tile's padded thread-index space (one segment per replicate, each starting at a wave boundary;
// they come first): slots below it hold loglambda pairs or padding, slots from it on unit pairs
BB_DEV int bs_lspan(const BRSeg* sg, int nseg) {
    int e = 0;
    for (int i = 0; i < nseg && sg[i].kind == SK_L; ++i) e = sg[i].tbeg + sg[i].span;
    return e;
}

// BS_PF0: between the M pass and the exchange -- the memory system idle until the G passes start -- the lines of the FIRST loglambda slot's
// state are pulled towards the CU: 4 bytes per lane by LDS-DMA into a dump area (no register, nothing waits for them).  1 = the window
// slot's lines (HBM: not touched for a whole window), 2 = all seven arrays.  (The same for the NEXT slot inside the G pass: measured
// slower, 67.1 -> 68.8 / 72.6 us, profiles/r04b_stream_fused -- the pass is bound by the memory system itself.)
#ifndef BS_PF0
#define BS_PF0 0
#endif
template <int KIND, int TT>
BB_DEV void bs_touch0(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BRLay& Y, const BBSlot wslot) {
#ifndef BB_EMU
    if (!BS_PF0) return;
    const BRSeg* sg = (const BRSeg*)(cx.lds + Y.seg);
    const int nseg = ((const int*)(cx.lds + Y.L.misc))[0];
    const int lspan = bs_lspan(sg, nseg), tid = threadIdx.x;
    if (tid >= lspan || tid >= sg[0].tbeg + sg[0].span || TT != 2 * bs_lpb<TT>()) return;
    const long long i0 = sg[0].lo + 2 * (long long)(tid - sg[0].tbeg), ih = i0 - sg[0].pad;
    auto touch = [&](const void* p) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)((float*)(cx.lds + Y.eps) + (tid & ~63)), 4, 0, 0);
    };
    if (A.opt == 0) {
        touch(S.hist + ((long long)wslot.slot * 2 + 0) * M.Dh + ih);
        touch(S.hist + ((long long)wslot.slot * 2 + 1) * M.Dh + ih);
    }
    if (BS_PF0 > 1) { touch(S.mu + i0); touch(S.om + i0); touch(S.acc_mu + i0); touch(S.acc_om + i0); touch(S.accl + 2 * i0); }
#else
    (void)cx; (void)M; (void)S; (void)A; (void)Y; (void)wslot;
#endif
}

// pair slot p of the tile's loglambda segments (lane q = bl LPB + kk of replicate r's segment owns (b, 2 kk), (b, 2 kk + 1) of that
// replicate): what br_desc's loglambda branch gives, without the search through all segments, with the lanes per barcode a constant
template <int KIND, int TT, bool CNT>
BB_DEV void bs_desc_l(const DevModel& M, const BRLay& Y, const BBTile& t, const BRSeg* sg, int p, BRSt<1>& st, const double* lds) {
    constexpr int LPB = bs_lpb<TT>();
    const int R = KIND <= 2 ? 1 : M.R;
    int r = 0;
    if (KIND >= 3) { while (r + 1 < R && p >= sg[r + 1].tbeg) ++r; }
    const int q = p - sg[r].tbeg, bl = q / LPB, kk = q - bl * LPB, t0 = 2 * kk;
    const int E = (KIND == 1 || KIND == 4) ? M.E : 1;
    st.meta[0] = 15;
    st.uo[0][0] = st.uo[0][1] = st.uo[0][2] = 0;
    st.thoff[0] = 0;
    if (q >= sg[r].span || t0 >= TT) return;          // (padding behind the segment; T = 6: a barcode's fourth lane)
    const int* rtb = (const int*)(lds + Y.rtab) + 4 * r;          // {tcum, -, T, first count}
    st.i0[0] = sg[r].lo + (long long)bl * TT + t0;
    st.zoff[0] = 2 * ((r * t.NB + bl) * (TT / 2) + kk);          // the pair's 16-byte entry in the sample table: [R][NB][T / 2] pairs
    int meta = SK_L | BRM_A0 | BRM_A1 | BRM_VALID | (kk > 0 ? BRM_PREV : 0) | (t0 + 2 < TT ? BRM_NEXT : 0) | (r << 12);
    st.pt[0] = rtb[0] + t0;
    if (bl >= t.nshift) {
        meta |= BRM_MUT;
        const int ml = bl - t.nshift, base = (KIND >= 3 ? r * t.NB + ml : ml) * E;
        st.thoff[0] = KIND >= 3 ? r * t.NB * E : (KIND == 2 ? ml - ((const int*)(lds + Y.gix))[ml] : 0);
        if (E > 1) {
            const int* envt = (const int*)(lds + Y.envt);
#pragma unroll
            for (int d = 0; d < 3; ++d) { const int tt = t0 - 1 + d; st.uo[0][d] = base + ((tt >= 0 && tt < TT - 1) ? envt[rtb[0] + tt + 1] : 0); }
        } else st.uo[0][0] = st.uo[0][1] = st.uo[0][2] = base;
    }
    st.meta[0] = meta;
    if (CNT) {
        const int co = rtb[3];
        const long long cb = (co >= 0 ? (long long)co : M.cnt_off[r]) + (t.b0 + bl) * TT + t0;
        st.cnt[0][0] = M.counts[cb];
        st.cnt[0][1] = M.counts[cb + 1];
    }
}

// ---- the sample of a pair: z = mu + softplus(omega) eps ------------------------------------------------------------------------------
BB_DEV bb_d2 bs_z(const bb_d2 mu, const bb_d2 om, const bb_d2 e, bb_d2* sp, bb_d2* sg) {
    double sp0, sg0, sp1, sg1;
    bb_softplus_sigmoid(om.x, &sp0, &sg0);
    bb_softplus_sigmoid(om.y, &sp1, &sg1);
    *sp = bb_d2{sp0, sp1};
    *sg = bb_d2{sg0, sg1};
    return bb_d2{fma(sp0, e.x, mu.x), fma(sp1, e.y, mu.y)};
}
// what a loglambda pair leaves for the NEXT step's M pass: its two forward differences in the thread's private LDS entry, lambda on the
// thread's running sums (zn: the next pair's first sample, by DPP)
template <int KIND>
BB_DEV void bs_put_l(double* lds, const BRLay& Y, int zoff, int meta, const bb_d2 z, double zn, BSG& g) {
    if (!(meta & BRM_VALID)) return;
    const bool hn = meta & BRM_NEXT;
    *(bb_d2*)(lds + Y.zl + zoff) = BS_ZKEEP ? z : bb_d2{z.y - z.x, hn ? zn - z.y : 0.0};
    // (one replicate: a thread's pairs share their S_t rows, lambda joins running sums; several: the M pass takes e^z again, per replicate)
    if (KIND <= 2) { g.lam0 += bb_exp(z.x); g.lam1 += bb_exp(z.y); }
}
// ... and a unit pair: the forms OTHER threads read, into the stage tables of buffer `buf` (an owner recomputes its own sample)
template <int KIND>
BB_DEV void bs_put_u(double* lds, const DevModel& M, const BRLay& Y, const RunArgs& A, const BRSt<1>& st, int buf, const bb_d2 z) {
    const BBLds& L = Y.L;
    const int meta = st.meta[0], kind = meta & 15;
    const bool a0 = meta & BRM_A0, a1 = meta & BRM_A1;
    if (!(meta & BRM_VALID)) return;
    if (kind < SK_GS) {
        const int raw = br_stage_raw<KIND>(kind), trn = br_stage_trn<KIND>(kind);
        if (raw < Y.nst) {
            double* dst = lds + BR_ST(Y, raw) + buf * Y.SU + st.zoff[0];
            if (a0) dst[0] = z.x;
            if (a1) dst[1] = z.y;
        }
        if (trn >= 0) {
            const double f = (kind == SK_LS_E || (KIND >= 2 && kind == SK_LS_R)) ? -2.0 : 1.0;      // logsigma: precision w = e^{-2 z}; logtau: e^{logtau}
            double* dw = lds + BR_ST(Y, trn) + buf * Y.SU + st.zoff[0];
            if (a0) dw[0] = bb_exp(f * z.x);
            if (a1) dw[1] = bb_exp(f * z.y);
        }
    } else if (A.count_globals) {      // replicated global latents (tile 0 only): they ride along in the tile's row
        double* dst = lds + L.wk + M.K + (kind == SK_GLS ? M.nt1 : 0) + st.zoff[0];
        if (a0) dst[0] = z.x;
        if (a1) dst[1] = z.y;
    }
}

// MS: ELBO terms of a freshly formed sample of a pair (the split of br_sample's MS form): per latent -(z - m)^2 / (2 v^2) + log sigma, per
// (b, t) R z - lambda, per unit - logsigma_eff x (time steps that use it)
template <int KIND, int TT>
BB_DEV double bs_el_pair(const double* lds, const DevModel& M, const BRLay& Y, const RunArgs& A, const BRSt<1>& st, const bb_d2 z, const bb_d2 sp) {
    const int meta = st.meta[0], kd = meta & 15;
    if (!(meta & BRM_VALID) || !(kd < SK_GS || A.count_globals)) return 0.0;          // (sharded run: rank 0 counts the replicated blocks)
    const bool a0 = meta & BRM_A0, a1 = meta & BRM_A1;
    double pm0, iv0, pm1, iv1, el = 0.0;
    br_pair_prior<KIND>(lds, Y, st, 0, a0, a1, &pm0, &iv0, &pm1, &iv1);
    if (a0) el += -0.5 * (z.x - pm0) * (z.x - pm0) * iv0 + bb_log(sp.x);
    if (a1) el += -0.5 * (z.y - pm1) * (z.y - pm1) * iv1 + bb_log(sp.y);
    if (kd == SK_L) {
        el += (double)st.cnt[0][0] * z.x - bb_exp(z.x);
        el += (double)st.cnt[0][1] * z.y - bb_exp(z.y);
    } else if (kd == SK_LS_E || (KIND >= 2 && kd == SK_LS_R)) {
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            if (!(x ? a1 : a0)) continue;
            int n = TT - 1;
            if (KIND == 1 || KIND == 4) {
                const int* envt = (const int*)(lds + Y.envt);
                const int e_ = (st.uo[0][2] >> (8 * x)) & 255, tc = KIND == 4 ? ((const int*)(lds + Y.rtab))[4 * st.pt[0]] : 0;
                n = 0;
                for (int tt = 0; tt < TT - 1; ++tt) n += envt[tc + tt + 1] == e_ ? 1 : 0;
            }
            el -= (x ? z.y : z.x) * (double)n;
        }
    }
    return el;
}

// ---- the launch's first sample (every later one is formed inside the G passes of the step before it) -------------------------------
template <int KIND, int TT, bool MS = false>
BB_DEV void bs_sample0(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BRLay& Y, int NB, int P, unsigned step, BSG* gv, int buf0 = -1, bool want_el = false) {
    double* lds = cx.lds;
    const BBLds& L = Y.L;
    const BBTile t = KIND == 2 ? br_tile_geno(M, S, cx.block, NB) : br_tile(M, A, cx.block, NB);
    const int g0 = KIND == 2 ? S.tile_g[cx.block] : 0, g1 = KIND == 2 ? S.tile_g[cx.block + 1] : 0;
    const BRSeg* sg = (const BRSeg*)(lds + Y.seg);
    const int nseg = ((const int*)(lds + L.misc))[0];
    const int lspan = bs_lspan(sg, nseg);
    BB_STAMP(cx, S, 20);
    BB_PASS(cx, tid) { BSG& g = BB_PSTATE(gv, tid); g.lam0 = g.lam1 = 0.0; g.el = 0.0; }
    for (int k = 0; k < P; ++k) {
        BB_PASS(cx, tid) {
            BSG& g = BB_PSTATE(gv, tid);
            if (tid + k * cx.nthr < lspan) bs_desc_l<KIND, TT, MS>(M, Y, t, sg, tid + k * cx.nthr, g.st, lds);
            else br_desc<KIND, 1, false, bs_lpb<TT>(), false>(M, Y, t, sg, nseg, g0, g1, tid + k * cx.nthr, g.st, 0, lds);
            const int meta = g.st.meta[0];
            g.z = bb_d2{0.0, 0.0};
            if (meta & BRM_VALID) {
                const bool a0 = meta & BRM_A0, a1 = meta & BRM_A1;
                const bb_d2 mu = br_load_pair<false>(S.mu, g.st.i0[0], a0, a1), om = br_load_pair<false>(S.om, g.st.i0[0], a0, a1);
                g.z = bs_z(mu, om, bs_draw(A.seed, g.st.i0[0], step), &g.sp, &g.sg);
                if (MS && want_el) g.el += bs_el_pair<KIND, TT>(lds, M, Y, A, g.st, g.z, g.sp);
            }
        }
        BB_PASS(cx, tid) {
            BSG& g = BB_PSTATE(gv, tid);
            const double zn = BS_NEXT(gv, tid, z.x);
            const int p = tid + k * cx.nthr, meta = g.st.meta[0];
            if (p < lspan) { if ((meta & 15) == SK_L) bs_put_l<KIND>(lds, Y, g.st.zoff[0], meta, g.z, zn, g); }
            else bs_put_u<KIND>(lds, M, Y, A, g.st, buf0 >= 0 ? buf0 : (int)(step & 1u), g.z);
        }
    }
    BB_SYNC(cx);
    BB_STAMP(cx, S, 21);
}

// ---- M: the loglambda pairs' moment contributions, summed per thread, then by class over the wave, one LDS entry per (wave, class, value) ----
template <int KIND, int TT, bool MS = false>
BB_DEV void bs_moments(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BRLay& Y, int NB, int P, unsigned step, BSG* gv, int buf_ = -1, bool want_el = false) {
    double* lds = cx.lds;
    const BBLds& L = Y.L;
    const BBTile t = KIND == 2 ? br_tile_geno(M, S, cx.block, NB) : br_tile(M, A, cx.block, NB);
    const BRSeg* sg = (const BRSeg*)(lds + Y.seg);
    const int nseg = ((const int*)(lds + L.misc))[0];
    constexpr int LPB = bs_lpb<TT>();
    const int buf = buf_ >= 0 ? buf_ : (int)(step & 1u), lspan = bs_lspan(sg, nseg);
    const int R = KIND <= 2 ? 1 : M.R;
    BB_STAMP(cx, S, 22);
    if (MS) {
        // the threads' ELBO terms of this sample (gathered when it was formed): one partial per wave, br_row_publish adds them in order
        BB_PASS(cx, tid) {
            BSG& g = BB_PSTATE(gv, tid);
#ifdef BB_EMU
            if (tid == 0) { double e = 0.0; if (want_el) for (int t2 = 0; t2 < cx.nthr; ++t2) e += BB_PSTATE(gv, t2).el; lds[L.part] = e; for (int w = 1; w < (cx.nthr + 63) / 64; ++w) lds[L.part + w] = 0.0; }
#else
            double e = br_row16_sum(want_el ? g.el : 0.0);
            const int lo_ = __double2loint(e), hi_ = __double2hiint(e);
            double w_ = 0.0;
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) w_ += __hiloint2double(__builtin_amdgcn_readlane(hi_, 16 * r4), __builtin_amdgcn_readlane(lo_, 16 * r4));
            if ((tid & 63) == 0) lds[L.part + (tid >> 6)] = w_;
#endif
        }
        BB_PASS(cx, tid) { BB_PSTATE(gv, tid).el = 0.0; }
    }
    // (several replicates: a thread's slots may sit in different replicates' segments -- one round per replicate; the DPP sums need whole
    //  waves, so every thread takes every round)
    for (int r = 0; r < R; ++r) {
        const int stride = Y.rw[r] + 4;
        BB_PASS(cx, tid) {
            BSG& g = BB_PSTATE(gv, tid);
#pragma unroll
            for (int q = 0; q < BR_NCV; ++q) g.cv[q] = 0.0;
            if (KIND <= 2) { g.cv[0] = g.lam0; g.cv[6] = g.lam1; g.lam0 = g.lam1 = 0.0; }          // (summed while the pairs were sampled)
            const int lo = (nseg > r && sg[r].kind == SK_L) ? sg[r].tbeg : 0, hi = (nseg > r && sg[r].kind == SK_L) ? sg[r].tbeg + sg[r].span : 0;
            for (int k = 0; k < P; ++k) {
                const int p = tid + k * cx.nthr;
                if (p >= hi) break;
                if (p < lo) continue;
                BRSt<1>& st = g.st;
                bs_desc_l<KIND, TT, false>(M, Y, t, sg, p, st, lds);
                const int meta = st.meta[0];
                if ((meta & 15) != SK_L || !(meta & BRM_VALID)) continue;
                const bb_d2 d = *(const bb_d2*)(lds + Y.zl + st.zoff[0]);
                const bool hn = meta & BRM_NEXT, mut = meta & BRM_MUT;
                double dm = d.x, dn = d.y;
                if (BS_ZKEEP) { dm = d.y - d.x; dn = hn ? lds[Y.zl + st.zoff[0] + 2] - d.y : 0.0; }
                if (KIND >= 3) { g.cv[0] += bb_exp(d.x); g.cv[6] += bb_exp(d.y); }
                if (mut) {
                    double sm, sn, wm, wn;
                    br_unit_sw<KIND>(lds, Y, buf, st.uo[0][1], KIND >= 2 ? st.thoff[0] : 0, &sm, &wm);
                    if (KIND == 1 || KIND == 4) br_unit_sw<KIND>(lds, Y, buf, st.uo[0][2], KIND >= 3 ? st.thoff[0] : 0, &sn, &wn);
                    else { sn = sm; wn = wm; }
                    dm -= sm; dn -= sn;
                    g.cv[1] += wm; g.cv[2] += wm * dm; g.cv[3] += wm * dm * dm;
                    if (hn) { g.cv[7] += wn; g.cv[8] += wn * dn; g.cv[9] += wn * dn * dn; }
                } else {
                    g.cv[4] += dm; g.cv[5] += dm * dm;
                    if (hn) { g.cv[10] += dn; g.cv[11] += dn * dn; }
                }
            }
#ifndef BB_EMU
            // a thread's pairs share their class (tid % LPB): lanes of equal class add up over the 16-lane row (DPP), the wave's four rows
            // over two lane exchanges; the wave's first LPB lanes store
            // (several replicates: the wave's four rows add up too, over two lane exchanges -- LDS is short there, see br_layout)
            const int lane = KIND <= 2 ? (tid & 15) : (tid & 63), grp = KIND <= 2 ? (tid >> 4) : (tid >> 6);
#pragma unroll
            for (int q = 0; q < BR_NCV; ++q) {
                double v = bs_class_sum<LPB>(g.cv[q]);
                if (KIND >= 3) { v += __shfl_xor(v, 16); v += __shfl_xor(v, 32); }
                if (lane < LPB) lds[Y.racc_r[r] + q * stride + grp * LPB + lane] = v;
            }
#endif
        }
#ifdef BB_EMU
        BB_PASS(cx, tid) {      // (the emulation adds a row's / a wave's lanes of one class in lane order)
            constexpr int W = KIND <= 2 ? 16 : 64;
            const int lane = tid & (W - 1), grp = tid / W;
            if (lane < LPB) {
                for (int q = 0; q < BR_NCV; ++q) {
                    double v = 0.0;
                    for (int i = lane; i < W && (tid & ~(W - 1)) + i < cx.nthr; i += LPB) v += BB_PSTATE(gv, (tid & ~(W - 1)) + i).cv[q];
                    lds[Y.racc_r[r] + q * stride + grp * LPB + lane] = v;
                }
            }
        }
#endif
    }
    BB_SYNC(cx);                     // barrier: the partial sums are in LDS
}

// state of a pair and its window slot in; out again
BB_DEV void bs_load_state(const DevModel& M, const DevState& S, const double* hs_m, const double* hs_o, long long i0, long long ih, bool a0, bool a1, BSG& g) {
    g.mu = br_load_pair<false>(S.mu, i0, a0, a1); g.om = br_load_pair<false>(S.om, i0, a0, a1);
    g.am = br_load_pair<false>(S.acc_mu, i0, a0, a1); g.ao = br_load_pair<false>(S.acc_om, i0, a0, a1);
    g.lo = bb_load_lo(S, i0, a0, a1);
    g.hm = g.ho = bb_d2{0, 0};
    if (hs_m) {
#if !defined(BB_EMU) && BS_NT_HIST
        if (a0 && a1) {
            typedef double bs_v2d __attribute__((ext_vector_type(2)));
            const bs_v2d x = __builtin_nontemporal_load((const bs_v2d*)(hs_m + ih)), y = __builtin_nontemporal_load((const bs_v2d*)(hs_o + ih));
            g.hm = bb_d2{x.x, x.y}; g.ho = bb_d2{y.x, y.y};
        } else
#endif
        { g.hm = br_load_pair<false>(hs_m, ih, a0, a1); g.ho = br_load_pair<false>(hs_o, ih, a0, a1); }
    }
}
// optimiser update of the pair from the likelihood + prior gradient (g0, g1) and the draw's a = eps sigmoid, h = sigmoid / softplus (g.a, g.h); stores
BB_DEV bool bs_apply_store(const DevModel& M, const DevState& S, const RunArgs& A, const BBSlot wslot, double* hs_m, double* hs_o,
                           long long i0, long long ih, bool a0, bool a1, double g0, double g1, BSG& g) {
    const double go0 = fma(g0, g.a.x, g.h.x), go1 = fma(g1, g.a.y, g.h.y);
    bb_d2 nhm = g.hm, nho = g.ho;
    if (a0) {
        bb_opt_apply(M, S, A, wslot, 0, ih, -g0, g.hm.x, &nhm.x, &g.mu.x, &g.am.x, &g.lo.x);
        bb_opt_apply(M, S, A, wslot, 1, ih, -go0, g.ho.x, &nho.x, &g.om.x, &g.ao.x, &g.lo.y);
    }
    if (a1) {
        bb_opt_apply(M, S, A, wslot, 0, ih + 1, -g1, g.hm.y, &nhm.y, &g.mu.y, &g.am.y, &g.lo.z);
        bb_opt_apply(M, S, A, wslot, 1, ih + 1, -go1, g.ho.y, &nho.y, &g.om.y, &g.ao.y, &g.lo.w);
    }
    br_store_pair<false>(S.mu, i0, a0, a1, g.mu);
    br_store_pair<false>(S.om, i0, a0, a1, g.om);
    br_store_pair<false>(S.acc_mu, i0, a0, a1, g.am);
    br_store_pair<false>(S.acc_om, i0, a0, a1, g.ao);
    bb_store_lo(S, i0, a0, a1, g.lo);
    if (hs_m) { br_store_pair_stream<false>(hs_m, ih, a0, a1, nhm); br_store_pair_stream<false>(hs_o, ih, a0, a1, nho); }
    const double chk = (a0 ? g.mu.x + g.om.x : 0.0) + (a1 ? g.mu.y + g.om.y : 0.0);
    return !(chk - chk == 0.0);
}

// MS, NS > 1: sample smp's gradient of the pair (likelihood + prior part gl, and gl a for omega) joins the running sums in memory; the last
// sample averages them, adds the entropy term and takes the optimiser step (bb_update_pair's arithmetic).  Returns "non-finite".
template <bool MS>
BB_DEV bool bs_finish_pair(const DevModel& M, const DevState& S, const RunArgs& A, const BBSlot wslot, double* hs_m, double* hs_o,
                           long long i0, long long ih, bool a0, bool a1, double gl0, double gl1, BSG& g, const BSMs& ms) {
    if (!MS || ms.NS == 1) return bs_apply_store(M, S, A, wslot, hs_m, hs_o, i0, ih, a0, a1, gl0, gl1, g);
    bb_d2 gm{gl0, gl1}, go{gl0 * g.a.x, gl1 * g.a.y};
    if (ms.smp > 0) {
        const bb_d2 pm = br_load_pair<false>(S.gacc_mu, i0, a0, a1), po = br_load_pair<false>(S.gacc_om, i0, a0, a1);
        gm.x += pm.x; gm.y += pm.y; go.x += po.x; go.y += po.y;
    }
    if (!ms.last) {
        br_store_pair<false>(S.gacc_mu, i0, a0, a1, gm);
        br_store_pair<false>(S.gacc_om, i0, a0, a1, go);
        return false;
    }
    const double invS = 1.0 / (double)ms.NS;
    // (bs_apply_store forms go = fma(g, a, h): hand it g = the averaged gradient and a such that g a = the averaged omega part)
    const double g0 = gm.x * invS, g1 = gm.y * invS;
    const double go0 = go.x * invS + g.h.x, go1 = go.y * invS + g.h.y;
    bb_d2 nhm = g.hm, nho = g.ho;
    if (a0) {
        bb_opt_apply(M, S, A, wslot, 0, ih, -g0, g.hm.x, &nhm.x, &g.mu.x, &g.am.x, &g.lo.x);
        bb_opt_apply(M, S, A, wslot, 1, ih, -go0, g.ho.x, &nho.x, &g.om.x, &g.ao.x, &g.lo.y);
    }
    if (a1) {
        bb_opt_apply(M, S, A, wslot, 0, ih + 1, -g1, g.hm.y, &nhm.y, &g.mu.y, &g.am.y, &g.lo.z);
        bb_opt_apply(M, S, A, wslot, 1, ih + 1, -go1, g.ho.y, &nho.y, &g.om.y, &g.ao.y, &g.lo.w);
    }
    br_store_pair<false>(S.mu, i0, a0, a1, g.mu);
    br_store_pair<false>(S.om, i0, a0, a1, g.om);
    br_store_pair<false>(S.acc_mu, i0, a0, a1, g.am);
    br_store_pair<false>(S.acc_om, i0, a0, a1, g.ao);
    bb_store_lo(S, i0, a0, a1, g.lo);
    if (hs_m) { br_store_pair_stream<false>(hs_m, ih, a0, a1, nhm); br_store_pair_stream<false>(hs_o, ih, a0, a1, nho); }
    const double chk = (a0 ? g.mu.x + g.om.x : 0.0) + (a1 ? g.mu.y + g.om.y : 0.0);
    return !(chk - chk == 0.0);
}
// (MS, a sample that does not update: only mu, omega are needed)
BB_DEV void bs_load_mu_om(const DevState& S, long long i0, bool a0, bool a1, BSG& g) {
    g.mu = br_load_pair<false>(S.mu, i0, a0, a1); g.om = br_load_pair<false>(S.om, i0, a0, a1);
    g.am = g.ao = g.hm = g.ho = bb_d2{0, 0};
    g.lo = bb_f4{0.f, 0.f, 0.f, 0.f};
}

// ---- G-L: every loglambda pair slot -- state in, the draw again, gradient, the barcode's unit sums, optimiser, out; the next sample ----
template <int KIND, int TT, bool MS = false>
BB_DEV void bs_update_l(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BRLay& Y, int NB, int P, unsigned step, const BBSlot wslot, int* bad_any, BSG* gv,
                        const BSMs ms_ = BSMs{1, 0, -1, true, false, 0u, 0u}) {
    const BSMs ms = (MS && ms_.buf >= 0) ? ms_ : bs_ms_plain(step);
    double* lds = cx.lds;
    const BBLds& L = Y.L;
    constexpr int LPB = bs_lpb<TT>();
    const BBTile t = KIND == 2 ? br_tile_geno(M, S, cx.block, NB) : br_tile(M, A, cx.block, NB);
    const int g0t = KIND == 2 ? S.tile_g[cx.block] : 0, g1t = KIND == 2 ? S.tile_g[cx.block + 1] : 0;
    const BRSeg* sg = (const BRSeg*)(lds + Y.seg);
    const int nseg = ((const int*)(lds + L.misc))[0];
    const int lspan = bs_lspan(sg, nseg), buf = ms.buf;
    const int PL = (lspan + cx.nthr - 1) / cx.nthr;           // slots with loglambda pairs: the same for every thread (DPP needs whole waves)
    const int E = (KIND == 1 || KIND == 4) ? M.E : 1;
    double* hs_m = nullptr;
    double* hs_o = nullptr;
    if (A.opt == 0) {
        hs_m = S.hist + ((long long)wslot.slot * 2 + 0) * M.Dh;
        hs_o = S.hist + ((long long)wslot.slot * 2 + 1) * M.Dh;
    }
    double* aq = lds + Y.hbuf;            // [SU] pairs (As, Qs) per unit, in the moment contributions' region (dead since the row sums)
    for (int k = 0; k < PL; ++k) {
        // A: state in, the step's draw again, the sample
        BB_PASS(cx, tid) {
            BSG& g = BB_PSTATE(gv, tid);
            const int p = tid + k * cx.nthr;
            g.ok = 0;
            g.z = bb_d2{0.0, 0.0};
            g.xa = g.xq = g.rm = g.rn = 0.0;
            if (p < lspan) {
                bs_desc_l<KIND, TT, true>(M, Y, t, sg, p, g.st, lds);
                if ((g.st.meta[0] & BRM_VALID) && (g.st.meta[0] & 15) == SK_L) {
                    g.ok = 1;
                    const long long i0 = g.st.i0[0];
                    if (MS && !ms.last) bs_load_mu_om(S, i0, true, true, g);
                    else bs_load_state(M, S, hs_m, hs_o, i0, i0 - sg[g.st.meta[0] >> 12].pad, true, true, g);
                    if (BS_ZKEEP) {
                        g.z = *(const bb_d2*)(lds + Y.zl + g.st.zoff[0]);
                        double sp0, sg0, sp1, sg1;
                        bb_softplus_sigmoid(g.om.x, &sp0, &sg0);
                        bb_softplus_sigmoid(g.om.y, &sp1, &sg1);
                        const double r0 = bb_rcp(sp0), r1 = bb_rcp(sp1);
                        g.a = bb_d2{(g.z.x - g.mu.x) * r0 * sg0, (g.z.y - g.mu.y) * r1 * sg1};          // eps = (z - mu) / softplus(omega)
                        g.h = bb_d2{sg0 * r0, sg1 * r1};
                    } else {
                        g.e = BS_G_INLINE_DRAW ? bs_draw_inline(A.seed, i0, step, (unsigned)ms.smp) : bs_draw(A.seed, i0, step, (unsigned)ms.smp);
                        g.z = bs_z(g.mu, g.om, g.e, &g.sp, &g.sg);
                        g.a = bb_d2{g.e.x * g.sg.x, g.e.y * g.sg.y};
                        g.h = bb_d2{g.sg.x * bb_rcp(g.sp.x), g.sg.y * bb_rcp(g.sp.y)};
                    }
                }
            }
        }
        // B: the neighbour pairs' samples, differences, residuals, gradient
        BB_PASS(cx, tid) {
            BSG& g = BB_PSTATE(gv, tid);
            const double zpv = BS_PREV(gv, tid, z.y), znv = BS_NEXT(gv, tid, z.x);
            if (g.ok) {
                const BRSt<1>& st = g.st;
                const int meta = st.meta[0], pt = st.pt[0];
                const bool hp = meta & BRM_PREV, hn = meta & BRM_NEXT, mut = meta & BRM_MUT;
                const double z0 = g.z.x, z1 = g.z.y;
                const double zp = hp ? zpv : z0, zn = hn ? znv : z1;
                double ap = z0 - zp, am = z1 - z0, an = zn - z1, wp, wm, wn;
                if (mut) {
                    double sp_, sm, sn;
                    br_unit_sw<KIND>(lds, Y, buf, st.uo[0][1], KIND >= 2 ? st.thoff[0] : 0, &sm, &wm);
                    if (KIND == 1 || KIND == 4) {
                        br_unit_sw<KIND>(lds, Y, buf, st.uo[0][0], KIND >= 3 ? st.thoff[0] : 0, &sp_, &wp);
                        br_unit_sw<KIND>(lds, Y, buf, st.uo[0][2], KIND >= 3 ? st.thoff[0] : 0, &sn, &wn);
                    } else { sp_ = sn = sm; wp = wn = wm; }
                    ap -= sp_; am -= sm; an -= sn;
                } else { wp = hp ? lds[L.wbar + pt - 1] : 0.0; wm = lds[L.wbar + pt]; wn = lds[L.wbar + pt + 1]; }
                const double cp = hp ? lds[L.cc + pt - 1] : 0.0, cm = lds[L.cc + pt], cn = hn ? lds[L.cc + pt + 1] : 0.0;
                const double rm = am - cm, rn = hn ? an - cn : 0.0;          // residuals of the pair's own two forward differences
                const double rp_ = hp ? wp * (ap - cp) : 0.0, rm_ = wm * rm, rn_ = hn ? wn * rn : 0.0;
                double pm0, iv0, pm1, iv1;
                br_pair_prior<KIND>(lds, Y, st, 0, true, true, &pm0, &iv0, &pm1, &iv1);
                const double l0 = bb_exp(z0), l1 = bb_exp(z1);
                g.g0 = ((double)st.cnt[0][0] - l0) + l0 * lds[Y.iG + pt] + rm_ - rp_ - (z0 - pm0) * iv0;
                g.g1 = ((double)st.cnt[0][1] - l1) + l1 * lds[Y.iG + pt + 1] + rn_ - rm_ - (z1 - pm1) * iv1;
                g.rm = mut ? rm : 0.0;
                g.rn = mut ? rn : 0.0;
                g.wm = wm;
            }
        }
        // C: the barcode's unit sums As = sum_t r, Qs = sum_t r^2 over the time steps that use the unit (multienv: per environment) --
        // the lanes of a barcode add up, its first lane stores
        for (int e = 0; e < E; ++e) {
            BB_PASS(cx, tid) {
                BSG& g = BB_PSTATE(gv, tid);
                // (stage index of a unit = mutant x E + environment: the difference t0 uses the unit uo[1], t0 + 1 the unit uo[2])
                const bool um = g.ok && (g.st.meta[0] & BRM_MUT) && (E == 1 || g.st.uo[0][1] % E == e);
                const bool un = g.ok && (g.st.meta[0] & BRM_MUT) && (g.st.meta[0] & BRM_NEXT) && (E == 1 || g.st.uo[0][2] % E == e);
                g.xa = (um ? g.rm : 0.0) + (un ? g.rn : 0.0);
                g.xq = (um ? g.rm * g.rm : 0.0) + (un ? g.rn * g.rn : 0.0);
            }
            BB_PASS(cx, tid) {
                BSG& g = BB_PSTATE(gv, tid);
                const double As = BS_GROUP_SUM(LPB, gv, tid, xa), Qs = BS_GROUP_SUM(LPB, gv, tid, xq);
                const int p = tid + k * cx.nthr;
                if (g.ok && (g.st.meta[0] & BRM_MUT) && (p & (LPB - 1)) == 0) {          // (segments start at wave boundaries)
                    const int u = g.st.uo[0][1] - (E > 1 ? g.st.uo[0][1] % E : 0) + e;          // stage index of unit (mutant [, replicate], e)
                    *(bb_d2*)(aq + 2 * u) = bb_d2{As, Qs};
                    // hierarchical kinds: d/ds_eff of the unit = w As -- its theta (genotype's; mutant's over the replicates) sums these
                    if (KIND >= 2) lds[Y.gas + u] = (E == 1 ? g.wm : lds[BR_ST(Y, 2) + buf * Y.SU + u]) * As;
                }
            }
        }
        // D: optimiser, everything out; then the NEXT step's sample of the updated pair
        BB_PASS(cx, tid) {
            BSG& g = BB_PSTATE(gv, tid);
            g.z = bb_d2{0.0, 0.0};
            if (g.ok) {
                const long long i0 = g.st.i0[0];
                if (bs_finish_pair<MS>(M, S, A, wslot, hs_m, hs_o, i0, i0 - sg[g.st.meta[0] >> 12].pad, true, true, g.g0, g.g1, g, ms)) *bad_any = 1;
                const bb_d2 en = BS_S_INLINE_DRAW ? bs_draw_inline(A.seed, i0, ms.nstep, ms.nstream) : bs_draw(A.seed, i0, ms.nstep, ms.nstream);
                g.z = bs_z(g.mu, g.om, en, &g.sp, &g.sg);
                if (MS && ms.el_next) g.el += bs_el_pair<KIND, TT>(lds, M, Y, A, g.st, g.z, g.sp);
            }
        }
        BB_PASS(cx, tid) {
            BSG& g = BB_PSTATE(gv, tid);
            const double zn = BS_NEXT(gv, tid, z.x);
            const int p = tid + k * cx.nthr;
            if (g.ok) bs_put_l<KIND>(lds, Y, g.st.zoff[0], g.st.meta[0], g.z, zn, g);
        }
    }
    BB_STAMP(cx, S, 27);             // (no barrier here: the unit pass forms its first slot's sample before it meets the loglambda lanes)
}

// ---- G-U: the unit pair slots (and the replicated global latents on tile 0) ----------------------------------------------------------
// part A of a slot: state in, the step's draw again, the owner's own sample -- nothing of it needs the loglambda lanes' sums
template <int KIND, int TT, bool MS = false>
BB_DEV bool bs_unit_a(const DevModel& M, const DevState& S, const RunArgs& A, const BRLay& Y, const BBTile& t, const BRSeg* sg, int nseg, int g0t, int g1t,
                      const double* hs_m, const double* hs_o, int p, int lspan, unsigned step, double* lds, BSG& g, const BSMs& ms) {
    g.ok = 0;
    if (p < lspan) return false;
    BRSt<1>& st = g.st;
    br_desc<KIND, 1, false, bs_lpb<TT>(), false>(M, Y, t, sg, nseg, g0t, g1t, p, st, 0, lds);
    const int meta = st.meta[0];
    if (!(meta & BRM_VALID)) return false;
    const bool a0 = meta & BRM_A0, a1 = meta & BRM_A1;
    const long long i0 = st.i0[0];
    if (MS && !ms.last) bs_load_mu_om(S, i0, a0, a1, g);
    else bs_load_state(M, S, hs_m, hs_o, i0, i0 - sg[meta >> 12].pad, a0, a1, g);
    g.e = BS_G_INLINE_DRAW ? bs_draw_inline(A.seed, i0, step, (unsigned)ms.smp) : bs_draw(A.seed, i0, step, (unsigned)ms.smp);
    g.z = bs_z(g.mu, g.om, g.e, &g.sp, &g.sg);          // the owner's own sample of this step, again
    g.a = bb_d2{g.e.x * g.sg.x, g.e.y * g.sg.y};
    g.h = bb_d2{g.sg.x * bb_rcp(g.sp.x), g.sg.y * bb_rcp(g.sp.y)};
    g.ok = 1;
    return true;
}
// part B: gradient from the units' sums and the staged forms of THIS step, optimiser, everything out; the next step's sample, staged
template <int KIND, int TT, bool MS = false>
BB_DEV bool bs_unit_b(const DevModel& M, const DevState& S, const RunArgs& A, const BRLay& Y, const BRSeg* sg, double* hs_m, double* hs_o,
                      const BBSlot wslot, unsigned step, double* lds, BSG& g, int NBs, const BSMs& ms) {
    const BBLds& L = Y.L;
    const int buf = ms.buf, nbuf = buf ^ 1;
    const int E = (KIND == 1 || KIND == 4) ? M.E : 1;
    const int* envt = (const int*)(lds + Y.envt);
    const double* aq = lds + Y.hbuf;
    const double* stg = lds + buf * Y.SU;
    const BRSt<1>& st = g.st;
    const int meta = st.meta[0], kind = meta & 15;
    const bool a0 = meta & BRM_A0, a1 = meta & BRM_A1;
    const long long i0 = st.i0[0], ih = i0 - sg[meta >> 12].pad;
    double pm0, iv0, pm1, iv1, gl0 = 0.0, gl1 = 0.0;
    br_pair_prior<KIND>(lds, Y, st, 0, a0, a1, &pm0, &iv0, &pm1, &iv1);
    if (KIND == 2 && kind == SK_TH_R) {
        // d/dtheta_g = sum over the genotype's mutants (consecutive units of this tile) of w As, left by their first loglambda lanes
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            if (!(x ? a1 : a0)) continue;
            const int first = st.uo[0][x] & 0xffff, n = st.uo[0][x] >> 16;
            const double* ge = lds + Y.gas + first;          // (four reads in flight per LDS round trip, as br_update's theta sum)
            double s = 0.0;
            int i = 0;
            for (; i + 4 <= n; i += 4) s += (ge[i] + ge[i + 1]) + (ge[i + 2] + ge[i + 3]);
            for (; i < n; ++i) s += ge[i];
            (x ? gl1 : gl0) = s;
        }
    } else if (KIND >= 3 && kind == SK_TH_R) {
        // replicate kinds: d/dtheta[e, m] = sum over the replicates of w As of unit (m, r, e)
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            if (!(x ? a1 : a0)) continue;
            const int j = st.zoff[0] + x;                      // theta index m E + e; unit (m, r, e): (r NB + m) E + e
            double s = 0.0;
            for (int r = 0; r < M.R; ++r) s += lds[Y.gas + r * NBs * E + j];
            (x ? gl1 : gl0) = s;
        }
    } else if (kind < SK_GS) {
        // Per unit u = (mutant [, environment]) the sums over the time steps that use it, As = sum r, Qs = sum r^2 (the loglambda lanes
        // left them), give d/ds_bc = w As, d/dlogsigma = w Qs - n; hierarchical: s_eff = theta + e^{logtau} theta_tilde, so
        // d/dtheta_tilde = w As e^{logtau}, d/dlogtau = w As e^{logtau} theta_tilde
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            if (!(x ? a1 : a0)) continue;
            const int j = st.zoff[0] + x;                      // stage index of the latent's unit
            const bb_d2 AQ = *(const bb_d2*)(aq + 2 * j);
            const double wv = stg[BR_ST(Y, KIND <= 1 ? 1 : 2) + j];
            int nn = TT - 1;
            if (E > 1) {
                const int e = (st.uo[0][2] >> (8 * x)) & 255, tc = KIND == 4 ? ((const int*)(lds + Y.rtab))[4 * st.pt[0]] : 0;          // (unit pairs: pt = replicate)
                nn = 0;
                for (int tt = 0; tt < TT - 1; ++tt) nn += envt[tc + tt + 1] == e ? 1 : 0;
            }
            double acc;
            if (KIND <= 1) acc = kind == SK_S ? wv * AQ.x : wv * AQ.y - (double)nn;
            else if (kind == SK_LS_R) acc = wv * AQ.y - (double)nn;
            else if (kind == SK_TT_R) acc = wv * AQ.x * stg[BR_ST(Y, 1) + j];                                          // e^{logtau}
            else acc = wv * AQ.x * stg[BR_ST(Y, 1) + j] * stg[BR_ST(Y, 0) + j];                                        // logtau: e^{logtau} theta_tilde
            (x ? gl1 : gl0) = acc;
        }
    } else {
        const double* gg = lds + L.gglob + (kind == SK_GLS ? M.nt1 : 0) + st.zoff[0];
        if (a0) gl0 = gg[0];
        if (a1) gl1 = gg[1];
    }
    // (the replicated global latents' sample is rank 0's draw, back with the totals; on one GPU that is this thread's own)
    double zv0 = g.z.x, zv1 = g.z.y;
    if (kind >= SK_GS) {
        const double* zz = lds + L.zgl + (kind == SK_GLS ? M.nt1 : 0) + st.zoff[0];
        if (a0) zv0 = zz[0];
        if (a1) zv1 = zz[1];
    }
    gl0 -= (zv0 - pm0) * iv0;
    gl1 -= (zv1 - pm1) * iv1;
    const bool bad = bs_finish_pair<MS>(M, S, A, wslot, hs_m, hs_o, i0, ih, a0, a1, gl0, gl1, g, ms);
    const bb_d2 en = BS_S_INLINE_DRAW ? bs_draw_inline(A.seed, i0, ms.nstep, ms.nstream) : bs_draw(A.seed, i0, ms.nstep, ms.nstream);
    const bb_d2 zn = bs_z(g.mu, g.om, en, &g.sp, &g.sg);
    bs_put_u<KIND>(lds, M, Y, A, st, nbuf, zn);
    if (MS && ms.el_next) g.el += bs_el_pair<KIND, TT>(lds, M, Y, A, st, zn, g.sp);
    return bad;
}
template <int KIND, int TT, bool MS = false>
BB_DEV void bs_update_u(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BRLay& Y, int NB, int P, unsigned step, const BBSlot wslot, int* bad_any, BSG* gv,
                        const BSMs ms_ = BSMs{1, 0, -1, true, false, 0u, 0u}) {
    const BSMs ms = (MS && ms_.buf >= 0) ? ms_ : bs_ms_plain(step);
    double* lds = cx.lds;
    const BBLds& L = Y.L;
    const BBTile t = KIND == 2 ? br_tile_geno(M, S, cx.block, NB) : br_tile(M, A, cx.block, NB);
    const int g0t = KIND == 2 ? S.tile_g[cx.block] : 0, g1t = KIND == 2 ? S.tile_g[cx.block + 1] : 0;
    const BRSeg* sg = (const BRSeg*)(lds + Y.seg);
    const int nseg = ((const int*)(lds + L.misc))[0];
    const int lspan = bs_lspan(sg, nseg), k0 = lspan / cx.nthr;
    double* hs_m = nullptr;
    double* hs_o = nullptr;
    if (A.opt == 0) {
        hs_m = S.hist + ((long long)wslot.slot * 2 + 0) * M.Dh;
        hs_o = S.hist + ((long long)wslot.slot * 2 + 1) * M.Dh;
    }
    // The first unit slot's part A runs BEFORE the barrier that makes the loglambda lanes' sums visible: most waves have no loglambda
    // pair in the segment's last, partly filled slot and would only wait there for the one wave that has
    BB_PASS(cx, tid) {
        BSG& g = BB_PSTATE(gv, tid);
        g.ok = 0;
        if (k0 < P) bs_unit_a<KIND, TT, MS>(M, S, A, Y, t, sg, nseg, g0t, g1t, hs_m, hs_o, tid + k0 * cx.nthr, lspan, step, lds, g, ms);
    }
    BB_SYNC(cx);                     // the units' sums (As, Qs; genotype model: w As) are in LDS
    BB_PASS(cx, tid) {
        BSG& g = BB_PSTATE(gv, tid);
        bool bad = false;
        if (g.ok) bad = bs_unit_b<KIND, TT, MS>(M, S, A, Y, sg, hs_m, hs_o, wslot, step, lds, g, NB, ms);
        for (int k = k0 + 1; k < P; ++k)
            if (bs_unit_a<KIND, TT, MS>(M, S, A, Y, t, sg, nseg, g0t, g1t, hs_m, hs_o, tid + k * cx.nthr, lspan, step, lds, g, ms))
                bad = bs_unit_b<KIND, TT, MS>(M, S, A, Y, sg, hs_m, hs_o, wslot, step, lds, g, NB, ms) || bad;
        if (bad) *bad_any = 1;
    }
    BB_SYNC(cx);                     // the next step's tables are complete
    BB_STAMP(cx, S, 28);
}

// One step of the MS form, as the emulation and the kernel both walk it: the bookkeeping of (step, sample) -> what the passes need
BB_DEV BSMs bs_ms_of(const RunArgs& A, unsigned long long step, int smp, int NS, bool el_this_step, bool el_next_step) {
    const bool last = smp == NS - 1;
    const unsigned long long xc = step * (unsigned long long)NS + (unsigned long long)smp;
    return BSMs{NS, smp, (int)(xc & 1ull), last, last ? el_next_step : el_this_step, (unsigned)(last ? step + 1ull : step), last ? 0u : (unsigned)(smp + 1)};
}

#ifndef BB_EMU
template <int KIND, int NT, int TT, bool MS = false>
__global__ void __launch_bounds__(NT) k_stream(const DevModel* __restrict__ Mp, const DevState* __restrict__ Sp, const BRLay* __restrict__ Yp,
                                               RunArgs A, int NB, int nsteps, int P) {
    const DevModel& M = *Mp;
    const DevState& S = *Sp;
    const BRLay& Y = *Yp;
    extern __shared__ __attribute__((aligned(16))) double bs_smem[];
    BBCtx cx{(int)blockDim.x, (int)blockIdx.x, bs_smem, nullptr};
    int* ok_slot = (int*)(bs_smem + Yp->L.misc) + 1;
    int* bad_any = (int*)(bs_smem + Yp->L.misc) + 3;
    const unsigned long long c0 = S.ctr[0], c1 = S.ctr[1];
    unsigned long long step0 = c0 > c1 ? c0 : c1;
    step0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(step0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)step0);
    br_tile_setup<KIND>(cx, M, S, A, Y, NB, KIND <= 2 ? NT / 16 : NT / 64);
    if (threadIdx.x == 0) *bad_any = 0;
    __syncthreads();
    const bool dead = *ok_slot == 0;
    BSG g;
    BRSt<1>* nost = nullptr;
    int done = 0;
    if (!dead) {
        const int NS = MS ? A.S : 1;
        // ELBO recording (MS): position inside the recording period and the ring slot, carried like the window slot (as k_res)
        int ec = (MS && A.elbo_every > 0) ? bb_uniform((int)(step0 % (unsigned long long)A.elbo_every)) : 1;
        int ring = (MS && A.elbo_every > 0) ? bb_uniform((int)(((step0 + (unsigned long long)A.elbo_every - 1ull) / (unsigned long long)A.elbo_every) % BB_ELBO_RING)) : 0;
        bs_sample0<KIND, TT, MS>(cx, M, S, A, Y, NB, P, (unsigned)step0, &g, (int)((step0 * (unsigned long long)NS) & 1ull), MS && A.elbo_every > 0 && ec == 0);
        BBSlotCtr sc = bb_slot_init(A, step0);
        for (; done < nsteps; ++done, bb_slot_next(A, sc)) {
            const unsigned long long step = step0 + (unsigned long long)done;
            const BBSlot wslot = bb_slot_now(A, sc);
            const bool want_el = MS && A.elbo_every > 0 && ec == 0;
            const int ec1 = (MS && A.elbo_every > 0) ? (ec + 1 == A.elbo_every ? 0 : ec + 1) : 1;
            bool stop = false;
            for (int smp = 0; smp < NS; ++smp) {
                const BSMs ms = MS ? bs_ms_of(A, step, smp, NS, want_el, MS && A.elbo_every > 0 && ec1 == 0) : bs_ms_plain((unsigned)step);
                const unsigned long long xc = MS ? step * (unsigned long long)NS + (unsigned long long)smp : step;
                bs_moments<KIND, TT, MS>(cx, M, S, A, Y, NB, P, (unsigned)step, &g, ms.buf, want_el);
                bs_touch0<KIND, TT>(cx, M, S, A, Y, wslot);
                br_row_publish<1, true, MS>(cx, M, S, Y, nost, A.xepoch0 + (unsigned)(xc + 1), want_el);
                br_xchg_lead<false>(cx, M, S, A, Y, xc, ok_slot);
                br_xchg_consume<KIND, 1, false, MS>(cx, M, S, A, Y, nost, xc, ok_slot, want_el, ring, smp);
                if (*ok_slot == 0) { stop = true; break; }
                bs_update_l<KIND, TT, MS>(cx, M, S, A, Y, NB, P, (unsigned)step, wslot, bad_any, &g, ms);
                bs_update_u<KIND, TT, MS>(cx, M, S, A, Y, NB, P, (unsigned)step, wslot, bad_any, &g, ms);
            }
            if (stop) break;
            if (MS && A.elbo_every > 0) {
                if (ec == 0) ring = ring + 1 == BB_ELBO_RING ? 0 : ring + 1;
                ec = ec1;
            }
        }
    }
    if (threadIdx.x == 0) {
        if (*bad_any) S.hstatus[1] = 1u;
        if (dead || *ok_slot == 0) S.hstatus[0] = 1u;
        if (blockIdx.x == 0) { S.ctr[0] = step0 + (unsigned long long)done; S.ctr[1] = step0 + (unsigned long long)done; }
    }
}
typedef void (*bb_stream_kernel)(const DevModel*, const DevState*, const BRLay*, RunArgs, int, int, int);
#endif
