// bb_stream.h -- the ADVI step loop as one resident launch for tiles whose state does NOT fit the register file.
//
// k_res (bb_resident.h) keeps a tile's variational parameters, optimiser accumulators and draw in registers for the whole run: 44
// registers per pair of latents, at most two to four pair slots per thread.  BASELINE config 5 on ONE GPU (200 000 barcodes x 8, 5 000
// genotypes: 2.19 M latents, 782 barcodes and ~4 300 pairs per tile) needs twice the chip's register file and used to fall back to
// two kernels per step -- 184 B per latent of traffic, six launches per step, 135.8 us (0.20 of the HBM roofline).
//
// k_stream keeps k_res's tile map, segment table, moment algebra and in-launch exchange, and STREAMS the per-pair state: a thread
// walks its P pair slots (p = tid + k NT) twice per step,
//   S   mu, omega in (16 B per latent), Philox draw, softplus / sigmoid, z = mu + sigma eps, lambda = e^z; z and the unit forms staged
//       in LDS (single-buffered: a barrier ends the step), the lambdas summed per thread
//   --  barrier 1
//   M   the loglambda pairs' differences and moment contributions (neighbours and unit forms from LDS), summed per thread over its
//       slots -- a thread's pairs share their time-pair class (NT is a multiple of the lanes per barcode) -- then by class over the
//       16-lane rows of the wave (DPP), one LDS entry per (row, class, value)
//   --  barrier 2 -- row sums, publish, exchange, F pass: k_res's own code (br_row_publish, bbp_*_tg, br_finish)
//   G   per slot: mu, omega, accumulators, window slot in (60 B per latent), the draw AGAIN (Philox is a pure function of (seed, latent,
//       step): recomputing it and the softplus costs ~300 VALU instructions per pair, keeping eps sigmoid and sigmoid / softplus in
//       memory across the exchange would cost 32 B per latent -- the launch is bandwidth-bound, the VALUs are not), lambda = e^z from the
//       staged z, gradient, optimiser, everything out (56 B per latent)
// = 132 B per latent and step against 112 algorithmic (96 + the accumulators' low-order floats) -- the S pass's second read of mu,
// omega is the difference, and hits the Infinity Cache.
//
// Shapes: one replicate (fitness, multienv, genotype kinds), an even number of time points whose lanes per barcode divide 16
// (T = 2, 4, 8, 16; instances for the BASELINE shapes' T = 8 and for 4), flat-index-aligned pairs (no AP), one GPU, one MC sample per
// step, no ELBO recording.  Everything else keeps its launch.
#pragma once
#include "bb_resident.h"

struct BSAcc { double cv[BR_NCV]; };      // a thread's moment contributions, summed over its pair slots

// the step's normals of pair i0 (flat-index-aligned): out of line, as br_draw_call, so that its temporaries stay out of the slot loops
#ifdef BB_EMU
static inline
#else
__device__ __attribute__((noinline))
#endif
bb_d2 bs_draw(unsigned long long seed, long long i0, unsigned step) {
    double a, b;
    bb_normal_pair(seed, (unsigned long long)(i0 >> 1), step, 0u, &a, &b);
    return bb_d2{a, b};
}

BB_DEV bb_d2 bs_draw_inline(unsigned long long seed, long long i0, unsigned step) {
    double a, b;
    bb_normal_pair(seed, (unsigned long long)(i0 >> 1), step, 0u, &a, &b);
    return bb_d2{a, b};
}
// (inline in the G pass: the pair's state loads stay in flight behind it -- a call drains them first; C5 93.1 -> 91.4 us)
#ifndef BS_G_INLINE_DRAW
#define BS_G_INLINE_DRAW 1
#endif
// BS_KEEP_AH = 1: eps sigmoid and sigmoid / softplus go through memory (S.asv, S.hsv) from the S pass to the G pass instead of being
// recomputed there (+32 B per latent of traffic, -330 VALU instructions per pair): measured SLOWER, C5 90.8 -> 108.2 us -- the G pass is
// bandwidth-bound (profiles/r03d_stream_c5)
#ifndef BS_KEEP_AH
#define BS_KEEP_AH 0
#endif
// BS_NT_HIST = 1: the window slot is read with the non-temporal policy (it is not touched again for a whole window; the state arrays,
// re-read every step, keep the Infinity Cache): C5 90.8 -> 83.9 us
#ifndef BS_NT_HIST
#define BS_NT_HIST 1
#endif
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
#endif

// ---- S: every pair slot -- state in, draw, sample, stage ---------------------------------------------------------------------
template <int KIND, int TT>
BB_DEV void bs_sample(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BRLay& Y, int NB, int P, unsigned step, BSAcc* accv) {
    double* lds = cx.lds;
    const BBLds& L = Y.L;
    const BBTile t = KIND == 2 ? br_tile_geno(M, S, cx.block, NB) : br_tile(M, A, cx.block, NB);
    const int g0 = KIND == 2 ? S.tile_g[cx.block] : 0, g1 = KIND == 2 ? S.tile_g[cx.block + 1] : 0;
    const BRSeg* sg = (const BRSeg*)(lds + Y.seg);
    const int nseg = ((const int*)(lds + L.misc))[0];
    BB_STAMP(cx, S, 20);
    BB_PASS(cx, tid) {
        BSAcc& acc = BB_PSTATE(accv, tid);
#pragma unroll
        for (int q = 0; q < BR_NCV; ++q) acc.cv[q] = 0.0;
        for (int k = 0; k < P; ++k) {
            BRSt<1> st;
            br_desc<KIND, 1, false, (TT + 1) / 2, false>(M, Y, t, sg, nseg, g0, g1, tid + k * cx.nthr, st, 0, lds);
            const int meta = st.meta[0];
            if (!(meta & BRM_VALID)) continue;
            const bool a0 = meta & BRM_A0, a1 = meta & BRM_A1;
            const int kind = meta & 15;
            const bb_d2 mu = br_load_pair<false>(S.mu, st.i0[0], a0, a1), om = br_load_pair<false>(S.om, st.i0[0], a0, a1);
            const bb_d2 e = bs_draw(A.seed, st.i0[0], step);
            double sp0, sg0, sp1, sg1;
            bb_softplus_sigmoid(om.x, &sp0, &sg0);
            bb_softplus_sigmoid(om.y, &sp1, &sg1);
            const double z0 = fma(sp0, e.x, mu.x), z1 = fma(sp1, e.y, mu.y);
            if (BS_KEEP_AH) {
                br_store_pair<false>(S.asv, st.i0[0], a0, a1, bb_d2{e.x * sg0, e.y * sg1});
                br_store_pair<false>(S.hsv, st.i0[0], a0, a1, bb_d2{sg0 * bb_rcp(sp0), sg1 * bb_rcp(sp1)});
            }
            const double f = (kind == SK_LS_E || (KIND >= 2 && kind == SK_LS_R)) ? -2.0 : 1.0;      // logsigma: precision w = e^{-2 z}; logtau: e^{logtau}
            const bool need_exp = kind == SK_L || br_stage_trn<KIND>(kind) >= 0;
            const double l0 = need_exp ? bb_exp(f * z0) : 0.0, l1 = need_exp ? bb_exp(f * z1) : 0.0;
            if (kind == SK_L) {
                double* zw = lds + Y.zl + st.zoff[0];
                zw[0] = z0;
                zw[1] = z1;
                acc.cv[0] += l0;
                acc.cv[6] += l1;
            } else if (kind < SK_GS) {
                const int raw = br_stage_raw<KIND>(kind), trn = br_stage_trn<KIND>(kind);
                double* dst = lds + BR_ST(Y, raw) + st.zoff[0];
                if (a0) dst[0] = z0;
                if (a1) dst[1] = z1;
                if (trn >= 0) {
                    double* dw = lds + BR_ST(Y, trn) + st.zoff[0];
                    if (a0) dw[0] = l0;
                    if (a1) dw[1] = l1;
                }
            } else {      // replicated global latents (tile 0 only): they ride along in the tile's row
                double* dst = lds + L.wk + M.K + (kind == SK_GLS ? M.nt1 : 0) + st.zoff[0];
                if (A.count_globals) {
                    if (a0) dst[0] = z0;
                    if (a1) dst[1] = z1;
                }
            }
        }
    }
    BB_SYNC(cx);                     // barrier 1: every z and unit form of the tile is staged
    BB_STAMP(cx, S, 21);
}

// ---- M: the loglambda pairs' moment contributions, summed per thread, then by class over the wave's rows, one LDS entry each ----
template <int KIND, int TT>
BB_DEV void bs_moments(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BRLay& Y, int NB, int P, BSAcc* accv) {
    double* lds = cx.lds;
    const BBLds& L = Y.L;
    const BBTile t = KIND == 2 ? br_tile_geno(M, S, cx.block, NB) : br_tile(M, A, cx.block, NB);
    const int g0 = KIND == 2 ? S.tile_g[cx.block] : 0, g1 = KIND == 2 ? S.tile_g[cx.block + 1] : 0;
    const BRSeg* sg = (const BRSeg*)(lds + Y.seg);
    const int nseg = ((const int*)(lds + L.misc))[0];
    constexpr int LPB = (TT + 1) / 2;
    const int stride = Y.rw[0] + 4;
    BB_PASS(cx, tid) {
        BSAcc& acc = BB_PSTATE(accv, tid);
        for (int k = 0; k < P; ++k) {
            const int p = tid + k * cx.nthr;
            if (nseg < 1 || sg[0].kind != SK_L || p >= sg[0].tbeg + sg[0].span) break;          // (the loglambda segment comes first: later slots hold unit pairs only)
            BRSt<1> st;
            br_desc<KIND, 1, false, (TT + 1) / 2, false>(M, Y, t, sg, nseg, g0, g1, p, st, 0, lds);
            const int meta = st.meta[0];
            if ((meta & 15) != SK_L || !(meta & BRM_VALID)) continue;
            const double* zb = lds + Y.zl + st.zoff[0];
            const bool hn = meta & BRM_NEXT, mut = meta & BRM_MUT;
            const double z0 = zb[0], z1 = zb[1];
            const double zn = hn ? zb[2] : z1;
            double dm = z1 - z0, dn = zn - z1;
            if (mut) {
                double sm, sn, wm, wn;
                br_unit_sw<KIND>(lds, Y, 0, st.uo[0][1], KIND >= 2 ? st.thoff[0] : 0, &sm, &wm);
                if (KIND == 1) br_unit_sw<KIND>(lds, Y, 0, st.uo[0][2], 0, &sn, &wn);
                else { sn = sm; wn = wm; }
                dm -= sm; dn -= sn;
                acc.cv[1] += wm; acc.cv[2] += wm * dm; acc.cv[3] += wm * dm * dm;
                if (hn) { acc.cv[7] += wn; acc.cv[8] += wn * dn; acc.cv[9] += wn * dn * dn; }
            } else {
                acc.cv[4] += dm; acc.cv[5] += dm * dm;
                if (hn) { acc.cv[10] += dn; acc.cv[11] += dn * dn; }
            }
        }
#ifndef BB_EMU
        // a thread's pairs share their class (tid % LPB): lanes of equal class in a 16-lane row add up, the row's first LPB lanes store
        const int lane16 = tid & 15, row = tid >> 4;
#pragma unroll
        for (int q = 0; q < BR_NCV; ++q) {
            const double v = bs_class_sum<LPB>(acc.cv[q]);
            if (lane16 < LPB) lds[Y.racc_r[0] + q * stride + row * LPB + lane16] = v;
        }
#endif
    }
#ifdef BB_EMU
    BB_PASS(cx, tid) {      // (the emulation adds a row's lanes of one class in lane order)
        const int lane16 = tid & 15, row = tid >> 4;
        if (lane16 < LPB) {
            for (int q = 0; q < BR_NCV; ++q) {
                double v = 0.0;
                for (int i = lane16; i < 16; i += LPB) v += BB_PSTATE(accv, (tid & ~15) + i).cv[q];
                lds[Y.racc_r[0] + q * stride + row * LPB + lane16] = v;
            }
        }
    }
#endif
    BB_SYNC(cx);                     // barrier 2: the partial sums are in LDS
}

// likelihood gradient of the two latents of a unit pair (the unit branch of br_update, one slot): the sums over the time steps that use
// the unit, As = w sum r, Qs = w sum r^2 - n, r = dl - s_eff - c_t
template <int KIND, int TT>
BB_DEV void bs_unit_grad(double* lds, const DevModel& M, const BRLay& Y, const BRSt<1>& st, int kind, bool a0, bool a1, double* g0, double* g1, double* z0, double* z1) {
    const BBLds& L = Y.L;
    const int* envt = (const int*)(lds + Y.envt);
    const double* stg = lds;
    const double* zbuf = lds + Y.zl;
    const int E = KIND == 1 ? M.E : 1;
    double gx[2] = {0.0, 0.0}, zx[2] = {0.0, 0.0};
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        if (!(x ? a1 : a0)) continue;
        const int j = st.zoff[0] + x;                      // stage index of the latent
        const int e = KIND == 2 ? 0 : (st.uo[0][2] >> (8 * x)) & 255, bl = st.uo[0][x];
        zx[x] = stg[BR_ST(Y, br_stage_raw<KIND>(kind)) + j];
        const int th = KIND <= 1 ? 0 : j - ((st.uo[0][2] >> (16 * x)) & 0xffff);
        double sv, wv;
        br_unit_sw<KIND>(lds, Y, 0, j, th, &sv, &wv);
        const double* zr = zbuf + Y.zr0[0] + bl * (TT + 1);
        double As = 0.0, Qs = 0.0;
        int nn = 0;
        double zrow[TT];
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) zrow[tt] = zr[tt];          // the whole row in flight at once
#pragma unroll
        for (int tt = 0; tt < TT - 1; ++tt) {
            const bool use = E == 1 || envt[tt + 1] == e;
            const double rr = use ? (zrow[tt + 1] - zrow[tt]) - sv - lds[L.cc + tt] : 0.0;
            As += rr; Qs += rr * rr; nn += use ? 1 : 0;
        }
        double acc;
        if (KIND <= 1) acc = kind == SK_S ? wv * As : wv * Qs - (double)nn;
        else if (kind == SK_LS_R) acc = wv * Qs - (double)nn;
        else if (kind == SK_TT_R) {
            acc = wv * As * stg[BR_ST(Y, 1) + j];                                            // e^{logtau}
            lds[Y.gas + j] = wv * As;                                                        // d/ds_eff: its genotype's theta sums these
        } else acc = wv * As * stg[BR_ST(Y, 1) + j] * stg[BR_ST(Y, 0) + j];                  // logtau: e^{logtau} theta_tilde
        gx[x] = acc;
    }
    *g0 = gx[0]; *g1 = gx[1]; *z0 = zx[0]; *z1 = zx[1];
}

// ---- G: every pair slot -- state and window slot in, the draw again, gradient, optimiser, everything out -----------------------
template <int KIND, int TT>
BB_DEV void bs_update(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BRLay& Y, int NB, int P, unsigned step, const BBSlot wslot, int* bad_any) {
    double* lds = cx.lds;
    const BBLds& L = Y.L;
    const BBTile t = KIND == 2 ? br_tile_geno(M, S, cx.block, NB) : br_tile(M, A, cx.block, NB);
    const int g0t = KIND == 2 ? S.tile_g[cx.block] : 0, g1t = KIND == 2 ? S.tile_g[cx.block + 1] : 0;
    const BRSeg* sg = (const BRSeg*)(lds + Y.seg);
    const int nseg = ((const int*)(lds + L.misc))[0];
    double* hs_m = nullptr;
    double* hs_o = nullptr;
    if (A.opt == 0) {
        hs_m = S.hist + ((long long)wslot.slot * 2 + 0) * M.Dh;
        hs_o = S.hist + ((long long)wslot.slot * 2 + 1) * M.Dh;
    }
    // genotype model: pass 0 everything but theta (the theta_tilde thread of every mutant leaves w As in LDS), pass 1 theta
    int th_lo = 0, th_hi = 0;
    if (KIND == 2) for (int i = 0; i < nseg; ++i) if (sg[i].kind == SK_TH_R) { th_lo = sg[i].tbeg; th_hi = sg[i].tbeg + sg[i].span; }
    for (int pass = 0; pass < (KIND == 2 ? 2 : 1); ++pass) {
        if (pass) BB_SYNC(cx);
        BB_PASS(cx, tid) {
            bool bad = false;
            for (int k = 0; k < P; ++k) {
                const int p = tid + k * cx.nthr;
                if (KIND == 2 && ((p >= th_lo && p < th_hi) != (pass == 1))) continue;      // (the theta segment's pairs: pass 1, everything else: pass 0)
                BRSt<1> st;
                br_desc<KIND, 1, false, (TT + 1) / 2, true>(M, Y, t, sg, nseg, g0t, g1t, p, st, 0, lds);
                const int meta = st.meta[0];
                if (!(meta & BRM_VALID)) continue;
                const int kind = meta & 15;
                const bool a0 = meta & BRM_A0, a1 = meta & BRM_A1;
                const long long i0 = st.i0[0];
                bb_d2 mu = br_load_pair<false>(S.mu, i0, a0, a1), om = br_load_pair<false>(S.om, i0, a0, a1);
                bb_d2 am = br_load_pair<false>(S.acc_mu, i0, a0, a1), ao = br_load_pair<false>(S.acc_om, i0, a0, a1);
                bb_f4 lo = bb_load_lo(S, i0, a0, a1);
                bb_d2 hm{0, 0}, ho{0, 0};
                const long long ih = i0 - sg[meta >> 12].pad;
                if (hs_m) {
#if !defined(BB_EMU) && BS_NT_HIST
                    if (a0 && a1) {
                        typedef double bs_v2d __attribute__((ext_vector_type(2)));
                        const bs_v2d x = __builtin_nontemporal_load((const bs_v2d*)(hs_m + ih)), y = __builtin_nontemporal_load((const bs_v2d*)(hs_o + ih));
                        hm = bb_d2{x.x, x.y}; ho = bb_d2{y.x, y.y};
                    } else
#endif
                    { hm = br_load_pair<false>(hs_m, ih, a0, a1); ho = br_load_pair<false>(hs_o, ih, a0, a1); }
                }
                // the draw of the S pass again (a pure function of seed, latent and step), and what of it the omega gradient needs
                bb_d2 av, hv;
                if (BS_KEEP_AH) {
                    av = br_load_pair<false>(S.asv, i0, a0, a1);
                    hv = br_load_pair<false>(S.hsv, i0, a0, a1);
                } else {
                    const bb_d2 e = BS_G_INLINE_DRAW ? bs_draw_inline(A.seed, i0, step) : bs_draw(A.seed, i0, step);
                    double sp0, sg0, sp1, sg1;
                    bb_softplus_sigmoid(om.x, &sp0, &sg0);
                    bb_softplus_sigmoid(om.y, &sp1, &sg1);
                    av = bb_d2{e.x * sg0, e.y * sg1};
                    hv = bb_d2{sg0 * bb_rcp(sp0), sg1 * bb_rcp(sp1)};
                }
                double pm0 = 0.0, pm1 = 0.0, iv0 = 0.0, iv1 = 0.0;
                double g0 = 0.0, g1 = 0.0, z0 = 0.0, z1 = 0.0;
                if (kind == SK_L) {
                    const double* zb = lds + Y.zl + st.zoff[0];
                    st.lam[0] = bb_d2{bb_exp(zb[0]), bb_exp(zb[1])};
                    br_l_grad<KIND, 1, false, false>(lds, Y, st, 0, 0, &g0, &g1);      // (prior term included)
                } else {
                    br_pair_prior<KIND>(lds, Y, st, 0, a0, a1, &pm0, &iv0, &pm1, &iv1);
                    if (KIND == 2 && kind == SK_TH_R) {
#pragma unroll
                        for (int x = 0; x < 2; ++x) {
                            if (!(x ? a1 : a0)) continue;
                            const int first = st.uo[0][x] & 0xffff, n = st.uo[0][x] >> 16;
                            double s = 0.0;
                            for (int i = 0; i < n; ++i) s += lds[Y.gas + first + i];
                            (x ? g1 : g0) = s;
                            (x ? z1 : z0) = lds[BR_ST(Y, 3) + st.zoff[0] + x];
                        }
                    } else if (kind < SK_GS) {
                        bs_unit_grad<KIND, TT>(lds, M, Y, st, kind, a0, a1, &g0, &g1, &z0, &z1);
                    } else {
                        const double* gg = lds + L.gglob + (kind == SK_GLS ? M.nt1 : 0) + st.zoff[0];
                        const double* zz = lds + L.zgl + (kind == SK_GLS ? M.nt1 : 0) + st.zoff[0];
                        if (a0) { g0 = gg[0]; z0 = zz[0]; }
                        if (a1) { g1 = gg[1]; z1 = zz[1]; }
                    }
                    g0 -= (z0 - pm0) * iv0;
                    g1 -= (z1 - pm1) * iv1;
                }
                const double go0 = fma(g0, av.x, hv.x), go1 = fma(g1, av.y, hv.y);
                bb_d2 nhm = hm, nho = ho;
                if (a0) {
                    bb_opt_apply(M, S, A, wslot, 0, ih, -g0, hm.x, &nhm.x, &mu.x, &am.x, &lo.x);
                    bb_opt_apply(M, S, A, wslot, 1, ih, -go0, ho.x, &nho.x, &om.x, &ao.x, &lo.y);
                }
                if (a1) {
                    bb_opt_apply(M, S, A, wslot, 0, ih + 1, -g1, hm.y, &nhm.y, &mu.y, &am.y, &lo.z);
                    bb_opt_apply(M, S, A, wslot, 1, ih + 1, -go1, ho.y, &nho.y, &om.y, &ao.y, &lo.w);
                }
                br_store_pair<false>(S.mu, i0, a0, a1, mu);
                br_store_pair<false>(S.om, i0, a0, a1, om);
                br_store_pair<false>(S.acc_mu, i0, a0, a1, am);
                br_store_pair<false>(S.acc_om, i0, a0, a1, ao);
                bb_store_lo(S, i0, a0, a1, lo);
                if (hs_m) { br_store_pair_stream<false>(hs_m, ih, a0, a1, nhm); br_store_pair_stream<false>(hs_o, ih, a0, a1, nho); }
                const double chk = (a0 ? mu.x + om.x : 0.0) + (a1 ? mu.y + om.y : 0.0);
                bad = bad || !(chk - chk == 0.0);
            }
            if (bad) *bad_any = 1;
        }
    }
    BB_SYNC(cx);                     // the step's LDS tables are free: the next S pass rewrites them
    BB_STAMP(cx, S, 28);
}

#ifndef BB_EMU
template <int KIND, int NT, int TT>
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
    br_tile_setup<KIND>(cx, M, S, A, Y, NB, NT / 16);
    if (threadIdx.x == 0) *bad_any = 0;
    __syncthreads();
    const bool dead = *ok_slot == 0;
    BSAcc acc;
    BRSt<1>* nost = nullptr;
    int done = 0;
    if (!dead) {
        BBSlotCtr sc = bb_slot_init(A, step0);
        for (; done < nsteps; ++done, bb_slot_next(A, sc)) {
            const unsigned long long step = step0 + (unsigned long long)done;
            const BBSlot wslot = bb_slot_now(A, sc);
            bs_sample<KIND, TT>(cx, M, S, A, Y, NB, P, (unsigned)step, &acc);
            bs_moments<KIND, TT>(cx, M, S, A, Y, NB, P, &acc);
            br_row_publish<1, true, false>(cx, M, S, Y, nost, A.xepoch0 + (unsigned)(step + 1));
            br_xchg_lead<false>(cx, M, S, A, Y, step, ok_slot);
            br_xchg_consume<KIND, 1, false, false>(cx, M, S, A, Y, nost, step, ok_slot);
            if (*ok_slot == 0) break;
            bs_update<KIND, TT>(cx, M, S, A, Y, NB, P, (unsigned)step, wslot, bad_any);
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
