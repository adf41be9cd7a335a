// bb_resident.h -- the ADVI step loop as one resident launch, second generation: the OWNER of a latent computes.
//
// k_persist (bb_persist.h) keeps theta in registers but still runs the (barcode, time) work as separate passes over
// LDS-staged tables: a step is ~12 workgroup barriers, and the S -> M and R/U -> G hand-offs go through LDS.  Here the
// thread that owns a pair of consecutive loglambda latents (b, t), (b, t+1) also does that pair's (b, t) work:
//
//   S  softplus / sigmoid, z = mu + sigma eps, lambda = e^z; z goes to LDS only for the two NEIGHBOUR pairs of the barcode
//      (they need l[t-1], l[t+2]) and for the per-unit gradient sums; unit latents (s_bc, logsigma_bc, ...) are staged in
//      the form their readers need (s raw, precision w = e^{-2 logsigma})
//   -- barrier 1 --
//   M  differences, a = dl - s_eff, the pair's moment contributions; lanes of a wave with the same time pair are summed
//      by a shuffle butterfly, one lane per (wave, time pair) leaves 12 partial sums in LDS
//   -- barrier 2 --
//      K row entries = fixed-order sums over the waves; the row is published straight from registers
//   X  exchange (bb_persist.h: tiles -> 8 group leaders -> every tile; cross-GPU inboxes when sharded); in its shadow the
//      next step's normals and the window-slot prefetch, as before
//   F  c_t, D_t, G_t/S_t, global-latent gradients -- ends with barrier 3
//   G  gradient of the pair's latents from REGISTERS (lambda, a, w) + a handful of per-time table reads, prior,
//      optimiser, window slot
//
// Three workgroup barriers per step (+ the leaders' two).  LDS tables that a later phase of the SAME step still reads
// while fast waves already write the next step's values (z, unit stages) are double-buffered by step parity.
//
// Lane mapping: every loglambda segment (one per replicate) starts at a wave boundary; LPB = ceil(T/2) lanes per barcode,
// lane = bl * LPB + k owns (b, 2k), (b, 2k+1).  Their moment contributions go to LDS transposed (one column entry per lane)
// and are summed by column walks of stride LPB.  Unit pairs follow, flat.  Needs: 2 <= T_r <= 16, not the ragged-method quirk;
// other shapes keep k_persist.  A loglambda pair is (b, 2k), (b, 2k+1): where T is even and an even number of latents precedes
// the block, that is a pair (2q, 2q+1) of the flat index (one Philox pair, 16-byte accesses) and the plain instances run; the
// AP ("any parity") instances take odd T (LPB = (T+1)/2, a barcode's last lane owns a single latent) and odd offsets, with the
// parity of every pair a run-time property (two draws, 8-byte accesses where odd).
//
// Written, like bb_persist.h, as passes over an explicit per-thread state so that the host emulation (tests) runs the
// same source; the wave-level operations (a DPP row sum, the LDS-DMA prefetch) have emulation twins that add / move the same
// numbers.
#pragma once
#include "bb_persist.h"

#define BR_NCV 12          // per pair: 2 time slots x (S, M0, M1, M2, N1, N2)
// Scheduling fence: the two latents of a pair run the same long fp64 chains (softplus / sigmoid / exp / sqrt / rcp); interleaved for
// ILP they double the live temporaries and the 128-VGPR budget of a 16-wave workgroup spills.  Four waves per SIMD hide the
// dependent-issue latency anyway, so the chains are kept one after the other.
#if defined(BB_EMU) || defined(BR_NO_FENCE)
#define BR_SCHED_FENCE() ((void)0)
#else
#define BR_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
#define BR_MAXT 16
// BR_UNIT_PRIO: waves that hold unit pairs (s_bc, logsigma_bc, theta ...) raise their issue priority for the G pass (1) or for the G and the
// following S pass (2).  Their G work is a chain of LDS round trips (a barcode's row per latent) that the SIMD's arbiter -- oldest wave first,
// and they are the tile's youngest -- lets run only after the loglambda waves are done: they leave G last by far (per_wave_stamps.txt:
// 8.3 k cycles against 3.4 - 5.5 k) and everybody waits for them at barrier 1.
#ifndef BR_UNIT_PRIO
#define BR_UNIT_PRIO 1
#endif
// BR_ROW_L2: a tile on its group leader's XCD stores its row with plain stores (br_row_publish); 0 = always write-through
#ifndef BR_ROW_L2
#define BR_ROW_L2 1
#endif
#ifndef BR_UNIT_PRIO_LEVEL
#define BR_UNIT_PRIO_LEVEL 3
#endif

// (the prior of the segment's block travels with it: read per latent in the G pass, it must come from LDS -- fetched through the
// model descriptor with a per-lane block index it was a chain of five dependent global loads per pair and step)
struct BRSeg { long long lo, hi; int tbeg, span, blk, kind, ldsoff, r, lpb, T; double pm, iv; const double* mean_e; const double* iv_e; long long blo; int rstride, pad; };   // 96 B = 12 doubles
#define BR_SEG_DOUBLES 12

struct BRLay {
    BBLds L;             // what the shared exchange / finish code reads: wk, zgl, Lt, invS, cc, wbar, gglob, Dt, elbt, misc, acc, red
    int zl, NBT;         // [2][NBT] staged loglambda samples, double-buffered by step parity; replicate r's rows start at zr0[r] and are
    int zr0[BB_MAX_REP]; // T_r + 1 doubles apart: with the natural stride T_r (8: 64 B) the unit threads' walks along their barcodes'
                         // rows (two barcodes per lane) land all 64 lanes in two banks -- measured: the unit waves' G pass took
                         // 14 k cycles against 5 k for the loglambda waves
    int zlw, stw;        // buffers of zl and of every stage table (2: double-buffered by step parity; k_stream: 1), stage-table stride zlw x SU
    int st[6], SU, nst;  // nst unit stage tables, each [2][SU]: fitness / multienv  0 = s, 1 = w = e^{-2 logsigma}, 2 = logsigma
                         //   hierarchical  0 = theta_tilde, 1 = e^logtau, 2 = w, 3 = theta, 4 = logtau, 5 = logsigma
    int eps;             // [P * NT] bb_d2: the next step's normals
    int hbuf;            // [P][2][NT] bb_d2: this step's TruncatedADAGrad window slot, fetched by LDS-DMA while the exchange is in flight
                         // (no registers held across the exchange); shares the moment contributions' region (dead by then) unless the
                         // cross-GPU inbox staging needs it at the same time
    int racc;            // the moment contributions, transposed: per replicate r [12][rw[r] + 4] at racc_r[r], one column entry per lane
    int racc_r[BB_MAX_REP], rw[BB_MAX_REP];   // of the replicate's loglambda segment (rw = its lanes, whole waves; + 4: the 12 columns start in different banks)
    int rowmap;          // [K] int pairs: {lanes per barcode | used << 24, LDS offset of the entry's column (value v of time-pair class k of replicate r)}
    int iG, csum;        // [Ttot] G_t / S_t;  [R] sum_t c_t
    int rtab;            // [R] int4: per replicate {first time point (tcum), offset of its rows in zl, time points, -} -- for the unit threads' walks
    int ftab;            // [Ttot] int4: what the F pass needs of time point j -- {LDS offset of its five moment totals or -1 (a replicate's last
                         // time point), its index inside s_pop, has a time point before it, -}
    int envt;            // [Ttot] ints: environment of every time point (multienv)
    int gas;             // genotype model: [SU] w As of every mutant of the tile, summed per genotype by the theta threads
    int gix;             // genotype model: int tables of the tile -- [SU] genotype of every mutant minus the tile's first, then [SU] per own genotype
                         // first local mutant | members << 16: k_stream forms its pair descriptors in every pass, and looked these up in device memory
                         // (two or three dependent loads per slot and pass)
    int seg;             // BRSeg table
    int total;
    int lpb[BB_MAX_REP];
};

// LDS offset of unit stage table i (the tables are stw = 2 SU apart): arithmetic on two uniform values -- indexed with a per-lane table
// number, Y.st[i] was a vector load from the layout record in device memory (~500 cycles) inside the S and G passes
#define BR_ST(Y, i) ((Y).st[0] + (i) * (Y).stw)
// lanes per barcode of a loglambda segment: one per pair of time points.  (Any count works: the moment contributions are summed
// by column walks over the segment's lanes, stride LPB, not by a butterfly over lane bits -- T = 6 used to idle one lane in four.)
static inline int br_lpb(int T) { return (T + 1) / 2; }
// k_stream (bb_stream.h): the lanes of a barcode exchange their samples and add their residuals up by DPP inside a 16-lane row, so a
// barcode takes a power-of-two number of lanes (T = 6: four, the last one idle)
BB_HD int br_lpb_stream(int T) { const int l = (T + 1) / 2; return l <= 1 ? 1 : (l <= 2 ? 2 : (l <= 4 ? 4 : 8)); }

// padded thread-index span of a tile: loglambda segments wave-aligned, LPB lanes per barcode, then the unit pairs
static inline long long br_tile_span(const DevModel& M, long long NB, bool globals, bool stream = false) {
    long long p = 0;
    for (int r = 0; r < M.R; ++r) p = ((p + 63) & ~63ll) + NB * (stream ? br_lpb_stream(M.T[r]) : br_lpb(M.T[r]));
    if (M.kind == 0 || M.kind == 1) p += 2 * (NB * M.E / 2 + 1);
    else if (M.kind == 2) p += 4 * (NB / 2 + 1);          // theta of the tile's own genotypes (at most NB), theta_tilde, logtau, logsigma
    else if (M.kind == 3) p += (NB / 2 + 1) + 3ll * M.R * (NB / 2 + 1);
    else p += (NB * M.E / 2 + 1) + 3ll * M.R * (NB * M.E / 2 + 1);
    if (globals) p += 2 * (M.nt1 / 2 + 1);
    return p;
}

static inline bool br_eligible(const DevModel& M) {
    if (M.kind == 2 && !M.geno_sorted) return false;   // (genotype model: tiles must own whole genotypes, see br_tile_geno)
    if (M.quirk || M.Ttot > 64) return false;
    for (int r = 0; r < M.R; ++r) if (M.T[r] < 2 || M.T[r] > BR_MAXT) return false;
    return true;
}
// pairs (b, 2k), (b, 2k+1) of this shape are not all pairs (2q, 2q+1) of the flat index: the AP instances
static inline bool br_any_parity(const DevModel& M) {
    if (M.blk_lo[BK_L] & 1) return true;
    for (int r = 0; r < M.R; ++r) if ((M.T[r] & 1) || (M.off_l[r] & 1)) return true;
    return false;
}

// The small tables of a tile -- everything whose size depends on the model's (K, nt1, Ttot, R) only -- come FIRST in LDS, in this order
// (round 4).  (Measured and dropped: telling the compiler these offsets and the model constants behind them as compile-time constants in
// the instances whose time-point count is a template argument -- __builtin_assume on every field, to take ~20 uniform values out of a
// step loop that holds more of them than there are SGPRs: the spill count did not move, 296 -> 300 v_readlane_b32, C2 11.87 -> 11.95 us,
// C4 unchanged; profiles/r04d_c2_experiments.)
struct BRHdr { int rowmap, tmap, wk, zgl, Lt, invS, cc, wbar, gglob, misc, part, Dt, elbt, iG, csum, ftab, rtab, envt, seg, end; };
#ifndef BB_EMU
__host__ __device__
#endif
inline constexpr BRHdr br_header(int K, int nt1, int Ttot, int R) {
    BRHdr h{};
    const int KK = K + 2 * nt1;
    int o = 0;
    h.rowmap = o;  o += K + 1;                // [K] int pairs
    h.tmap = o;    o += (K + 1) / 2 + 1;
    h.wk = o;      o += KK;
    h.zgl = o;     o += 2 * nt1;
    h.Lt = o;      o += Ttot;
    h.invS = o;    o += Ttot;
    h.cc = o;      o += Ttot + 1;
    h.wbar = o;    o += Ttot + 1;
    h.gglob = o;   o += 2 * nt1;
    h.misc = o;    o += 16 + BB_MAX_REP;
    h.part = o;    o += 32;
    h.Dt = o;      o += Ttot + 1;
    h.elbt = o;    o += Ttot;
    h.iG = o;      o += Ttot;
    h.csum = o;    o += BB_MAX_REP;
    o = (o + 1) & ~1;
    h.ftab = o;    o += 2 * Ttot;
    h.rtab = o;    o += 2 * R;                // [R] {tcum, zr0, T, cnt_off (or -1: beyond 31 bits, read from the model record)}
    h.envt = o;    o += (Ttot + 1) / 2 + 1;
    o = (o + 1) & ~1;
    h.seg = o;     o += BR_SEG_DOUBLES * (4 + 4 * R + 1);      // (br_build_segs: at most R + 1 + 3 R + 2 segments, + the end marker)
    h.end = (o + 1) & ~1;
    return h;
}

static inline
#ifndef BB_EMU
__host__ __device__
#endif
BRLay br_layout(const DevModel& M, int NB, int NT, int P, int xg_rows, bool own_hbuf = false, bool stream = false) {
    BRLay Y;
    const int X = (M.kind == 1) ? M.E : (M.kind == 4 ? M.E * M.R : M.R);
    const int KK = M.K + 2 * M.nt1;
    int o = 0;
    int lmax = 1;
    for (int r = 0; r < BB_MAX_REP; ++r) { Y.lpb[r] = r < M.R ? (stream ? br_lpb_stream(M.T[r]) : br_lpb(M.T[r])) : 1; if (Y.lpb[r] > lmax) lmax = Y.lpb[r]; }
    BBLds& L = Y.L;
    L = BBLds{};
    Y.NBT = 0;
    for (int r = 0; r < BB_MAX_REP; ++r) { Y.zr0[r] = Y.NBT; if (r < M.R) Y.NBT += NB * (M.T[r] + 1); }
    Y.NBT = (Y.NBT + 1) & ~1;
    // (k_stream, bb_stream.h: no z rows at all -- the region holds the loglambda pairs' samples, one 16-byte entry per pair, [R][NB][T / 2];
    //  its unit stage tables are double-buffered like k_res's, but only the forms OTHER threads read are staged)
    if (stream) { long long sp = 0; for (int r = 0; r < M.R; ++r) sp += (long long)NB * (M.T[r] / 2); Y.NBT = (int)(2 * sp + 2); }      // ([R][NB][T / 2] pairs)
    {
        const BRHdr H = br_header(M.K, M.nt1, M.Ttot, M.R);
        Y.rowmap = H.rowmap; L.tmap = H.tmap; L.wk = H.wk; L.zgl = H.zgl; L.Lt = H.Lt; L.invS = H.invS; L.cc = H.cc; L.GG = H.wbar; L.wbar = H.wbar;
        L.gglob = H.gglob; L.misc = H.misc; L.part = H.part; L.Dt = H.Dt; L.elbt = H.elbt; Y.iG = H.iG; Y.csum = H.csum; Y.ftab = H.ftab;
        Y.rtab = H.rtab; Y.envt = H.envt; Y.seg = H.seg; L.seg = H.seg;
        o = H.end;
    }
    Y.zlw = stream ? 1 : 2;
    Y.zl = o;      o += Y.zlw * Y.NBT;
    Y.SU = NB * X;
    Y.nst = stream ? (M.kind <= 1 ? 2 : 4) : (M.kind <= 1 ? 3 : 6);
    Y.stw = 2 * Y.SU;
    for (int i = 0; i < 6; ++i) { Y.st[i] = o; if (i < Y.nst) o += Y.stw; }
    o = (o + 1) & ~1;
    // (k_res: two terms per mutant as one 16-byte entry, br_theta_pre; k_stream: w As of every unit of the hierarchical kinds, bs_update_l)
    Y.gas = o;     o += stream ? (M.kind >= 2 ? Y.SU : 0) : ((M.kind == 2 || M.kind == 3) ? 2 * Y.SU : 0);
    Y.gix = o;     o += M.kind == 2 ? Y.SU + 1 : 0;
    o = (o + 1) & ~1;
    // One transient region, users that never overlap in time: the transposed moment contributions (M pass -> row sums), then --
    // from the publish of the tile's row to the next step's S / G passes -- the drawn-ahead normals and the prefetched window slot.
    o = (o + 1) & ~1;
    Y.racc = o;
    int racc_total = 0;
    for (int r = 0; r < BB_MAX_REP; ++r) {
        Y.racc_r[r] = Y.racc + racc_total;
        // (k_stream: a thread adds its pair slots' contributions up in registers, the 16-lane rows of a wave reduce them by class, so a
        //  column holds one entry per (row of 16 lanes, class) instead of one per lane of the segment)
        // (several replicates, round 4: LDS is short there -- the four rows of a wave add up as well, one entry per (wave, class); the two lane
        //  exchanges per value that costs took C5's M pass from 7.6 k to 12.6 k cycles, so one replicate keeps the entry per row)
        Y.rw[r] = r < M.R ? (stream ? (NT / (M.R == 1 ? 16 : 64)) * Y.lpb[r] : (int)(((long long)NB * Y.lpb[r] + 63) & ~63ll)) : 0;
        if (r < M.R) racc_total += BR_NCV * (Y.rw[r] + 4);
    }
    // Staging of the cross-GPU inbox rows (bbp_consume<true>; xg_rows = 8 x world rows of KK entries, summed in chunks of whole rows):
    // in use between the publish and the F pass, while the drawn-ahead normals (and, pf = 0, the window slot) wait in the transient
    // region -- whatever that region has to spare behind them is the staging; only the missing part is added.
    // (Only the ready-word form of the cross-GPU consume stages rows, bbp_consume<true>: builds with BR_TG = 0.  The tagged form polls
    //  the inbox straight into registers, bbp_consume_tgx.)
    const int xg_want = (!BR_TG && xg_rows > 0) ? (xg_rows * KK < (BB_NQ + 1) * NT ? xg_rows * KK : (BB_NQ + 1) * NT) : 0;
    int busy;           // doubles of the transient region alive during the exchange
    if (stream) {
        // (no normals, no window slot in LDS: they go through registers.  hbuf = the units' sums (As, Qs) [SU] pairs, written by the loglambda
        //  lanes in the G pass -- the contributions are dead by then -- and read by the unit threads)
        Y.hbuf = Y.racc;
        busy = racc_total > 2 * Y.SU ? racc_total : 2 * Y.SU;
        o += busy;
        Y.eps = o;     o += NT / 2;          // (4 bytes per thread: where the LDS-DMA touches of the first slot's lines land, bs_touch0)
    } else if (own_hbuf) {
        // the window slot is fetched while the moment contributions are alive (RunArgs.pf = 1, 2): a region of its own
        Y.eps = Y.racc;
        busy = 2 * P * NT;
        { int need = racc_total > busy ? racc_total : busy; if (need < busy + xg_want) need = busy + xg_want; o += need; }
        o = (o + 1) & ~1;
        Y.hbuf = o;    o += 4 * P * NT;
    } else {
        Y.hbuf = Y.racc;
        Y.eps = Y.hbuf + 4 * P * NT;
        busy = 6 * P * NT;
        { int need = racc_total > busy ? racc_total : busy; if (need < busy + xg_want) need = busy + xg_want; o += need; }
    }
    L.acc = Y.racc + busy;
    L.acc_cap = xg_want;
    { const int nch = xg_rows / 8 > 5 ? xg_rows / 8 : 5;               // (bbp_consume<.., WIDE> / bbp_consume_tg: the other thread groups' partial sums; bbp_consume_tgx: one chunk sum per source rank)
      L.red = o;     o += 2 * 128 + nch * ((KK + 63) & ~63) + 16; }
    (void)lmax;
    L.total = Y.total = (o + 1) & ~1;
    return Y;
}

template <int P>
struct BRSt {
    bb_d2 mu[P], om[P], am[P], ao[P];   // variational parameters and optimiser accumulators
    bb_d2 a[P], h[P];                   // eps * sigmoid(omega), sigmoid / softplus of the current draw
    bb_d2 z[P];                         // the pair's samples (alive from S to M only: the G pass re-reads what it needs from LDS)
    bb_d2 lam[P];                       // loglambda pairs: e^z
    long long i0[P];
    int meta[P];                        // seg kind | a0 << 4 | a1 << 5 | valid << 6 | mutant << 7 | has_prev << 8 | has_next << 9 | seg index << 12
    int zoff[P];                        // loglambda: offset of z0 inside one zl buffer; unit pairs: index of latent 0 inside its stage table
    int uo[P][3];                       // loglambda, mutant: stage-table index of the unit the backward / inner / forward difference uses;
                                        // unit pairs: [0], [1] = zl-buffer offset of the barcode row of latent 0 / 1, [2] = env of latent 0 | env of latent 1 << 8
    int pt[P];                          // loglambda: tcum[r] + t0; unit pairs: tcum[r]
    int rb[P];                          // loglambda: where the pair's 12 moment contributions go (racc_r[r] + lane position in its segment);
                                        // hierarchical unit pairs and loglambda pairs: see thoff
    int thoff[P];                       // hierarchical models: stage index of a unit minus thoff = index of its theta (r NB E_)
    unsigned cnt[P][2];                 // loglambda: the two counts
    bb_f4 lo[P];                        // low-order parts of the four running window sums (bb_opt_apply)
    bb_d2 gp[P];                        // loglambda pairs: the part of the gradient that needs no totals (br_grad_pre)
    bb_d2 gm[P], go[P];                 // MS instances (several MC samples per step): running sums of d/dmu, d/domega over the samples
    double el;                          // MS instances, recording steps: the thread's ELBO terms of the current sample
};

enum { BRM_A0 = 1 << 4, BRM_A1 = 1 << 5, BRM_VALID = 1 << 6, BRM_MUT = 1 << 7, BRM_PREV = 1 << 8, BRM_NEXT = 1 << 9 };

// Pair accesses.  A loglambda pair (b, 2k), (b, 2k+1) normally sits at an even flat index: one Philox pair, 16-byte accesses.  It
// does not where an odd number of latents precedes the loglambda block (genotype layout: n_geno + 3 n_bc; replicate layouts with R
// even and n_bc odd) or the number of time points is odd (a barcode's row then starts at alternating parity, and its last lane
// owns a single latent).  Those shapes run the AP ("any parity") instances: parity checked per pair -- two 8-byte accesses and two
// Philox pairs' halves for the normals (br_draw_call) where it is odd.  The even shapes' instances carry none of that.
template <bool AP> BB_DEV bool br_pair_aligned(long long i0) { return !AP || !(i0 & 1); }
template <bool AP> BB_DEV bb_d2 br_load_pair(const double* base, long long i0, bool a0, bool a1) {
    if (a0 && a1 && br_pair_aligned<AP>(i0)) return *(const bb_d2*)(base + i0);
    return bb_d2{a0 ? base[i0] : 0.0, a1 ? base[i0 + 1] : 0.0};
}
// window slots are written once and read again a whole window (100 steps) later: BR_NT_STORE = 1 stores them non-temporally,
// BR_NT_LOAD = 2 fetches them with the nt policy (experiment switches; defaults below)
#ifndef BR_NT_LOAD
#define BR_NT_LOAD 0
#endif
#ifndef BR_PUB_COALESCED
#define BR_PUB_COALESCED 1
#endif
#ifndef BR_SKIP_EXP
#define BR_SKIP_EXP 1
#endif
#ifndef BR_NT_STORE
#define BR_NT_STORE 1
#endif
template <bool AP> BB_DEV void br_store_pair_stream(double* base, long long i0, bool a0, bool a1, bb_d2 v) {
#if !defined(BB_EMU) && BR_NT_STORE
    if (a0 && a1 && br_pair_aligned<AP>(i0)) {
        typedef double bb_v2d __attribute__((ext_vector_type(2)));
        bb_v2d w = {v.x, v.y};
        __builtin_nontemporal_store(w, (bb_v2d*)(base + i0));
        return;
    }
    if (a0) __builtin_nontemporal_store(v.x, base + i0);
    if (a1) __builtin_nontemporal_store(v.y, base + i0 + 1);
#else
    if (a0 && a1 && br_pair_aligned<AP>(i0)) { *(bb_d2*)(base + i0) = v; return; }
    if (a0) base[i0] = v.x;
    if (a1) base[i0 + 1] = v.y;
#endif
}
template <bool AP> BB_DEV void br_store_pair(double* base, long long i0, bool a0, bool a1, bb_d2 v) {
    if (a0 && a1 && br_pair_aligned<AP>(i0)) { *(bb_d2*)(base + i0) = v; return; }
    if (a0) base[i0] = v.x;
    if (a1) base[i0 + 1] = v.y;
}

// Tile map of k_res.  The group leaders of the exchange (tiles 0 .. 7) do extra work between their publish and everybody's
// consume; with nbl < NB barcodes they reach their publish early enough to have drawn their next normals before their members'
// rows arrive, and the group rows appear one draw (~5 k cycles) sooner for all tiles.
BB_HD BBTile br_tile(const DevModel& M, const RunArgs& A, int block, int NB) {
    if (A.nbl <= 0) return bb_tile(M, A, block, NB);
    const int nlead = bbp_groups(A);
    BBTile t;
    t.NB = NB;
    const int cap = block < nlead ? A.nbl : NB;
    t.b0 = block < nlead ? A.b_lo + (long long)block * A.nbl : A.b_lo + (long long)nlead * A.nbl + (long long)(block - nlead) * NB;
    long long b1 = t.b0 + cap < A.b_hi ? t.b0 + cap : A.b_hi;
    if (t.b0 > A.b_hi) t.b0 = A.b_hi;
    t.nbt = (int)(b1 > t.b0 ? b1 - t.b0 : 0);
    long long ns = M.nn - t.b0;
    t.nshift = (int)(ns < 0 ? 0 : (ns > t.nbt ? t.nbt : ns));
    t.m0 = t.b0 + t.nshift - M.nn;
    t.nmt = t.nbt - t.nshift;
    return t;
}

// Genotype model: the tiles' cuts come from a host-built table and fall on genotype boundaries (geno_idx is non-decreasing), so
// that a tile holds ALL mutants of the genotypes [tile_g[i], tile_g[i + 1]) it owns: d/dtheta_g = sum over the genotype's mutants
// of w As is then a sum inside the tile, and theta_g is sampled, staged and updated by that tile alone
// (/root/reference/src/model_fitness_normal_hierarchical_genotypes.jl:209-243: s_eff = theta[geno] + exp(logtau) theta_tilde).
BB_HD BBTile br_tile_geno(const DevModel& M, const DevState& S, int block, int NB) {
    BBTile t;
    t.NB = NB;
    t.b0 = S.tile_b[block];
    t.nbt = (int)(S.tile_b[block + 1] - t.b0);
    long long ns = M.nn - t.b0;
    t.nshift = (int)(ns < 0 ? 0 : (ns > t.nbt ? t.nbt : ns));
    t.m0 = t.b0 + t.nshift - M.nn;
    t.nmt = t.nbt - t.nshift;
    return t;
}

// ---- segment table of a tile in the padded thread-index space (one thread) --------------------------------------
template <int KIND>
BB_HD int br_build_segs(BRSeg* sg, const DevModel& M, const BRLay& Y, const BBTile& t, bool globals, int g0 = 0, int g1 = 0) {
    int n = 0, cur = 0;
    auto add = [&](int blk, int kind, long long lo, long long cnt, int ldsoff, int r, int lpb, int T) {
        if (cnt <= 0) return;
        BRSeg s;
        s.lo = lo; s.hi = lo + cnt; s.blk = blk; s.kind = kind; s.ldsoff = ldsoff; s.r = r; s.lpb = lpb; s.T = T;
        s.rstride = kind == SK_L ? Y.rw[r] + 4 : 0; s.pad = (int)bb_hdelta(M, blk, r);      // (pad: the segment's place in a window row)
             // (loglambda: stride between the 12 columns of the transposed moment contributions)
        s.pm = M.pri[blk].mean; s.iv = M.pri[blk].inv_var; s.mean_e = M.pri[blk].mean_e; s.iv_e = M.pri[blk].inv_var_e; s.blo = M.blk_lo[blk];
        if (kind == SK_L) { cur = (cur + 63) & ~63; s.span = (int)(cnt / T) * lpb; }   // (T == 0 only for non-loglambda segments)
        else s.span = bb_seg_pairs(lo, lo + cnt);
        s.tbeg = cur;
        cur += s.span;
        sg[n++] = s;
    };
    for (int r = 0; r < M.R; ++r)
        add(BK_L, SK_L, M.off_l[r] + t.b0 * M.T[r], (long long)t.nbt * M.T[r], Y.zr0[r], r, Y.lpb[r], M.T[r]);
    if (KIND == 2) {   // theta of the tile's genotypes (stage table 3, index = genotype - g0), then per mutant tt / lt / ls
        add(BK_S, SK_TH_R, M.blk_lo[BK_S] + g0, g1 - g0, 0, 0, 0, 0);
        add(BK_TT, SK_TT_R, M.blk_lo[BK_TT] + t.m0, t.nmt, 0, 0, 0, M.T[0]);
        add(BK_LT, SK_LT_R, M.blk_lo[BK_LT] + t.m0, t.nmt, 0, 0, 0, M.T[0]);
        add(BK_LS, SK_LS_R, M.blk_lo[BK_LS] + t.m0, t.nmt, 0, 0, 0, M.T[0]);
    } else if (t.nmt > 0) {
        if (KIND == 0 || KIND == 1) {
            const int E = KIND == 1 ? M.E : 1;
            add(BK_S, SK_S, M.blk_lo[BK_S] + t.m0 * E, (long long)t.nmt * E, 0, 0, 0, M.T[0]);
            add(BK_LS, SK_LS_E, M.blk_lo[BK_LS] + t.m0 * E, (long long)t.nmt * E, 0, 0, 0, M.T[0]);
        } else {   // replicate (E_ = 1) and multienv_replicate: theta[e, m]; per replicate tt / lt / ls [e, m, r], env fastest;
                   // ldsoff = stage index of the segment's first latent (unit (ml, r, e): (r NB + ml) E_ + e; theta: ml E_ + e)
            const int E_ = KIND == 4 ? M.E : 1;
            add(BK_S, SK_TH_R, M.blk_lo[BK_S] + t.m0 * E_, (long long)t.nmt * E_, 0, 0, 0, 0);
            for (int r = 0; r < M.R; ++r) {
                const long long o = ((long long)r * M.nb + t.m0) * E_;
                add(BK_TT, SK_TT_R, M.blk_lo[BK_TT] + o, (long long)t.nmt * E_, r * t.NB * E_, r, 0, M.T[r]);
                add(BK_LT, SK_LT_R, M.blk_lo[BK_LT] + o, (long long)t.nmt * E_, r * t.NB * E_, r, 0, M.T[r]);
                add(BK_LS, SK_LS_R, M.blk_lo[BK_LS] + o, (long long)t.nmt * E_, r * t.NB * E_, r, 0, M.T[r]);
            }
        }
    }
    if (globals) {
        add(BK_SPOP, SK_GS, M.blk_lo[BK_SPOP], M.blk_hi[BK_SPOP] - M.blk_lo[BK_SPOP], 0, 0, 0, 0);
        add(BK_LSPOP, SK_GLS, M.blk_lo[BK_LSPOP], M.blk_hi[BK_LSPOP] - M.blk_lo[BK_LSPOP], 0, 0, 0, 0);
    }
    sg[n].tbeg = cur;
    return n;
}

// ---- what a thread knows about pair p of its tile's padded index space (segment, latent index, LDS offsets, counts): slot k of st ----
// (LPBC > 0: the lanes per barcode as a compile-time constant -- k_stream recomputes descriptors per pass and a run-time integer
//  division costs ~30 instructions; CNT = false: the pair's counts are not fetched)
template <int KIND, int P, bool AP = false, int LPBC = 0, bool CNT = true>
BB_DEV void br_desc(const DevModel& M, const BRLay& Y, const BBTile& t, const BRSeg* sg, int nseg, int g0, int g1, int p, BRSt<P>& st, int k, const double* lds) {
    const int* gx = (const int*)(lds + Y.gix);          // (genotype model: the tile's genotype tables, br_tile_setup)
    const int* rtb = (const int*)(lds + Y.rtab);        // per replicate {tcum, zr0, T, cnt_off}: by lane-indexed loads from the model record these
    const int* envt = (const int*)(lds + Y.envt);       // were dependent vector loads from device memory in front of everything else
    const int E = (KIND == 1 || KIND == 4) ? M.E : 1;       // units per (mutant [, replicate]): environments
    int si = -1;
    // (k_stream, a descriptor per slot and pass: most slots sit in the first segment -- the loglambda slab -- and skip the search)
    if (LPBC && nseg > 0 && p >= sg[0].tbeg && p < sg[0].tbeg + sg[0].span) si = 0;
    else for (int i = 0; i < nseg; ++i) if (p >= sg[i].tbeg && p < sg[i].tbeg + sg[i].span) si = i;
    int meta = 15;                          // (kind 15: no segment -- SK_L is 0)
    long long i0 = 0;
    st.zoff[k] = 0; st.uo[k][0] = st.uo[k][1] = st.uo[k][2] = 0; st.pt[k] = 0; st.cnt[k][0] = st.cnt[k][1] = 0u;
    st.rb[k] = 0; st.thoff[k] = 0;
    if (si >= 0) {
        const BRSeg s = sg[si];
        meta = s.kind | (si << 12);
        if (s.kind == SK_L) {
            const int q = p - s.tbeg, bl = LPBC ? q / LPBC : q / s.lpb, kk = q - bl * (LPBC ? LPBC : s.lpb);
            st.rb[k] = Y.racc_r[s.r] + q;           // (every lane of the segment's waves has a column entry; idle lanes write zeros)
            if (2 * kk < s.T) {
                const int t0 = 2 * kk;
                i0 = s.lo + (long long)bl * s.T + t0;
                meta |= BRM_A0 | (t0 + 1 < s.T ? BRM_A1 : 0) | BRM_VALID | (t0 > 0 ? BRM_PREV : 0) | (t0 + 2 < s.T ? BRM_NEXT : 0);   // (odd T: the last lane owns one latent)
                st.zoff[k] = s.ldsoff + bl * (s.T + 1) + t0;
                st.pt[k] = rtb[4 * s.r] + t0;
                if (bl >= t.nshift) {
                    meta |= BRM_MUT;
                    const int ml = bl - t.nshift;
                    // stage index of the unit (ml [, r] [, e]) -- fitness: ml; multienv: ml E + e; replicate: r NB + ml;
                    // multienv_replicate: (r NB + ml) E + e -- for the differences t0-1, t0, t0+1 (environment of t + 1)
                    const int base = KIND >= 3 ? (s.r * t.NB + ml) * E : ml * E;
                    // (genotype model: unit ml, theta index = its genotype's position among the tile's own = ml - thoff)
                    st.thoff[k] = KIND >= 3 ? s.r * t.NB * E : (KIND == 2 ? ml - gx[ml] : 0);
                    for (int d = 0; d < 3; ++d) {
                        const int tt = t0 - 1 + d;
                        const int e = (E > 1 && tt >= 0 && tt < s.T - 1) ? envt[rtb[4 * s.r] + tt + 1] : 0;
                        st.uo[k][d] = base + e;
                    }
                }
                if (CNT) {
                    const int co = rtb[4 * s.r + 3];
                    const long long cb = (co >= 0 ? (long long)co : M.cnt_off[s.r]) + t.b0 * s.T + (long long)bl * s.T + t0;
                    st.cnt[k][0] = M.counts[cb];
                    st.cnt[k][1] = t0 + 1 < s.T ? M.counts[cb + 1] : 0u;
                }
            }
        } else {
            const int q = p - s.tbeg;
            i0 = 2 * ((s.lo >> 1) + q);
            const bool a0 = i0 >= s.lo, a1 = i0 + 1 < s.hi;
            meta |= (a0 ? BRM_A0 : 0) | (a1 ? BRM_A1 : 0) | BRM_VALID;
            st.zoff[k] = s.ldsoff + (int)(i0 - s.lo);      // stage index of latent 0 (may sit one before the segment: never stored)
            st.thoff[k] = s.ldsoff;                          // (hierarchical: r NB E_; the theta of unit j is j - thoff)
            if (KIND == 2 && s.kind == SK_TH_R) {
                // theta of genotype g: its mutants are the consecutive local units [first, first + n) -- uo[x] = first | n << 16
                for (int x = 0; x < 2; ++x) {
                    const long long gl = (i0 - s.lo) + x;             // (own genotype number gl of the tile's g1 - g0)
                    int v = 0;
                    if (gl >= 0 && gl < g1 - g0) {
                        if (gl < Y.SU) v = gx[Y.SU + (int)gl];
                        else {          // (more own genotypes than the table holds -- genotypes without mutants: from device memory)
                            const long long g = g0 + gl;
                            const int n = M.geno_ptr[g + 1] - M.geno_ptr[g];
                            v = (n > 0 ? (int)(M.geno_mem[M.geno_ptr[g]] - t.m0) : 0) | (n << 16);
                        }
                    }
                    st.uo[k][x] = v;
                }
            } else if (s.kind < SK_GS) {
                // unit (ml, e) of latent x: index j = (i0 - lo) + x = ml * E + e inside the segment; its barcode's local index
                int env = 0;
                for (int x = 0; x < 2; ++x) {
                    int j = (int)(i0 - s.lo) + x;
                    if (j < 0) j = 0;
                    const int ml = j / E, e = j - ml * E;
                    st.uo[k][x] = t.nshift + ml;
                    // (genotype model, E == 1: the unit's theta index instead of its environment, 16 bits each)
                    if (KIND == 2) env |= ((j < t.nmt ? gx[j] : 0) & 0xffff) << (16 * x);
                    else env |= e << (8 * x);
                }
                st.uo[k][2] = env;
                st.pt[k] = s.r;
            }
        }
    }
    st.i0[k] = i0;
    st.meta[k] = meta;
}

// ---- prologue: segment table, row map, per-thread metadata, state into registers --------------------------------
// ---- tile setup shared by k_res and k_stream: segment table, zeroed LDS tables, row map, F-pass table (ends inside a pass: the caller
// meets at a barrier before anything reads them).  ncol: column entries per time-pair class in the transposed contributions
// (k_res: the tile's barcodes; k_stream: the rows of 16 lanes)
// ---- the tile-independent LDS descriptor tables, one entry each (device: a thread per entry; host: host_tables) ----------------
// row j of a tile's moment row = sum over the threads of time-pair class k (tid % LPB == k) of their value v: everything the row sums
// need in ONE LDS read {lanes per barcode | used << 24, column's LDS offset} -- looked up by replicate in the layout record they were
// three dependent vector loads from device memory behind the code's LDS read, in a pass that is all latency; tmap[j] = the time point
// whose normaliser S_t row entry j is, or -1
BB_HD void br_table_row(const DevModel& M, const BRLay& Y, int j, int* rm2, int* tmapj) {
    int code = 0, col = 0;
    for (int r = 0; r < M.R; ++r) {
        const int T = M.T[r], q0 = j - M.kq[r];
        if (q0 < 0 || q0 >= 6 * T - 5) continue;
        int tt, q;
        if (q0 < T) { tt = q0; q = 0; } else { tt = (q0 - T) / 5; q = 1 + (q0 - T) - 5 * tt; }
        code = Y.lpb[r] | (1 << 24);
        col = Y.racc_r[r] + ((tt & 1) * 6 + q) * (Y.rw[r] + 4) + (tt >> 1);
    }
    rm2[0] = code;
    rm2[1] = col;
    int tj = -1;
    for (int r = 0; r < M.R; ++r) { const int tt = j - M.kq[r]; if (tt >= 0 && tt < M.T[r]) tj = M.tcum[r] + tt; }
    *tmapj = tj;
}
// F-pass table: a lane of the F pass looked its replicate up in the model record -- a chain of four dependent vector loads from
// device memory (~2 k cycles) in a pass the whole tile waits for
BB_HD void br_table_time(const DevModel& M, const BBLds& L, int j, int* ft) {
    int r = 0;
    while (r + 1 < M.R && j >= M.tcum[r + 1]) ++r;
    const int tt = j - M.tcum[r], T = M.T[r];
    ft[0] = tt < T - 1 ? L.wk + M.kq[r] + T + 5 * tt : -1;
    ft[1] = M.off_t[r] + tt;
    ft[2] = tt > 0 ? 1 : 0;
    ft[3] = 0;
}
BB_HD void br_table_rep(const DevModel& M, const BRLay& Y, int r, int* rt) {
    rt[0] = M.tcum[r]; rt[1] = Y.zr0[r]; rt[2] = M.T[r];
    rt[3] = M.cnt_off[r] < (1ll << 31) ? (int)M.cnt_off[r] : -1;          // (first count of the replicate)
}

template <int KIND>
BB_DEV void br_tile_setup(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BRLay& Y, int NB, int ncol_or_neg) {
    double* lds = cx.lds;
    const BBLds& L = Y.L;
    const BBTile t = KIND == 2 ? br_tile_geno(M, S, cx.block, NB) : br_tile(M, A, cx.block, NB);
    const int g0 = KIND == 2 ? S.tile_g[cx.block] : 0, g1 = KIND == 2 ? S.tile_g[cx.block + 1] : 0;
    BRSeg* sg = (BRSeg*)(lds + Y.seg);
    int* li = (int*)(lds + L.misc);
    const int KK = M.K + 2 * M.nt1;
    BB_PASS(cx, tid) {
        // li[1] = exchange ok word; it starts at 0 ("leave") while an earlier launch's timeout is unacknowledged by the host
        if (S.segtab) {
            // the tile's segment table came from the host (thread 0 building it here took 3.4 us of every launch): one coalesced copy
            const double* src = S.segtab + (long long)cx.block * S.segtab_stride;
            const int nd = S.segtab_stride - 1;
            for (int i = tid; i < nd; i += cx.nthr) lds[Y.seg + i] = src[i];
            if (tid == 0) li[0] = ((const int*)(src + nd))[0];
        } else if (tid == 0) li[0] = br_build_segs<KIND>(sg, M, Y, t, cx.block == 0, g0, g1);
        if (tid == 0) { li[1] = bb_get_word(S.gbar + 1) == 0u ? 1 : 0; li[2] = ncol_or_neg < 0 ? t.nbt : ncol_or_neg; }
        if (tid == 0) {
            // Same-XCD first hop (BR_ROW_L2): li[4] = 0 not known yet / 1 this tile runs on its group leader's XCD / -1 it does not;
            // li[5] = own XCC id, li[6] = this launch's tag, li[7] = groups.  The leaders say where they run.
            const int xcc = br_xcc_id(), NG = bbp_groups(A);
            li[4] = (BR_ROW_L2 && A.row_l2) ? 0 : -1; li[5] = xcc; li[6] = (int)A.launch_tag; li[7] = NG;
            S.xsel[cx.block] = li[4];
            if (BR_ROW_L2 && cx.block < NG) bb_set_word64(S.xtab + cx.block, ((unsigned long long)A.launch_tag << 32) | (unsigned)xcc);
        }
        for (int k = tid; k < KK; k += cx.nthr) lds[L.wk + k] = 0.0;
        for (int i = tid; i < Y.zlw * Y.NBT; i += cx.nthr) lds[Y.zl + i] = 0.0;
        for (int i = tid; i < Y.nst * Y.stw; i += cx.nthr) lds[Y.st[0] + i] = 0.0;
        if (tid <= M.Ttot) { lds[L.cc + tid] = 0.0; lds[L.wbar + tid] = 0.0; lds[L.Dt + tid] = 0.0; }
        if (tid < M.Ttot) ((int*)(lds + Y.envt))[tid] = (KIND == 1 || KIND == 4) ? M.env_idx[tid] : 0;
        // (the per-replicate table here, in front of the barrier: the pair descriptors read it)
        if (S.ldstab) { for (int i = tid; i < 4 * M.R; i += cx.nthr) ((int*)(lds + Y.rtab))[i] = S.ldstab[3 * M.K + 4 * M.Ttot + i]; }
        else for (int rr = tid; rr < M.R; rr += cx.nthr) br_table_rep(M, Y, rr, (int*)(lds + Y.rtab) + 4 * rr);
        if (KIND == 2) {
            int* gx = (int*)(lds + Y.gix);
            for (int j = tid; j < t.nmt; j += cx.nthr) gx[j] = M.geno_idx[t.m0 + j] - g0;
            for (int j = tid; j < g1 - g0 && j < Y.SU; j += cx.nthr) {
                const int g = g0 + j, n = M.geno_ptr[g + 1] - M.geno_ptr[g];
                gx[Y.SU + j] = (n > 0 ? (int)(M.geno_mem[M.geno_ptr[g]] - t.m0) : 0) | (n << 16);
            }
        }
    }
    BB_STAMP_RT(cx, S, 7);          // (thread 0: the segment table is built)
    BB_SYNC(cx);
    BB_STAMP_RT(cx, S, 8);
    int* rm = (int*)(lds + Y.rowmap);
    if (S.ldstab) {
        // the tables are the same for every tile: built once on the host, one coalesced copy here
        const int K = M.K, Tt = M.Ttot, R = M.R;
        BB_PASS(cx, tid) {
            for (int i = tid; i < 2 * K; i += cx.nthr) rm[i] = S.ldstab[i];
            for (int i = tid; i < K; i += cx.nthr) ((int*)(lds + L.tmap))[i] = S.ldstab[2 * K + i];
            for (int i = tid; i < 4 * Tt; i += cx.nthr) ((int*)(lds + Y.ftab))[i] = S.ldstab[3 * K + i];
        }
    } else {
        BB_PASS(cx, tid) {
            for (int j = tid; j < M.K; j += cx.nthr) br_table_row(M, Y, j, rm + 2 * j, (int*)(lds + L.tmap) + j);
            for (int j = tid; j < M.Ttot; j += cx.nthr) br_table_time(M, L, j, (int*)(lds + Y.ftab) + 4 * j);
        }
    }
    BB_STAMP_RT(cx, S, 9);          // (thread 0's share of the LDS tables)
}

template <int KIND, int P, bool AP = false>
BB_DEV void br_prologue(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BRLay& Y, int NB, BRSt<P>* stv) {
    double* lds = cx.lds;
    const BBLds& L = Y.L;
    const BBTile t = KIND == 2 ? br_tile_geno(M, S, cx.block, NB) : br_tile(M, A, cx.block, NB);
    const int g0 = KIND == 2 ? S.tile_g[cx.block] : 0, g1 = KIND == 2 ? S.tile_g[cx.block + 1] : 0;
    const BRSeg* sg = (const BRSeg*)(lds + Y.seg);
    const int* li = (const int*)(lds + L.misc);
    br_tile_setup<KIND>(cx, M, S, A, Y, NB, -1);
    BB_PASS(cx, tid) {
        const int nseg = li[0];
        BRSt<P>& st = BB_PSTATE(stv, tid);
#pragma unroll
        for (int k = 0; k < P; ++k) {
            br_desc<KIND, P, AP>(M, Y, t, sg, nseg, g0, g1, tid + k * cx.nthr, st, k, lds);
            const long long i0 = st.i0[k];
            const int meta = st.meta[k];
            const bool a0 = meta & BRM_A0, a1 = meta & BRM_A1;
            st.mu[k] = br_load_pair<AP>(S.mu, i0, a0, a1);
            st.om[k] = br_load_pair<AP>(S.om, i0, a0, a1);
            st.am[k] = br_load_pair<AP>(S.acc_mu, i0, a0, a1);
            st.ao[k] = br_load_pair<AP>(S.acc_om, i0, a0, a1);
            st.lo[k] = bb_load_lo(S, i0, a0, a1);
            st.a[k] = st.h[k] = st.z[k] = st.lam[k] = st.gp[k] = bb_d2{0.0, 0.0};
        }
    }
    BB_STAMP_RT(cx, S, 10);         // (descriptors formed, state loads issued)
    BB_SYNC(cx);
}

// ---- next step's standard normals (out of line on the GPU, as in bb_persist.h) -----------------------------------
template <int P> struct BRIdx { long long i0[P]; int meta[P]; };

template <int P, bool MAYODD>
#ifdef BB_EMU
static inline
#else
__device__ __attribute__((noinline))
#endif
void br_draw_call(bb_d2* eps, int nthr, int tid, unsigned long long seed, unsigned step, unsigned stream,
                  long long i0a, long long i0b, long long i0c, long long i0d, int ma, int mb, int mc, int md) {
    // (the pair indices as scalar arguments: a struct of three or four of them went through scratch memory at every call -- C3's
    //  instance wrote and re-read 36 B per thread and step, 4 MB of the 21.9 MB the counters saw)
    BRIdx<P> ix;
    { const long long i0v[4] = {i0a, i0b, i0c, i0d}; const int mv[4] = {ma, mb, mc, md};
#pragma unroll
      for (int k = 0; k < P; ++k) { ix.i0[k] = i0v[k]; ix.meta[k] = mv[k]; } }
#pragma unroll
    for (int k = 0; k < P; ++k) {
        if (!(ix.meta[k] & BRM_VALID)) continue;
        // Where the loglambda block starts at an odd flat index (an odd number of latents in front of it) the thread's two
        // latents are the SECOND of one Philox pair and the FIRST of the next: two draws, in the exchange's shadow like the one.
        // One call site run once or twice (a second inlined copy enlarged the function's register footprint, which the step
        // loop pays for around the call: C3's instance 60 -> 96 spilled registers).
        const bool odd = MAYODD && (ix.i0[k] & 1);
        double e0 = 0.0, e1 = 0.0;
#pragma nounroll
        for (int rep = 0; rep < (odd ? 2 : 1); ++rep) {
            double a, b;
            bb_normal_pair(seed, (unsigned long long)(ix.i0[k] >> 1) + (unsigned long long)rep, step, stream, &a, &b);
            if (rep == 0) { e0 = odd ? b : a; e1 = b; }
            else e1 = a;
        }
        eps[k * nthr + tid] = bb_d2{e0, e1};              // read back by the same thread: no barrier needed
    }
}

template <int KIND, int P, bool AP = false>
BB_DEV void br_draw_ahead(BBCtx& cx, const RunArgs& A, const BRLay& Y, BRSt<P>* stv, unsigned long long step, unsigned stream = 0u) {
    bb_d2* eps = (bb_d2*)(cx.lds + Y.eps);
    BB_PASS(cx, tid) {
        BRSt<P>& st = BB_PSTATE(stv, tid);
        static_assert(P <= 4, "br_draw_call takes four pair slots");
        br_draw_call<P, AP>(eps, cx.nthr, tid, A.seed, (unsigned)step, stream, st.i0[0], P > 1 ? st.i0[P > 1 ? 1 : 0] : 0, P > 2 ? st.i0[P > 2 ? 2 : 0] : 0,
                            P > 3 ? st.i0[P > 3 ? 3 : 0] : 0, st.meta[0], P > 1 ? st.meta[P > 1 ? 1 : 0] : 0, P > 2 ? st.meta[P > 2 ? 2 : 0] : 0, P > 3 ? st.meta[P > 3 ? 3 : 0] : 0);
    }
}

// stage tables of the unit kinds (BRLay.st): raw sample, transformed form (-1: none)
template <int KIND> BB_DEV int br_stage_raw(int kind) {
    if (KIND <= 1) return kind == SK_S ? 0 : 2;
    return kind == SK_TH_R ? 3 : (kind == SK_TT_R ? 0 : (kind == SK_LT_R ? 4 : 5));
}
template <int KIND> BB_DEV int br_stage_trn(int kind) {
    if (KIND <= 1) return kind == SK_LS_E ? 1 : -1;
    return kind == SK_LT_R ? 1 : (kind == SK_LS_R ? 2 : -1);
}
// effective fitness and precision of the unit with stage index o (theta index o - thoff), from the stage buffer `sb` of this step
template <int KIND>
BB_DEV void br_unit_sw(const double* lds, const BRLay& Y, int buf, int o, int thoff, double* s, double* w) {
    const double* b = lds + buf * Y.SU;
    if (KIND <= 1) { *s = b[BR_ST(Y, 0) + o]; *w = b[BR_ST(Y, 1) + o]; }
    else { *s = fma(b[BR_ST(Y, 1) + o], b[BR_ST(Y, 0) + o], b[BR_ST(Y, 3) + o - thoff]); *w = b[BR_ST(Y, 2) + o]; }
}

// ---- S: draw, stage ---------------------------------------------------------------------------------------------------
// The long fp64 chains (softplus / sigmoid, exp) run for ALL pair slots without a branch, so that the compiler may interleave
// the slots' chains; only the stores depend on what the pair is.
// A pair's entry in a row of the TruncatedADAGrad window = flat index - its segment's difference (DevModel.Dh, bb_hdelta).  Only a
// SHARDED handle has differences, and only the cross-GPU instances (HD = XG) run on one: the single-GPU instances skip the LDS read
// and the 64-bit subtraction at compile time (with them, or behind a run-time flag: C2 80.7 -> 78.5 / 79.2 k steps/s).
#ifdef BB_EMU
#define BR_HDELTA(HD, lds, Y, meta) (((const BRSeg*)((lds) + (Y).seg))[(meta) >> 12].pad)
#else
#define BR_HDELTA(HD, lds, Y, meta) ((HD) ? ((const BRSeg*)((lds) + (Y).seg))[(meta) >> 12].pad : 0)
#endif
template <int P, bool HD = false>
BB_DEV void br_prefetch_slot(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BRLay& Y, BRSt<P>* stv, int slot);
template <int KIND, int P> BB_DEV void br_pair_prior(const double* lds, const BRLay& Y, const BRSt<P>& st, int k, bool a0, bool a1, double* pm0, double* iv0, double* pm1, double* iv1);
// MS (instances that take several MC samples per step and record the ELBO): want_el -- this sample's ELBO terms are gathered
//   per latent      -(z - m)^2 / (2 v^2) + log sigma            (prior quadratic + entropy term; bb_sample_pair)
//   per (b, t)      R z - lambda                                  (Poisson; bb_pass_moments)
//   per unit        - logsigma_eff x (time steps that use it)      (normaliser of the fitness likelihood; bb_effective_tables)
// into BRSt.el; br_moments sums them over the tile (row entry K - 2), br_finish adds the per-time terms and stores the estimate.
template <int KIND, int P, bool MS = false, bool HD = false>
BB_DEV void br_sample(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BRLay& Y, BRSt<P>* stv, int buf, int slot, bool prefetch = true, bool want_el = false) {
    double* lds = cx.lds;
    const BBLds& L = Y.L;
    BB_STAMP(cx, S, 20);
    BB_STAMP_WAVE(cx, S, A, 1);
    if (A.pf == 1 && prefetch) br_prefetch_slot<P, HD>(cx, M, S, A, Y, stv, slot);
    BB_PASS(cx, tid) {
        BRSt<P>& st = BB_PSTATE(stv, tid);
#pragma unroll
        for (int k = 0; k < P; ++k) {
            // a pair slot that no lane of this wave uses (the tail of the tile's last slot) costs nothing
#ifdef BB_EMU
            if (!(st.meta[k] & BRM_VALID)) continue;
#else
            if (P > 1 && __builtin_amdgcn_ballot_w64((st.meta[k] & BRM_VALID) != 0) == 0ull) continue;
#endif
            const bb_d2 e = ((const bb_d2*)(lds + Y.eps))[k * cx.nthr + tid];
            double sp0, sg0, sp1, sg1;
            bb_softplus_sigmoid(st.om[k].x, &sp0, &sg0);
            bb_softplus_sigmoid(st.om[k].y, &sp1, &sg1);
            st.z[k] = bb_d2{fma(sp0, e.x, st.mu[k].x), fma(sp1, e.y, st.mu[k].y)};
            st.a[k] = bb_d2{e.x * sg0, e.y * sg1};
            st.h[k] = bb_d2{sg0 * bb_rcp(sp0), sg1 * bb_rcp(sp1)};
            if (MS && want_el) {
                if (k == 0) st.el = 0.0;
                const int meta = st.meta[k], kd = meta & 15;
                const bool counted = (meta & BRM_VALID) && (kd < SK_GS || A.count_globals);      // (sharded run: rank 0 counts the replicated blocks)
                if (counted) {
                    const bool a0 = meta & BRM_A0, a1 = meta & BRM_A1;
                    double pm0, iv0, pm1, iv1;
                    br_pair_prior<KIND>(lds, Y, st, k, a0, a1, &pm0, &iv0, &pm1, &iv1);
                    if (a0) st.el += -0.5 * (st.z[k].x - pm0) * (st.z[k].x - pm0) * iv0 + bb_log(sp0);
                    if (a1) st.el += -0.5 * (st.z[k].y - pm1) * (st.z[k].y - pm1) * iv1 + bb_log(sp1);
                }
            }
            BR_SCHED_FENCE();
            // loglambda: lambda = e^z; logsigma_bc: precision w = e^{-2 z}; (others: unused)
            const int kd = st.meta[k] & 15;
            const double f = (kd == SK_LS_E || (KIND >= 2 && kd == SK_LS_R)) ? -2.0 : 1.0;      // logtau: e^{logtau}
#if BR_SKIP_EXP && !defined(BB_EMU)
            // a wave whose pairs are all s_bc / theta / theta_tilde -- nobody reads their e^z -- skips the chain (C2
            // 83.9 -> 84.7 k steps/s, C4 and C3 unchanged: profiles/r03g_parallel_leaders/skip_unused_exp.txt)
            if (__builtin_amdgcn_ballot_w64((st.meta[k] & BRM_VALID) && (kd == SK_L || (kd < SK_GS && br_stage_trn<KIND>(kd) >= 0))) == 0ull) st.lam[k] = bb_d2{0.0, 0.0};
            else
#endif
            st.lam[k] = bb_d2{bb_exp(f * st.z[k].x), bb_exp(f * st.z[k].y)};
            BR_SCHED_FENCE();
            if (MS && want_el && (st.meta[k] & BRM_VALID)) {
                const int meta = st.meta[k];
                if (kd == SK_L) {
                    st.el += (double)st.cnt[k][0] * st.z[k].x - st.lam[k].x;
                    if (meta & BRM_A1) st.el += (double)st.cnt[k][1] * st.z[k].y - st.lam[k].y;
                } else if (kd == SK_LS_E || (KIND >= 2 && kd == SK_LS_R)) {
                    // the unit's log sigma, once per time step that uses the unit (multienv kinds: the steps INTO its environment)
                    const BRSeg* sgk = (const BRSeg*)(lds + Y.seg) + (meta >> 12);
                    const int T1 = sgk->T - 1;
#pragma unroll
                    for (int x = 0; x < 2; ++x) {
                        if (!(meta & (x ? BRM_A1 : BRM_A0))) continue;
                        int n = T1;
                        if (KIND == 1 || KIND == 4) {
                            const int* envt = (const int*)(lds + Y.envt);
                            const int e_ = (st.uo[k][2] >> (8 * x)) & 255, tc = KIND == 4 ? M.tcum[st.pt[k]] : 0;
                            n = 0;
                            for (int tt = 0; tt < T1; ++tt) n += envt[tc + tt + 1] == e_ ? 1 : 0;
                        }
                        st.el -= (x ? st.z[k].y : st.z[k].x) * (double)n;
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int meta = st.meta[k];
            if (!(meta & BRM_VALID)) continue;
            const int kind = meta & 15;
            if (kind == SK_L) {
                double* zw = lds + Y.zl + buf * Y.NBT + st.zoff[k];                // (rows are T + 1 apart: 8-byte aligned only)
                zw[0] = st.z[k].x;
                zw[1] = st.z[k].y;
            } else if (kind < SK_GS) {
                // unit latents: the raw sample (the G pass needs it for the prior term) and, where the (b, t) owners need another
                // form, that form: logsigma -> w = e^{-2 logsigma}, logtau -> e^{logtau}
                const int raw = br_stage_raw<KIND>(kind), trn = br_stage_trn<KIND>(kind);
                double* dst = lds + BR_ST(Y, raw) + buf * Y.SU + st.zoff[k];
                if (meta & BRM_A0) dst[0] = st.z[k].x;
                if (meta & BRM_A1) dst[1] = st.z[k].y;
                if (trn >= 0) {
                    double* dw = lds + BR_ST(Y, trn) + buf * Y.SU + st.zoff[k];
                    if (meta & BRM_A0) dw[0] = st.lam[k].x;
                    if (meta & BRM_A1) dw[1] = st.lam[k].y;
                }
            } else {      // replicated global latents (tile 0 only): they ride along in the tile's row, every other tile adds +0.0
                double* dst = lds + L.wk + M.K + (kind == SK_GLS ? M.nt1 : 0) + st.zoff[k];
                if (A.count_globals) {     // (sharded run: rank 0's draw is THE draw)
                    if (meta & BRM_A0) dst[0] = st.z[k].x;
                    if (meta & BRM_A1) dst[1] = st.z[k].y;
                }
            }
        }
    }
    BB_STAMP_WAVE(cx, S, A, 0);
#ifndef BB_EMU
    if (BR_UNIT_PRIO == 2) __builtin_amdgcn_s_setprio(0);
#endif
    BB_SYNC(cx);                     // barrier 1: neighbours' z and the unit stages are visible
    BB_STAMP(cx, S, 21);
}

// sum of s over the 16 lanes of a DPP row (every lane gets it)
#ifndef BB_EMU
template <int CTRL>
__device__ __forceinline__ double br_dpp(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double br_row16_sum(double s) {
    s += br_dpp<0x128>(s);   // row_ror:8
    s += br_dpp<0x124>(s);   // row_ror:4
    s += br_dpp<0x122>(s);   // row_ror:2
    s += br_dpp<0x121>(s);   // row_ror:1
    return s;
}
#endif

// ---- M: the pairs' differences and moment contributions, summed in the thread over its pair slots (they share the time pair),
// transposed into LDS; after the barrier 16 lanes per row entry walk their column and a DPP row sum finishes the entry, which
// goes straight to the tile's published row ------------------------------------------------------------------------------------
// ---- the tile's row from the transposed contributions (after barrier 2): 16 lanes per row entry walk their column, a DPP row sum
// finishes the entry; the row goes out (self-validating entries: from LDS as whole lines; else sc1 stores + drain + ready word) ----
template <int P, bool TG = false, bool MS = false>
BB_DEV void br_row_publish(BBCtx& cx, const DevModel& M, const DevState& S, const BRLay& Y, BRSt<P>* stv, unsigned epoch, bool want_el = false) {
    double* lds = cx.lds;
    BB_STAMP(cx, S, 23);
    const int KK = M.K + 2 * M.nt1, KS = bb_row_stride(KK);
    const int* rm = (const int*)(lds + Y.rowmap);
    const int nbt = ((const int*)(lds + Y.L.misc))[2];          // barcodes of this tile: column entries per time-pair class
    BB_PASS(cx, tid) {
        const int c = tid & 15;
        if (MS && tid == 0) {
            double e = 0.0;
            if (want_el) {
#ifdef BB_EMU
                if (stv) for (int t2 = 0; t2 < cx.nthr; ++t2) e += BB_PSTATE(stv, t2).el;
                else for (int w = 0; w < (cx.nthr + 63) / 64; ++w) e += lds[Y.L.part + w];          // (k_stream: no register state; bs_moments left the sums)
#else
                for (int w = 0; w < (cx.nthr >> 6); ++w) e += lds[Y.L.part + w];
#endif
            }
            if (TG && BR_PUB_COALESCED) lds[Y.L.wk + M.K - 2] = e;
            else if (TG) bb_gran_st(S.grow + (long long)cx.block * KS + M.K - 2, e, epoch);
            else bb_st<true>(S.prow + (long long)cx.block * KK + M.K - 2, e);
        }
        for (int j = tid >> 4; j < M.K; j += cx.nthr >> 4) {
            if (MS && j == M.K - 2) continue;
            const int code = rm[2 * j], lpb = code & 255;
            const double* col = lds + rm[2 * j + 1];
            double s = 0.0;
#ifdef BB_EMU
            if (c == 0 && (code >> 24)) for (int e = 0; e < nbt; ++e) s += col[e * lpb];
#else
            // (More reads in flight per lane make this walk SLOWER, three forms tried -- source-level unrolling twice, one asm block of
            //  four ds_read_b64: pub 2.26 k -> 2.71 k cycles, profiles/r03b_tagged_rows/column_walk_*: it is bound by the LDS pipeline
            //  under the stride-lpb bank pattern, not by the chain of round trips.)
            if (code >> 24) for (int e = c; e < nbt; e += 16) s += col[e * lpb];
            s = br_row16_sum(s);
#endif
            if (c == 0) {
                if (TG && BR_PUB_COALESCED) lds[Y.L.wk + j] = s;
                else if (TG) bb_gran_st(S.grow + (long long)cx.block * KS + j, s, epoch);
                else bb_st<true>(S.prow + (long long)cx.block * KK + j, s);
            }
        }
        if (!(TG && BR_PUB_COALESCED)) for (int j = M.K + tid; j < KK; j += cx.nthr) {
            if (TG) bb_gran_st(S.grow + (long long)cx.block * KS + j, lds[Y.L.wk + j], epoch);
            else bb_st<true>(S.prow + (long long)cx.block * KK + j, lds[Y.L.wk + j]);
        }
    }
    // (Measured and dropped, round 3, profiles/r03b_tagged_rows: a light "sentinel" first stage of the polls -- one entry per row until it carries the tag, then the sweep:
    //  C2 78.8 -> 75.2 k steps/s, the extra round trip costs more than the lighter polling saves.)
    if (TG) {
        // self-validating entries: no drain, no ready word.  The barrier stays for LDS alone: the contributions' region is about
        // to take the next normals and the window slot
        BB_SYNC(cx);
        if (BR_PUB_COALESCED) {
            // the row leaves as whole lines from the first waves (one store instruction per 64 entries) instead of one 16-byte
            // partial line write per entry from a dozen waves
            // Same-XCD first hop: a tile's row is read by ONE tile, its group's leader -- and the dispatcher deals workgroups to the XCDs in
            // turn, so leader g and its members g + NG j (NG = 8, 16, 32) share an XCD and with it an L2.  A PLAIN store leaves the entry in
            // that L2, where the leader's sc1 polls find it after an L2 round trip instead of a trip through the fabric: C2 11.35 -> 10.89 us per
            // step (profiles/r04i_same_xcd_first_hop).  Nothing rests on the dispatch order: a member stores write-through (sc1) until it has
            // read, from its leader's entry of this launch, that both run on the same XCC, and for good if they do not.
            int* li = (int*)(lds + Y.L.misc);
            BB_PASS(cx, tid) {
                if (BR_ROW_L2 && li[4] > 0) { for (int j = tid; j < KK; j += cx.nthr) bb_gran_st_l2(S.grow + (long long)cx.block * KS + j, lds[Y.L.wk + j], epoch); }
                else for (int j = tid; j < KK; j += cx.nthr) bb_gran_st(S.grow + (long long)cx.block * KS + j, lds[Y.L.wk + j], epoch);
                if (BR_ROW_L2 && tid == 0 && li[4] == 0) {
                    int g = cx.block;
                    while (g >= li[7]) g -= li[7];          // (at most 31 rounds, the first steps of a launch only)
                    const unsigned long long v = bb_get_word64(S.xtab + g);
                    if ((unsigned)(v >> 32) == (unsigned)li[6]) { li[4] = (int)(unsigned)v == li[5] ? 1 : -1; S.xsel[cx.block] = li[4]; }
                }
            }
        }
        BB_STAMP(cx, S, 24);
        BB_STAMP_RT(cx, S, 29);
        return;
    }
    bb_drain_and_meet(cx);           // every storing wave has drained its write-through stores
    BB_STAMP(cx, S, 24);
    BB_STAMP_RT(cx, S, 29);
    BB_PASS(cx, tid) { if (tid == 0) bb_set_word(S.rdy + 32 * cx.block, epoch); }
}

template <int KIND, int P, int TT> BB_DEV void br_theta_pre(BBCtx& cx, const DevModel& M, const BRLay& Y, BRSt<P>* stv, int buf);
template <int KIND, int P, bool TG = false, bool MS = false>
BB_DEV void br_moments(BBCtx& cx, const DevModel& M, const DevState& S, const BRLay& Y, BRSt<P>* stv, int buf, unsigned epoch, bool want_el = false) {
    double* lds = cx.lds;
    // (genotype / replicate model: the units' terms of d/dtheta, by their theta_tilde threads -- idle in this pass -- in front of barrier 2, so
    //  that the theta threads can add them up in the exchange's shadow: br_theta_pre, br_theta_sum)
    if (KIND == 2) br_theta_pre<KIND, P, 0>(cx, M, Y, stv, buf);
    if (MS && want_el) {
        // the threads' ELBO terms: summed per wave (DPP rows, then the four rows in order), one partial per wave in LDS; the row
        // pass below adds the waves in order -> row entry K - 2
        BB_PASS(cx, tid) {
            BRSt<P>& st = BB_PSTATE(stv, tid);
#ifdef BB_EMU
            (void)st;                                  // (emulation: summed thread by thread below)
#else
            double e = br_row16_sum(st.el);
            const int lo = __double2loint(e), hi = __double2hiint(e);
            double w = 0.0;
#pragma unroll
            for (int r = 0; r < 4; ++r) w += __hiloint2double(__builtin_amdgcn_readlane(hi, 16 * r), __builtin_amdgcn_readlane(lo, 16 * r));
            if ((tid & 63) == 0) lds[Y.L.part + (tid >> 6)] = w;
#endif
        }
    }
    BB_PASS(cx, tid) {
        BRSt<P>& st = BB_PSTATE(stv, tid);
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int meta = st.meta[k];
            if ((meta & 15) != SK_L) continue;
            double cv[BR_NCV];
#pragma unroll
            for (int q = 0; q < BR_NCV; ++q) cv[q] = 0.0;
            if (meta & BRM_VALID) {
            const double* zb = lds + Y.zl + buf * Y.NBT + st.zoff[k];
            const bool hn = meta & BRM_NEXT, mut = meta & BRM_MUT;
            const double z0 = st.z[k].x, z1 = st.z[k].y;
            const double zn = hn ? zb[2] : z1;
            double dm = z1 - z0, dn = zn - z1;       // the pair's two forward differences (the backward one of z0 belongs to the previous pair)
            cv[0] += st.lam[k].x;
            cv[6] += st.lam[k].y;
            if (mut) {
                double sm, sn, wm, wn;
                br_unit_sw<KIND>(lds, Y, buf, st.uo[k][1], KIND >= 2 ? st.thoff[k] : 0, &sm, &wm);
                if (KIND == 1 || KIND == 4) br_unit_sw<KIND>(lds, Y, buf, st.uo[k][2], KIND >= 3 ? st.thoff[k] : 0, &sn, &wn);
                else { sn = sm; wn = wm; }
                dm -= sm; dn -= sn;
                cv[1] += wm; cv[2] += wm * dm; cv[3] += wm * dm * dm;
                if (hn) { cv[7] += wn; cv[8] += wn * dn; cv[9] += wn * dn * dn; }
            } else {
                cv[4] += dm; cv[5] += dm * dm;
                if (hn) { cv[10] += dn; cv[11] += dn * dn; }
            }
            }
            {   // (idle lanes of the segment write zeros: every column entry of its nbt * LPB lanes is fresh each step)
                const BRSeg* sgk = (const BRSeg*)(lds + Y.seg) + (meta >> 12);
                const int stride = sgk->rstride;
#pragma unroll
                for (int q = 0; q < BR_NCV; ++q) lds[st.rb[k] + q * stride] = cv[q];
            }
        }
    }
    BB_SYNC(cx);                     // barrier 2: the contributions are in LDS
    br_row_publish<P, TG, MS>(cx, M, S, Y, stv, epoch, want_el);
}

// ---- window-slot prefetch (in the exchange's shadow): LDS-DMA, 16 bytes per lane straight into LDS, no register held across the
// exchange.  Every pair slot fetches both halves of its 16-byte pair (the row is padded: an edge pair's outside half is a
// neighbour's or the padding, never used). --------------------------------------------------------------------------------------
template <int P, bool HD>
BB_DEV void br_prefetch_slot(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BRLay& Y, BRSt<P>* stv, int slot) {
    if (A.opt != 0) return;
    BB_PASS(cx, tid) {
        BRSt<P>& st = BB_PSTATE(stv, tid);
#pragma unroll
        for (int k = 0; k < P; ++k) {
            if (!(st.meta[k] & BRM_VALID)) continue;
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                const double* src = S.hist + ((long long)slot * 2 + which) * M.Dh + (st.i0[k] - BR_HDELTA(HD, cx.lds, Y, st.meta[k]));
                bb_d2* dst = (bb_d2*)(cx.lds + Y.hbuf) + (k * 2 + which) * cx.nthr;
#ifdef BB_EMU
                dst[tid] = bb_d2{src[0], src[1]};
#else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                                 (__attribute__((address_space(3))) void*)(dst + (tid & ~63)), 16, 0, BR_NT_LOAD);
#endif
            }
        }
    }
}

// ---- F: everything that depends only on the totals (tiny; ends with barrier 3) --------------------------------------------
template <int KIND, bool MS = false>
BB_DEV void br_finish(BBCtx& cx, const DevModel& M, const DevState& S, const BRLay& Y, const RunArgs* Ap = nullptr, bool want_el = false, int ring = 0, int smp = 0) {
    double* lds = cx.lds;
    const BBLds& L = Y.L;
    BB_STAMP(cx, S, 25);
    BB_STAMP_RT(cx, S, 30);
    BB_PASS(cx, tid) {
        for (int j = tid; j < M.Ttot; j += cx.nthr) {
            const int* ft = (const int*)(lds + Y.ftab) + 4 * j;
            const int mo = ft[0], zo = ft[1], hasprev = ft[2];
            double Dt = 0.0, c = 0.0, wb = 0.0;
            if (mo >= 0) {
                const double nn = (double)M.nn;
                const double* mm = lds + mo;
                const double M0 = mm[0], M1 = mm[1], N1 = mm[3], N2 = mm[4];
                const double sbar = lds[L.zgl + zo], ls = lds[L.zgl + M.nt1 + zo];
                wb = bb_exp(-2.0 * ls);
                c = lds[L.Lt + j + 1] - lds[L.Lt + j] - sbar;
                const double quadN = N2 - 2.0 * c * N1 + nn * c * c;
                Dt = (M1 - c * M0) + wb * (N1 - c * nn);
                lds[L.gglob + zo] = -Dt;
                lds[L.gglob + M.nt1 + zo] = wb * quadN - nn;
                if (MS && want_el) {
                    const double M2 = mm[2], quadM = M2 - 2.0 * c * M1 + c * c * M0;
                    lds[L.elbt + j] = -0.5 * (quadM + wb * quadN) - nn * ls;
                }
            } else if (MS && want_el) lds[L.elbt + j] = 0.0;
            lds[L.cc + j] = c;
            lds[L.wbar + j] = wb;
            lds[L.Dt + j] = Dt;
            // G_t / S_t with G_t = D_{t-1} - D_t (D == 0 at t == T-1 and before t == 0).  Ttot <= 64 <= nthr: all of this is wave 0,
            // whose LDS operations complete in order (D_{t-1} was stored by lane j-1 in the instruction above); the emulation runs
            // the threads in index order
            lds[Y.iG + j] = ((hasprev ? lds[L.Dt + j - 1] : 0.0) - Dt) * lds[L.invS + j];
        }
    }
    BB_SYNC(cx);                     // barrier 3
    BB_STAMP(cx, S, 26);
    if (MS && want_el && cx.block == 0) {
        // ELBO estimate of this sample (bb_block_update): tile and theta-block terms came back with the totals (row entries K - 2, K - 1)
        BB_PASS(cx, tid) {
            if (tid == 0) {
                const RunArgs& A = *Ap;
                double v = lds[L.wk + M.K - 2] + lds[L.wk + M.K - 1] + A.elbo_const;
                for (int j = 0; j < M.Ttot; ++j) v += lds[L.elbt + j];
                double* slot = S.elbo_ring + ring;
                *slot = (smp == 0 ? 0.0 : *slot) + v / (double)A.S;
            }
        }
    }
}

// ---- G: gradients from registers, prior, optimiser, window slot -----------------------------------------------------------
// Likelihood gradient of a loglambda pair's two latents, with r = a - c_t, a = dl - s_eff:
//   d/dl_t = (R_t - lambda_t) + lambda_t G_t / S_t + w r|_t... - prior
// Everything that needs no totals -- the counts, the prior term, w a of the three differences (mutants: w comes from the unit's
// own latent) -- is formed in the EXCHANGE'S SHADOW (br_grad_pre, after the tile's row is out; the SIMDs idle there otherwise) and
// waits in two registers; after the totals three fused multiply-adds per latent finish it.  Neutral barcodes' precision is a
// global latent's sample that comes back with the totals: their w a terms are formed afterwards.
template <int KIND, int P>
BB_DEV void br_pair_prior(const double* lds, const BRLay& Y, const BRSt<P>& st, int k, bool a0, bool a1, double* pm0, double* iv0, double* pm1, double* iv1) {
    // Vector form from the segment (LDS, one address per wave mostly), Matrix form per element
    const BRSeg* sgk = (const BRSeg*)(lds + Y.seg) + (st.meta[k] >> 12);
    *pm0 = *pm1 = sgk->pm;
    *iv0 = *iv1 = sgk->iv;
    if (sgk->mean_e) {
        const long long j = st.i0[k] - sgk->blo;
        if (a0) { *pm0 = sgk->mean_e[j]; *iv0 = sgk->iv_e[j]; }
        if (a1) { *pm1 = sgk->mean_e[j + 1]; *iv1 = sgk->iv_e[j + 1]; }
    }
}
// the three differences of the pair's latents with their neighbours (minus the units' s_eff for mutants) and the units' precisions
template <int KIND, int P>
BB_DEV void br_pair_diffs(const double* lds, const BRLay& Y, const BRSt<P>& st, int k, int buf, double* z0, double* z1,
                          double* ap, double* am, double* an, double* wp, double* wm, double* wn) {
    const int meta = st.meta[k];
    const bool hp = meta & BRM_PREV, hn = meta & BRM_NEXT;
    const double* zb = lds + Y.zl + buf * Y.NBT + st.zoff[k];
    *z0 = zb[0]; *z1 = zb[1];
    const double zp = hp ? zb[-1] : *z0, zn = hn ? zb[2] : *z1;
    *ap = *z0 - zp; *am = *z1 - *z0; *an = zn - *z1;
    *wp = *wm = *wn = 0.0;
    if (meta & BRM_MUT) {
        double sp, sm, sn;
        br_unit_sw<KIND>(lds, Y, buf, st.uo[k][1], KIND >= 2 ? st.thoff[k] : 0, &sm, wm);
        if (KIND == 1 || KIND == 4) {
            br_unit_sw<KIND>(lds, Y, buf, st.uo[k][0], KIND >= 3 ? st.thoff[k] : 0, &sp, wp);
            br_unit_sw<KIND>(lds, Y, buf, st.uo[k][2], KIND >= 3 ? st.thoff[k] : 0, &sn, wn);
        } else { sp = sn = sm; *wp = *wn = *wm; }
        *ap -= sp; *am -= sm; *an -= sn;
    }
}
// (Only where the pair state leaves room: one pair slot per thread, fitness / multienv kinds -- C2 15.4 -> 15.1 us per step; with
//  three slots the two extra registers per slot spill, C3 22.7 -> 24.6 us.)
template <int KIND, int P> BB_DEV constexpr bool br_has_pre() { return P == 1 && KIND <= 1; }
template <int KIND, int P, bool AP = false>
BB_DEV void br_grad_pre(BBCtx& cx, const BRLay& Y, BRSt<P>* stv, int buf) {
    if (!br_has_pre<KIND, P>()) return;
    const double* lds = cx.lds;
    BB_PASS(cx, tid) {
        BRSt<P>& st = BB_PSTATE(stv, tid);
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int meta = st.meta[k];
            if ((meta & 15) != SK_L || !(meta & BRM_VALID)) continue;
            const bool hp = meta & BRM_PREV, hn = meta & BRM_NEXT;
            double pm0, iv0, pm1, iv1, z0, z1, ap, am, an, wp, wm, wn;
            br_pair_prior<KIND>(lds, Y, st, k, true, AP ? (st.meta[k] & BRM_A1) != 0 : true, &pm0, &iv0, &pm1, &iv1);
            br_pair_diffs<KIND>(lds, Y, st, k, buf, &z0, &z1, &ap, &am, &an, &wp, &wm, &wn);
            double A0 = ((double)st.cnt[k][0] - st.lam[k].x) - (z0 - pm0) * iv0;
            double A1 = ((double)st.cnt[k][1] - st.lam[k].y) - (z1 - pm1) * iv1;
            if (meta & BRM_MUT) {
                const double ma = (AP && !(meta & BRM_A1)) ? 0.0 : wm * am;        // (odd T: the last lane's single latent has no difference after it)
                A0 += ma - (hp ? wp * ap : 0.0);
                A1 += (hn ? wn * an : 0.0) - ma;
            }
            st.gp[k] = bb_d2{A0, A1};
        }
    }
}
// Genotype model, d/dtheta_g = sum over the genotype's mutants m of w_m As_m, As_m = sum_t r_t, r_t = (z_{t+1} - z_t) - s_eff - c_t:
//   w As = w [(z_{T-1} - z_0) - (T - 1) s_eff]  -  w sum_t c_t
// and only the c_t need the totals.  BR_TH_PRE = 1: every mutant's two terms w [..], w are formed IN THE EXCHANGE'S SHADOW (z rows and unit
// forms are in LDS since barrier 1) and left in LDS; at the start of the G pass the theta thread adds its members' up, d/dtheta =
// SA - SW sum_t c_t, and goes with all other pairs.  0 = round 3: the theta_tilde thread of every mutant leaves w As in LDS, one more
// barrier, the theta threads add their members up and update in a second pass -- every wave waited for that chain at the next step's
// barrier 1 (C5's rank shape, 25 000 x 8 / 626 genotypes: S 14.7 k cycles against C2's 10.0 k).
#ifndef BR_TH_PRE
#define BR_TH_PRE 1
#endif
// (replicate model, BR_TH_PRE3: the same for d/dtheta_m = sum over the replicates of w As of unit (m, r) -- the theta thread no longer walks
//  R barcode rows, it adds R entries up)
#ifndef BR_TH_PRE3
#define BR_TH_PRE3 1
#endif
template <int KIND> BB_DEV constexpr bool br_th_pre() { return (KIND == 2 && BR_TH_PRE) || (KIND == 3 && BR_TH_PRE3); }
template <int KIND, int P, int TT>
BB_DEV void br_theta_pre(BBCtx& cx, const DevModel& M, const BRLay& Y, BRSt<P>* stv, int buf) {
    if (!br_th_pre<KIND>()) return;
    double* lds = cx.lds;
    BB_PASS(cx, tid) {
        BRSt<P>& st = BB_PSTATE(stv, tid);
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int meta = st.meta[k];
            if ((meta & 15) != SK_TT_R || !(meta & BRM_VALID)) continue;
            // the theta_tilde thread of a unit (idle here) leaves the unit's two terms; its theta thread adds its members' up at the start
            // of the G pass (a theta thread walking its ~40 members' rows itself, here, took 12 k cycles and held its tile's exchange up:
            // profiles/r04c_theta_pre)
            const int* rt = (const int*)(lds + Y.rtab) + 4 * (KIND == 2 ? 0 : st.pt[k]);          // (unit pairs: pt = replicate)
            const int T = TT ? TT : rt[2];
            const double* zbuf = lds + Y.zl + buf * Y.NBT + (KIND == 2 ? Y.zr0[0] : rt[1]);
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                if (!(meta & (x ? BRM_A1 : BRM_A0))) continue;
                const int j = st.zoff[k] + x, bl = st.uo[k][x];
                const int th = KIND == 2 ? j - ((st.uo[k][2] >> (16 * x)) & 0xffff) : st.thoff[k];
                double sv, wv;
                br_unit_sw<KIND>(lds, Y, buf, j, th, &sv, &wv);
                const double* zr = zbuf + bl * (T + 1);
                *(bb_d2*)(lds + Y.gas + 2 * j) = bb_d2{wv * ((zr[T - 1] - zr[0]) - (double)(T - 1) * sv), wv};
            }
        }
    }
}

// ... and the theta threads add their members' entries up IN THE EXCHANGE'S SHADOW (BR_TH_SUM_F = 1): the entries are written during the M
// pass (the theta_tilde threads are idle there) and visible since barrier 2; the sums wait in two register pairs (gp, lam: a theta pair
// uses neither).  0: at the start of the G pass (C5's rank shape: S 13.7 k cycles -- the tile waits for that chain at barrier 1).
// (Beside the F pass instead -- entries written in the shadow, summed behind the consume's barrier: F 1.5 k -> 3.3 k cycles, 12.77 -> 11.9 us.)
#ifndef BR_TH_SUM_F
#define BR_TH_SUM_F 1
#endif
template <int KIND, int P>
BB_DEV void br_theta_sum(BBCtx& cx, const DevModel& M, const BRLay& Y, BRSt<P>* stv) {
    if (!(KIND == 2 && BR_TH_PRE && BR_TH_SUM_F)) return;
    const double* lds = cx.lds;
    BB_PASS(cx, tid) {
        BRSt<P>& st = BB_PSTATE(stv, tid);
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int meta = st.meta[k];
            if ((meta & 15) != SK_TH_R || !(meta & BRM_VALID)) continue;
            bb_d2 sa{0.0, 0.0}, sw{0.0, 0.0};
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                if (!(meta & (x ? BRM_A1 : BRM_A0))) continue;
                const int first = st.uo[k][x] & 0xffff, n = st.uo[k][x] >> 16;
                const bb_d2* ge = (const bb_d2*)(lds + Y.gas) + first;
                double a = 0.0, w = 0.0;
                int i = 0;
                for (; i + 4 <= n; i += 4) {          // (four 16-byte entries in flight per LDS round trip)
                    const bb_d2 e0 = ge[i], e1 = ge[i + 1], e2 = ge[i + 2], e3 = ge[i + 3];
                    a += (e0.x + e1.x) + (e2.x + e3.x);
                    w += (e0.y + e1.y) + (e2.y + e3.y);
                }
                for (; i < n; ++i) { a += ge[i].x; w += ge[i].y; }
                if (x) { sa.y = a; sw.y = w; } else { sa.x = a; sw.x = w; }
            }
            st.gp[k] = sa;
            st.lam[k] = sw;
        }
    }
}

template <int KIND, int P, bool AP = false, bool PRE = (P == 1 && KIND <= 1)>
BB_DEV void br_l_grad(const double* lds, const BRLay& Y, const BRSt<P>& st, int k, int buf, double* g0, double* g1) {
    const BBLds& L = Y.L;
    const int meta = st.meta[k], pt = st.pt[k];
    const bool hp = meta & BRM_PREV, hn = meta & BRM_NEXT;
    const double cp = hp ? lds[L.cc + pt - 1] : 0.0, cm = lds[L.cc + pt], cn = hn ? lds[L.cc + pt + 1] : 0.0;
    if (!PRE) {    // everything here, after the totals
        double pm0, iv0, pm1, iv1, z0, z1, ap, am, an, wp, wm, wn;
        br_pair_prior<KIND>(lds, Y, st, k, true, AP ? (st.meta[k] & BRM_A1) != 0 : true, &pm0, &iv0, &pm1, &iv1);
        br_pair_diffs<KIND>(lds, Y, st, k, buf, &z0, &z1, &ap, &am, &an, &wp, &wm, &wn);
        if (!(meta & BRM_MUT)) { wp = hp ? lds[L.wbar + pt - 1] : 0.0; wm = lds[L.wbar + pt]; wn = lds[L.wbar + pt + 1]; }
        const double rp_ = hp ? wp * (ap - cp) : 0.0, rm_ = (AP && !(meta & BRM_A1)) ? 0.0 : wm * (am - cm), rn_ = hn ? wn * (an - cn) : 0.0;
        const double l0 = st.lam[k].x, l1 = st.lam[k].y;
        *g0 = ((double)st.cnt[k][0] - l0) + l0 * lds[Y.iG + pt] + rm_ - rp_ - (z0 - pm0) * iv0;
        *g1 = ((double)st.cnt[k][1] - l1) + l1 * lds[Y.iG + pt + 1] + rn_ - rm_ - (z1 - pm1) * iv1;
        return;
    }
    double rp, rm, rn;               // what the totals add to w r of the three differences
    if (meta & BRM_MUT) {
        double s_, wp, wm, wn;
        br_unit_sw<KIND>(lds, Y, buf, st.uo[k][1], KIND >= 2 ? st.thoff[k] : 0, &s_, &wm);
        if (KIND == 1 || KIND == 4) {
            br_unit_sw<KIND>(lds, Y, buf, st.uo[k][0], KIND >= 3 ? st.thoff[k] : 0, &s_, &wp);
            br_unit_sw<KIND>(lds, Y, buf, st.uo[k][2], KIND >= 3 ? st.thoff[k] : 0, &s_, &wn);
        } else { wp = wn = wm; }
        rp = -(wp * cp); rm = -(wm * cm); rn = -(wn * cn);
    } else {
        double z0, z1, ap, am, an, wp, wm, wn;
        br_pair_diffs<KIND>(lds, Y, st, k, buf, &z0, &z1, &ap, &am, &an, &wp, &wm, &wn);
        rp = hp ? lds[L.wbar + pt - 1] * (ap - cp) : 0.0;
        rm = (AP && !(meta & BRM_A1)) ? 0.0 : lds[L.wbar + pt] * (am - cm);
        rn = hn ? lds[L.wbar + pt + 1] * (an - cn) : 0.0;
    }
    *g0 = fma(st.lam[k].x, lds[Y.iG + pt], st.gp[k].x) + (rm - rp);
    *g1 = fma(st.lam[k].y, lds[Y.iG + pt + 1], st.gp[k].y) + (rn - rm);
}

// TT > 0: the number of time points is a compile-time constant -- the unit threads then fetch their barcodes' whole rows at once
// (with the row walk as a runtime loop, one LDS round trip per time step, the unit waves' G pass took 13 k cycles against 4 k
// for the loglambda waves and the whole tile waited for them).
// MS instances, NS > 1 MC samples per step (AdvancedVI's ELBO estimator, Turing.ADVI(samples_per_step, ..), src/vi.jl:98): sample smp's
// gradient joins running sums; the last sample averages them, adds the entropy term and updates -- the arithmetic of bb_update_pair.
template <int KIND, int P, int TT = 0, bool AP = false, bool MS = false, bool HD = false>
BB_DEV void br_update(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BRLay& Y, BRSt<P>* stv,
                      const BBSlot wslot, int buf, int NBs, int smp = 0, int NS = 1) {
    double* lds = cx.lds;
    const BBLds& L = Y.L;
    // (Measured and dropped: letting a barcode's loglambda lanes also form its units' sums As, Qs -- a DPP sum over the LPB lanes,
    //  two numbers per unit through LDS, one more barrier -- so that the unit threads need not walk the barcode's row: the G pass
    //  is bound by the SIMDs' total VALU work, not by the unit waves; C2 15.3 -> 15.5 us per step, C3 unchanged.)
    BB_STAMP_WAVE(cx, S, A, 2);
#ifndef BB_EMU
    if (BR_UNIT_PRIO) {
        bool u = false;
#pragma unroll
        for (int k = 0; k < P; ++k) u = u || ((stv->meta[k] & BRM_VALID) && (stv->meta[k] & 15) != SK_L);
        if (__builtin_amdgcn_ballot_w64(u) != 0ull) __builtin_amdgcn_s_setprio(BR_UNIT_PRIO_LEVEL);
    }
#endif
    const double* zbuf = lds + Y.zl + buf * Y.NBT;
    // Genotype model: two passes.  Pass 0 updates everything but theta, and the theta_tilde thread of every mutant leaves w As in
    // LDS; after one more barrier the theta threads add up their genotypes' members (consecutive units of this tile) and update.
    for (int pass = 0; pass < ((KIND == 2 && !BR_TH_PRE) ? 2 : 1); ++pass) {
    if (pass) BB_SYNC(cx);
    BB_PASS(cx, tid) {
        BRSt<P>& st = BB_PSTATE(stv, tid);
        double* hs_m = nullptr;
        double* hs_o = nullptr;
        if (A.opt == 0) {
            hs_m = S.hist + ((long long)wslot.slot * 2 + 0) * M.Dh;
            hs_o = S.hist + ((long long)wslot.slot * 2 + 1) * M.Dh;
        }
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int meta = st.meta[k];
            if (!(meta & BRM_VALID)) continue;
            const int kind = meta & 15;
            if (KIND == 2 && !BR_TH_PRE && (kind == SK_TH_R) != (pass == 1)) continue;
            const bool a0 = meta & BRM_A0, a1 = meta & BRM_A1;
            double pm0 = 0.0, pm1 = 0.0, iv0 = 0.0, iv1 = 0.0;       // (loglambda: the prior term is in st.gp already)
            if (kind != SK_L) br_pair_prior<KIND>(lds, Y, st, k, a0, a1, &pm0, &iv0, &pm1, &iv1);
            double g0 = 0.0, g1 = 0.0, z0 = 0.0, z1 = 0.0;
            if (kind == SK_L) {
                br_l_grad<KIND, P, AP>(lds, Y, st, k, buf, &g0, &g1);
            } else if (KIND == 2 && kind == SK_TH_R) {
                const double* stg = lds + buf * Y.SU;
                double csum = 0.0;
                if (BR_TH_PRE) { const int T = TT ? TT : M.T[0]; for (int tt = 0; tt < T - 1; ++tt) csum += lds[L.cc + tt]; }
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    if (!(x ? a1 : a0)) continue;
                    double acc = 0.0;
                    const int first = st.uo[k][x] & 0xffff, n = st.uo[k][x] >> 16;
                    if (BR_TH_PRE && BR_TH_SUM_F) acc = (x ? st.gp[k].y : st.gp[k].x) - (x ? st.lam[k].y : st.lam[k].x) * csum;          // (br_theta_sum, beside the F pass)
                    else if (BR_TH_PRE) {          // (br_theta_pre: the members' terms w [..] and w, left in LDS before the exchange's barriers)
                        // (four 16-byte entries in flight per round trip: one after the other the ~40 members of a genotype are 40 LDS
                        //  round trips, 5 k cycles, that the whole tile waits for at the next barrier)
                        const bb_d2* ge = (const bb_d2*)(lds + Y.gas) + first;
                        double sa = 0.0, sw = 0.0;
                        int i = 0;
                        for (; i + 4 <= n; i += 4) {
                            const bb_d2 e0 = ge[i], e1 = ge[i + 1], e2 = ge[i + 2], e3 = ge[i + 3];
                            sa += (e0.x + e1.x) + (e2.x + e3.x);
                            sw += (e0.y + e1.y) + (e2.y + e3.y);
                        }
                        for (; i < n; ++i) { sa += ge[i].x; sw += ge[i].y; }
                        acc = sa - sw * csum;
                    } else for (int i = 0; i < n; ++i) acc += lds[Y.gas + first + i];
                    (x ? g1 : g0) = acc;
                    (x ? z1 : z0) = stg[BR_ST(Y, 3) + st.zoff[k] + x];
                }
            } else if (kind < SK_GS) {
                // Unit latents.  Per unit u = (mutant [, replicate] [, environment]) the sums over the time steps that use it,
                //   As = w sum r,  Qs = w sum r^2 - n,  r = dl - s_eff - c_t,
                // give  d/ds_bc = As, d/dlogsigma = Qs;  hierarchical: s_eff = theta + e^{logtau} theta_tilde, so
                //   d/dtheta_tilde = As e^{logtau},  d/dlogtau = As e^{logtau} theta_tilde,  d/dtheta = sum over replicates of As.
                const int* envt = (const int*)(lds + Y.envt);
                const double* stg = lds + buf * Y.SU;
                const int E = (KIND == 1 || KIND == 4) ? M.E : 1;
                double gx[2] = {0.0, 0.0}, zx[2] = {0.0, 0.0};
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    if (!(x ? a1 : a0)) continue;
                    const int j = st.zoff[k] + x;                      // stage index of the latent
                    const int e = KIND == 2 ? 0 : (st.uo[k][2] >> (8 * x)) & 255, bl = st.uo[k][x];
                    zx[x] = stg[BR_ST(Y, br_stage_raw<KIND>(kind)) + j];
                    // replicates whose rows the latent's gradient sums over: its own; theta: all of them
                    const bool is_th = KIND >= 3 && kind == SK_TH_R;
                    if (KIND == 3 && BR_TH_PRE3 && is_th) {          // (br_theta_pre: the units' terms w [..], w are in LDS since the exchange's shadow)
                        double acc = 0.0;
                        for (int r = 0; r < M.R; ++r) {
                            const int* rt = (const int*)(lds + Y.rtab) + 4 * r;
                            const int T = TT ? TT : rt[2];
                            double csum = 0.0;
                            for (int tt = 0; tt < T - 1; ++tt) csum += lds[L.cc + rt[0] + tt];
                            const bb_d2 ge = *(const bb_d2*)(lds + Y.gas + 2 * (r * NBs + j));
                            acc += ge.x - ge.y * csum;
                        }
                        gx[x] = acc;
                        continue;
                    }
                    const int r0 = KIND <= 1 ? 0 : (is_th ? 0 : st.pt[k]), r1 = KIND <= 1 ? 1 : (is_th ? M.R : r0 + 1);
                    double acc = 0.0;
                    for (int r = r0; r < r1; ++r) {
                        // stage index of the unit (bl's mutant, replicate r, environment e) and its s_eff, w
                        const int o = is_th ? (r * NBs + (j / E)) * E + e : j;
                        const int th = KIND <= 1 ? 0 : (KIND == 2 ? j - ((st.uo[k][2] >> (16 * x)) & 0xffff) : (is_th ? r * NBs * E : st.thoff[k]));
                        double sv, wv;
                        br_unit_sw<KIND>(lds, Y, buf, o, th, &sv, &wv);
                        // (per-replicate values from a small LDS table: indexed by a per-lane replicate in the model / layout records they were
                        //  dependent vector loads from device memory in front of every row walk -- C3's G pass)
                        const int* rt = (const int*)(lds + Y.rtab) + 4 * r;
                        const int T = TT ? TT : rt[2], tc = KIND <= 1 ? 0 : rt[0];
                        const double* zr = zbuf + (KIND <= 1 ? Y.zr0[0] : rt[1]) + bl * (T + 1);
                        double As = 0.0, Qs = 0.0;
                        int nn = 0;
                        // (Measured and dropped, round 4: d/ds_bc from the telescoped sum As = (z_{T-1} - z_0) - (T - 1) s - sum_t c_t -- two reads of
                        //  the barcode's row instead of T for the s_bc threads: the unit waves stay the last to leave the G pass, C2 11.87 -> 12.06 us,
                        //  profiles/r04d_c2_experiments)
                        if (TT) {
                            double zrow[TT ? TT : 1];
#pragma unroll
                            for (int tt = 0; tt < TT; ++tt) zrow[tt] = zr[tt];          // the whole row in flight at once
#pragma unroll
                            for (int tt = 0; tt < TT - 1; ++tt) {
                                const bool use = E == 1 || envt[tc + tt + 1] == e;
                                const double rr = use ? (zrow[tt + 1] - zrow[tt]) - sv - lds[L.cc + tc + tt] : 0.0;
                                As += rr; Qs += rr * rr; nn += use ? 1 : 0;
                            }
                        } else {
                            for (int tt = 0; tt < T - 1; ++tt) {
                                if (E > 1 && envt[tc + tt + 1] != e) continue;
                                const double rr = (zr[tt + 1] - zr[tt]) - sv - lds[L.cc + tc + tt];
                                As += rr; Qs += rr * rr; ++nn;
                            }
                        }
                        if (KIND <= 1) acc = kind == SK_S ? wv * As : wv * Qs - (double)nn;
                        else if (is_th) acc += wv * As;
                        else if (kind == SK_LS_R) acc = wv * Qs - (double)nn;
                        else if (kind == SK_TT_R) {
                            acc = wv * As * stg[BR_ST(Y, 1) + j];                                                // e^{logtau}
                            if (KIND == 2 && !BR_TH_PRE) lds[Y.gas + j] = wv * As;                           // d/ds_eff: its genotype's theta sums these
                        }
                        else acc = wv * As * stg[BR_ST(Y, 1) + j] * stg[BR_ST(Y, 0) + j];                            // logtau: e^{logtau} theta_tilde
                    }
                    gx[x] = acc;
                }
                g0 = gx[0]; g1 = gx[1]; z0 = zx[0]; z1 = zx[1];
            } else {
                // the replicated global latents' sample came back with the totals (rank 0's draw)
                const double* gg = lds + L.gglob + (kind == SK_GLS ? M.nt1 : 0) + st.zoff[k];
                const double* zz = lds + L.zgl + (kind == SK_GLS ? M.nt1 : 0) + st.zoff[k];
                if (a0) { g0 = gg[0]; z0 = zz[0]; }
                if (a1) { g1 = gg[1]; z1 = zz[1]; }
            }
            g0 -= (z0 - pm0) * iv0;
            g1 -= (z1 - pm1) * iv1;
            double go0, go1;
            if (MS && NS > 1) {
                go0 = g0 * st.a[k].x; go1 = g1 * st.a[k].y;
                if (smp > 0) { g0 += st.gm[k].x; g1 += st.gm[k].y; go0 += st.go[k].x; go1 += st.go[k].y; }
                if (smp < NS - 1) { st.gm[k] = bb_d2{g0, g1}; st.go[k] = bb_d2{go0, go1}; continue; }
                const double invS = 1.0 / (double)NS;
                g0 *= invS; g1 *= invS;
                go0 = go0 * invS + st.h[k].x; go1 = go1 * invS + st.h[k].y;
            } else { go0 = fma(g0, st.a[k].x, st.h[k].x); go1 = fma(g1, st.a[k].y, st.h[k].y); }
            bb_d2 hm{0, 0}, ho{0, 0};
            if (hs_m) {
#ifndef BB_EMU
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's LDS-DMA of the window slot has landed (br_prefetch_slot)
#endif
                hm = ((const bb_d2*)(lds + Y.hbuf))[(k * 2 + 0) * cx.nthr + tid];
                ho = ((const bb_d2*)(lds + Y.hbuf))[(k * 2 + 1) * cx.nthr + tid];
            }
            bb_d2 nhm = hm, nho = ho;
            const long long ih = st.i0[k] - BR_HDELTA(HD, lds, Y, meta);          // the pair's entry in a row of the window
            if (a0) {
                bb_opt_apply(M, S, A, wslot, 0, ih, -g0, hm.x, &nhm.x, &st.mu[k].x, &st.am[k].x, &st.lo[k].x);
                BR_SCHED_FENCE();
                bb_opt_apply(M, S, A, wslot, 1, ih, -go0, ho.x, &nho.x, &st.om[k].x, &st.ao[k].x, &st.lo[k].y);
                BR_SCHED_FENCE();
            }
            if (a1) {
                bb_opt_apply(M, S, A, wslot, 0, ih + 1, -g1, hm.y, &nhm.y, &st.mu[k].y, &st.am[k].y, &st.lo[k].z);
                BR_SCHED_FENCE();
                bb_opt_apply(M, S, A, wslot, 1, ih + 1, -go1, ho.y, &nho.y, &st.om[k].y, &st.ao[k].y, &st.lo[k].w);
                BR_SCHED_FENCE();
            }
            if (hs_m) { br_store_pair_stream<AP>(hs_m, ih, a0, a1, nhm); br_store_pair_stream<AP>(hs_o, ih, a0, a1, nho); }
        }
    }
    }
    // (the same thread has just read its entries of the slot buffer: its LDS-DMA of the next step's slot may overwrite them)
    if (A.pf == 2) br_prefetch_slot<P, HD>(cx, M, S, A, Y, stv, wslot.slot + 1 == A.W ? 0 : wslot.slot + 1);
#ifndef BB_EMU
    if (BR_UNIT_PRIO == 1) __builtin_amdgcn_s_setprio(0);
#endif
    BB_STAMP_WAVE(cx, S, A, 3);
    BB_STAMP(cx, S, 28);
}

// ---- epilogue: state back to memory, step counter, status words ----------------------------------------------------------
template <int KIND, int P, bool AP = false>
BB_DEV void br_epilogue(BBCtx& cx, const DevState& S, BRSt<P>* stv, unsigned long long step_end, bool timed_out) {
    BB_PASS(cx, tid) {
        BRSt<P>& st = BB_PSTATE(stv, tid);
        bool bad = false;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const int meta = st.meta[k];
            if (!(meta & BRM_VALID)) continue;
            const bool a0 = meta & BRM_A0, a1 = meta & BRM_A1;
            br_store_pair<AP>(S.mu, st.i0[k], a0, a1, st.mu[k]);
            br_store_pair<AP>(S.om, st.i0[k], a0, a1, st.om[k]);
            br_store_pair<AP>(S.acc_mu, st.i0[k], a0, a1, st.am[k]);
            br_store_pair<AP>(S.acc_om, st.i0[k], a0, a1, st.ao[k]);
            bb_store_lo(S, st.i0[k], a0, a1, st.lo[k]);
            const double chk = (a0 ? st.mu[k].x + st.om[k].x : 0.0) + (a1 ? st.mu[k].y + st.om[k].y : 0.0);
            bad = bad || !(chk - chk == 0.0);
        }
        if (bad) S.hstatus[1] = 1u;
        if (timed_out && tid == 0) S.hstatus[0] = 1u;
        if (cx.block == 0 && tid == 0) { S.ctr[0] = step_end; S.ctr[1] = step_end; }
    }
    BB_SYNC(cx);
}

// a tile's step between its moments and its update, in three parts (the emulation runs part 2 of all tiles between parts 1 and 3)
// `xc` numbers the exchanges of a handle's life: the step, or step x S + sample in the instances that take several samples per
// step; it gives the parity of the double-buffered tables and rows and the rows' epoch.
template <int KIND, int P, bool AP = false, bool HD = false>
BB_DEV void br_xchg_publish(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BRLay& Y, BRSt<P>* stv, unsigned long long xc, int slot,
                            unsigned long long next_step, unsigned next_stream = 0u, bool prefetch = true) {
    // (the tile's row went out at the end of br_moments)  The window slot first: LDS-DMA is slow to land (~3 k cycles for a
    // tile's 32 KB) and loads return in order, so it must be out of the way before this wave polls and reads the group rows
    if (A.pf == 0 && prefetch) br_prefetch_slot<P, HD>(cx, M, S, A, Y, stv, slot);
    br_grad_pre<KIND, P, AP>(cx, Y, stv, (int)(xc & 1));          // what of this step's gradient needs no totals
    br_theta_sum<KIND, P>(cx, M, Y, stv);          // (genotype model: the theta threads add their members' terms up -- left in front of barrier 2)
    // (replicate model: theta adds R entries at the start of the G pass; its units' terms are formed HERE -- in the M pass they cost C3 more
    //  than they save: M 2.0 k -> 2.9 k cycles, 17.1 -> 17.5 us)
    if (KIND == 3) br_theta_pre<KIND, P, 0>(cx, M, Y, stv, (int)(xc & 1));
    br_draw_ahead<KIND, P, AP>(cx, A, Y, stv, next_step, next_stream);         // the next normals, in the shadow of the rows' flight
}
template <bool XG>
BB_DEV void br_xchg_lead(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BRLay& Y, unsigned long long xc, int* ok_slot) {
    const unsigned epoch = A.xepoch0 + (unsigned)(xc + 1);
    // (fetching the members' rows in one round trip -- two waves, 16 members each, partial sums through LDS -- measured SLOWER,
    //  as round 1 had found for k_persist: 7.6 k cycles against 4.8 k for the leader's read + sum + publish)
    if (cx.block < bbp_groups(A)) {
        if (BR_TG) bbp_leader_reduce_tg<XG>(cx, M, S, A, Y.L, (int)(xc & 1), epoch, ok_slot);
        else bbp_leader_reduce<XG>(cx, M, S, A, Y.L, (int)(xc & 1), epoch, ok_slot, epoch);
    }
}
template <int KIND, int P, bool XG, bool MS = false>
BB_DEV void br_xchg_consume(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BRLay& Y, BRSt<P>* stv,
                            unsigned long long xc, int* ok_slot, bool want_el = false, int ring = 0, int smp = 0, int slot = -1) {
    const unsigned epoch = A.xepoch0 + (unsigned)(xc + 1);
    if (XG && BR_TG) bbp_consume_tgx(cx, M, S, A, Y.L, (int)(xc & 1), epoch, ok_slot);
    else if (BR_TG) bbp_consume_tg(cx, M, S, A, Y.L, (int)(xc & 1), epoch, ok_slot);
    else bbp_consume<XG, !XG>(cx, M, S, A, Y.L, (int)(xc & 1), epoch, ok_slot, epoch);
    // pf = 3 (BB_TUNE_PF only): the window slot fetched HERE, behind the exchange and in front of the F pass and the first gradients.
    // Measured on C3 (no LDS room for pf = 1): 52.7k steps/s against 56.3k with pf = 0 -- the fetch is exposed, not hidden
    if (A.pf == 3 && slot >= 0) br_prefetch_slot<P, XG>(cx, M, S, A, Y, stv, slot);
    br_finish<KIND, MS>(cx, M, S, Y, &A, want_el, ring, smp);
}

#ifndef BB_EMU
// MS = false: one MC sample per step, no ELBO recording (the lean instances every BASELINE shape runs).  MS = true: A.S >= 1 samples
// per step (each with its own exchange) and, every A.elbo_every steps, the ELBO estimate into the ring.
template <int KIND, int P, int NT, bool XG = false, int TT = 0, bool AP = false, bool MS = false>
__global__ void __launch_bounds__(NT) k_res(const DevModel* __restrict__ Mp, const DevState* __restrict__ Sp, const BRLay* __restrict__ Yp,
                                            RunArgs A, int NB, int nsteps) {
    const DevModel& M = *Mp;
    const DevState& S = *Sp;
    const BRLay& Y = *Yp;
    extern __shared__ __attribute__((aligned(16))) double br_smem[];
    BBCtx cx{(int)blockDim.x, (int)blockIdx.x, br_smem, nullptr};
    BRSt<P> st;
    int* ok_slot = (int*)(br_smem + Yp->L.misc) + 1;
    const unsigned long long c0 = S.ctr[0], c1 = S.ctr[1];
    unsigned long long step0 = c0 > c1 ? c0 : c1;
    // (uniform: tell the compiler, so that everything derived from the step number -- epoch, parity, window slot -- is scalar work)
    step0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(step0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)step0);
    BB_STAMP_RT(cx, S, 2);
    br_prologue<KIND, P, AP>(cx, M, S, A, Y, NB, &st);
    BB_STAMP_RT(cx, S, 3);
    const bool dead = *ok_slot == 0;
    int done = 0;
    if (!dead) {
        br_draw_ahead<KIND, P, AP>(cx, A, Y, &st, step0);
        BB_STAMP_RT(cx, S, 4);
        BBSlotCtr sc = bb_slot_init(A, step0);
        if (A.pf == 2) br_prefetch_slot<P, XG>(cx, M, S, A, Y, &st, sc.slot);
        const int NS = MS ? A.S : 1;
        // ELBO recording (MS): position inside the recording period and the ring slot, carried like the window slot
        int ec = (MS && A.elbo_every > 0) ? bb_uniform((int)(step0 % (unsigned long long)A.elbo_every)) : 1;
        int ring = (MS && A.elbo_every > 0) ? bb_uniform((int)(((step0 + (unsigned long long)A.elbo_every - 1ull) / (unsigned long long)A.elbo_every) % BB_ELBO_RING)) : 0;      // (of the next recording step)
        for (; done < nsteps; ++done, bb_slot_next(A, sc)) {
            const unsigned long long step = step0 + (unsigned long long)done;
            const BBSlot wslot = bb_slot_now(A, sc);
            const bool want_el = MS && A.elbo_every > 0 && ec == 0;
            bool stop = false;
            for (int smp = 0; smp < NS; ++smp) {
                const unsigned long long xc = MS ? step * (unsigned long long)NS + (unsigned long long)smp : step;
                const int buf = (int)(xc & 1);
                const bool last = smp == NS - 1;
                // Several pair slots: the pair descriptors are opaque to the compiler at every step.  Otherwise it hoists the ~15
                // predicates on each of them (kind, valid, mutant, has neighbour ...) out of the step loop as 64-bit lane masks --
                // scalar registers the loop does not have, so that they lived in VGPR lanes (two v_readlane_b32 per use) and took
                // vector registers from the pair state: C3's instance 58 -> 9 spilled registers, 22.4 -> 20.1 us per step.  (One
                // slot of the fitness / multienv kinds: no gain, -0.4 %.)
                if (P > 1 || KIND >= 2) {
#pragma unroll
                    for (int k = 0; k < P; ++k) asm volatile("" : "+v"(st.meta[k]));
                }
                br_sample<KIND, P, MS, XG>(cx, M, S, A, Y, &st, buf, wslot.slot, last, want_el);
                br_moments<KIND, P, BR_TG != 0, MS>(cx, M, S, Y, &st, buf, A.xepoch0 + (unsigned)(xc + 1), want_el);      // (tagged rows on the first hop of the sharded instances too)
                br_xchg_publish<KIND, P, AP, XG>(cx, M, S, A, Y, &st, xc, wslot.slot, last ? step + 1 : step, last ? 0u : (unsigned)(smp + 1), last);
                br_xchg_lead<XG>(cx, M, S, A, Y, xc, ok_slot);
                br_xchg_consume<KIND, P, XG, MS>(cx, M, S, A, Y, &st, xc, ok_slot, want_el, ring, smp, last ? wslot.slot : -1);
                if (*ok_slot == 0) { stop = true; break; }                 // uniform: read after barrier 3
                br_update<KIND, P, TT, AP, MS, XG>(cx, M, S, A, Y, &st, wslot, buf, NB, smp, NS);
            }
            if (stop) break;
            if (MS && A.elbo_every > 0) {
                if (ec == 0) ring = ring + 1 == BB_ELBO_RING ? 0 : ring + 1;
                ec = ec + 1 == A.elbo_every ? 0 : ec + 1;
            }
        }
    }
    BB_STAMP_RT(cx, S, 5);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (pf = 2: the slot fetched for a step this launch does not take has landed)
    br_epilogue<KIND, P, AP>(cx, S, &st, step0 + (unsigned long long)done, dead || *ok_slot == 0);
    BB_STAMP_RT(cx, S, 6);
}
#endif
