// bb_persist.h -- the ADVI step loop as ONE resident launch.
//
// The two-kernel step (bb_block.h) re-reads theta, the optimiser accumulators and the saved draw from
// the Infinity Cache every sweep: 184 B per latent per step against 96 algorithmic, and k_update is
// bandwidth-bound at the CU (profiles/r01b).  Here every workgroup keeps its tile's variational
// parameters and accumulators in REGISTERS and the staged samples / per-unit tables in LDS for the
// whole run; per step only the TruncatedADAGrad window slot streams through HBM (32 B per latent) and
// the K moment rows cross workgroups.  One grid barrier per step (all tiles resident: one workgroup per
// CU, grid <= CUs) sits where the kernel boundary was; the exchange follows the release/acquire recipe
// of the CDNA guide (write-through stores, every storing wave drains vmcnt, one lane arrives on an
// agent-scope counter, one lane polls relaxed with s_sleep, one agent acquire, then loads) and every
// spin is bounded (a timeout word makes all workgroups leave and bb_run report an error).
//
// The programme is split into prologue / sample / update / epilogue functions over an explicit
// per-thread state so that the host emulation can run the same source (sample for all tiles, then
// update for all tiles, per step).
#pragma once
#include "bb_block.h"
#include <string.h>

// The resident launch's LDS carve-up is computed once on the host and read from device memory where needed: recomputing it in
// every pass of every step (ten times ~70 scalar instructions) was a visible share of each wave's issue slots.
// (only where it pays: the 1024-thread instances, whose 128-VGPR / SGPR budget is the tight one; the 512-thread instances
// measured 2 % slower with it and keep computing)
template <int KIND>
BB_DEV BBLds bbp_layout(const BBCtx& cx, const DevModel& M, int NB) {
    if (cx.lay) return *(const BBLds*)cx.lay;
    return bb_lds_layout(M.R, M.E, KIND, M.Ttot, M.nt1, M.K, NB, cx.nthr, 1);
}

template <int P>
struct BBPst {
    bb_d2 mu[P], om[P], am[P], ao[P];   // variational parameters and optimiser accumulators of P pairs
    bb_d2 a[P], h[P];                   // current draw: eps*sigmoid(omega), sigmoid/softplus (z itself stays staged in LDS)
    bb_d2 hm[P], ho[P];                 // this step's TruncatedADAGrad window slot, fetched while the exchange is in flight
    long long i0[P];                    // first latent of each pair
    int meta[P];                        // segment index | a0 << 8 | a1 << 9 | valid << 10
    bb_f4 lo[P];                        // low-order parts of the four running window sums (bb_opt_apply)
};

#ifdef BB_EMU
#define BB_PSTATE(stv, tid) ((stv)[tid])
#else
#define BB_PSTATE(stv, tid) ((stv)[0])
#endif

struct BBPair { BBSeg s; long long i0; bool a0, a1, valid; };

BB_DEV BBPair bb_pair_of(const BBSeg* sg, int nseg, int p) {
    BBPair q;
    q.valid = p < sg[nseg].pbeg;
    int si = 0;
    if (q.valid) while (si + 1 < nseg && p >= sg[si + 1].pbeg) ++si;
    q.s = sg[si];
    q.i0 = 2 * ((q.s.lo >> 1) + (p - q.s.pbeg));
    q.a0 = q.valid && q.i0 >= q.s.lo;
    q.a1 = q.valid && q.i0 + 1 < q.s.hi;
    return q;
}

// the pair a thread owns never changes during a launch: found once, rebuilt from two registers afterwards
template <int P>
BB_DEV BBPair bb_pair_cached(const BBSeg* sg, const BBPst<P>& st, int k) {
    BBPair q;
    const int m = st.meta[k];
    q.s = sg[m & 255];
    q.i0 = st.i0[k];
    q.a0 = (m >> 8) & 1;
    q.a1 = (m >> 9) & 1;
    q.valid = (m >> 10) & 1;
    return q;
}

BB_DEV bb_d2 bb_load_pair(const double* base, long long i0, bool a0, bool a1) {
    if (a0 && a1) return *(const bb_d2*)(base + i0);
    return bb_d2{a0 ? base[i0] : 0.0, a1 ? base[i0 + 1] : 0.0};
}
BB_DEV void bb_store_pair(double* base, long long i0, bool a0, bool a1, bb_d2 v) {
    if (a0 && a1) { *(bb_d2*)(base + i0) = v; return; }
    if (a0) base[i0] = v.x;
    if (a1) base[i0 + 1] = v.y;
}

// per-thread LDS slots behind the layout's total: the drawn-ahead normals (bbp_draw_ahead, 16 B per pair), then the
// pairs' barcode counts (constants of the run; meaningful for loglambda pairs; 8 B per pair)
struct alignas(8) bb_u2 { unsigned x, y; };
BB_DEV bb_d2* bbp_eps(BBCtx& cx, const BBLds& L) { return (bb_d2*)(cx.lds + L.total); }
template <int P> BB_DEV bb_u2* bbp_cnt(BBCtx& cx, const BBLds& L) { return (bb_u2*)(cx.lds + L.total + 2 * P * cx.nthr); }

BB_DEV unsigned bb_get_word(const unsigned* word) {
#ifdef BB_EMU
    return *word;
#else
    return __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// ---- prologue: segment table, state into registers -------------------------------------------------
template <int KIND, int P>
BB_DEV void bbp_prologue(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, int NB, BBPst<P>* stv) {
    const BBLds L = bbp_layout<KIND>(cx, M, NB);
    double* lds = cx.lds;
    const BBTile t = bb_tile(M, A, cx.block, NB);
    BBSeg* sg = (BBSeg*)(lds + L.seg);
    int* li = (int*)(lds + L.misc);
    BB_PASS(cx, tid) {
        // li[1] = exchange ok word; it starts at 0 ("leave") while an earlier launch's timeout is unacknowledged by the host
        if (tid == 0) { li[0] = bb_build_segs<KIND>(sg, M, L, t, cx.block == 0); li[1] = bb_get_word(S.gbar + 1) == 0u ? 1 : 0; }
        for (int k = tid; k < M.K + 2 * M.nt1; k += cx.nthr) lds[L.wk + k] = 0.0;
        if (KIND <= 1)      // neutral units: no own fitness, no own precision (the mutants' entries are refreshed every step)
            for (int u = tid; u < NB * bb_xdim<KIND>(M); u += cx.nthr) { lds[L.seff + u] = 0.0; lds[L.weff + u] = 0.0; }
    }
    BB_SYNC(cx);
    BB_PASS(cx, tid) {
        BBPst<P>& st = BB_PSTATE(stv, tid);
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const BBPair q = bb_pair_of(sg, li[0], tid + k * cx.nthr);
            {
                int si = 0;
                while (si + 1 < li[0] && q.valid && tid + k * cx.nthr >= sg[si + 1].pbeg) ++si;
                st.i0[k] = q.i0;
                st.meta[k] = si | ((int)q.a0 << 8) | ((int)q.a1 << 9) | ((int)q.valid << 10);
            }
            st.mu[k] = bb_load_pair(S.mu, q.i0, q.a0, q.a1);
            st.om[k] = bb_load_pair(S.om, q.i0, q.a0, q.a1);
            st.am[k] = bb_load_pair(S.acc_mu, q.i0, q.a0, q.a1);
            st.ao[k] = bb_load_pair(S.acc_om, q.i0, q.a0, q.a1);
            st.lo[k] = bb_load_lo(S, q.i0, q.a0, q.a1);
            st.a[k] = st.h[k] = st.hm[k] = st.ho[k] = bb_d2{0.0, 0.0};
            bb_u2 c{0u, 0u};
            if (q.valid && q.s.kind == SK_L) {
                const long long base = M.cnt_off[q.s.r] + t.b0 * M.T[q.s.r] + (q.i0 - q.s.lo);
                if (q.a0) c.x = M.counts[base];
                if (q.a1) c.y = M.counts[base + 1];
            }
            bbp_cnt<P>(cx, L)[k * cx.nthr + tid] = c;
        }
    }
    BB_SYNC(cx);
}

// ---- next step's standard normals, drawn while the tile waits at the exchange -------------------------------
// The draw depends only on (seed, latent index, step), never on theta: Philox + Box-Muller (about half of the
// S pass) run in the exchange's shadow and wait in LDS (16 B per pair, behind the layout's total).  Kept out of
// line on purpose: inlined into the step loop it raised the kernel's spills from 45 to 143 VGPRs (-25 % steps/s);
// as a call the allocator treats it as its own region (24 spills, +10 % steps/s over drawing inside the S pass).

// (the pairs' indices travel BY VALUE: handing the callee a pointer to the register state would pin that whole
// struct in scratch memory for the entire launch)
template <int P> struct BBPairIdx { long long i0[P]; int meta[P]; };

template <int P>
#ifdef BB_EMU
static inline
#else
__device__ __attribute__((noinline))
#endif
void bbp_draw_ahead_call(bb_d2* eps, int nthr, int tid, unsigned long long seed, unsigned step, BBPairIdx<P> ix) {
#pragma unroll
    for (int k = 0; k < P; ++k) {
        if (!((ix.meta[k] >> 10) & 1)) continue;          // no pair in this slot
        double e0, e1;
        bb_normal_pair(seed, (unsigned long long)(ix.i0[k] >> 1), step, 0u, &e0, &e1);
        eps[k * nthr + tid] = bb_d2{e0, e1};              // read back by the same thread: no barrier needed
    }
}

template <int KIND, int P>
BB_DEV void bbp_draw_ahead(BBCtx& cx, const DevModel& M, const RunArgs& A, int NB, BBPst<P>* stv, unsigned long long step) {
    const BBLds L = bbp_layout<KIND>(cx, M, NB);
    bb_d2* eps = bbp_eps(cx, L);
    BB_PASS(cx, tid) {
        BBPst<P>& st = BB_PSTATE(stv, tid);
        BBPairIdx<P> ix;
#pragma unroll
        for (int k = 0; k < P; ++k) { ix.i0[k] = st.i0[k]; ix.meta[k] = st.meta[k]; }
        bbp_draw_ahead_call<P>(eps, cx.nthr, tid, A.seed, (unsigned)step, ix);
    }
}

// ---- the totals-independent part of every residual, a_tb = (l[t+1] - l[t]) - s_eff, tabulated while the rows fly -------
// (b, t) lanes as in the moment pass; read after the exchange by the R/U and G passes (two LDS reads per residual
// instead of four).  The ragged replicate method's neutral term depends on the sampled global latents, which only
// come back with the totals: that one case keeps forming its residuals inline.
template <int KIND>
BB_DEV void bbp_residual_ahead(BBCtx& cx, const DevModel& M, int NB, const RunArgs& A) {
    if (KIND == 3 && M.quirk) return;
    const BBLds L = bbp_layout<KIND>(cx, M, NB);
    const BBTile t = bb_tile(M, A, cx.block, NB);
    double* lds = cx.lds;
    const int X = bb_xdim<KIND>(M);
    for (int r = 0; r < M.R; ++r) {
        const int T = M.T[r], tc = M.tcum[r];
        const double* zl = lds + L.zl + NB * tc;
        BB_PASS(cx, tid) {
            for (int i = tid; i < t.nbt * T; i += cx.nthr) {
                const int bl = (int)bb_umulhi((unsigned)i, M.Tmagic[r]), tt = i - bl * T;
                if (tt < T - 1) {
                    double a = zl[i + 1] - zl[i];
                    if (bl >= t.nshift) a -= lds[L.seff + bl * X + bb_xof<KIND>(M, r, tt)];
                    lds[L.res + NB * tc + i] = a;
                }
            }
        }
    }
}

// ---- first half of a step: draw, stage, moments, publish the tile's K partial rows ---------------------
template <int KIND, int P>
BB_DEV void bbp_sample(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, int NB, BBPst<P>* stv,
                       unsigned long long step) {
    const BBLds L = bbp_layout<KIND>(cx, M, NB);
    double* lds = cx.lds;
    const BBTile t = bb_tile(M, A, cx.block, NB);
    const BBSeg* sg = (const BBSeg*)(lds + L.seg);
    const int* li = (const int*)(lds + L.misc);
    BB_STAMP(cx, S, 20);
    // (the tile's row lds[L.wk ..] was zeroed by the previous step's G pass / the prologue: no pass + barrier for it here)
    BB_PASS(cx, tid) {
        BBPst<P>& st = BB_PSTATE(stv, tid);
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const BBPair q = bb_pair_cached(sg, st, k);
            if (!q.valid) continue;
            double sp0, sg0, sp1, sg1;
            const bb_d2 e = bbp_eps(cx, L)[k * cx.nthr + tid];         // drawn during the previous exchange (bbp_draw_ahead)
            const double e0 = e.x, e1 = e.y;
            bb_softplus_sigmoid(st.om[k].x, &sp0, &sg0);
            bb_softplus_sigmoid(st.om[k].y, &sp1, &sg1);
            const double z0 = fma(sp0, e0, st.mu[k].x), z1 = fma(sp1, e1, st.mu[k].y);
            st.a[k] = bb_d2{e0 * sg0, e1 * sg1};
            st.h[k] = bb_d2{sg0 * bb_rcp(sp0), sg1 * bb_rcp(sp1)};
            if (q.s.kind >= SK_GS) {       // replicated global latents (tile 0 only): they ride along in the tile's
                double* dst = lds + L.wk + M.K + (q.s.kind == SK_GLS ? M.nt1 : 0);   // row, every other tile adds +0.0
                if (A.count_globals) {     // (sharded run: every rank's tile 0 holds the replicated blocks, rank 0's draw is THE draw)
                    if (q.a0) dst[q.i0 - q.s.lo] = z0;
                    if (q.a1) dst[q.i0 + 1 - q.s.lo] = z1;
                }
            } else {
                if (q.a0) lds[q.s.ldsoff + (q.i0 - q.s.lo)] = z0;
                if (q.a1) lds[q.s.ldsoff + (q.i0 + 1 - q.s.lo)] = z1;
                if (KIND <= 1) {
                    // fitness / multienv: a unit's effective fitness IS its s sample and its precision exp(-2 logsigma):
                    // the owning thread fills the tables here, the E pass and its barrier are not needed (the neutral
                    // units' zero entries were set once in the prologue)
                    const int X = bb_xdim<KIND>(M);
                    const long long u0 = (long long)t.nshift * X + (q.i0 - q.s.lo);
                    if (q.s.kind == SK_S) {
                        if (q.a0) lds[L.seff + u0] = z0;
                        if (q.a1) lds[L.seff + u0 + 1] = z1;
                    } else if (q.s.kind == SK_LS_E) {
                        if (q.a0) lds[L.weff + u0] = bb_exp(-2.0 * z0);
                        if (q.a1) lds[L.weff + u0 + 1] = bb_exp(-2.0 * z1);
                    }
                }
            }
        }
    }
    BB_SYNC(cx);
    BB_STAMP(cx, S, 21);
    if (KIND > 1) {
        BB_PASS(cx, tid) { bb_effective_tables<KIND>(cx, tid, M, S, L, t, false); }
        BB_SYNC(cx);
    }
    BB_STAMP(cx, S, 22);
    bb_pass_moments<KIND, true>(cx, M, S, L, t, NB, false);
    BB_STAMP(cx, S, 23);
    BB_STAMP(cx, S, 24);
}

// ---- window-slot prefetch: issued right after the tile has arrived at the exchange, so that the cold HBM
// lines (and their address translations) are in flight while the workgroup waits for the other tiles ----
template <int KIND, int P>
BB_DEV void bbp_prefetch_slot(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, int NB, BBPst<P>* stv,
                              int slot) {
    const BBLds L = bbp_layout<KIND>(cx, M, NB);
    double* lds = cx.lds;
    const BBSeg* sg = (const BBSeg*)(lds + L.seg);
    const int* li = (const int*)(lds + L.misc);
    BB_PASS(cx, tid) {
        BBPst<P>& st = BB_PSTATE(stv, tid);
        if (A.opt == 0) {
            const double* hs_m = S.hist + ((long long)slot * 2 + 0) * M.Dh;
            const double* hs_o = S.hist + ((long long)slot * 2 + 1) * M.Dh;
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const BBPair q = bb_pair_cached(sg, st, k);
                const long long ih = q.i0 - q.s.pad;
                st.hm[k] = bb_load_pair(hs_m, ih, q.a0, q.a1);
                st.ho[k] = bb_load_pair(hs_o, ih, q.a0, q.a1);
            }
        }
    }
}

// ---- exchange of the tiles' rows (K moment rows + the 2 nt1 sampled global latents) ----------------------------
// Two levels keep the bytes small and the summation order fixed: tile g (g < NG = min(8, tiles)) leads group
// g = {tiles b : b % NG == g} (on MI355X workgroups are dealt round-robin over the 8 XCDs, so a group shares an
// L2 -- speed only, never correctness), sums its members' rows in member order and publishes the group row;
// every tile then adds the NG group rows in group order.
// Hand-off (CDNA guide, R1 with the acquire replaced by sc1 loads): a row is stored write-through (sc1), every
// storing wave drains vmcnt, the workgroup meets at a barrier, ONE lane stores the row's ready word = this
// step's epoch (sc1).  A reader polls ONLY ready words (a few lanes: polling the rows themselves from 256 CUs
// was ~15 MB of memory-side traffic per round), then reads the row once with sc1 loads in a single batch.
// Ready words carry base + step + 1 (RunArgs.xepoch0; never 0 on fresh memory), which only ever grows over the life of a handle --
// no zeroing between launches, restarts bump the base; group
// rows and their ready words are double-buffered by step parity (a slow reader of step s must not meet step
// s+1's row); member rows need no double buffer (a member rewrites its row only after it has read every group
// row of the previous step, which the leaders publish only after reading all member rows).  Every poll is bounded.
typedef unsigned long long bb_u64;

#define BB_NG_MAX 32       /* groups of the exchange's first hop: RunArgs.ng = 8 (the cross-GPU inbox protocol is laid out for 8), 16, or -- self-validating rows only -- 32 */
BB_HD int bbp_groups(const RunArgs& A) { return A.nblk < A.ng ? A.nblk : A.ng; }

// Poll *word until it equals epoch; false = gave up (timeout word set).
BB_DEV bool bb_wait_word(const unsigned* word, unsigned epoch, unsigned* tmo, unsigned limit) {
#ifdef BB_EMU
    (void)tmo; (void)limit;
    return *word == epoch;           // the emulation runs the phases in order: the row must already be there
#else
    for (unsigned spins = 0; __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != epoch; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 1023u) == 1023u) {
            if (__hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || spins > limit) {
                __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
    return true;
#endif
}

BB_DEV void bb_drain_and_meet(BBCtx& cx) {
#ifndef BB_EMU
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // every storing wave: its write-through stores have landed
#endif
    BB_SYNC(cx);
}

BB_DEV void bb_set_word(unsigned* word, unsigned v) {
#ifdef BB_EMU
    *word = v;
#else
    __hip_atomic_store(word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// entries between the rows of two tiles in DevState.grow.  (Measured and dropped, round 4: every row on a 128-byte line of its own, stride
// rounded up to 8 entries -- C2's leaders then walk their 32 members' rows 8 KB apart, a power of two, and the step went 10.80 -> 10.93 us;
// rows of neighbouring tiles, which run on different XCDs, therefore share a line at their ends: each L2 writes back the bytes it dirtied.)
BB_HD int bb_row_stride(int KK) { return KK; }
BB_DEV void bb_set_word64(unsigned long long* word, unsigned long long v) {
#ifdef BB_EMU
    *word = v;
#else
    __hip_atomic_store(word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
BB_DEV unsigned long long bb_get_word64(const unsigned long long* word) {
#ifdef BB_EMU
    return *word;
#else
    return __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
// the XCD (accelerator complex die) this wave runs on: its workgroups share an L2
BB_DEV int br_xcc_id() {
#ifdef BB_EMU
    return 0;
#else
    return (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15u);          // hwreg(HW_REG_XCC_ID, 0, 4)
#endif
}

// ---- self-validating rows (k_res on one GPU, BR_TG) -------------------------------------------------------------------
// A row entry travels as ONE 16-byte write-through store {lo32, tag, hi32, tag}: two 8-byte granules that each carry the step's
// tag (= the ready words' epoch: base + step + 1, only ever growing).  A reader takes a value only when both tags match, so the
// row needs no ready word: the producer neither drains its stores nor meets at a barrier nor stores a flag, and a reader's poll
// of the row IS its read -- two round trips and two workgroup barriers less per hop than the ready-word protocol
// (tools/probe/xchg_probe.hip: 4.2 -> 3.7 us for the bare two-hop chain on an idle chip).  8-byte granules are written and read
// whole by the hardware (observed, as the CDNA guide's R2 form); the two halves of the 16-byte store may land apart -- hence a
// tag in each.  The loads are inline asm (no builtin gives a 16-byte sc1 load): the compiler does not know they are
// asynchronous, so the wait behind them takes the destination registers as operands and nothing that reads them can move up.
struct alignas(16) bb_gran { unsigned lo, t0, hi, t1; };
#ifndef BR_TG
#define BR_TG 1
#endif
#ifndef BB_EMU
typedef unsigned bb_v4u __attribute__((ext_vector_type(4)));
#endif
// SYS: system scope (sc0 sc1) -- an entry of a PEER's inbox, written over xGMI.  Each 8-byte half carries the tag, so the entry
// validates itself as long as the fabric keeps aligned 8-byte writes whole (every PCIe / xGMI transport does); no ordering between
// entries, or between the halves, is assumed.
template <bool SYS = false>
BB_DEV void bb_gran_st(bb_gran* p, double v, unsigned tag) {
#ifdef BB_EMU
    unsigned long long b;
    memcpy(&b, &v, 8);
    *p = bb_gran{(unsigned)b, tag, (unsigned)(b >> 32), tag};
#else
    bb_v4u g = {(unsigned)__double2loint(v), tag, (unsigned)__double2hiint(v), tag};
    if (SYS) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(g) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(g) : "memory");
#endif
}
// the same entry by a PLAIN store: the line stays (dirty) in the storing XCD's L2 -- only a reader on the SAME XCD ever sees it
BB_DEV void bb_gran_st_l2(bb_gran* p, double v, unsigned tag) {
#ifdef BB_EMU
    bb_gran_st(p, v, tag);
#else
    bb_v4u g = {(unsigned)__double2loint(v), tag, (unsigned)__double2hiint(v), tag};
    asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(g) : "memory");
#endif
}
// The entries base[off + i * stride], i < 8: poll until those with i < n carry `tag` in both halves, then their values in order
// (i >= n: +0.0; their loads still run -- the row buffers are allocated with the slack for it).  `base`, `stride` uniform, `off`
// per lane: ONE address register for the eight loads in flight (scalar base + 32-bit lane offset, advanced between the loads) --
// with eight 64-bit lane addresses beside the 32 destination registers the step loop of the 1024-thread instances spilled.  One
// call per thread with a row entry; the lanes of a wave leave together.  false = gave up.
template <bool SYS = false>
BB_DEV bool bb_gran_poll8(const bb_gran* base, unsigned off, unsigned stride, int n, unsigned tag, bool active, unsigned* tmo, unsigned limit, double* out) {
#ifdef BB_EMU
    (void)tmo; (void)limit;
    bool good = true;
    for (int i = 0; i < 8; ++i) {
        out[i] = 0.0;
        if (!active || i >= n) continue;
        const bb_gran g = base[off + i * stride];
        good = good && g.t0 == tag && g.t1 == tag;
        const unsigned long long b = (unsigned long long)g.lo | ((unsigned long long)g.hi << 32);
        memcpy(&out[i], &b, 8);
    }
    return good;                          // the emulation runs the phases in order: the rows must already be there
#else
    bb_v4u g0, g1, g2, g3, g4, g5, g6, g7;
    bool ok = true;
    // (uniform by construction; the readfirstlanes tell the compiler so -- the "s" operands must be scalar registers)
    const unsigned sb = (unsigned)__builtin_amdgcn_readfirstlane((int)(stride * 16u));
    const unsigned long long bp = (unsigned long long)base;
    const unsigned long long sbase = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(bp >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)bp);
    for (unsigned spins = 0;; ++spins) {
        unsigned vo = off * 16u;
#define BB_POLL8(SC) \
        asm volatile("global_load_dwordx4 %0, %8, %9 " SC "\n\t" "v_add_u32 %8, %10, %8\n\t" \
                     "global_load_dwordx4 %1, %8, %9 " SC "\n\t" "v_add_u32 %8, %10, %8\n\t" \
                     "global_load_dwordx4 %2, %8, %9 " SC "\n\t" "v_add_u32 %8, %10, %8\n\t" \
                     "global_load_dwordx4 %3, %8, %9 " SC "\n\t" "v_add_u32 %8, %10, %8\n\t" \
                     "global_load_dwordx4 %4, %8, %9 " SC "\n\t" "v_add_u32 %8, %10, %8\n\t" \
                     "global_load_dwordx4 %5, %8, %9 " SC "\n\t" "v_add_u32 %8, %10, %8\n\t" \
                     "global_load_dwordx4 %6, %8, %9 " SC "\n\t" "v_add_u32 %8, %10, %8\n\t" \
                     "global_load_dwordx4 %7, %8, %9 " SC "\n\t" \
                     "s_waitcnt vmcnt(0)" \
                     : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3), "=&v"(g4), "=&v"(g5), "=&v"(g6), "=&v"(g7), "+v"(vo) \
                     : "s"(sbase), "s"(sb) : "memory")
        if (SYS) BB_POLL8("sc0 sc1"); else BB_POLL8("sc1");
#undef BB_POLL8
        const bool good = ((g0.y == tag && g0.w == tag) || n < 1) && ((g1.y == tag && g1.w == tag) || n < 2) && ((g2.y == tag && g2.w == tag) || n < 3) &&
                          ((g3.y == tag && g3.w == tag) || n < 4) && ((g4.y == tag && g4.w == tag) || n < 5) && ((g5.y == tag && g5.w == tag) || n < 6) &&
                          ((g6.y == tag && g6.w == tag) || n < 7) && ((g7.y == tag && g7.w == tag) || n < 8);
        if (__builtin_amdgcn_ballot_w64(active && !good) == 0ull) break;
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 255u) == 255u) {
            if (__hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || spins > limit) {
                __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = false;
                break;
            }
        }
    }
#define BB_GV(g) __hiloint2double((int)g.z, (int)g.x)
    out[0] = n > 0 ? BB_GV(g0) : 0.0; out[1] = n > 1 ? BB_GV(g1) : 0.0; out[2] = n > 2 ? BB_GV(g2) : 0.0; out[3] = n > 3 ? BB_GV(g3) : 0.0;
    out[4] = n > 4 ? BB_GV(g4) : 0.0; out[5] = n > 5 ? BB_GV(g5) : 0.0; out[6] = n > 6 ? BB_GV(g6) : 0.0; out[7] = n > 7 ? BB_GV(g7) : 0.0;
#undef BB_GV
    return ok;
#endif
}

BB_DEV long long bbx_slot(const RunArgs& A, int par, int src, int g);
// leader of group g = tile g: its members' rows (16 at most per batch of two polls), summed in member order, out as the group row.
// XG: the first hop is the same (the members are this rank's own tiles); the group row then goes into EVERY rank's inbox as
// tagged entries written with system-scope stores (bb_gran_st<true>): no drain, no meet, no ready words on the cross-GPU hop either.
// PAR (the sharded instances: 8 groups, up to 33 members): the members go in chunks of eight to as many thread groups as the tile
// has, all polling at once; the chunk sums cross through LDS and are added in chunk order -- one round of polls instead of four.
#ifndef BR_LEAD_PAR
#define BR_LEAD_PAR 1          /* 1: the one-GPU instances too (measured: C2 80.8 -> 82.6 k steps/s, C4 90.7 -> 93.9 k at 16 groups; 0 = round-3 form before) */
#endif
template <bool XG = false>
BB_DEV void bbp_leader_reduce_tg(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BBLds& L, int par, unsigned epoch, int* ok) {
    const int KK = M.K + 2 * M.nt1, KS = bb_row_stride(KK), NG = bbp_groups(A), g = cx.block;
    const int members = (A.nblk - g + NG - 1) / NG;
    constexpr bool PAR = XG || BR_LEAD_PAR;
    const int KKP = (KK + 63) & ~63, chunks = (members + 7) >> 3;
    if (PAR && chunks > 1 && cx.nthr >= 2 * KKP) {
        double* lds = cx.lds;
        int NQ = cx.nthr / KKP;
        if (NQ > chunks) NQ = chunks;
        BB_PASS(cx, tid) {
            int q = 0;
            for (int t = KKP; t <= tid && q < NQ; t += KKP) ++q;
            const int kk = tid - q * KKP;
            if (q < NQ) {
                const bool act = kk < KK;
                const int k = act ? kk : KK - 1;
                for (int c = q; c < chunks; c += NQ) {
                    double v[8];
                    const int m0 = 8 * c, n = members - m0 < 8 ? members - m0 : 8;
                    if (!bb_gran_poll8(S.grow + (long long)g * KS, (unsigned)(m0 * NG * KS + k), (unsigned)(NG * KS), n, epoch, act, S.gbar + 1, A.spin_limit, v)) *ok = 0;
                    double s = 0.0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) s += v[i];
                    if (act) lds[L.red + c * KKP + kk] = s;
                }
            }
        }
        BB_SYNC(cx);
        BB_PASS(cx, tid) {
            if (tid < KK) {
                double s = lds[L.red + tid];
                for (int c = 1; c < chunks; ++c) s += lds[L.red + c * KKP + tid];
                if (XG) { for (int r = 0; r < A.world; ++r) bb_gran_st<true>(S.xgr[r] + bbx_slot(A, par, A.rank, g) * KK + tid, s, epoch); }
                else bb_gran_st(S.gxrow + ((long long)par * NG + g) * KK + tid, s, epoch);
            }
        }
        BB_SYNC(cx);          // (the chunk sums' LDS is the consume's as well; with separate regions and no barrier here C2 ran 1 % SLOWER:
                              //  the leader's other waves running ahead slow the chain everybody waits for -- as round 1 found for k_persist)
        BB_STAMP(cx, S, 18);
        return;
    }
    BB_PASS(cx, tid) {
        if (tid < KKP) {             // (whole waves: the lanes of a wave poll together)
            const bool act = tid < KK;
            const int k = act ? tid : KK - 1;
            double s = 0.0;
            for (int m0 = 0; m0 < members; m0 += 8) {
                double v[8];
                const int n = members - m0 < 8 ? members - m0 : 8;
                if (!bb_gran_poll8(S.grow + (long long)g * KS, (unsigned)(m0 * NG * KS + k), (unsigned)(NG * KS), n, epoch, act, S.gbar + 1, A.spin_limit, v)) *ok = 0;
#pragma unroll
                for (int i = 0; i < 8; ++i) s += v[i];
            }
            if (XG) {
                if (act) for (int r = 0; r < A.world; ++r) bb_gran_st<true>(S.xgr[r] + bbx_slot(A, par, A.rank, g) * KK + k, s, epoch);
            } else if (act) bb_gran_st(S.gxrow + ((long long)par * NG + g) * KK + k, s, epoch);
        }
    }
    BB_STAMP(cx, S, 18);
}

// every tile: the NG group rows, eight per thread group -- groups [8 q, 8 q + 8) by threads [q KKP, q KKP + KK), q < NG / 8 <= 4, all
// polling at once; the partial sums of q >= 1 cross through LDS and are added in group order: ((q0 + q1) + q2) + q3
BB_DEV void bbp_consume_tg(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BBLds& L, int par, unsigned epoch, int* ok) {
    double* lds = cx.lds;
    const int KK = M.K + 2 * M.nt1, NG = bbp_groups(A);
    const int KKP = (KK + 63) & ~63, NQ = (NG + 7) >> 3;
    double s0 = 0.0;
    BB_PASS(cx, tid) {
        const int q = (tid >= KKP ? 1 : 0) + (tid >= 2 * KKP ? 1 : 0) + (tid >= 3 * KKP ? 1 : 0) + (tid >= 4 * KKP ? 1 : 0), kk = tid - q * KKP;      // (no integer division in the step loop)
        if (q < NQ) {
            const bool act = kk < KK;
            const int k = act ? kk : KK - 1, g0 = 8 * q;
            double v[8];
            const int n = NG - g0 < 8 ? NG - g0 : 8;
            if (!bb_gran_poll8(S.gxrow + (long long)par * NG * KK, (unsigned)(g0 * KK + k), (unsigned)KK, n, epoch, act, S.gbar + 1, A.spin_limit, v)) *ok = 0;
            double s = 0.0;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += v[i];
            if (q > 0) { if (act) lds[L.red + (q - 1) * KKP + kk] = s; }
            else {
#ifdef BB_EMU
                if (act) lds[L.red + 3 * KKP + kk] = s;
#else
                s0 = s;
#endif
            }
        }
    }
    BB_STAMP(cx, S, 1);
    BB_SYNC(cx);
    BB_PASS(cx, tid) {
        if (tid < KK) {
#ifdef BB_EMU
            s0 = lds[L.red + 3 * KKP + tid];
#endif
            double s = s0;
            for (int q = 1; q < NQ; ++q) s += lds[L.red + (q - 1) * KKP + tid];
            if (tid < M.K) bb_put_total(M, L, lds, tid, s);
            else lds[L.zgl + (tid - M.K)] = s;
        }
    }
    if (!(KK <= 64 && M.Ttot <= 64)) BB_SYNC(cx);
}

// Sharded k_res: the 8 x world group rows of this rank's own inbox (tagged entries, written by every rank's leaders over xGMI, polled
// here as local memory with system-scope loads).  Rows come in chunks of eight -- chunk c = rows [8 c, 8 c + 8) = source rank c's
// eight groups -- each summed in row order by one thread group; the chunk sums cross through LDS and are added in chunk order:
// the order depends on nothing but (rank, group), so every rank forms bit-identical totals whatever its tile geometry.
BB_DEV void bbp_consume_tgx(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BBLds& L, int par, unsigned epoch, int* ok) {
    double* lds = cx.lds;
    const int KK = M.K + 2 * M.nt1, KKP = (KK + 63) & ~63, chunks = A.world;
    int NQ = cx.nthr / KKP;
    if (NQ > chunks) NQ = chunks;
    if (NQ < 1) NQ = 1;
    const bb_gran* in = S.xgr[A.rank] + (long long)par * A.world * 8 * KK;
    BB_PASS(cx, tid) {
        int q = 0;
        for (int t = KKP; t <= tid && q < NQ; t += KKP) ++q;             // (no integer division in the step loop)
        const int kk = tid - q * KKP;
        if (q < NQ) {
            const bool act = kk < KK;
            const int k = act ? kk : KK - 1;
            for (int c = q; c < chunks; c += NQ) {
                double v[8];
                if (!bb_gran_poll8<true>(in, (unsigned)(c * 8 * KK + k), (unsigned)KK, 8, epoch, act, S.gbar + 1, A.spin_limit, v)) *ok = 0;
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < 8; ++i) s += v[i];
                if (act) lds[L.red + c * KKP + kk] = s;
            }
        }
    }
    BB_STAMP(cx, S, 1);
    BB_SYNC(cx);
    BB_PASS(cx, tid) {
        if (tid < KK) {
            double s = lds[L.red + tid];
            for (int c = 1; c < chunks; ++c) s += lds[L.red + c * KKP + tid];
            if (tid < M.K) bb_put_total(M, L, lds, tid, s);
            else lds[L.zgl + (tid - M.K)] = s;
        }
    }
    if (!(KK <= 64 && M.Ttot <= 64)) BB_SYNC(cx);
}

// ---- cross-GPU leg (XG launches): a group leader stores its group row into EVERY rank's inbox (its own included)
// with system-scope write-through stores over xGMI, drains, then sets the row's ready word there; every tile then
// waits for the 8 x world rows of its own rank's inbox -- local memory -- and adds them in (rank, group) order, so all
// ranks form bit-identical totals.  Inbox words carry the ABSOLUTE step number + 1: they only ever grow, need no
// zeroing between launches (a peer may already be writing while this rank is still launching) and a slot's parity
// double buffer is safe for the same reason as within one GPU (nobody can publish step s + 2 before everybody has
// consumed step s).  Inboxes live in fine-grained memory, so neither side's L2 keeps a stale line.
BB_DEV void bb_st_sys(double* p, double v) {
#ifdef BB_EMU
    *p = v;
#else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}
BB_DEV double bb_ld_sys(const double* p) {
#ifdef BB_EMU
    return *p;
#else
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}
BB_DEV void bb_set_word_sys(unsigned* word, unsigned v) {
#ifdef BB_EMU
    *word = v;
#else
    __hip_atomic_store(word, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}
BB_DEV bool bb_wait_word_sys(const unsigned* word, unsigned epoch, unsigned* tmo, unsigned limit) {
#ifdef BB_EMU
    (void)tmo; (void)limit;
    return *word == epoch;
#else
    for (unsigned spins = 0; __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != epoch; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 1023u) == 1023u) {
            if (__hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || spins > limit) {
                __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
    return true;
#endif
}
// slot of (parity, source rank, group) in an inbox
BB_DEV long long bbx_slot(const RunArgs& A, int par, int src, int g) { return ((long long)par * A.world + src) * 8 + g; }

// this tile's row (complete in lds[L.wk .. + K + 2 nt1)) -> S.prow[b], then its ready word
BB_DEV void bbp_publish_row(BBCtx& cx, const DevModel& M, const DevState& S, const BBLds& L, unsigned epoch) {
    const int KK = M.K + 2 * M.nt1;
    BB_PASS(cx, tid) {
        for (int k = tid; k < KK; k += cx.nthr) bb_st<true>(S.prow + (long long)cx.block * KK + k, cx.lds[L.wk + k]);
    }
    if (KK <= 64) {
        // one wave stored the whole row: its own drain orders the ready word behind the row, no workgroup barrier
#ifndef BB_EMU
        if (threadIdx.x < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    } else {
        bb_drain_and_meet(cx);
    }
    BB_STAMP(cx, S, 7);
    BB_PASS(cx, tid) { if (tid == 0) bb_set_word(S.rdy + 32 * cx.block, epoch); }
}

// leader of group g = tile g: wait for its members' rows, sum them in member order, publish the group row
template <bool XG = false>
BB_DEV void bbp_leader_reduce(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BBLds& L, int par,
                              unsigned epoch, int* ok, unsigned abs_epoch = 0u) {
    const int KK = M.K + 2 * M.nt1, NG = bbp_groups(A), g = cx.block;
    const int members = (A.nblk - g + NG - 1) / NG;
    BB_PASS(cx, tid) {
        if (tid < members && !bb_wait_word(S.rdy + 32 * (g + tid * NG), epoch, S.gbar + 1, A.spin_limit)) *ok = 0;
    }
    BB_SYNC(cx);      // (kept although one wave does all of the leader's work when KK <= 64: letting the tile's other waves
    BB_STAMP(cx, S, 17);   //  run ahead beside that wave's dependent chain cost 2 %)
    // thread k owns row entry k: its members' values come straight into registers, 16 loads in flight (more would raise the kernel's register peak), and are
    // added in member order (coalesced across k; no LDS staging, no extra barrier)
    BB_PASS(cx, tid) {
        for (int k = tid; k < KK; k += cx.nthr) {
            double s = 0.0;
            for (int m0 = 0; m0 < members; m0 += 16) {
                double v[16];
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    v[i] = (m0 + i < members) ? bb_ld<true>(S.prow + (long long)(g + (m0 + i) * NG) * KK + k) : 0.0;
#pragma unroll
                for (int i = 0; i < 16; ++i) s += v[i];
            }
            if (XG) {
                for (int r = 0; r < A.world; ++r) bb_st_sys(S.xout[r] + bbx_slot(A, par, A.rank, g) * KK + k, s);
            } else {
                bb_st<true>(S.xrow + ((long long)par * NG + g) * KK + k, s);
            }
        }
    }
    bb_drain_and_meet(cx);
    BB_PASS(cx, tid) {
        if (XG) { if (tid < A.world) bb_set_word_sys(S.xout_rdy[tid] + 32 * bbx_slot(A, par, A.rank, g), abs_epoch); }
        else if (tid == 0) bb_set_word(S.rdy + 32 * (A.nblk + par * NG + g), epoch);
    }
    BB_STAMP(cx, S, 18);
}

// every tile: wait for the NG group rows, add them in group order -> totals in lds[L.wk], global samples in lds[L.zgl]
// WIDE (k_res on one GPU): up to 16 groups, read by two thread groups of KKP = 64 or 128 lanes -- half h polls and reads groups
// [8 h, 8 h + 8), eight loads in flight per lane as in the narrow form (sixteen would raise the kernel's register peak: measured,
// the G pass doubled); half 1's partial sums cross through LDS and one workgroup barrier, half 0 adds them: (g0 + .. + g7) +
// (g8 + .. + g15).  A leader then has 16 members: one round of loads instead of two.
template <bool XG = false, bool WIDE = false>
BB_DEV void bbp_consume(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, const BBLds& L, int par,
                        unsigned epoch, int* ok, unsigned abs_epoch = 0u) {
    double* lds = cx.lds;
    const int KK = M.K + 2 * M.nt1, NG = bbp_groups(A);
    if (WIDE && !XG) {
        // (the host runs this form only where the tile has a thread per row entry, KK <= nthr, and asks for 16 groups only where
        //  KK <= 128 and the tile has 2 KKP threads: try_resident.  A strided general form beside or instead of it costs the
        //  C2 instance 3 %: the step loop is short of scalar registers as it is.)
        const int KKP = KK <= 64 ? 64 : 128;
        BB_PASS(cx, tid) {
            const int half = tid / KKP, k = tid - half * KKP, g0 = 8 * half;
            if (half < 2 && k < 8 && g0 + k < NG && !bb_wait_word(S.rdy + 32 * (A.nblk + par * NG + g0 + k), epoch, S.gbar + 1, A.spin_limit)) *ok = 0;
        }
        if (KK <= 64) {
            // each half is one wave: it has left its poll loop before it loads (the wait also keeps the compiler from hoisting the loads)
#ifndef BB_EMU
            if (threadIdx.x < 128) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        } else {
            BB_SYNC(cx);
        }
        BB_STAMP(cx, S, 1);
        BB_PASS(cx, tid) {
            const int half = tid / KKP, k = tid - half * KKP, g0 = 8 * half;
            if (half == 1 && k < KK) {
                double v[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) v[g] = g0 + g < NG ? bb_ld<true>(S.xrow + ((long long)par * NG + g0 + g) * KK + k) : 0.0;
                double s = 0.0;
#pragma unroll
                for (int g = 0; g < 8; ++g) s += v[g];
                lds[L.red + k] = s;
            }
        }
        double s0 = 0.0;
        BB_PASS(cx, tid) {
            if (tid < KK) {
                double v[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) v[g] = g < NG ? bb_ld<true>(S.xrow + ((long long)par * NG + g) * KK + tid) : 0.0;
                double s = 0.0;
#pragma unroll
                for (int g = 0; g < 8; ++g) s += v[g];
#ifdef BB_EMU
                lds[L.red + KKP + tid] = s;
#else
                s0 = s;
#endif
            }
        }
        BB_SYNC(cx);
        BB_PASS(cx, tid) {
            if (tid < KK) {
#ifdef BB_EMU
                s0 = lds[L.red + KKP + tid];
#endif
                const double s = s0 + (NG > 8 ? lds[L.red + tid] : 0.0);
                if (tid < M.K) bb_put_total(M, L, lds, tid, s);
                else lds[L.zgl + (tid - M.K)] = s;
            }
        }
        if (!(KK <= 64 && M.Ttot <= 64)) BB_SYNC(cx);
        return;
    }
    if (XG) {
        // the 8 x world rows of this rank's own inbox (slot order = (source rank, group) = summation order)
        const int rows = 8 * A.world;
        const double* in = S.xout[A.rank] + (long long)par * rows * KK;
        const unsigned* in_rdy = S.xout_rdy[A.rank] + 32ll * par * rows;
        BB_PASS(cx, tid) {
            for (int j = tid; j < rows; j += cx.nthr)
                if (!bb_wait_word_sys(in_rdy + 32 * j, abs_epoch, S.gbar + 1, A.spin_limit)) *ok = 0;
        }
        BB_SYNC(cx);
        BB_STAMP(cx, S, 1);
        // all threads fetch (row, entry) items into the moment pass's LDS scratch, whole rows per chunk; entry k's
        // running sum lives in L.red[k] across chunks
        double* stage = lds + L.acc;
        double* sums = lds + L.red;
        const int cap = L.acc_cap, RC = cap / KK < rows ? cap / KK : rows;
        for (int r0 = 0; r0 < rows; r0 += RC) {
            const int nr = rows - r0 < RC ? rows - r0 : RC;
            BB_PASS(cx, tid) {
                for (int i = tid; i < nr * KK; i += cx.nthr) stage[i] = bb_ld_sys(in + (long long)r0 * KK + i);
            }
            BB_SYNC(cx);
            BB_PASS(cx, tid) {
                for (int k = tid; k < KK; k += cx.nthr) {
                    double s = r0 == 0 ? 0.0 : sums[k];
                    for (int r = 0; r < nr; ++r) s += stage[r * KK + k];
                    if (r0 + nr < rows) sums[k] = s;
                    else if (k < M.K) bb_put_total(M, L, lds, k, s);
                    else lds[L.zgl + (k - M.K)] = s;
                }
            }
            BB_SYNC(cx);
        }
        return;
    }
    BB_PASS(cx, tid) {
        if (tid < NG && !bb_wait_word(S.rdy + 32 * (A.nblk + par * NG + tid), epoch, S.gbar + 1, A.spin_limit)) *ok = 0;
    }
    if (KK <= 64) {
        // the polling lanes and the reading lanes are one wave: it has left every poll loop before it loads (the wait
        // also keeps the compiler from hoisting the row loads); the other waves meet it at the barrier below
#ifndef BB_EMU
        if (threadIdx.x < 64) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    } else {
        BB_SYNC(cx);
    }
    BB_STAMP(cx, S, 1);
    BB_PASS(cx, tid) {
        for (int k = tid; k < KK; k += cx.nthr) {
            double v[8];
#pragma unroll
            for (int g = 0; g < 8; ++g) v[g] = g < NG ? bb_ld<true>(S.xrow + ((long long)par * NG + g) * KK + k) : 0.0;
            double s = 0.0;
#pragma unroll
            for (int g = 0; g < 8; ++g) s += v[g];
            if (k < M.K) bb_put_total(M, L, lds, k, s);
            else lds[L.zgl + (k - M.K)] = s;
        }
    }
    // the F pass that follows (bbp_finish) reads these totals with threads j < Ttot: when they and the summing threads
    // k < KK are all wave 0, that wave's own program order is enough; F ends with the barrier everybody needs
    if (!(KK <= 64 && M.Ttot <= 64)) BB_SYNC(cx);
}

// ---- F pass: everything that depends only on the totals (ends with a workgroup barrier) -----------------------
template <int KIND>
BB_DEV void bbp_finish(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, int NB) {
    const BBLds L = bbp_layout<KIND>(cx, M, NB);
    BB_STAMP(cx, S, 25);
    bb_finalize_finish<KIND>(cx, M, S, A, L);
}

// ---- second half: residuals, per-latent gradient, optimiser in registers ----------
template <int KIND, int P>
BB_DEV void bbp_update(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, int NB, BBPst<P>* stv,
                       const BBSlot wslot) {
    const BBLds L = bbp_layout<KIND>(cx, M, NB);
    double* lds = cx.lds;
    const BBTile t = bb_tile(M, A, cx.block, NB);
    const BBSeg* sg = (const BBSeg*)(lds + L.seg);
    const int* li = (const int*)(lds + L.misc);
    BB_STAMP(cx, S, 26);
    bb_pass_residuals_units<KIND, true>(cx, M, S, L, t, NB);
    BB_STAMP(cx, S, 27);
    BB_PASS(cx, tid) {
        BBPst<P>& st = BB_PSTATE(stv, tid);
        // the totals in lds[L.wk ..] have been consumed (F pass): clear the row for the next step's partial sums
        for (int k = tid; k < M.K + 2 * M.nt1; k += cx.nthr) lds[L.wk + k] = 0.0;
        double* hs_m = nullptr;
        double* hs_o = nullptr;
        if (A.opt == 0) {
            hs_m = S.hist + ((long long)wslot.slot * 2 + 0) * M.Dh;
            hs_o = S.hist + ((long long)wslot.slot * 2 + 1) * M.Dh;
        }
#pragma unroll
        for (int k = 0; k < P; ++k) {
            if (k == 0) BB_STAMP_W(cx, S, 29);
            const BBPair q = bb_pair_cached(sg, st, k);
            if (!q.valid) continue;
            const long long blo = M.blk_lo[q.s.blk];
            // the draw is still staged in LDS (tile latents) / came back with the totals (replicated global latents)
            const double* zsrc = q.s.kind >= SK_GS ? lds + L.zgl + (q.s.kind == SK_GLS ? M.nt1 : 0) : lds + q.s.ldsoff;
            double pm, iv, g0 = 0.0, g1 = 0.0;
            const bb_u2 cu = bbp_cnt<P>(cx, L)[k * cx.nthr + tid];
            const bb_d2 cnt{(double)cu.x, (double)cu.y};
            if (q.a0) {
                const double z0 = zsrc[q.i0 - q.s.lo];
                bb_prior_of(M, q.s.blk, q.i0 - blo, &pm, &iv);
                g0 = bb_glik<KIND, true>(lds, M, L, t, NB, q.s, q.i0 - q.s.lo, z0, cnt.x) - (z0 - pm) * iv;
            }
            if (q.a1) {
                const double z1 = zsrc[q.i0 + 1 - q.s.lo];
                bb_prior_of(M, q.s.blk, q.i0 + 1 - blo, &pm, &iv);
                g1 = bb_glik<KIND, true>(lds, M, L, t, NB, q.s, q.i0 + 1 - q.s.lo, z1, cnt.y) - (z1 - pm) * iv;
            }
            if (k == 0) BB_STAMP_W(cx, S, 30);
            const double go0 = fma(g0, st.a[k].x, st.h[k].x), go1 = fma(g1, st.a[k].y, st.h[k].y);
            const bb_d2 hm = hs_m ? st.hm[k] : bb_d2{0, 0}, ho = hs_m ? st.ho[k] : bb_d2{0, 0};
            bb_d2 nhm = hm, nho = ho;
            if (q.a0) {
                bb_opt_apply(M, S, A, wslot, 0, q.i0 - q.s.pad, -g0, hm.x, &nhm.x, &st.mu[k].x, &st.am[k].x, &st.lo[k].x);
                bb_opt_apply(M, S, A, wslot, 1, q.i0 - q.s.pad, -go0, ho.x, &nho.x, &st.om[k].x, &st.ao[k].x, &st.lo[k].y);
            }
            if (q.a1) {
                bb_opt_apply(M, S, A, wslot, 0, q.i0 + 1 - q.s.pad, -g1, hm.y, &nhm.y, &st.mu[k].y, &st.am[k].y, &st.lo[k].z);
                bb_opt_apply(M, S, A, wslot, 1, q.i0 + 1 - q.s.pad, -go1, ho.y, &nho.y, &st.om[k].y, &st.ao[k].y, &st.lo[k].w);
            }
            if (k == 0) BB_STAMP_W(cx, S, 31);
            if (hs_m) { const long long ih = q.i0 - q.s.pad; bb_store_pair(hs_m, ih, q.a0, q.a1, nhm); bb_store_pair(hs_o, ih, q.a0, q.a1, nho); }
        }
        BB_STAMP_W(cx, S, 19);
    }
    BB_SYNC(cx);   // the LDS tables are rewritten by the next step's sample half
    BB_STAMP(cx, S, 28);
}

// ---- epilogue: state back to memory, step counter, status words -------------------------------------------
template <int KIND, int P>
BB_DEV void bbp_epilogue(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, int NB, BBPst<P>* stv,
                         unsigned long long step_end, bool timed_out) {
    const BBLds L = bbp_layout<KIND>(cx, M, NB);
    double* lds = cx.lds;
    const BBSeg* sg = (const BBSeg*)(lds + L.seg);
    const int* li = (const int*)(lds + L.misc);
    BB_PASS(cx, tid) {
        BBPst<P>& st = BB_PSTATE(stv, tid);
        bool bad = false;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const BBPair q = bb_pair_cached(sg, st, k);
            if (!q.valid) continue;
            bb_store_pair(S.mu, q.i0, q.a0, q.a1, st.mu[k]);
            bb_store_pair(S.om, q.i0, q.a0, q.a1, st.om[k]);
            bb_store_pair(S.acc_mu, q.i0, q.a0, q.a1, st.am[k]);
            bb_store_pair(S.acc_om, q.i0, q.a0, q.a1, st.ao[k]);
            bb_store_lo(S, q.i0, q.a0, q.a1, st.lo[k]);
            const double chk = (q.a0 ? st.mu[k].x + st.om[k].x : 0.0) + (q.a1 ? st.mu[k].y + st.om[k].y : 0.0);
            bad = bad || !(chk - chk == 0.0);        // NaN or +-Inf anywhere in the pair's variational parameters
        }
        // host-mapped status words (read by bb_run after the stream has drained, no copy): rare stores, any writer will do
        if (bad) S.hstatus[1] = 1u;
        if (timed_out && tid == 0) S.hstatus[0] = 1u;
        if (cx.block == 0 && tid == 0) { S.ctr[0] = step_end; S.ctr[1] = step_end; }
    }
    BB_SYNC(cx);
}

#ifndef BB_EMU
template <int KIND, int P, int NT, bool XG = false>
__global__ void __launch_bounds__(NT) k_persist(const DevModel* __restrict__ Mp, const DevState* __restrict__ Sp, const BBLds* __restrict__ Lp,
                                                  RunArgs A, int NB, int nsteps) {
    const DevModel& M = *Mp;   // descriptors live in device memory: scalar loads on demand instead of ~1.5 KB of
    const DevState& S = *Sp;   // kernel arguments held (and spilled) in SGPRs across the whole step loop
    extern __shared__ __attribute__((aligned(16))) double bbp_smem[];
    BBCtx cx{(int)blockDim.x, (int)blockIdx.x, bbp_smem, NT > 512 ? (const void*)Lp : nullptr};
    BBPst<P> st;
    int* ok_slot = (int*)(bbp_smem + Lp->misc) + 1;
    // the step comes from the device counter, not from the host: launches of one bb_run queue back to back, and one that
    // follows a timed-out launch must neither skip steps nor run at all (the timeout word stays set until the host clears it)
    const unsigned long long c0 = S.ctr[0], c1 = S.ctr[1];
    unsigned long long step0 = c0 > c1 ? c0 : c1;
    // (uniform: tell the compiler, so that everything derived from the step number -- epoch, parity, window slot -- is scalar work)
    step0 = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(step0 >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)step0);   // (the two-kernel step ping-pongs the counter: the current step is the larger word)
    bbp_prologue<KIND, P>(cx, M, S, A, NB, &st);
    const bool dead = *ok_slot == 0;               // uniform: written by one thread before the prologue's barriers
    int done = 0;
    if (!dead) {
    bbp_draw_ahead<KIND, P>(cx, M, A, NB, &st, step0);
    BBSlotCtr sc = bb_slot_init(A, step0);
    for (; done < nsteps; ++done, bb_slot_next(A, sc)) {
        const unsigned long long step = step0 + (unsigned long long)done;
        const BBSlot wslot = bb_slot_now(A, sc);
        bbp_sample<KIND, P>(cx, M, S, A, NB, &st, step);
        {
            const BBLds L = bbp_layout<KIND>(cx, M, NB);
            const unsigned epoch = A.xepoch0 + (unsigned)(step + 1);   // ready / inbox words never restart
            const int par = (int)(step & 1);
            bbp_publish_row(cx, M, S, L, epoch);                       // wk is complete: bb_pass_moments ended with a barrier
            bbp_draw_ahead<KIND, P>(cx, M, A, NB, &st, step + 1);      // the next step's normals, in the shadow of the rows' flight
            if ((int)blockIdx.x < bbp_groups(A)) bbp_leader_reduce<XG>(cx, M, S, A, L, par, epoch, ok_slot, epoch);
            bbp_prefetch_slot<KIND, P>(cx, M, S, A, NB, &st, wslot.slot);    // cold window lines fly while the rows arrive
            bbp_residual_ahead<KIND>(cx, M, NB, A);                    // ... and the totals-independent half of the residuals is tabulated
            bbp_consume<XG>(cx, M, S, A, L, par, epoch, ok_slot, epoch);
            bbp_finish<KIND>(cx, M, S, A, NB);
            if (*ok_slot == 0) break;                                  // uniform: read after the F pass's barrier
        }
        bbp_update<KIND, P>(cx, M, S, A, NB, &st, wslot);
    }
    }
    bbp_epilogue<KIND, P>(cx, M, S, A, NB, &st, step0 + (unsigned long long)done, dead || *ok_slot == 0);
}
#endif
