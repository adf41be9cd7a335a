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

template <int P>
struct BBPst {
    bb_d2 mu[P], om[P], am[P], ao[P];   // variational parameters and optimiser accumulators of P pairs
    bb_d2 z[P], a[P], h[P];             // current draw: z, eps*sigmoid(omega), sigmoid/softplus
    bb_d2 hm[P], ho[P];                 // this step's TruncatedADAGrad window slot, fetched while the exchange is in flight
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

BB_DEV bb_d2 bb_load_pair(const double* base, long long i0, bool a0, bool a1) {
    if (a0 && a1) return *(const bb_d2*)(base + i0);
    return bb_d2{a0 ? base[i0] : 0.0, a1 ? base[i0 + 1] : 0.0};
}
BB_DEV void bb_store_pair(double* base, long long i0, bool a0, bool a1, bb_d2 v) {
    if (a0 && a1) { *(bb_d2*)(base + i0) = v; return; }
    if (a0) base[i0] = v.x;
    if (a1) base[i0 + 1] = v.y;
}

// ---- prologue: segment table, state into registers -------------------------------------------------
template <int KIND, int P>
BB_DEV void bbp_prologue(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, int NB, BBPst<P>* stv) {
    const BBLds L = bb_lds_layout(M.R, M.E, KIND, M.Ttot, M.nt1, M.K, NB, cx.nthr);
    double* lds = cx.lds;
    const BBTile t = bb_tile(M, A, cx.block, NB);
    BBSeg* sg = (BBSeg*)(lds + L.seg);
    int* li = (int*)(lds + L.misc);
    BB_PASS(cx, tid) {
        if (tid == 0) li[0] = bb_build_segs<KIND>(sg, M, L, t, cx.block == 0);
    }
    BB_SYNC(cx);
    BB_PASS(cx, tid) {
        BBPst<P>& st = BB_PSTATE(stv, tid);
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const BBPair q = bb_pair_of(sg, li[0], tid + k * cx.nthr);
            st.mu[k] = bb_load_pair(S.mu, q.i0, q.a0, q.a1);
            st.om[k] = bb_load_pair(S.om, q.i0, q.a0, q.a1);
            st.am[k] = bb_load_pair(S.acc_mu, q.i0, q.a0, q.a1);
            st.ao[k] = bb_load_pair(S.acc_om, q.i0, q.a0, q.a1);
            st.z[k] = st.a[k] = st.h[k] = st.hm[k] = st.ho[k] = bb_d2{0.0, 0.0};
        }
    }
    BB_SYNC(cx);
}

// ---- first half of a step: draw, stage, moments, publish the tile's K partial rows ---------------------
template <int KIND, int P>
BB_DEV void bbp_sample(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, int NB, BBPst<P>* stv,
                       unsigned long long step) {
    const BBLds L = bb_lds_layout(M.R, M.E, KIND, M.Ttot, M.nt1, M.K, NB, cx.nthr);
    double* lds = cx.lds;
    const BBTile t = bb_tile(M, A, cx.block, NB);
    const BBSeg* sg = (const BBSeg*)(lds + L.seg);
    const int* li = (const int*)(lds + L.misc);
    const int par = (int)(step & 1);
    double* zg = S.zg + (long long)par * 2 * M.nt1;
    BB_STAMP(cx, S, 20);
    BB_PASS(cx, tid) {
        BBPst<P>& st = BB_PSTATE(stv, tid);
        for (int k = tid; k < M.K; k += cx.nthr) lds[L.wk + k] = 0.0;
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const BBPair q = bb_pair_of(sg, li[0], tid + k * cx.nthr);
            if (!q.valid) continue;
            double e0, e1, sp0, sg0, sp1, sg1;
            bb_normal_pair(A.seed, (unsigned long long)(q.i0 >> 1), (unsigned)step, 0u, &e0, &e1);
            bb_softplus_sigmoid(st.om[k].x, &sp0, &sg0);
            bb_softplus_sigmoid(st.om[k].y, &sp1, &sg1);
            const double z0 = fma(sp0, e0, st.mu[k].x), z1 = fma(sp1, e1, st.mu[k].y);
            st.z[k] = bb_d2{z0, z1};
            st.a[k] = bb_d2{e0 * sg0, e1 * sg1};
            st.h[k] = bb_d2{sg0 * bb_rcp(sp0), sg1 * bb_rcp(sp1)};
            if (q.s.kind >= SK_GS) {       // replicated global latents: every tile's finalize needs them
                double* dst = zg + (q.s.kind == SK_GLS ? M.nt1 : 0);
                if (q.a0) bb_st<true>(dst + (q.i0 - q.s.lo), z0);
                if (q.a1) bb_st<true>(dst + (q.i0 + 1 - q.s.lo), z1);
            } else {
                if (q.a0) lds[q.s.ldsoff + (q.i0 - q.s.lo)] = z0;
                if (q.a1) lds[q.s.ldsoff + (q.i0 + 1 - q.s.lo)] = z1;
            }
        }
    }
    BB_SYNC(cx);
    BB_STAMP(cx, S, 21);
    BB_PASS(cx, tid) { bb_effective_tables<KIND>(cx, tid, M, S, L, t, false); }
    BB_SYNC(cx);
    BB_STAMP(cx, S, 22);
    bb_pass_moments<KIND>(cx, M, S, L, t, NB, false);
    BB_STAMP(cx, S, 23);
    BB_PASS(cx, tid) {
        double* dst = S.partials + (long long)par * M.K * A.nblk;
        for (int k = tid; k < M.K; k += cx.nthr) bb_st<true>(dst + (long long)k * A.nblk + cx.block, lds[L.wk + k]);
    }
    BB_STAMP(cx, S, 24);
    // (the exchange that follows drains vmcnt and synchronises the workgroup)
}

// ---- window-slot prefetch: issued right after the tile has arrived at the exchange, so that the cold HBM
// lines (and their address translations) are in flight while the workgroup waits for the other tiles ----
template <int KIND, int P>
BB_DEV void bbp_prefetch_slot(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, int NB, BBPst<P>* stv,
                              unsigned long long step) {
    const BBLds L = bb_lds_layout(M.R, M.E, KIND, M.Ttot, M.nt1, M.K, NB, cx.nthr);
    double* lds = cx.lds;
    const BBSeg* sg = (const BBSeg*)(lds + L.seg);
    const int* li = (const int*)(lds + L.misc);
    BB_PASS(cx, tid) {
        BBPst<P>& st = BB_PSTATE(stv, tid);
        if (A.opt == 0) {
            const int slot = bb_slot_of(A, step).slot;
            const double* hs_m = S.hist + ((long long)slot * 2 + 0) * M.Dp;
            const double* hs_o = S.hist + ((long long)slot * 2 + 1) * M.Dp;
#pragma unroll
            for (int k = 0; k < P; ++k) {
                const BBPair q = bb_pair_of(sg, li[0], tid + k * cx.nthr);
                st.hm[k] = bb_load_pair(hs_m, q.i0, q.a0, q.a1);
                st.ho[k] = bb_load_pair(hs_o, q.i0, q.a0, q.a1);
            }
        }
    }
}

// ---- exchange: the tiles of group g (= workgroups b with b % NG == g; on MI355X the dispatcher deals
// workgroups round-robin over the 8 XCDs, so a group shares an L2 -- speed only, never correctness) are
// summed in member order by whichever member arrives last; every tile then reads NG rows instead of nblk.
BB_DEV int bbp_groups(int nblk) { return nblk < 8 ? nblk : 8; }

BB_DEV void bbp_reduce_group(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, int NB, int kind, int par, int g) {
    const BBLds L = bb_lds_layout(M.R, M.E, kind, M.Ttot, M.nt1, M.K, NB, cx.nthr);
    double* lds = cx.lds;
    const int NG = bbp_groups(A.nblk);
    const int members = (A.nblk - g + NG - 1) / NG;
    const double* src = S.partials + (long long)par * M.K * A.nblk;
    double* dst = S.xsum + (long long)par * M.K * NG;
    double* tmp = lds + L.red;                      // 16 K doubles available, 4 K used
    BB_PASS(cx, tid) {
        for (int w = tid; w < M.K * 4; w += cx.nthr) {
            const int k = w >> 2, c = w & 3;
            const int per = (members + 3) / 4;
            double v[8];
            double s = 0.0;
            for (int m0 = c * per; m0 < (c + 1) * per && m0 < members; m0 += 8) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int m = m0 + i;
                    v[i] = (m < (c + 1) * per && m < members) ? bb_ld<true>(src + (long long)k * A.nblk + g + (long long)m * NG) : 0.0;
                }
                s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            }
            tmp[w] = s;
        }
    }
    BB_SYNC(cx);
    BB_PASS(cx, tid) {
        for (int k = tid; k < M.K; k += cx.nthr)
            bb_st<true>(dst + (long long)k * NG + g, (tmp[4 * k] + tmp[4 * k + 1]) + (tmp[4 * k + 2] + tmp[4 * k + 3]));
    }
    BB_SYNC(cx);
}

// ---- second half: totals, global finish, residuals, per-latent gradient, optimiser in registers ----------
template <int KIND, int P>
BB_DEV void bbp_update(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, int NB, BBPst<P>* stv,
                       unsigned long long step) {
    const BBLds L = bb_lds_layout(M.R, M.E, KIND, M.Ttot, M.nt1, M.K, NB, cx.nthr);
    double* lds = cx.lds;
    const BBTile t = bb_tile(M, A, cx.block, NB);
    const BBSeg* sg = (const BBSeg*)(lds + L.seg);
    const int* li = (const int*)(lds + L.misc);
    const int par = (int)(step & 1);
    RunArgs Af = A;
    Af.red = S.xsum + (long long)par * M.K * bbp_groups(A.nblk);
    Af.nred = bbp_groups(A.nblk);
    BB_STAMP(cx, S, 25);
    bb_finalize<true>(cx, M, S, Af, L, S.zg + (long long)par * 2 * M.nt1);
    BB_STAMP(cx, S, 26);
    bb_pass_residuals_units<KIND>(cx, M, S, L, t, NB);
    BB_STAMP(cx, S, 27);
    const BBSlot wslot = bb_slot_of(A, step);
    BB_PASS(cx, tid) {
        BBPst<P>& st = BB_PSTATE(stv, tid);
        double* hs_m = nullptr;
        double* hs_o = nullptr;
        if (A.opt == 0) {
            hs_m = S.hist + ((long long)wslot.slot * 2 + 0) * M.Dp;
            hs_o = S.hist + ((long long)wslot.slot * 2 + 1) * M.Dp;
        }
#pragma unroll
        for (int k = 0; k < P; ++k) {
            if (k == 0) BB_STAMP_W(cx, S, 29);
            const BBPair q = bb_pair_of(sg, li[0], tid + k * cx.nthr);
            if (!q.valid) continue;
            const long long blo = M.blk_lo[q.s.blk];
            double pm, iv, g0 = 0.0, g1 = 0.0;
            if (q.a0) {
                bb_prior_of(M, q.s.blk, q.i0 - blo, &pm, &iv);
                g0 = bb_glik<KIND>(lds, M, L, t, NB, q.s, q.i0 - q.s.lo, st.z[k].x) - (st.z[k].x - pm) * iv;
            }
            if (q.a1) {
                bb_prior_of(M, q.s.blk, q.i0 + 1 - blo, &pm, &iv);
                g1 = bb_glik<KIND>(lds, M, L, t, NB, q.s, q.i0 + 1 - q.s.lo, st.z[k].y) - (st.z[k].y - pm) * iv;
            }
            if (k == 0) BB_STAMP_W(cx, S, 30);
            const double go0 = fma(g0, st.a[k].x, st.h[k].x), go1 = fma(g1, st.a[k].y, st.h[k].y);
            const bb_d2 hm = hs_m ? st.hm[k] : bb_d2{0, 0}, ho = hs_m ? st.ho[k] : bb_d2{0, 0};
            bb_d2 nhm = hm, nho = ho;
            if (q.a0) {
                bb_opt_apply(M, S, A, wslot, 0, q.i0, -g0, hm.x, &nhm.x, &st.mu[k].x, &st.am[k].x);
                bb_opt_apply(M, S, A, wslot, 1, q.i0, -go0, ho.x, &nho.x, &st.om[k].x, &st.ao[k].x);
            }
            if (q.a1) {
                bb_opt_apply(M, S, A, wslot, 0, q.i0 + 1, -g1, hm.y, &nhm.y, &st.mu[k].y, &st.am[k].y);
                bb_opt_apply(M, S, A, wslot, 1, q.i0 + 1, -go1, ho.y, &nho.y, &st.om[k].y, &st.ao[k].y);
            }
            if (k == 0) BB_STAMP_W(cx, S, 31);
            if (hs_m) { bb_store_pair(hs_m, q.i0, q.a0, q.a1, nhm); bb_store_pair(hs_o, q.i0, q.a0, q.a1, nho); }
        }
        BB_STAMP_W(cx, S, 19);
    }
    BB_SYNC(cx);   // the LDS tables are rewritten by the next step's sample half
    BB_STAMP(cx, S, 28);
}

// ---- epilogue: state back to memory, step counter -----------------------------------------------------
template <int KIND, int P>
BB_DEV void bbp_epilogue(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, int NB, BBPst<P>* stv,
                         unsigned long long step_end) {
    const BBLds L = bb_lds_layout(M.R, M.E, KIND, M.Ttot, M.nt1, M.K, NB, cx.nthr);
    double* lds = cx.lds;
    const BBSeg* sg = (const BBSeg*)(lds + L.seg);
    const int* li = (const int*)(lds + L.misc);
    BB_PASS(cx, tid) {
        BBPst<P>& st = BB_PSTATE(stv, tid);
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const BBPair q = bb_pair_of(sg, li[0], tid + k * cx.nthr);
            if (!q.valid) continue;
            bb_store_pair(S.mu, q.i0, q.a0, q.a1, st.mu[k]);
            bb_store_pair(S.om, q.i0, q.a0, q.a1, st.om[k]);
            bb_store_pair(S.acc_mu, q.i0, q.a0, q.a1, st.am[k]);
            bb_store_pair(S.acc_om, q.i0, q.a0, q.a1, st.ao[k]);
        }
        if (cx.block == 0 && tid == 0) { S.ctr[0] = step_end; S.ctr[1] = step_end; }
    }
    BB_SYNC(cx);
}

#ifndef BB_EMU
// Exchange of the moment rows between the resident workgroups (zeroed counters before every launch;
// epoch = step index within the launch + 1).  words: bar[0] = groups done, bar[1] = timeout, bar[32 (g+1)] =
// arrivals of group g.  Every wait is bounded; on timeout all workgroups leave the step loop.
__device__ __forceinline__ bool bb_wait_ge(unsigned* word, unsigned target, unsigned* tmo) {
    unsigned spins = 0;
    while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) == 0u) {
            if (__hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || spins > (1u << 24)) {
                __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
    return true;
}

template <class F>
__device__ __forceinline__ bool bb_exchange(BBCtx& cx, const DevModel& M, const DevState& S, const RunArgs& A, int NB, int kind,
                                            int par, unsigned epoch, int* slot /* LDS: [0] ok, [1] reducer */, F after_arrival) {
    unsigned* bar = S.gbar;
    const int NG = bbp_groups(A.nblk);
    const int g = cx.block % NG;
    const unsigned members = (unsigned)((A.nblk - g + NG - 1) / NG);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave: write-through rows have landed
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned ticket = __hip_atomic_fetch_add(bar + 32 * (g + 1), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        slot[1] = (ticket == members * epoch - 1u);        // last arriver of the group in this epoch
        slot[0] = 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // rows are stored sc1 + drained and read with sc1 loads only
    }
    __syncthreads();
    if (slot[1]) {                                         // uniform per workgroup
        bbp_reduce_group(cx, M, S, A, NB, kind, par, g);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    after_arrival();                                       // loads issued here fly while the tile waits
    if (threadIdx.x == 0) {
        slot[0] = bb_wait_ge(bar, (unsigned)NG * epoch, bar + 1) ? 1 : 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __syncthreads();
    return slot[0] != 0;
}

template <int KIND, int P, int NT>
__global__ void __launch_bounds__(NT) k_persist(const DevModel* __restrict__ Mp, const DevState* __restrict__ Sp, RunArgs A, int NB,
                                                  unsigned long long step0, int nsteps) {
    const DevModel& M = *Mp;   // descriptors live in device memory: scalar loads on demand instead of ~1.5 KB of
    const DevState& S = *Sp;   // kernel arguments held (and spilled) in SGPRs across the whole step loop
    extern __shared__ __attribute__((aligned(16))) double bbp_smem[];
    BBCtx cx{(int)blockDim.x, (int)blockIdx.x, bbp_smem};
    BBPst<P> st;
    int* ok_slot = (int*)(bbp_smem + bb_lds_layout(M.R, M.E, KIND, M.Ttot, M.nt1, M.K, NB, cx.nthr).misc) + 1;
    bbp_prologue<KIND, P>(cx, M, S, A, NB, &st);
    int done = 0;
    for (; done < nsteps; ++done) {
        const unsigned long long step = step0 + (unsigned long long)done;
        bbp_sample<KIND, P>(cx, M, S, A, NB, &st, step);
        if (!bb_exchange(cx, M, S, A, NB, KIND, (int)(step & 1), (unsigned)(done + 1), ok_slot,
                         [&]() { bbp_prefetch_slot<KIND, P>(cx, M, S, A, NB, &st, step); })) break;
        bbp_update<KIND, P>(cx, M, S, A, NB, &st, step);
    }
    bbp_epilogue<KIND, P>(cx, M, S, A, NB, &st, step0 + (unsigned long long)done);
}
#endif
