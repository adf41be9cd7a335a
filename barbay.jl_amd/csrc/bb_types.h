// bb_types.h -- POD descriptors shared by the host engine and the device block programs.
#pragma once
#include <stdint.h>

#define BB_MAX_REP 16

// `~` blocks of the model family, in a fixed kind numbering (not source order).
enum BBBlockKind {
    BK_SPOP = 0,   // s_pop        population mean fitness per time step (global)
    BK_LSPOP = 1,  // logsigma_pop (global)
    BK_S = 2,      // s_bc (fitness, multienv) or theta (genotype: per genotype; replicate: per mutant)
    BK_TT = 3,     // theta_tilde
    BK_LT = 4,     // logtau
    BK_LS = 5,     // logsigma_bc
    BK_L = 6,      // loglambda
    BK_COUNT = 7
};

// moment rows per (replicate, time step): S_t, then for t < T-1: M0 M1 M2 N1 N2
#define BB_NQ 6

struct DevPrior {
    double mean, inv_var;          // Vector form
    const double* mean_e;          // Matrix form (device), indexed by i - blk_lo; nullptr = Vector form
    const double* inv_var_e;
};

struct DevModel {
    int kind, R, E, G;
    long long nn, nb, B, D;
    long long Dp;                     // D rounded up to a multiple of 8
    // The TruncatedADAGrad window [W][2][Dh].  A whole-problem handle: Dh = Dp, entry i of a row is latent i.  A SHARDED handle
    // keeps only the latents it ever updates -- its barcodes' latents, the replicated blocks, the genotype block: one contiguous
    // range per (block, replicate), the ranges a tile's segment table is cut from -- packed one after the other (bb_engine.hip,
    // hist_rows): latent i of block blk, replicate r sits at entry i - bb_hdelta(M, blk, r).  All deltas are even (pairs stay
    // whole and 16-byte aligned) and zero on a whole-problem handle.
    long long Dh;
    long long hd0[BK_COUNT], hd1[BK_COUNT];   // per-mutant and replicated blocks: delta = hd0[blk] + r hd1[blk]
    long long hdl[BB_MAX_REP];                // loglambda slab of replicate r
    int T[BB_MAX_REP];
    unsigned Tmagic[BB_MAX_REP];      // floor(2^32 / T) + 1: n / T == umulhi(n, magic) for n < 2^16
    unsigned Tmagic1[BB_MAX_REP];     // same for T - 1
    long long off_l[BB_MAX_REP];      // flat index of replicate r's loglambda slab
    int off_t[BB_MAX_REP];            // first time step of replicate r inside s_pop / logsigma_pop
    long long cnt_off[BB_MAX_REP];    // offset of replicate r in counts
    int kq[BB_MAX_REP];               // first moment row of replicate r
    int kqa[BB_MAX_REP];              // first row of replicate r's (t, j) neutral table (ragged-method pairing only)
    int quirk;                        // 1 = ragged replicate method: neutral element (t, b) pairs population index (t + (T-1) b) div n_neutral
    int tcum[BB_MAX_REP];             // sum_{r' < r} T_r'
    int Ttot, nt1, K;
    long long blk_lo[BK_COUNT], blk_hi[BK_COUNT];
    DevPrior pri[BK_COUNT];
    const int* env_idx;               // device [T]
    const int* geno_idx;              // device [nb]
    const int* geno_ptr;              // device [G+1]   CSR over genotypes
    const int* geno_mem;              // device [nb]    mutant indices grouped by genotype
    int geno_sorted;                  // 1 = geno_idx is non-decreasing: a genotype's mutants are consecutive, so tiles / shards cut at genotype
                                      // boundaries own their genotypes' theta outright (the resident launch of the genotype model needs it)
    const unsigned* counts;           // device, uint32, same indexing as loglambda minus blk_lo
};

#define BB_MAX_WORLD 16               // ranks of one resident multi-GPU run (one xGMI hive holds 8)

struct bb_gran;
struct DevState {
    double *mu, *om;                  // [D] variational parameters theta = [mu; omega]
    double *zsv, *asv, *hsv;          // [D] per-sample scratch: z, eps*sigmoid(omega) (= dz/domega), sigmoid/softplus (= dH/domega)
    double *acc_mu, *acc_om;          // [D] optimiser accumulators
    double *hist;                     // [W][2][Dp] TruncatedADAGrad window of squared gradients
    const double *optc;               // [8] the optimiser's constants {eta, tau, pre, post}: read by scalar loads at every update (bb_opt_apply) --
                                      // as kernel arguments they sat in VGPR lanes inside the resident launches' step loop (the loop holds more
                                      // uniform values than there are SGPRs) and every update paid eight v_readlane_b32 for them
    float *accl;                      // [2 D] low-order parts of the running window sums: (acc_mu[i], accl[2 i]) and (acc_om[i], accl[2 i + 1]) are
                                      // compensated (two-sum) accumulators, bb_opt_apply
    double *gacc_mu, *gacc_om;        // [D] S > 1 accumulation / gradient export
    double *partials;                 // [2][K][nblk] (second half: odd steps of the persistent launch)
    double *totals;                   // [K]
    double *zg;                       // [2][2 nt1] sampled global latents (second half: odd steps, persistent launch)
    double *ztheta;                   // [G] sampled genotype fitness (genotype model)
    double *ds;                       // [nb] dlogp/ds_eff per mutant (genotype model)
    double *gsum;                     // [G] per-genotype sums of ds
    double *geno_el;                  // [geno blocks] ELBO partials of the theta block
    double *elbo_ring;                // [BB_ELBO_RING]
    double *elbo_sample;              // [S]
    unsigned long long *ctr;          // [2] device-side step counter (ping-pong)
    const double *eps_in;             // [S][D] caller-supplied draws (test hook) or nullptr
    unsigned *gbar;                   // [32 * 10] words of the persistent launch, one 128-B line each: [1] = timeout word (sticky until the host clears it)
    unsigned *hstatus;                // host-mapped status words the host reads after a run without a copy: [0] timeout, [1] non-finite state
    double *prow;                     // [nblk][K + 2 nt1] rows of the tiles (persistent launch)
    double *xrow;                     // [2][8][K + 2 nt1] group rows, double-buffered by step parity
    struct bb_gran *grow, *gxrow;     // k_res on one GPU: the same rows as self-validating 16-byte entries (bb_persist.h, BR_TG): [nblk][bb_row_stride(K + 2 nt1)], [2][16][K + 2 nt1]
    unsigned *rdy;                    // [32 * (nblk + 16)] ready words, one 128-B line each: tiles, then [2][8] groups
    int *xsel;                        // [nblk + 8] what each tile of the last resident launch decided about its row stores: 1 plain (same XCD as its leader), -1 write-through, 0 not decided
    unsigned long long *xtab;         // [BB_NG_MAX] k_res / k_stream: where the group leaders run -- {launch tag << 32 | XCC id}, written by every leader at the start of
                                      //   a launch; a member whose own XCC id is the leader's stores its row with plain stores (br_row_publish)
    const double *segtab;             // k_res / k_stream: the tiles' segment tables, built on the host (bb_engine.hip, host_tables): [tiles][segtab_stride] doubles,
    int segtab_stride;                //   each = (4 + 4 R + 1) BRSeg records + one word with their count; nullptr: thread 0 of a tile builds its own
    const int *ldstab;                // ... and the tile-independent LDS descriptor tables [rowmap 2 K | tmap K | ftab 4 Ttot | rtab 4 R]
    const long long *tile_b;          // genotype model, k_res: [tiles + 1] first barcode of every tile (cuts fall on genotype boundaries)
    const int *tile_g;                // genotype model, k_res: [tiles + 1] first genotype every tile owns
    unsigned long long *stamps;       // [nblk + 8][32] s_memtime stamps, then [nblk + 8][4][16] per-wave stamps (diagnostic build -DBB_STAMPS only)
    // cross-GPU leg of the resident launch's exchange (bb_p2p_*): every rank owns an INBOX -- group rows
    // [2 parity][world][8 groups][K + 2 nt1] and their ready words [2][world][8] (one 128-B line each) -- in
    // fine-grained memory that its peers map through IPC handles; xout[r] / xout_rdy[r] are rank r's inbox as
    // seen from here (r == own rank: the local inbox itself)
    double *xout[BB_MAX_WORLD];
    unsigned *xout_rdy[BB_MAX_WORLD];
    struct bb_gran *xgr[BB_MAX_WORLD]; // k_res: the same group rows [2][world][8][K + 2 nt1] as self-validating 16-byte entries (no ready words; bb_persist.h)
};

#define BB_ELBO_RING 4096

struct RunArgs {
    long long b_lo, b_hi;             // barcode shard [b_lo, b_hi)
    int rank, world;                  // of the sharded run (0, 1 otherwise)
    unsigned xepoch0;                 // base of the ready / inbox words of the resident launch: they carry base + step + 1 and only ever grow (bb_persist.h)
    unsigned spin_limit;              // polls of one ready word before a resident launch gives up (the first launch of a sharded run gets more)
    unsigned launch_tag;              // counts the handle's resident launches (never 0): what a launch's entries in DevState.xtab carry
    int row_l2;                       // 1: tiles on their group leader's XCD store their row with plain stores (BB_TUNE_ROW_L2=0 turns it off)
    int nblk;                         // blocks of the barcode grid
    int nblk_alloc;                   // tiles the exchange / stamp buffers were sized for (+ 8)
    int ng;                           // groups of the exchange's first hop (8; k_res on one GPU: 16 where the tile has the threads for it)
    int pf;                           // k_res: when a step's TruncatedADAGrad window slot is fetched into LDS -- 0 in the exchange's shadow, 1 at the start of the
                                      // step's S pass, 2 at the end of the previous step's G pass (1, 2: the slot buffer has an LDS region of its own)
    int nbl;                          // k_res: barcodes of each of the first min(8, nblk) tiles -- the exchange's group leaders get smaller tiles (0: all tiles alike)
    int par;                          // which ctr[] word holds the current step
    int sample, S;
    int first_sample, last_sample;    // of this step
    int apply;                        // 1 = optimiser update, 0 = export gradient to gacc_*
    int with_elbo;
    int count_globals;                // this rank adds the replicated blocks' ELBO terms
    int opt, W, resum_every, elbo_every;
    double eta, tau, pre, post;
    unsigned long long seed;
    const double* red;                // moment rows to finalise: [K][nred]
    int nred;
    double elbo_const;
};
