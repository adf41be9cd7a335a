/* barbay_hip.h -- C ABI of the MI355X-native ADVI engine behind BarBay.vi.advi().
 *
 * Drop-in boundary: the one line this library replaces in the reference is
 *     q = Turing.vi(bayes_model, advi; optimizer=opt)          (src/vi.jl:201)
 * Everything before it (src/vi.jl:103-198) produces the inputs described by
 * bb_model_desc / bb_advi_opts; everything after it (src/vi.jl:203-234,
 * src/utils.jl:1042-1078, 1409-1462) reads only q.dist.m, q.dist.σ and
 * q.transform.ranges_out, which bb_get_posterior / bb_get_layout return.
 *
 * Conventions
 *  - plain C, no C++ exceptions cross the boundary; every int-returning entry
 *    point returns BB_OK (0) or a negative BB_ERR_* code, with a thread-local
 *    message available from bb_last_error().
 *  - the caller owns every host array it passes; the library copies what it
 *    needs during the call and never retains a host pointer.
 *  - the flat latent vector is the concatenation of the model's `~` blocks in
 *    source order with Julia column-major indexing.
 *  - one handle = one host thread at a time; it drives one GPU, or bb_advi_opts.n_devices GPUs of the node.
 */
#ifndef BARBAY_HIP_H
#define BARBAY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BB_OK 0
#define BB_ERR_INVALID (-1)   /* bad argument / inconsistent description     */
#define BB_ERR_DEVICE (-2)    /* HIP runtime error                           */
#define BB_ERR_COMM (-3)      /* RCCL error / communicator not initialised   */
#define BB_ERR_UNSUPPORTED (-4)
#define BB_ERR_NONFINITE (-5) /* bb_run: the steps were taken, but the variational parameters (or the exchanged
                                 moments) went NaN / Inf on the way -- the reference has no such guard and would
                                 return the NaN posterior silently                      */

/* model kinds: BarBay.model.* entries on the hot path */
#define BB_MODEL_FITNESS 0    /* fitness_normal            src/model_fitness_normal.jl:120-272 */
#define BB_MODEL_MULTIENV 1   /* multienv_fitness_normal   src/model_multienv_fitness_normal.jl:133-303 */
#define BB_MODEL_GENOTYPE 2   /* genotype_fitness_normal   src/model_fitness_normal_hierarchical_genotypes.jl:151-330 */
#define BB_MODEL_REPLICATE 3  /* replicate_fitness_normal  src/model_fitness_normal_hierarchical_replicates.jl:145-332 (3-D)
                                 and :407-638 (ragged); equal n_time[] == the 3-D method */

#define BB_MODEL_MULTIENV_REPLICATE 4 /* multienv_replicate_fitness_normal
                                 src/model_multienv_fitness_normal_hierarchical_replicates.jl:158-363 (3-D), :449-687 (ragged) */

/* bb_model_desc.flags */
#define BB_FLAG_RAGGED_METHOD 1 /* BB_MODEL_REPLICATE called as the reference's Vector{Matrix} method
                                  (src/model_fitness_normal_hierarchical_replicates.jl:407-638): its neutral
                                  likelihood pairs data element (t, b) with population index
                                  (t + (T_r-1) b) div n_neutral (`repeat(.., inner=n_neutral)`, :599-605, against a
                                  time-fastest data vector :549).  Reproduced as written when this flag is set;
                                  without it the 3-D method's self-consistent pairing (:307-311) is evaluated. */

/* optimisers selectable at src/vi.jl:99 (AdvancedVI 0.2) */
#define BB_OPT_TRUNCATED_ADAGRAD 0
#define BB_OPT_DECAYED_ADAGRAD 1

/* A prior argument of a model (`VecOrMat{Float64}`, model_fitness_normal.jl:125-129):
 * n == 1  : Vector form [mean, std] shared by the block;
 * n == len: Matrix form, mean[i], std[i] per element of the block. */
typedef struct bb_prior {
    const double* mean;
    const double* std;
    int64_t n;
} bb_prior;

/* What `model(R, n_t, n_neutral, n_bc; kwargs...)` receives (src/vi.jl:172-178),
 * in the layout src/utils.jl:48-61 (DataArrays) hands over. */
typedef struct bb_model_desc {
    int32_t kind;            /* BB_MODEL_*                                                */
    int32_t n_rep;           /* replicates (1 unless BB_MODEL_REPLICATE / _MULTIENV_REPLICATE) */
    int64_t n_neutral;       /* neutral barcodes: columns 0..n_neutral-1 (utils.jl:431)   */
    int64_t n_bc;            /* mutant barcodes                                           */
    const int32_t* n_time;   /* [n_rep] time points per replicate                         */
    const int64_t* counts;   /* replicate-major; each T_r x B column-major (t fastest)    */
    const int64_t* totals;   /* replicate-major; [T_r] = row sums of counts               */
    int32_t n_env;           /* BB_MODEL_MULTIENV: number of distinct environments        */
    const int32_t* env_idx;  /* [sum_r T_r] 0-based env of each time point, replicate-major
                                (indexin(envs, unique(envs)); the 3-D multienv_replicate method
                                repeats its one env list per replicate)                   */
    int32_t n_geno;          /* BB_MODEL_GENOTYPE: number of distinct genotypes           */
    const int32_t* geno_idx; /* [n_bc] 0-based genotype of each mutant                    */
    bb_prior s_pop_prior;        /* default [0,2] */
    bb_prior logsigma_pop_prior; /* default [0,1] */
    bb_prior s_bc_prior;         /* default [0,2]; theta prior for the hierarchical models */
    bb_prior logsigma_bc_prior;  /* default [0,1] */
    bb_prior loglambda_prior;    /* default [3,3] */
    bb_prior logtau_prior;       /* default [-2,1]; Vector form only (as in the reference) */
    int32_t flags;               /* BB_FLAG_*                                              */
} bb_model_desc;

/* Turing.ADVI(samples_per_step, max_iters) + optimiser + engine options. */
typedef struct bb_advi_opts {
    int32_t samples_per_step; /* S >= 1                                                  */
    int32_t optimizer;        /* BB_OPT_*                                                */
    double eta;               /* both optimisers, default 0.1                            */
    double tau;               /* TruncatedADAGrad, default 40                            */
    int32_t window;           /* TruncatedADAGrad n, default 100                         */
    int32_t resum_every;      /* TruncatedADAGrad: 1 = re-add the whole window every step
                                 (same arithmetic as the reference's sum(g2)); k > 1 =
                                 running sum, exact re-add every k steps; 0 (default) =
                                 running sum, never re-added: a compensated (two-sum)
                                 accumulator whose low-order part is a float, so what is
                                 lost per step is 2^-77 of the sum AT THAT STEP -- it stays
                                 behind when the sum later falls by orders of magnitude.
                                 Measured against the correctly rounded window sum
                                 (tools/window_sum_accuracy.py ->
                                 profiles/window_sum_accuracy_10k.json): 0 at 100 steps,
                                 3e-11 at 1 000, 7e-8 at 5 000, 3e-8 at 10 000 (relative,
                                 worst latent); the step size eta / (tau + sqrt(s)) moves
                                 by at most 1e-8 of itself.  k > 1 bounds it.             */
    double pre;               /* DecayedADAGrad, default 1.0                             */
    double post;              /* DecayedADAGrad, default 0.9                             */
    uint64_t seed;            /* Philox key (DESIGN.md "RNG stream")                     */
    int32_t device;           /* HIP device ordinal                                      */
    int32_t rank;             /* barcode shard owned by this handle                      */
    int32_t world_size;       /* number of shards (1 = whole problem)                    */
    int32_t steps_per_graph;  /* steps captured per hipGraph (0 = default, <0 = eager)   */
    int32_t elbo_every;       /* evaluate the ELBO every k-th step (0 = never)           */
    int32_t launch_mode;      /* 0 = auto; 1 = two kernels per sample (graph / eager);
                                 2 = ONE resident launch for the whole step loop (k_res /
                                 k_stream / k_persist, bb_stats.resident_kernel; the exchange
                                 of the moment rows runs inside it) -- on one GPU also with
                                 samples_per_step > 1 and ELBO recording (k_res's MS
                                 instances), on a sharded handle with the rows pushed into
                                 the peers' inboxes; error if the shape has no such launch  */
    int32_t n_devices;        /* > 1: ONE handle drives this many GPUs from the calling host
                                 thread: the barcodes shard over the devices,
                                 the resident launches run concurrently and exchange their
                                 group rows through peer-mapped inboxes (xGMI); rank /
                                 world_size must then be 0 / 1.  0 or 1 = one device.       */
    const int32_t* device_ids;/* [n_devices] HIP ordinals, or NULL = 0 .. n_devices-1       */
} bb_advi_opts;

typedef struct bb_handle bb_handle;

typedef struct bb_block_range {
    char name[24];  /* s_pop, logsigma_pop, s_bc, logsigma_bc, theta, theta_tilde, logtau, loglambda */
    int64_t lo;     /* 0-based, half-open: q.transform.ranges_out[i] == lo+1 : hi                     */
    int64_t hi;
} bb_block_range;

typedef struct bb_stats {
    int64_t n_latents;         /* D                                                      */
    int64_t n_moments;         /* K doubles all-reduced per MC sample                    */
    int64_t steps_done;
    int64_t shard_lo, shard_hi;/* barcode range owned by this handle                     */
    int64_t bytes_per_step;    /* algorithmic HBM bytes per step on this shard           */
    int64_t bytes_sample, bytes_update; /* its split over the two kernels                */
    double last_run_ms;        /* HIP-event time of the last bb_run                      */
    double avg_sample_ms;      /* per-launch averages from the last bb_run_profiled      */
    double avg_update_ms;
    int32_t n_blocks, block_threads, lds_bytes;
    int32_t persistent_pairs;  /* > 0: bb_run uses the resident launch with this many latent pairs per thread */
    int32_t launches_last_run; /* kernel launches of the last bb_run (resident launch: <= 4096 steps each)       */
    int32_t resident_kernel;   /* which resident launch bb_run uses: 0 none (two kernels per sample), 1 k_persist
                                  (LDS-staged passes), 2 k_res (the owner of a latent computes; bb_resident.h), 3 k_stream
                                  (k_res's tile map with the per-pair state streamed: tiles beyond the register file; bb_stream.h) */
    int32_t geno_lo, geno_hi;  /* genotype model: the genotypes whose theta THIS handle owns (0 .. n_geno unless the run is
                                  sharded and geno_idx is non-decreasing: shards are then cut at genotype boundaries, and after
                                  a resident run only the owner's copy of theta_g is current)                          */
    int64_t device_bytes;      /* device memory this handle allocated (a multi-device handle: all its shards)          */
    int64_t window_row;        /* entries of one row of the TruncatedADAGrad window: n_latents rounded up to 8 on a
                                  whole-problem handle; a shard keeps only the latents it updates (folded rows)        */
    int32_t rows_same_xcd;     /* k_res / k_stream: tiles of the last launch that found themselves on their exchange group
                                  leader's XCD and stored their row through the L2 they share (of n_blocks; 0 with
                                  BB_TUNE_ROW_L2=0 in the environment, which keeps every row store write-through)        */
    int32_t reserved0;
} bb_stats;

const char* bb_version(void);
const char* bb_last_error(void);

void bb_default_opts(bb_advi_opts* opts);

/* Build device state for one model instance.  Validates the description
 * (shapes, totals == row sums, index ranges) and copies everything it needs. */
int bb_create(const bb_model_desc* model, const bb_advi_opts* opts, bb_handle** out);
void bb_destroy(bb_handle* h);

int64_t bb_num_latents(const bb_handle* h);
/* blocks[0..*n_blocks): one range per `~` block in source order (<= 8). */
int bb_get_layout(const bb_handle* h, bb_block_range* blocks, int32_t* n_blocks);

/* Turing.meanfield initialisation (mu0 = randn(D), sigma0 = softplus.(randn(D)))
 * drawn from the engine's own init streams; resets optimiser state and step. */
int bb_init_meanfield(bb_handle* h);
/* Explicit variational parameters theta = [mu; omega], sigma = softplus(omega);
 * resets optimiser state and step. */
int bb_set_params(bb_handle* h, const double* mu, const double* omega);
int bb_get_params(bb_handle* h, double* mu, double* omega);
/* Genotype model: the reference hands barcodes over in order of appearance (utils.data_to_arrays, src/utils.jl:692-731), so a
 * genotype's mutants are scattered over geno_idx.  The library then groups them itself (stable sort of the mutants by genotype:
 * the resident launch and genotype-aligned shards need consecutive runs), works in that order and presents the CALLER's order at
 * every entry point that takes or returns a latent vector.  Likewise the loglambda block: where n_geno + n_bc is odd it would start at
 * an odd flat index, so internally it sits in front of the theta block (the resident launch's 16-byte pairs stay whole).  caller_index[i] = the caller's flat index of the handle's internal
 * latent i (identity when nothing was regrouped); the engine's normal stream (bb_debug_normals, bb_elbo_grad with eps = NULL,
 * bb_run) is keyed by the INTERNAL index.  caller_index: [bb_num_latents(h)]. */
int bb_get_permutation(bb_handle* h, int64_t* caller_index);
/* Sharded runs: the CALLER's flat indices of the latents this handle owns -- its barcodes' loglambda and per-mutant latents, genotype model:
 * theta of its own genotypes (bb_stats.geno_lo / geno_hi) -- i.e. what a gather of the ranks' posteriors takes from this rank; the
 * replicated global blocks are not listed (every rank holds them).  idx: [bb_num_latents(h)], *n entries are written. */
int bb_get_owned(bb_handle* h, int64_t* idx, int64_t* n);

/* AdvancedVI.optimize!: n_steps iterations of
 *   grad(-ELBO) with S reparameterised samples -> optimiser -> theta -= delta. */
int bb_run(bb_handle* h, int64_t n_steps);
/* After BB_ERR_DEVICE from a resident launch (an exchange timed out: bb_last_error says so) the handle's step counter is the
 * device's, but a step may have been left half-way: with samples_per_step > 1 the gradient sums of the samples already taken are
 * discarded and that step's exchange numbers are issued again.  Re-initialise (bb_init_meanfield / bb_set_params) or set
 * launch_mode = 1 before running on; a retry without either is deterministic but unspecified.  BB_ERR_NONFINITE: the steps were taken. */
/* Same arithmetic, launched eagerly with a HIP event pair around every kernel
 * so that per-kernel durations can be reported (bb_stats.avg_*_ms). */
int bb_run_profiled(bb_handle* h, int64_t n_steps);

/* q.dist.m and q.dist.sigma (= softplus(omega)) -- what utils.advi_to_df reads
 * (src/utils.jl:1060).  With world_size > 1 only this handle's shard (and the
 * replicated global blocks) is meaningful; see bb_get_stats().shard_*. */
int bb_get_posterior(bb_handle* h, double* mean, double* sigma);

/* Deterministic test hook: ELBO estimate and its gradient at theta = [mu; omega]
 * with caller-supplied standard-normal draws eps (S x D row-major; NULL = the
 * Philox stream of the current step).  Does not touch optimiser state. */
int bb_elbo_grad(bb_handle* h, const double* mu, const double* omega, const double* eps,
                 int32_t n_samples, double* elbo, double* grad_mu, double* grad_omega);

/* ---- cross-GPU leg of the resident launch (sharded runs, one process per GPU) ----------------------
 * Without it a sharded run steps with two kernels + one ncclAllReduce per MC sample (bb_comm_init).
 * With it the whole step loop of every rank is ONE launch: group leaders push their moment rows into
 * every rank's INBOX over xGMI (peer-mapped fine-grained memory) and every rank adds the same rows in
 * the same order.  Protocol, on every rank, same order:
 *   bb_p2p_export(h, handle)            this rank's inbox as an IPC handle (BB_P2P_HANDLE_BYTES)
 *   (caller all-gathers the handles, rank-major)
 *   bb_p2p_import(h, handles)           maps the peers' inboxes
 *   bb_p2p_selftest(h, &ok)             tokens through every mapped inbox, bounded wait
 *   (caller ANDs `ok` over the ranks)
 *   bb_p2p_enable(h, all_ok)            non-zero: bb_run uses the resident launch from now on;
 *                                       BB_ERR_UNSUPPORTED if this shard cannot (caller ANDs again and
 *                                       calls bb_p2p_enable(h, 0) everywhere if any rank refused)
 * All ranks must then call bb_run with the same step counts.  No reference counterpart. */
#define BB_P2P_HANDLE_BYTES 64
int bb_p2p_export(bb_handle* h, void* handle_out);
int bb_p2p_import(bb_handle* h, const void* handles);
int bb_p2p_selftest(bb_handle* h, int32_t* ok);
int bb_p2p_enable(bb_handle* h, int32_t on);

/* log p(data, z) of the model (normalisers included) and its gradient at a point z
 * of the flat latent vector -- the `logdensity_and_gradient` service an HMC / NUTS
 * sampler needs (the reference's MCMC entry, src/mcmc.jl:86-160, samples the same
 * Turing model).  logp / grad may be NULL.  Does not touch the variational state.
 * On a sharded handle the gradient is this shard's part (global blocks replicated). */
int bb_logdensity_grad(bb_handle* h, const double* z, double* logp, double* grad);

/* ELBO estimates recorded by bb_run (elbo_every > 0): values of steps
 * first_step, first_step + elbo_every, ... ; NaN where not recorded/kept. */
int bb_get_elbo_trace(bb_handle* h, int64_t first_step, int64_t n, double* out);

/* `process_hierarchical_samples!` of utils.advi_to_df (src/utils.jl:1284-1343) on the device, for the hierarchical
 * models (genotype, replicate, multienv_replicate): for every unit of the theta_tilde block, n_samples draws of
 * theta + exp(logtau) * theta_tilde from the current mean-field posterior; median (reported by the reference under
 * the column name `mean`) and corrected std.  n_samples <= 16384.  median / std: [bb_hier_units(h)].
 * On a shard of a sharded run (world_size > 1) only the shard's own entries of the parameter arrays are current: pass the gathered
 * vector through bb_set_params first (a multi-device handle, n_devices > 1, does that itself). */
int64_t bb_hier_units(const bb_handle* h);
int bb_hier_fitness(bb_handle* h, int32_t n_samples, uint64_t seed, double* median, double* std);

/* The engine's normal stream for (step, stream) over latents [lo, hi), for checks. */
int bb_debug_normals(bb_handle* h, int64_t step, uint32_t stream, int64_t lo, int64_t hi, double* out);

/* s_memtime stamps at the pass boundaries of the last launches, [n_blocks][32]; all zero unless
 * the library was built with -DBB_STAMPS (diagnostic build, never the shipped one). */
int bb_debug_stamps(bb_handle* h, uint64_t* out, int64_t n);

int bb_get_stats(bb_handle* h, bb_stats* out);
/* The kernel bb_run launches on this handle, as text: the selected template instance with its arguments in declaration order --
 * "k_res<KIND,P,NT,XG,TT,AP,MS>" (bb_resident.h), "k_stream<KIND,NT,TT>" (bb_stream.h), "k_persist<KIND,P,NT[,XG]>" (bb_persist.h) --
 * or "k_sample<KIND> + k_update<KIND>" for the two-kernel step.  buf: [len], always terminated.  No reference counterpart. */
int bb_kernel_name(bb_handle* h, char* buf, int64_t len);

/* ---- sharded execution -------------------------------------------------------------
 * Barcodes shard over world_size handles (one process per GPU).  Per MC sample the
 * only exchange is the sum of K = bb_stats.n_moments doubles.
 *
 * (1) in-library: RCCL over xGMI.  Rank 0 makes an id, the caller broadcasts it
 *     (e.g. torch.distributed), every rank calls bb_comm_init; bb_run then issues one
 *     ncclAllReduce per sample on the engine's stream. */
#define BB_COMM_ID_BYTES 128
int bb_comm_make_id(void* id_out /* BB_COMM_ID_BYTES */);
int bb_comm_init(bb_handle* h, const void* id /* BB_COMM_ID_BYTES */);
/* (2) split-phase, caller-supplied reducer (MPI, gloo, Julia Distributed ...):
 *     bb_step_moments runs the sampling sweep of the next MC sample and returns this
 *     shard's K partial moments; the caller sums them over shards and passes the
 *     totals to bb_step_apply, which finishes the sample (and the step after the
 *     S-th sample). */
int bb_step_moments(bb_handle* h, double* partial /* K */);
int bb_step_apply(bb_handle* h, const double* total /* K */);

#ifdef __cplusplus
}
#endif
#endif /* BARBAY_HIP_H */
