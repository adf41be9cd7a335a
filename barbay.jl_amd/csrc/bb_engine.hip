// bb_engine.hip -- host side of the C ABI declared in include/barbay_hip.h.
//
// Owns device memory, the HIP stream, captured hipGraphs of the step loop, and (optionally) an
// RCCL communicator.  The compute is the block programs of bb_block.h launched as kernels.
//
// Built twice from this one source:
//   hipcc --offload-arch=gfx950           -> libbarbay_hip.so   (the product)
//   g++ -DBB_EMU -x c++                   -> tests/_emu/libbb_emu.so (sequential host emulation of
//                                            the same block programs; test-only debugging aid)
#include "../../include/barbay_hip.h"
#include "bb_block.h"
#include "bb_persist.h"
#include "bb_resident.h"
#include "bb_stream.h"
#include "bb_inst.h"
#include "bb_hier.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <type_traits>
#include <vector>

#ifndef BB_EMU
#include <dlfcn.h>
#endif

// ------------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int bb_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
extern "C" const char* bb_last_error(void) { return g_err; }
extern "C" const char* bb_version(void) {
#ifdef BB_EMU
    return "barbay_hip 0.1 (host emulation, tests only)";
#else
    return "barbay_hip 0.1 (gfx950)";
#endif
}

// ------------------------------------------------------------------------------------------------
// backend: device memory + kernel launches
// ------------------------------------------------------------------------------------------------
#ifdef BB_EMU
typedef int bbStream;
#define BB_CHECK(x) (x)
static int dmalloc(void** p, size_t n) { *p = calloc(1, n ? n : 1); return *p ? 0 : BB_ERR_DEVICE; }
static void dfree(void* p) { free(p); }
static int h2d(void* d, const void* h, size_t n, bbStream) { memcpy(d, h, n); return 0; }
static int d2h(void* h, const void* d, size_t n, bbStream) { memcpy(h, d, n); return 0; }
static int d2d(void* d, const void* s, size_t n, bbStream) { memcpy(d, s, n); return 0; }
static int dzero(void* d, size_t n, bbStream) { memset(d, 0, n); return 0; }
static int dsync(bbStream) { return 0; }
template <class F>
static void emu_launch(int nblocks, int nthr, size_t lds_doubles, F f) {
    std::vector<double> lds(lds_doubles + 64);
    for (int b = 0; b < nblocks; ++b) {
        BBCtx cx{nthr, b, lds.data()};
        f(cx);
    }
}
#else
typedef hipStream_t bbStream;
#define BB_HIP(call)                                                                              \
    do {                                                                                          \
        hipError_t _e = (call);                                                                   \
        if (_e != hipSuccess) return bb_fail(BB_ERR_DEVICE, "%s: %s", #call, hipGetErrorString(_e)); \
    } while (0)
static int dmalloc(void** p, size_t n) { BB_HIP(hipMalloc(p, n ? n : 1)); return 0; }
static void dfree(void* p) { (void)hipFree(p); }
static int h2d(void* d, const void* h, size_t n, bbStream s) {
    BB_HIP(hipMemcpyAsync(d, h, n, hipMemcpyHostToDevice, s));
    BB_HIP(hipStreamSynchronize(s));
    return 0;
}
static int d2h(void* h, const void* d, size_t n, bbStream s) {
    BB_HIP(hipMemcpyAsync(h, d, n, hipMemcpyDeviceToHost, s));
    BB_HIP(hipStreamSynchronize(s));
    return 0;
}
static int d2d(void* d, const void* s_, size_t n, bbStream s) {
    BB_HIP(hipMemcpyAsync(d, s_, n, hipMemcpyDeviceToDevice, s));
    return 0;
}
static int dzero(void* d, size_t n, bbStream s) { BB_HIP(hipMemsetAsync(d, 0, n, s)); return 0; }
static int dsync(bbStream s) { BB_HIP(hipStreamSynchronize(s)); return 0; }

extern __shared__ __attribute__((aligned(16))) double bb_smem[];

// (descriptors by pointer: scalar loads on demand; by value they cost dozens of SGPR spills per kernel)
template <int KIND>
__global__ void __launch_bounds__(1024) k_sample(const DevModel* __restrict__ Mp, const DevState* __restrict__ Sp, RunArgs A, int NB) {
    const DevModel& M = *Mp;
    const DevState& S = *Sp;
    BBCtx cx{(int)blockDim.x, (int)blockIdx.x, bb_smem};
    bb_block_sample<KIND>(cx, M, S, A, NB);
}
template <int KIND>
__global__ void __launch_bounds__(1024) k_update(const DevModel* __restrict__ Mp, const DevState* __restrict__ Sp, RunArgs A, int NB) {
    const DevModel& M = *Mp;
    const DevState& S = *Sp;
    BBCtx cx{(int)blockDim.x, (int)blockIdx.x, bb_smem};
    bb_block_update<KIND>(cx, M, S, A, NB);
}
typedef void (*bb_step_kernel)(const DevModel*, const DevState*, RunArgs, int);
static bb_step_kernel sample_kernel(int kind) {
    switch (kind) { case 0: return k_sample<0>; case 1: return k_sample<1>; case 2: return k_sample<2>; case 3: return k_sample<3>; default: return k_sample<4>; }
}
static bb_step_kernel update_kernel(int kind) {
    switch (kind) { case 0: return k_update<0>; case 1: return k_update<1>; case 2: return k_update<2>; case 3: return k_update<3>; default: return k_update<4>; }
}
__global__ void __launch_bounds__(256) k_geno(DevModel M, DevState S, RunArgs A, int do_update, int do_sample, int upd_par) {
    BBCtx cx{(int)blockDim.x, (int)blockIdx.x, bb_smem};
    bb_block_geno(cx, M, S, A, (int)gridDim.x, do_update, do_sample, upd_par);
}
__global__ void __launch_bounds__(256) k_geno_sum(DevModel M, DevState S, long long m_lo, long long m_hi) {
    BBCtx cx{(int)blockDim.x, (int)blockIdx.x, bb_smem};
    bb_block_geno_sum(cx, M, S, (int)gridDim.x, m_lo, m_hi);
}
__global__ void __launch_bounds__(256) k_reduce(DevModel M, DevState S, int nblk, int ngeno_blocks) {
    BBCtx cx{(int)blockDim.x, (int)blockIdx.x, bb_smem};
    bb_block_reduce(cx, M, S, nblk, ngeno_blocks);
}
__global__ void __launch_bounds__(256) k_theta_pack(DevModel M, DevState S, double* buf, int g_lo, int g_hi, int W, int unpack) {
    BBCtx cx{(int)blockDim.x, (int)blockIdx.x, bb_smem};
    bb_block_theta_pack(cx, M, S, buf, g_lo, g_hi, W, unpack, (int)gridDim.x);
}
__global__ void __launch_bounds__(256) k_init(DevModel M, DevState S, unsigned long long seed) {
    BBCtx cx{(int)blockDim.x, (int)blockIdx.x, bb_smem};
    bb_block_init(cx, M, S, seed, (int)gridDim.x);
}
__global__ void __launch_bounds__(1024) k_hier(HierArgs H) {
    BBCtx cx{(int)blockDim.x, (int)blockIdx.x, bb_smem};
    bb_block_hier(cx, H, (int)gridDim.x);
}
// transport probe of the cross-GPU leg: this rank's token into every peer's inbox, then every peer's token here
__global__ void __launch_bounds__(64) k_p2p_probe_seq(DevState S, int rank, int world, size_t probe_words_off, unsigned seq, unsigned* result) {
    const int r = threadIdx.x;
    if (r < world) {
        unsigned* out = S.xout_rdy[r] + probe_words_off + 32 * rank;
        __hip_atomic_store(out, 0xB0000000u | (seq << 8) | (unsigned)rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned* in = S.xout_rdy[rank] + probe_words_off + 32 * r;
        const unsigned want = 0xB0000000u | (seq << 8) | (unsigned)r;
        unsigned seen = 0, spins = 0;
        while ((seen = __hip_atomic_load(in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) != want && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(2);
        result[r] = seen == want ? 1u : 0u;
    }
}

__global__ void __launch_bounds__(256) k_normals(unsigned long long seed, unsigned step, unsigned stream, long long lo,
                                                 long long hi, double* out) {
    BBCtx cx{(int)blockDim.x, (int)blockIdx.x, bb_smem};
    bb_block_normals(cx, seed, step, stream, lo, hi, out, (int)gridDim.x);
}

// ---- RCCL, bound at run time so that the library loads (and N = 1 runs) without it -------------
typedef struct { char internal[128]; } bb_ncclUniqueId;
typedef void* bb_ncclComm_t;
struct RcclApi {
    void* lib = nullptr;
    int (*GetUniqueId)(bb_ncclUniqueId*) = nullptr;
    int (*CommInitRank)(bb_ncclComm_t*, int, bb_ncclUniqueId, int) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, bb_ncclComm_t, hipStream_t) = nullptr;
    int (*CommDestroy)(bb_ncclComm_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
static RcclApi g_rccl;
static int rccl_load() {
    if (g_rccl.lib) return 0;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        g_rccl.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.lib) break;
    }
    if (!g_rccl.lib) return bb_fail(BB_ERR_COMM, "cannot load librccl: %s", dlerror());
    g_rccl.GetUniqueId = (int (*)(bb_ncclUniqueId*))dlsym(g_rccl.lib, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(bb_ncclComm_t*, int, bb_ncclUniqueId, int))dlsym(g_rccl.lib, "ncclCommInitRank");
    g_rccl.AllReduce = (int (*)(const void*, void*, size_t, int, int, bb_ncclComm_t, hipStream_t))dlsym(g_rccl.lib, "ncclAllReduce");
    g_rccl.CommDestroy = (int (*)(bb_ncclComm_t))dlsym(g_rccl.lib, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char* (*)(int))dlsym(g_rccl.lib, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllReduce || !g_rccl.CommDestroy)
        return bb_fail(BB_ERR_COMM, "librccl lacks a required symbol");
    return 0;
}
#endif

// ------------------------------------------------------------------------------------------------
// handle
// ------------------------------------------------------------------------------------------------
// Every entry point that takes a handle runs with the handle's device current and restores the caller's on exit: the
// caller may have switched devices since bb_create (torch.cuda.set_device, other handles on other GPUs in this process).
#ifdef BB_EMU
struct DevGuard { explicit DevGuard(int) {} };
#else
struct DevGuard {
    int prev = -1, dev;
    explicit DevGuard(int d) : dev(d) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != dev) (void)hipSetDevice(dev); }
    ~DevGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
    DevGuard(const DevGuard&) = delete;
    DevGuard& operator=(const DevGuard&) = delete;
};
#endif
#define BB_ENTER(h) DevGuard bb_dev_guard_((h)->o.device)

struct bb_handle {
    DevModel M{};
    DevState S{};
    bb_advi_opts o{};
    std::vector<bb_block_range> blocks;
    std::vector<void*> owned;          // device allocations
    long long dev_bytes = 0;           // ... and their total size
    int NB = 0, nthr = 0, nblk = 0, ngeno_blk = 0;
    size_t lds_doubles = 0;            // dynamic LDS of the two-kernel path
    size_t lds_doubles_p0 = 0;         // ... of the resident launch (adds the lambda table)
    size_t lds_doubles_p = 0;          // ... plus the drawn-ahead normals and the cached counts (16 + 8 B per pair)
    long long b_lo = 0, b_hi = 0;      // barcode shard
    int res_ng = 8;                    // k_res: groups of the exchange's first hop (RunArgs.ng)
    int cus = 256;                     // compute units of the device (one resident workgroup each)
    int g_lo = 0, g_hi = 0;            // genotype model: the genotypes whose theta this shard owns (all of them unless cut at genotype boundaries)
    std::vector<int> geno_ptr_h;       // genotype model: CSR offsets over genotypes (sorted geno_idx: first mutant of every genotype)
    long long* d_tile_b = nullptr;     // genotype model, k_res: the tile table (DevState.tile_b / tile_g)
    int* d_tile_g = nullptr;
    long long step = 0;                // host mirror of the device step counter
    int sample = 0;                    // next MC sample inside the current step (split-phase API)
    double elbo_const = 0.0;
    std::vector<double> ld_omega, ld_zero;   // bb_logdensity_grad: constant omega / eps arguments
    // cross-GPU leg of the resident launch (bb_p2p_*)
    void* p2p_inbox = nullptr;               // this rank's inbox (fine-grained device memory)
    size_t p2p_rows_bytes = 0, p2p_bytes = 0, p2p_gran_off = 0;
    double* d_segtab = nullptr;        // host-built tables of the resident launches (host_tables)
    int* d_ldstab = nullptr;
    size_t segtab_cap = 0, ldstab_cap = 0;
    void* p2p_peer[BB_MAX_WORLD] = {};       // peers' inboxes as mapped here
    bool p2p_ready = false, p2p_on = false;
    unsigned p2p_seq = 0;                    // probe sequence number (tokens only ever grow)
    unsigned epoch0 = 0;                     // base of the ready / inbox words: grows whenever the step counter restarts
    long long req_steps = 0;                 // steps ASKED of the resident launch since the last restart (equal on all ranks, whatever a timeout left undone)
    bool p2p_first = false;                  // the next resident launch is the first since the cross-GPU leg was switched on (longer poll limit)
    unsigned* hstatus = nullptr;             // host-mapped status words (DevState.hstatus is their device address)
    double* bak_mu = nullptr;          // bb_elbo_grad: saved parameters
    double* bak_om = nullptr;
    double* eps_buf = nullptr;         // device copy of caller-supplied draws
    size_t eps_cap = 0;
    double* dbg_buf = nullptr;
    size_t dbg_cap = 0;
    bbStream stream{};
    double last_run_ms = 0, avg_sample_ms = 0, avg_update_ms = 0;
    int launches_last_run = 0;
    int64_t bytes_sample = 0, bytes_update = 0;
#ifndef BB_EMU
    hipGraphExec_t graph = nullptr;
    int graph_steps = 0;
    unsigned launch_seq = 0;           // resident launches of this handle so far (RunArgs.launch_tag)
    bool graph_failed = false;         // capture / instantiation failed once (e.g. a collective that cannot be captured): stay eager
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bb_ncclComm_t comm = nullptr;
#endif
    int persist_P = 0;                 // pairs per thread of the persistent launch (0 = not eligible)
    int res_P = 0;                     // > 0: the launch is k_res (bb_resident.h, owner-computes) with this many pair slots per thread
    int res_pf = 0;                    // ... when it fetches a step's window slot (RunArgs.pf)
    bool res_stream = false;           // ... the launch is k_stream (bb_stream.h): res_P pair slots per thread, state streamed from memory
    int res_NB = 0, res_NBL = 0, res_nblk = 0;   // ... its own tile map: barcodes per tile, per leader tile (0: uniform), tiles
    BRLay Yh{};                        // its LDS carve-up (host copy) and device copy
    BRLay* dY = nullptr;
    DevModel* dM = nullptr;            // device copies of the descriptors for the persistent launch
    DevState* dS = nullptr;
    BBLds* dL = nullptr;               // ... and of the resident launch's LDS carve-up (host copy: Lp)
    BBLds Lp{};
    // single-process multi-device handle (bb_advi_opts.n_devices > 1): all the work is in the shards, one per device
    std::vector<bb_handle*> shards;
    bool group_resident = false;       // the shards run resident launches with in-process peer-mapped inboxes
    double* theta_buf = nullptr;       // genotype model: staging of the theta rows gathered from their owners (theta_sync_*)
    bool theta_stale = false;          // a resident run left the theta_g this shard does not own out of date
    bool in_group = false;             // this handle is a shard of a group: its peers' inboxes are plain pointers, not IPC mappings
    // genotype model handed over with geno_idx NOT in consecutive runs (utils.data_to_arrays keeps barcodes in order of appearance,
    // src/utils.jl:692-731): the handle works on the mutants grouped by genotype and presents the caller's order at the ABI.
    // cidx[i] = the caller's flat index of internal latent i (empty: identity); perm_m[m'] = the caller's mutant of internal mutant m'
    std::vector<long long> cidx;
    std::vector<int> perm_m;
    bool force_reduce = false;         // BB_FORCE_ALLREDUCE=1: run the collective path even with one rank (tests)
    bool use_reduce() const { return o.world_size > 1 || M.kind == BB_MODEL_GENOTYPE || force_reduce; }
};

template <class T>
static int dalloc(bb_handle* h, T** p, size_t count) {
    void* q = nullptr;
    int rc = dmalloc(&q, count * sizeof(T));
    if (rc) return rc;
    h->owned.push_back(q);
    h->dev_bytes += (long long)(count * sizeof(T));
    *p = (T*)q;
    return dzero(q, count * sizeof(T), h->stream);
}

extern "C" void bb_default_opts(bb_advi_opts* o) {
    memset(o, 0, sizeof *o);
    o->samples_per_step = 1;
    o->optimizer = BB_OPT_TRUNCATED_ADAGRAD;
    o->eta = 0.1;
    o->tau = 40.0;
    o->window = 100;
    o->resum_every = 0;
    o->pre = 1.0;
    o->post = 0.9;
    o->seed = 0;
    o->device = 0;
    o->rank = 0;
    o->world_size = 1;
    o->steps_per_graph = 0;
    o->elbo_every = 0;
    o->n_devices = 1;
    o->device_ids = nullptr;
}

static void add_block(bb_handle* h, const char* name, int kind, long long n, long long* off) {
    bb_block_range b;
    memset(&b, 0, sizeof b);
    snprintf(b.name, sizeof b.name, "%s", name);
    b.lo = *off;
    b.hi = *off + n;
    h->blocks.push_back(b);
    h->M.blk_lo[kind] = b.lo;
    h->M.blk_hi[kind] = b.hi;
    *off += n;
}

static int upload_prior(bb_handle* h, int kind, const bb_prior* p, double dmean, double dstd, const char* name,
                        bool vector_only, double* sum_log_std) {
    const long long n = h->M.blk_hi[kind] - h->M.blk_lo[kind];
    DevPrior& dp = h->M.pri[kind];
    dp.mean_e = nullptr;
    dp.inv_var_e = nullptr;
    if (!p || !p->mean || !p->std || p->n == 0) {
        dp.mean = dmean;
        dp.inv_var = 1.0 / (dstd * dstd);
        *sum_log_std += (double)n * log(dstd);
        return 0;
    }
    if (p->n == 1 || (n == 1 && p->n == 1)) {
        if (!(p->std[0] > 0)) return bb_fail(BB_ERR_INVALID, "%s: std must be > 0", name);
        dp.mean = p->mean[0];
        dp.inv_var = 1.0 / (p->std[0] * p->std[0]);
        *sum_log_std += (double)n * log(p->std[0]);
        return 0;
    }
    if (vector_only) return bb_fail(BB_ERR_INVALID, "%s accepts only the Vector form [mean, std]", name);
    if (p->n != n) return bb_fail(BB_ERR_INVALID, "%s: Matrix form needs %lld rows, got %lld", name, n, (long long)p->n);
    std::vector<double> iv((size_t)n);
    for (long long i = 0; i < n; ++i) {
        if (!(p->std[i] > 0)) return bb_fail(BB_ERR_INVALID, "%s: std[%lld] must be > 0", name, i);
        iv[(size_t)i] = 1.0 / (p->std[i] * p->std[i]);
        *sum_log_std += log(p->std[i]);
    }
    double *dm = nullptr, *di = nullptr;
    int rc;
    if ((rc = dalloc(h, &dm, (size_t)n))) return rc;
    if ((rc = dalloc(h, &di, (size_t)n))) return rc;
    if ((rc = h2d(dm, p->mean, (size_t)n * 8, h->stream))) return rc;
    if ((rc = h2d(di, iv.data(), (size_t)n * 8, h->stream))) return rc;
    dp.mean = 0;
    dp.inv_var = 0;
    dp.mean_e = dm;
    dp.inv_var_e = di;
    return 0;
}

struct bb_handle;
static int group_create(const bb_model_desc* md, const bb_advi_opts* opts, bb_handle** out);
// Rows of the TruncatedADAGrad window on a SHARDED handle (DevModel.Dh): the window is 2 x window doubles per latent -- a hundred
// times everything else a handle holds -- and a shard only ever updates its own barcodes' latents, the replicated blocks and (genotype
// model) the genotype block: one contiguous range of the flat vector per (block, replicate), the ranges every tile's segment table
// is cut from (bb_build_segs / br_build_segs).  A row packs those ranges one after the other; a segment carries the difference
// between a latent's flat index and its entry (bb_hdelta), so the kernels pay one subtraction per pair.  Differences are even: pairs
// stay whole and 16-byte aligned.  BB_NO_HIST_PACK=1 keeps full rows.
static void hist_rows(bb_handle* h) {
    DevModel& M = h->M;
    M.Dh = M.Dp;
    for (int k = 0; k < BK_COUNT; ++k) M.hd0[k] = M.hd1[k] = 0;
    for (int r = 0; r < BB_MAX_REP; ++r) M.hdl[r] = 0;
    const char* ev = getenv("BB_NO_HIST_PACK");
    if (h->o.world_size <= 1 || (ev && atoi(ev) > 0) || M.Dp >= (1ll << 31)) return;      // (a segment keeps its difference in an int)
    const long long b0 = h->b_lo, nbt = h->b_hi - h->b_lo;
    const long long m0 = std::max(h->b_lo, M.nn) - M.nn, nmt = (std::max(h->b_hi, M.nn) - M.nn) - m0;
    long long c = 0;          // next free entry of the row
    // one range [a, a + len): starts on an entry of a's parity; two entries of slack (an edge pair's other half is read, never used)
    auto place = [&](long long a, long long len) { const long long at = c + ((a - c) & 1); c = at + len + 2; return a - at; };
    // R ranges of one block, `stride` apart in the flat vector, `len` long: packed `lp` apart with lp of the stride's parity
    auto place_r = [&](int blk, long long a0, long long stride, long long len, int R) {
        const long long lp = len + 2 + ((stride - (len + 2)) & 1);
        const long long at = c + ((a0 - c) & 1);
        M.hd0[blk] = a0 - at;
        M.hd1[blk] = stride - lp;
        c = at + (long long)R * lp;
    };
    M.hd0[BK_SPOP] = place(M.blk_lo[BK_SPOP], M.blk_hi[BK_SPOP] - M.blk_lo[BK_SPOP]);
    M.hd0[BK_LSPOP] = place(M.blk_lo[BK_LSPOP], M.blk_hi[BK_LSPOP] - M.blk_lo[BK_LSPOP]);
    if (M.kind == BB_MODEL_FITNESS || M.kind == BB_MODEL_MULTIENV) {
        M.hd0[BK_S] = place(M.blk_lo[BK_S] + m0 * M.E, nmt * M.E);
        M.hd0[BK_LS] = place(M.blk_lo[BK_LS] + m0 * M.E, nmt * M.E);
    } else if (M.kind == BB_MODEL_GENOTYPE) {
        M.hd0[BK_S] = place(M.blk_lo[BK_S], M.G);      // (every rank updates every theta_g on the all-reduce step; the resident launch only its own)
        M.hd0[BK_TT] = place(M.blk_lo[BK_TT] + m0, nmt);
        M.hd0[BK_LT] = place(M.blk_lo[BK_LT] + m0, nmt);
        M.hd0[BK_LS] = place(M.blk_lo[BK_LS] + m0, nmt);
    } else {
        const long long E_ = M.kind == BB_MODEL_MULTIENV_REPLICATE ? M.E : 1;
        M.hd0[BK_S] = place(M.blk_lo[BK_S] + m0 * E_, nmt * E_);
        for (int blk : {BK_TT, BK_LT, BK_LS}) place_r(blk, M.blk_lo[blk] + m0 * E_, M.nb * E_, nmt * E_, M.R);
    }
    for (int r = 0; r < M.R; ++r) M.hdl[r] = place(M.off_l[r] + b0 * M.T[r], nbt * M.T[r]);
    const long long Dh = (c + 7) & ~7ll;
    if (Dh >= M.Dp) {          // nothing gained (a shard that owns almost everything): full rows, no differences
        for (int k = 0; k < BK_COUNT; ++k) M.hd0[k] = M.hd1[k] = 0;
        for (int r = 0; r < BB_MAX_REP; ++r) M.hdl[r] = 0;
        return;
    }
    M.Dh = Dh;
}

static int create_inner(const bb_model_desc* md, const bb_advi_opts* opts, bb_handle** out);
// bb_create -> create_inner (and the shards of a multi-device handle): lay the loglambda block out in FRONT of the per-genotype / per-mutant
// blocks (the handle's internal order; the caller's stays the reference's source order) -- see bb_create
static thread_local bool g_loglambda_first = false;
static int ensure_scratch(bb_handle* h);
static void owned_ranges(const bb_handle* sh, std::vector<std::pair<long long, long long>>& out);
static RunArgs make_args(const bb_handle* h, long long step, int sample, int S, bool apply, bool with_elbo);
static int theta_sync_local(bb_handle* const* hs, int n);
#ifndef BB_EMU
static int launch_check();
#endif

// ------------------------------------------------------------------------------------------------
// persistent launch (bb_persist.h)
// ------------------------------------------------------------------------------------------------
#ifndef BB_EMU
static bb_persist_kernel persist_kernel(int kind, int P, int nthr, bool xg = false, const char** nm = nullptr) {
#ifdef BB_FAST_BUILD   /* experiment builds (tools/xp.py): only the instances the C2 / C4 workloads use, in this one translation unit */
    if (nm) *nm = "(experiment build)";
    if (!xg && nthr > 512 && P == 1 && kind == 0) return k_persist<0, 1, 1024>;
    return nullptr;
#else
    return bb_persist_instance(kind, P, nthr, xg, nm);
#endif
}
#endif

// the time-point count where all replicates share it (the compile-time-T instances), else 0
static int uniform_T(const DevModel& M) {
    for (int r = 1; r < M.R; ++r) if (M.T[r] != M.T[0]) return 0;
    return M.T[0];
}

#ifndef BB_EMU
static bb_res_kernel res_kernel(int kind, int P, int nthr, bool xg, int T, bool ap, bool ms = false, const char** nm = nullptr) {
#ifdef BB_FAST_BUILD
    if (nm) *nm = "(experiment build)";
    if (xg || ms) return nullptr;
    if (ap) {
        if (nthr > 512 && P == 1 && kind == 0) return k_res<0, 1, 1024, false, 0, true>;
        if (nthr > 256 && nthr <= 512 && P == 3 && kind == 3) return k_res<3, 3, 512, false, 0, true>;
        return nullptr;
    }
    if (nthr > 512 && P == 1 && kind == 0) return T == 8 ? k_res<0, 1, 1024, false, 8> : k_res<0, 1, 1024, false>;
    if (nthr > 512 && P == 1 && kind == 1) return T == 6 ? k_res<1, 1, 1024, false, 6> : k_res<1, 1, 1024, false>;
    if (nthr > 256 && nthr <= 512 && P == 2 && kind == 0) return T == 8 ? k_res<0, 2, 512, false, 8> : k_res<0, 2, 512, false>;
    if (nthr > 256 && nthr <= 512 && P == 3 && kind == 3) return T == 6 ? k_res<3, 3, 512, false, 6> : k_res<3, 3, 512, false>;
    if (nthr > 256 && nthr <= 512 && P == 2 && kind == 3) return T == 6 ? k_res<3, 2, 512, false, 6> : k_res<3, 2, 512, false>;
    if (nthr > 512 && P == 1 && kind == 2) return T == 8 ? k_res<2, 1, 1024, false, 8> : k_res<2, 1, 1024, false>;
    return nullptr;
#else
    switch (kind) {
    case 0: return bb_res_instance_k0(P, nthr, xg, T, ap, ms, nm);
    case 1: return bb_res_instance_k1(P, nthr, xg, T, ap, ms, nm);
    case 2: return bb_res_instance_k2(P, nthr, xg, T, ap, ms, nm);
    case 3: return bb_res_instance_k3(P, nthr, xg, T, ap, ms, nm);
    default: return bb_res_instance_k4(P, nthr, xg, T, ap, ms, nm);
    }
#endif
}
#endif

#ifndef BB_EMU
static bb_stream_kernel stream_kernel(int kind, int nthr, int T, const char** nm = nullptr, bool ms = false) {
#ifdef BB_FAST_BUILD
    if (nm) *nm = "(experiment build)";
    if (ms) return nullptr;
    if (nthr == 1024 && kind == 0 && T == 8) return k_stream<0, 1024, 8>;
    if (nthr == 1024 && kind == 2 && T == 8) return k_stream<2, 1024, 8>;
    if (nthr == 1024 && kind == 3 && T == 6) return k_stream<3, 1024, 6>;
    return nullptr;
#else
    return ms ? bb_stream_instance_ms(kind, nthr, T, nm) : bb_stream_instance(kind, nthr, T, nm);
#endif
}
#endif

// pairs a tile can hold: every segment contributes count/2 + 1 at most
static long long tile_pairs_bound(const DevModel& M, long long NB) {
    long long p = 0;
    for (int r = 0; r < M.R; ++r) p += NB * M.T[r] / 2 + 1;
    if (M.kind == 0 || M.kind == 1) p += 2 * (NB * M.E / 2 + 1);
    else if (M.kind == 2) p += 3 * (NB / 2 + 1);
    else if (M.kind == 3) p += (NB / 2 + 1) + 3ll * M.R * (NB / 2 + 1);
    else p += (NB * M.E / 2 + 1) + 3ll * M.R * (NB * M.E / 2 + 1);
    p += 2 * (M.nt1 / 2 + 1);
    return p;
}

// device copies of the descriptors (kernels read them through pointers); re-sent whenever the host copy changes
static int sync_descriptors(bb_handle* h) {
    int rc;
    if (!h->dM && ((rc = dalloc(h, &h->dM, 1)) || (rc = dalloc(h, &h->dS, 1)) || (rc = dalloc(h, &h->dL, 1)))) return rc;
    if ((rc = h2d(h->dM, &h->M, sizeof(DevModel), h->stream)) || (rc = h2d(h->dS, &h->S, sizeof(DevState), h->stream))) return rc;
    h->Lp = bb_lds_layout(h->M.R, h->M.E, h->M.kind, h->M.Ttot, h->M.nt1, h->M.K, h->NB, h->nthr, 1);   // the resident launch's carve-up
    if ((rc = h2d(h->dL, &h->Lp, sizeof(BBLds), h->stream))) return rc;
    if (!h->dY && (rc = dalloc(h, &h->dY, 1))) return rc;
    return h2d(h->dY, &h->Yh, sizeof(BRLay), h->stream);
}

// Genotype model: the tile table of k_res (br_tile_geno) -- tiles of at most NB barcodes (the first eight: NBL, if > 0) whose cuts
// inside the mutants fall on genotype boundaries; tg[i] = first genotype tile i owns.  False if a genotype does not fit a tile.
static bool build_geno_tiles(const bb_handle* h, int NB, int NBL, std::vector<long long>& tb, std::vector<int>& tg) {
    const DevModel& M = h->M;
    const std::vector<int>& ptr = h->geno_ptr_h;
    auto geno_of = [&](long long m) { return (int)(std::upper_bound(ptr.begin(), ptr.end(), (int)m) - ptr.begin()) - 1; };
    auto gcut = [&](long long b) { return (b <= M.nn || b <= h->b_lo) ? h->g_lo : (b >= h->b_hi ? h->g_hi : geno_of(b - M.nn)); };
    tb.clear();
    tg.clear();
    long long b = h->b_lo;
    while (b < h->b_hi) {
        const int cap = (NBL > 0 && tb.size() < (size_t)h->res_ng) ? NBL : NB;
        long long e = std::min<long long>(b + cap, h->b_hi);
        if (e < h->b_hi && e > M.nn) {
            e = M.nn + ptr[(size_t)geno_of(e - M.nn)];       // back to the first mutant of the genotype the cut fell into
            if (e <= b) return false;
        }
        tb.push_back(b);
        tg.push_back(gcut(b));
        b = e;
    }
    if (tb.empty()) { tb.push_back(h->b_lo); tg.push_back(h->g_lo); }
    tb.push_back(h->b_hi);
    tg.push_back(h->g_hi);
    for (size_t i = 0; i + 1 < tg.size(); ++i) if (tg[i + 1] - tg[i] > NB) return false;      // (theta stage table: NB entries)
    return NB < 32768;
}

// the owner-computes launch (bb_resident.h) where the shape allows it; BB_NO_RES=1 keeps k_persist (A/B runs)
// any_parity: also the AP instances (odd time-point counts, odd loglambda offset).  Measured (C2-sized fitness_normal with 7 / 5
// time points, the fifth model): they are 1 - 5 % SLOWER than k_persist -- parity is a run-time property of every pair there,
// 45 spilled registers and two Philox draws in divergent lanes -- so setup_persistent asks for them only where k_persist cannot
// run: the genotype model (whose other choice is the two-kernel step) or after k_persist has refused the shape.
// The tables a tile of k_res / k_stream needs before its first step, built HERE instead of in every launch's prologue (where
// thread 0 of each tile spent 3.4 us on its segment table behind a chain of scalar loads, and the per-lane descriptor loads of the
// LDS tables another 1.8 us -- profiles/r03z_round3_final/fixed_cost.txt): per tile its segment table (br_build_segs, the same code),
// once the tile-independent LDS tables (br_table_*).  BB_NO_HOST_TABLES=1: the kernels build them (A/B).
template <int KIND>
static void host_tables_kind(bb_handle* h, const RunArgs& A, const DevState& Sh, int nblk, int stride, std::vector<double>& tab) {
    for (int b = 0; b < nblk; ++b) {
        const BBTile t = KIND == 2 ? br_tile_geno(h->M, Sh, b, h->res_NB) : br_tile(h->M, A, b, h->res_NB);
        const int g0 = KIND == 2 ? Sh.tile_g[b] : 0, g1 = KIND == 2 ? Sh.tile_g[b + 1] : 0;
        double* rec = tab.data() + (size_t)b * stride;
        const int n = br_build_segs<KIND>((BRSeg*)rec, h->M, h->Yh, t, b == 0, g0, g1);
        ((int*)(rec + stride - 1))[0] = n;
    }
}
static bool host_tables(bb_handle* h, const std::vector<long long>& tb, const std::vector<int>& tg) {
    h->S.segtab = nullptr;
    h->S.ldstab = nullptr;
    h->S.segtab_stride = 0;
    const char* ev = getenv("BB_NO_HOST_TABLES");
    if (ev && atoi(ev) > 0) return true;
    const DevModel& M = h->M;
    const int nblk = h->res_nblk, stride = BR_SEG_DOUBLES * (4 + 4 * M.R + 1) + 1;
    RunArgs A = make_args(h, 0, 0, 1, true, false);
    A.nblk = nblk; A.nbl = h->res_NBL; A.ng = h->res_ng;
    DevState Sh = h->S;
    Sh.tile_b = tb.data();
    Sh.tile_g = tg.data();
    std::vector<double> tab((size_t)nblk * stride, 0.0);
    switch (M.kind) {
    case 0: host_tables_kind<0>(h, A, Sh, nblk, stride, tab); break;
    case 1: host_tables_kind<1>(h, A, Sh, nblk, stride, tab); break;
    case 2: host_tables_kind<2>(h, A, Sh, nblk, stride, tab); break;
    case 3: host_tables_kind<3>(h, A, Sh, nblk, stride, tab); break;
    default: host_tables_kind<4>(h, A, Sh, nblk, stride, tab);
    }
    const int K = M.K, Tt = M.Ttot, R = M.R;
    std::vector<int> img((size_t)3 * K + 4 * Tt + 4 * R + 4, 0);
    for (int j = 0; j < K; ++j) br_table_row(M, h->Yh, j, &img[2 * j], &img[2 * K + j]);
    for (int j = 0; j < Tt; ++j) br_table_time(M, h->Yh.L, j, &img[3 * K + 4 * j]);
    for (int r = 0; r < R; ++r) br_table_rep(M, h->Yh, r, &img[3 * K + 4 * Tt + 4 * r]);
    // (the handle keeps one buffer of each, sized for the largest tile map it has seen)
    if (tab.size() > h->segtab_cap) { double* d = nullptr; if (dalloc(h, &d, tab.size())) return false; h->d_segtab = d; h->segtab_cap = tab.size(); }
    if (img.size() > h->ldstab_cap) { int* d = nullptr; if (dalloc(h, &d, img.size())) return false; h->d_ldstab = d; h->ldstab_cap = img.size(); }
    if (h2d(h->d_segtab, tab.data(), tab.size() * 8, h->stream) || h2d(h->d_ldstab, img.data(), img.size() * 4, h->stream)) return false;
    h->S.segtab = h->d_segtab;
    h->S.segtab_stride = stride;
    h->S.ldstab = h->d_ldstab;
    return true;
}

static bool try_resident(bb_handle* h, bool any_parity) {
    const char* ev = getenv("BB_NO_RES");
    if (ev && atoi(ev) > 0) return false;
    if (!br_eligible(h->M)) return false;
    if (!any_parity && br_any_parity(h->M)) return false;
    // tile map: leaders (tiles 0 .. 7) hold `frac` of a tile's barcodes (br_tile); BB_TUNE_LEAD=100 keeps all tiles alike
    const long long nbar = std::max<long long>(h->b_hi - h->b_lo, 1);
    // The two-kernel step may run more tiles than the device has compute units (its tiles must fit LDS with ITS tables: config 5 on one
    // GPU runs 512 of them); a resident launch needs every tile resident -- one per compute unit: its own base map then
    int NB0 = h->NB, nblk0 = h->nblk;
    if (nblk0 > h->cus && !getenv("BB_TUNE_NB")) { NB0 = (int)((nbar + h->cus - 1) / h->cus); nblk0 = (int)((nbar + NB0 - 1) / NB0); }
    // BB_TUNE_RES_NB (tests): the resident launch's own tile size -- a cut of a BASELINE problem with the full-size tile geometry even where
    // the two-kernel step's tables would not fit such a tile (config 5 on one GPU: 782 barcodes per tile)
    const char* res_nb = getenv("BB_TUNE_RES_NB");
    if (res_nb && atoi(res_nb) > 0) { NB0 = atoi(res_nb); nblk0 = (int)((nbar + NB0 - 1) / NB0); }
    int NB = NB0, NBL = 0, nblk = nblk0;
    // Groups of the exchange's first hop.  Round 2: 16 on one GPU (a leader then fetched its 16 members' rows in one round of loads, the
    // consume ran on two thread groups).  Round 3, tagged rows + leaders that poll their members in chunks of eight on as many thread
    // groups as the tile has (bbp_leader_reduce_tg, PAR): 8 groups of 32 are ONE round of polls where the tile has four thread groups, and
    // every tile's consume is one thread group's eight loads -- C2 82.7 -> 83.5 k steps/s, C4 94.1 -> 94.9 k, C3 57.1 -> 57.7 k against 16
    // (profiles/r03g_parallel_leaders).  The cross-GPU inbox protocol is laid out for 8.  BB_TUNE_NG overrides (8 or 16).
    if (h->M.K + 2 * h->M.nt1 > h->nthr) return false;      // (one thread per row entry: bbp_consume_tg / _tgx, the leaders' chunk sums)
    {
        const int KK = h->M.K + 2 * h->M.nt1;
        const bool can16 = !h->p2p_on && nblk0 >= 64 && KK <= 128 && h->nthr >= 2 * (KK <= 64 ? 64 : 128);
        int ng = (can16 && !(BR_TG && BR_LEAD_PAR && h->nthr >= 4 * ((KK + 63) & ~63))) ? 16 : 8;
        if ((ev = getenv("BB_TUNE_NG")) && (atoi(ev) == 8 || (atoi(ev) == 16 && !h->p2p_on && KK <= 128 && h->nthr >= 2 * (KK <= 64 ? 64 : 128)))) ng = atoi(ev);
#if BR_TG
        // self-validating rows: a leader takes its members' rows in batches of eight loads per lane -- 32 groups of 8 on a full grid:
        // one batch, one round trip (the tile's consume then runs on four thread groups)
        if (ev && atoi(ev) == 32 && !h->p2p_on && nblk0 >= 64 && h->nthr >= 4 * ((KK + 63) & ~63)) ng = 32;
#endif
        h->res_ng = ng;
    }
    const int NGh = h->res_ng;
    int pct = (ev = getenv("BB_TUNE_LEAD")) ? atoi(ev) : 65;
    if (pct < 10 || pct > 100) pct = 100;
    const bool nb_fixed = getenv("BB_TUNE_NB") != nullptr || (res_nb && atoi(res_nb) > 0);
    ev = getenv("BB_TUNE_LEAD");
    if (pct < 100 && nblk0 >= 2 * NGh && (!nb_fixed || ev)) {
        if (!nb_fixed) {
            const double tiles = (double)nblk0 - (double)NGh * (1.0 - pct / 100.0);      // in units of a full tile
            NB = (int)std::ceil((double)nbar / tiles);
        }
        NBL = std::max(1, (int)(NB * (pct / 100.0)));
        const long long rest = nbar - (long long)NGh * NBL;
        nblk = NGh + (int)((std::max<long long>(rest, 0) + NB - 1) / NB);
        const long long p_uni = (br_tile_span(h->M, NB0, true) + h->nthr - 1) / h->nthr, p_new = (br_tile_span(h->M, NB, true) + h->nthr - 1) / h->nthr;
        // stay uniform where rounding pushed the map over the grid that fits, or the slightly larger tiles need another pair slot
        if (nblk > nblk0 + (nb_fixed ? 8 : 0) || p_new > p_uni) { NB = NB0; NBL = 0; nblk = nblk0; }
    }
    std::vector<long long> tb;
    std::vector<int> tg;
    if (h->M.kind == BB_MODEL_GENOTYPE) {
        // cuts on genotype boundaries leave tiles partly empty: grow the tile until the map fits the grid again
        const int limit = std::max(nblk0, std::min(nblk0 + 8, h->cus));      // (the exchange buffers hold nblk + 8 tiles)
        bool ok = false;
        for (int grow = 0; grow <= NB / 2 + 8 && !ok; ++grow) {
            const int nb = NB + grow, nbl = NBL > 0 ? std::max(1, (int)((long long)NBL * nb / NB)) : 0;
            if (build_geno_tiles(h, nb, nbl, tb, tg) && (int)tb.size() - 1 <= limit) { ok = true; NB = nb; NBL = nbl; }
            else if (nb_fixed && NBL > 0 && build_geno_tiles(h, nb, 0, tb, tg) && (int)tb.size() - 1 <= limit) { ok = true; NB = nb; NBL = 0; }
        }
        if (!ok) return false;
        nblk = (int)tb.size() - 1;
        if (h->p2p_on && nblk < 8) return false;
    }
    if (!h->p2p_on && h->M.Dh != h->M.Dp) return false;      // (packed window rows: only the cross-GPU instances read the segments' differences)
    int P = (int)((br_tile_span(h->M, NB, true) + h->nthr - 1) / h->nthr);
    // more pair slots than the register file holds: k_stream (bb_stream.h) -- the same tile map, the per-pair state streamed
    bool stream = false;
    const bool force_stream = (ev = getenv("BB_TUNE_STREAM")) && atoi(ev) > 0;        // (tests: small shapes through k_stream)
    if (force_stream || P > (h->nthr > 512 ? 2 : (h->nthr > 256 ? 3 : 4))) {
        const int T0 = uniform_T(h->M);
        const bool nostream = (ev = getenv("BB_NO_STREAM")) && atoi(ev) > 0;
        // every replicate the same even T (instances: 4, 6, 8), flat-index-aligned pairs, one GPU; several samples per step / the ELBO trace:
        // the MS instances (T = 6, 8 at 1024 threads; tests: 512)
        stream = !nostream && !br_any_parity(h->M) && (T0 == 8 || T0 == 6 || T0 == 4) && h->nthr % 64 == 0 && !h->p2p_on;
        if (stream) {          // (a barcode takes a power-of-two number of lanes there: T = 6 four)
            P = (int)((br_tile_span(h->M, NB, true, true) + h->nthr - 1) / h->nthr);
            if (P > 64) stream = false;
        }
        if (!stream && !force_stream) return false;
        if (!stream) { P = (int)((br_tile_span(h->M, NB, true) + h->nthr - 1) / h->nthr); if (P > (h->nthr > 512 ? 2 : (h->nthr > 256 ? 3 : 4))) return false; }
    }
    // When the window slot is fetched (RunArgs.pf).  In the exchange's shadow (round 2) its 32 B per latent of HBM reads compete with
    // the exchange's own loads and stores: at the start of the S pass instead, C2 73.1 -> 77.7 k steps/s, C4 87.1 -> 89.1 k
    // (profiles/r03b_tagged_rows/prefetch_timing_on_lean_kernel.txt) -- where the slot buffer fits beside the moment contributions
    int pf = (ev = getenv("BB_TUNE_PF")) ? atoi(ev) : 1;
    if (pf < 0 || pf > 3 || h->o.optimizer != BB_OPT_TRUNCATED_ADAGRAD) pf = 0;
    if (stream) pf = 0;
    BRLay Y = br_layout(h->M, NB, h->nthr, P, h->p2p_on ? 8 * h->o.world_size : 0, pf == 1 || pf == 2, stream);
    if (pf == 3) Y = br_layout(h->M, NB, h->nthr, P, h->p2p_on ? 8 * h->o.world_size : 0, false);
    else if (pf != 0 && (size_t)Y.total * 8 > 160 * 1024) { pf = 0; Y = br_layout(h->M, NB, h->nthr, P, h->p2p_on ? 8 * h->o.world_size : 0, false); }      // (no room for a slot buffer of its own: in the exchange's shadow; pf = 3, behind the exchange, measured 6% slower on C3)
    if ((size_t)Y.total * 8 > 160 * 1024) return false;
#ifndef BB_EMU
    const bool ms = h->o.samples_per_step != 1 || h->o.elbo_every != 0;
    const void* k = stream ? (const void*)stream_kernel(h->M.kind, h->nthr, uniform_T(h->M), nullptr, ms)
                           : (const void*)res_kernel(h->M.kind, P, h->nthr, h->p2p_on, uniform_T(h->M), br_any_parity(h->M), ms);
    if (!k) return false;
    const int lds = Y.total * 8;
    if (lds > 64 * 1024 && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess) return false;
    int per_cu = 0;
    hipDeviceProp_t pr;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, h->nthr, (size_t)lds) != hipSuccess ||
        hipGetDeviceProperties(&pr, h->o.device) != hipSuccess || (long long)per_cu * pr.multiProcessorCount < nblk) return false;
#endif
    if (h->M.kind == BB_MODEL_GENOTYPE) {
        if (!h->d_tile_b && (dalloc(h, &h->d_tile_b, (size_t)h->nblk + 10) || dalloc(h, &h->d_tile_g, (size_t)h->nblk + 10))) return false;
        if (h2d(h->d_tile_b, tb.data(), tb.size() * 8, h->stream) || h2d(h->d_tile_g, tg.data(), tg.size() * 4, h->stream)) return false;
        h->S.tile_b = h->d_tile_b;
        h->S.tile_g = h->d_tile_g;
    }
    h->Yh = Y;
    h->res_P = P;
    h->res_NB = NB;
    h->res_NBL = NBL;
    h->res_nblk = nblk;
    h->res_pf = pf;
    h->res_stream = stream;
    if (stream && h->o.samples_per_step > 1 && ensure_scratch(h)) { h->res_P = 0; return false; }      // (the samples' gradient sums live in gacc_mu / gacc_om)
    h->lds_doubles_p = (size_t)Y.total;
    if (!host_tables(h, tb, tg)) { h->res_P = 0; return false; }
    return true;
}

static int setup_persistent(bb_handle* h) {
    h->persist_P = 0;
    const char* ev = getenv("BB_NO_PERSIST");
    const bool want = h->o.launch_mode != 1 && !(ev && atoi(ev) > 0 && h->o.launch_mode == 0);
    const char* why = nullptr;
    // several MC samples per step (Turing.ADVI(samples_per_step, ..), src/vi.jl:98) and ELBO recording: k_res's MS instances (round 4: sharded too --
    // every sample is an exchange of its own, the inbox epochs count exchanges)
    const bool ms = h->o.samples_per_step != 1 || h->o.elbo_every != 0;
    if (h->force_reduce || (h->o.world_size != 1 && !h->p2p_on)) why = "sharded run";
    else if (h->p2p_on && h->nblk < 8) why = "fewer than 8 tiles on this rank";   // (k_res's own tile map never has fewer tiles than this one)
    h->res_P = 0;
    const bool ap_first = ms || h->M.kind == BB_MODEL_GENOTYPE || (getenv("BB_TUNE_AP") && atoi(getenv("BB_TUNE_AP")) > 0);
    if (!why && want && try_resident(h, ap_first)) { h->persist_P = h->res_P; return 0; }
    if (!why && ms) why = h->o.samples_per_step != 1 ? "samples_per_step != 1 and the shape has no owner-computes instance" : "ELBO recording is on and the shape has no owner-computes instance";
    if (!why && h->M.kind == BB_MODEL_GENOTYPE)
        why = h->M.geno_sorted ? "genotype model: no tile map with whole genotypes per tile fits the device" : "genotype model: geno_idx is not in consecutive runs (a tile must hold whole genotypes)";
    int P = 0;
    if (!why) {
        P = (int)((tile_pairs_bound(h->M, h->NB) + h->nthr - 1) / h->nthr);
        if (P == 3 && h->nthr > 512) P = 4;      // (no 3-pair instance at 1024 threads; 4 is refused just below)
        if (P > (h->nthr > 512 ? 2 : 4)) why = "tile too large for the register-resident state";
        if (h->p2p_on && P != 1) why = "sharded resident launch holds one pair per thread";
        h->lds_doubles_p = h->lds_doubles_p0 + (size_t)3 * P * h->nthr;        // drawn-ahead normals (16 B / pair) + cached counts (8 B)
    }
#ifndef BB_EMU
    if (!why && want) {
        bb_persist_kernel k = persist_kernel(h->M.kind, P, h->nthr, h->p2p_on);
        const int lds = (int)(h->lds_doubles_p * 8);
        if (lds > 160 * 1024) why = "tile does not fit LDS with the lambda table";
        else if (!k) why = "no kernel instance";
        else {
            if (lds > 64 * 1024 && hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
                why = "cannot raise dynamic LDS";
            int per_cu = 0, cus = 0;
            hipDeviceProp_t pr;
            if (!why && hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)k, h->nthr, (size_t)lds) == hipSuccess &&
                hipGetDeviceProperties(&pr, h->o.device) == hipSuccess) {
                cus = pr.multiProcessorCount;
                if ((long long)per_cu * cus < h->nblk) why = "grid does not fit resident on the device";
            } else if (!why) why = "occupancy query failed";
        }
    }
#endif
    if (why) {
        // k_persist cannot (tile too large for its state, no instance, sharded with more than one pair per thread ...): k_res's
        // any-parity instances as the second chance
        const bool structural = ms || h->force_reduce || (h->o.world_size != 1 && !h->p2p_on) || (h->p2p_on && h->nblk < 8);
        if (want && !structural && !ap_first && try_resident(h, true)) { h->persist_P = h->res_P; return 0; }
        if (h->o.launch_mode == 2) return bb_fail(BB_ERR_UNSUPPORTED, "launch_mode = 2 (persistent) not possible: %s", why);
        return 0;
    }
    if (want) h->persist_P = P;
    return 0;
}

#ifdef BB_EMU
// Host emulation of the resident launch, phase by phase so that several handles (the ranks of a sharded run, all in
// this process) can be stepped in lock step: phase 0 prologue, 1 sample + publish + draw-ahead, 2 leaders,
// 3 consume + update, 4 epilogue.
struct EmuPersist {
    bb_handle* h = nullptr;
    std::vector<double> lds;
    std::vector<unsigned char> st;      // BBPst<P>[nblk * nthr]
    RunArgs A;
    int ok = 1;
};

template <int KIND, int PP>
static void emu_persist_phase(EmuPersist& E, int phase, long long it, long long nsteps) {
    bb_handle* h = E.h;
    const RunArgs& A = E.A;
    BBPst<PP>* st = (BBPst<PP>*)E.st.data();
    auto cxof = [&](int b) { return BBCtx{h->nthr, b, E.lds.data() + (size_t)b * (h->lds_doubles_p + 64), &h->Lp}; };
    const BBLds L = bb_lds_layout(h->M.R, h->M.E, KIND, h->M.Ttot, h->M.nt1, h->M.K, h->NB, h->nthr, 1);
    const unsigned long long step = (unsigned long long)(h->step + it);
    const unsigned abs_epoch = A.xepoch0 + (unsigned)(step + 1), epoch = abs_epoch;
    const int par = (int)(step & 1);
    const bool xg = h->p2p_on;
    if (phase == 0) {
        for (int b = 0; b < h->nblk; ++b) {
            BBCtx cx = cxof(b);
            bbp_prologue<KIND, PP>(cx, h->M, h->S, A, h->NB, st + (size_t)b * h->nthr);
            bbp_draw_ahead<KIND, PP>(cx, h->M, A, h->NB, st + (size_t)b * h->nthr, (unsigned long long)h->step);
        }
    } else if (phase == 1) {
        for (int b = 0; b < h->nblk; ++b) {
            BBCtx cx = cxof(b);
            bbp_sample<KIND, PP>(cx, h->M, h->S, A, h->NB, st + (size_t)b * h->nthr, step);
            bbp_publish_row(cx, h->M, h->S, L, epoch);
            bbp_draw_ahead<KIND, PP>(cx, h->M, A, h->NB, st + (size_t)b * h->nthr, step + 1);
        }
    } else if (phase == 2) {
        for (int g = 0; g < bbp_groups(A); ++g) {
            BBCtx cx = cxof(g);
            if (xg) bbp_leader_reduce<true>(cx, h->M, h->S, A, L, par, epoch, &E.ok, abs_epoch);
            else bbp_leader_reduce<false>(cx, h->M, h->S, A, L, par, epoch, &E.ok, abs_epoch);
        }
    } else if (phase == 3) {
        for (int b = 0; b < h->nblk; ++b) {
            BBCtx cx = cxof(b);
            bbp_prefetch_slot<KIND, PP>(cx, h->M, h->S, A, h->NB, st + (size_t)b * h->nthr, bb_slot_of(A, step).slot);
            bbp_residual_ahead<KIND>(cx, h->M, h->NB, A);
            if (xg) bbp_consume<true>(cx, h->M, h->S, A, L, par, epoch, &E.ok, abs_epoch);
            else bbp_consume<false>(cx, h->M, h->S, A, L, par, epoch, &E.ok, abs_epoch);
            bbp_finish<KIND>(cx, h->M, h->S, A, h->NB);
            bbp_update<KIND, PP>(cx, h->M, h->S, A, h->NB, st + (size_t)b * h->nthr, bb_slot_of(A, step));
        }
    } else {
        for (int b = 0; b < h->nblk; ++b) {
            BBCtx cx = cxof(b);
            bbp_epilogue<KIND, PP>(cx, h->M, h->S, A, h->NB, st + (size_t)b * h->nthr, (unsigned long long)(h->step + nsteps), E.ok == 0);
        }
    }
}

template <int KIND, int PP, bool AP>
static void emu_res_phase(EmuPersist& E, int phase, long long it, long long nsteps) {
    bb_handle* h = E.h;
    const RunArgs& A = E.A;
    BRSt<PP>* st = (BRSt<PP>*)E.st.data();
    auto cxof = [&](int b) { return BBCtx{h->nthr, b, E.lds.data() + (size_t)b * (h->lds_doubles_p + 64), nullptr}; };
    const BRLay& Y = h->Yh;
    // `it` counts the exchanges of this run: step (h->step + it / NS), sample it % NS (the MS instances' arithmetic; NS == 1 otherwise)
    const int NS = A.S < 1 ? 1 : A.S;
    const bool MSrun = NS > 1 || A.elbo_every > 0;
    const unsigned long long step = (unsigned long long)(h->step + it / NS);
    const int smp = (int)(it % NS);
    const bool last = smp == NS - 1;
    const unsigned long long xc = MSrun ? step * (unsigned long long)NS + (unsigned long long)smp : step;
    const int buf = (int)(xc & 1);
    const bool want_el = MSrun && A.elbo_every > 0 && step % (unsigned long long)A.elbo_every == 0;
    const int ring = A.elbo_every > 0 ? (int)((step / (unsigned long long)A.elbo_every) % BB_ELBO_RING) : 0;
    const bool xg = h->p2p_on;
    const BBSlot wslot = bb_slot_of(A, step);
    for (int b = 0; b < (phase == 2 ? bbp_groups(A) : h->res_nblk); ++b) {
        BBCtx cx = cxof(b);
        BRSt<PP>* sb = st + (size_t)b * h->nthr;
        if (phase == 0) {
            br_prologue<KIND, PP, AP>(cx, h->M, h->S, A, Y, h->res_NB, sb);
            br_draw_ahead<KIND, PP, AP>(cx, A, Y, sb, (unsigned long long)h->step);
            if (A.pf == 2) br_prefetch_slot<PP>(cx, h->M, h->S, A, Y, sb, bb_slot_of(A, (unsigned long long)h->step).slot);
        } else if (phase == 1) {
            if (MSrun) br_sample<KIND, PP, true>(cx, h->M, h->S, A, Y, sb, buf, wslot.slot, last, want_el);
            else br_sample<KIND, PP, false>(cx, h->M, h->S, A, Y, sb, buf, wslot.slot);
            const unsigned epoch = A.xepoch0 + (unsigned)(xc + 1);
            if (!BR_TG) { if (MSrun) br_moments<KIND, PP, false, true>(cx, h->M, h->S, Y, sb, buf, epoch, want_el); else br_moments<KIND, PP, false, false>(cx, h->M, h->S, Y, sb, buf, epoch); }
            else { if (MSrun) br_moments<KIND, PP, true, true>(cx, h->M, h->S, Y, sb, buf, epoch, want_el); else br_moments<KIND, PP, true, false>(cx, h->M, h->S, Y, sb, buf, epoch); }
            br_xchg_publish<KIND, PP, AP>(cx, h->M, h->S, A, Y, sb, xc, wslot.slot, last ? step + 1 : step, last ? 0u : (unsigned)(smp + 1), last);
        } else if (phase == 2) {
            if (xg) br_xchg_lead<true>(cx, h->M, h->S, A, Y, xc, &E.ok);
            else br_xchg_lead<false>(cx, h->M, h->S, A, Y, xc, &E.ok);
        } else if (phase == 3) {
            if (xg && MSrun) br_xchg_consume<KIND, PP, true, true>(cx, h->M, h->S, A, Y, sb, xc, &E.ok, want_el, ring, smp, last ? wslot.slot : -1);
            else if (xg) br_xchg_consume<KIND, PP, true, false>(cx, h->M, h->S, A, Y, sb, xc, &E.ok, false, 0, 0, wslot.slot);
            else if (MSrun) br_xchg_consume<KIND, PP, false, true>(cx, h->M, h->S, A, Y, sb, xc, &E.ok, want_el, ring, smp, last ? wslot.slot : -1);
            else br_xchg_consume<KIND, PP, false, false>(cx, h->M, h->S, A, Y, sb, xc, &E.ok, false, 0, 0, wslot.slot);
            // (the compile-time-T forms of the G pass where the product has them, so that the emulation covers that code too)
            if (MSrun) br_update<KIND, PP, 0, AP, true>(cx, h->M, h->S, A, Y, sb, wslot, buf, h->res_NB, smp, NS);
            else if (!AP && uniform_T(h->M) == 8) br_update<KIND, PP, 8, false>(cx, h->M, h->S, A, Y, sb, wslot, buf, h->res_NB);
            else if (!AP && uniform_T(h->M) == 6) br_update<KIND, PP, 6, false>(cx, h->M, h->S, A, Y, sb, wslot, buf, h->res_NB);
            else br_update<KIND, PP, 0, AP>(cx, h->M, h->S, A, Y, sb, wslot, buf, h->res_NB);
        } else {
            br_epilogue<KIND, PP, AP>(cx, h->S, sb, (unsigned long long)(h->step + nsteps), E.ok == 0);
        }
    }
}

template <int KIND, int TT, bool MS>
static void emu_stream_phase(EmuPersist& E, int phase, long long it, long long nsteps) {
    bb_handle* h = E.h;
    const RunArgs& A = E.A;
    BSG* gs = (BSG*)E.st.data();
    auto cxof = [&](int b) { return BBCtx{h->nthr, b, E.lds.data() + (size_t)b * (h->lds_doubles_p + 64), nullptr}; };
    const BRLay& Y = h->Yh;
    // `it` counts the exchanges of this run: step (h->step + it / NS), sample it % NS (as emu_res_phase)
    const int NS = MS ? (A.S < 1 ? 1 : A.S) : 1;
    const unsigned long long step = (unsigned long long)(h->step + it / NS);
    const int smp = (int)(it % NS);
    const unsigned long long xc = step * (unsigned long long)NS + (unsigned long long)smp;
    auto rec = [&](unsigned long long st_) { return MS && A.elbo_every > 0 && st_ % (unsigned long long)A.elbo_every == 0; };
    const bool want_el = rec(step);
    const int ring = A.elbo_every > 0 ? (int)((step / (unsigned long long)A.elbo_every) % BB_ELBO_RING) : 0;
    const BSMs ms = MS ? bs_ms_of(A, step, smp, NS, want_el, rec(step + 1)) : bs_ms_plain((unsigned)step);
    BRSt<1>* nost = nullptr;
    for (int b = 0; b < (phase == 2 ? bbp_groups(A) : h->res_nblk); ++b) {
        BBCtx cx = cxof(b);
        BSG* gb = gs + (size_t)b * h->nthr;
        int* bad_any = (int*)(cx.lds + Y.L.misc) + 3;
        if (phase == 0) {
            br_tile_setup<KIND>(cx, h->M, h->S, A, Y, h->res_NB, KIND <= 2 ? h->nthr / 16 : h->nthr / 64);
            *bad_any = 0;
            // (the launch's first sample; later ones: inside the G passes)
            bs_sample0<KIND, TT, MS>(cx, h->M, h->S, A, Y, h->res_NB, h->res_P, (unsigned)h->step, gb, (int)(((unsigned long long)h->step * (unsigned long long)NS) & 1ull), rec((unsigned long long)h->step));
        } else if (phase == 1) {
            bs_moments<KIND, TT, MS>(cx, h->M, h->S, A, Y, h->res_NB, h->res_P, (unsigned)step, gb, ms.buf, want_el);
            br_row_publish<1, true, MS>(cx, h->M, h->S, Y, nost, A.xepoch0 + (unsigned)(xc + 1), want_el);
        } else if (phase == 2) {
            br_xchg_lead<false>(cx, h->M, h->S, A, Y, xc, &E.ok);
        } else if (phase == 3) {
            br_xchg_consume<KIND, 1, false, MS>(cx, h->M, h->S, A, Y, nost, xc, &E.ok, want_el, ring, smp);
            bs_update_l<KIND, TT, MS>(cx, h->M, h->S, A, Y, h->res_NB, h->res_P, (unsigned)step, bb_slot_of(A, step), bad_any, gb, ms);
            bs_update_u<KIND, TT, MS>(cx, h->M, h->S, A, Y, h->res_NB, h->res_P, (unsigned)step, bb_slot_of(A, step), bad_any, gb, ms);
        } else {
            if (*bad_any) h->S.hstatus[1] = 1u;
            if (b == 0) { h->S.ctr[0] = (unsigned long long)(h->step + nsteps); h->S.ctr[1] = h->S.ctr[0]; }
        }
    }
}

static void emu_persist_dispatch(EmuPersist& E, int phase, long long it, long long nsteps) {
    if (E.h->res_P && E.h->res_stream) {
        const int T = uniform_T(E.h->M);
        auto byT = [&](auto kindc) {
            constexpr int KIND = decltype(kindc)::value;
            const bool ms = E.h->o.samples_per_step != 1 || E.h->o.elbo_every != 0;
            if (T == 8) ms ? emu_stream_phase<KIND, 8, true>(E, phase, it, nsteps) : emu_stream_phase<KIND, 8, false>(E, phase, it, nsteps);
            else if (T == 6) ms ? emu_stream_phase<KIND, 6, true>(E, phase, it, nsteps) : emu_stream_phase<KIND, 6, false>(E, phase, it, nsteps);
            else ms ? emu_stream_phase<KIND, 4, true>(E, phase, it, nsteps) : emu_stream_phase<KIND, 4, false>(E, phase, it, nsteps);
        };
        switch (E.h->M.kind) {
        case 0: byT(std::integral_constant<int, 0>{}); break;
        case 1: byT(std::integral_constant<int, 1>{}); break;
        case 2: byT(std::integral_constant<int, 2>{}); break;
        case 3: byT(std::integral_constant<int, 3>{}); break;
        default: byT(std::integral_constant<int, 4>{});
        }
        return;
    }
    if (E.h->res_P) {
        const bool ap = br_any_parity(E.h->M);
        auto byP = [&](auto kindc) {
            constexpr int KIND = decltype(kindc)::value;
            switch (E.h->res_P) {
            case 1: ap ? emu_res_phase<KIND, 1, true>(E, phase, it, nsteps) : emu_res_phase<KIND, 1, false>(E, phase, it, nsteps); break;
            case 2: ap ? emu_res_phase<KIND, 2, true>(E, phase, it, nsteps) : emu_res_phase<KIND, 2, false>(E, phase, it, nsteps); break;
            case 3: ap ? emu_res_phase<KIND, 3, true>(E, phase, it, nsteps) : emu_res_phase<KIND, 3, false>(E, phase, it, nsteps); break;
            default: ap ? emu_res_phase<KIND, 4, true>(E, phase, it, nsteps) : emu_res_phase<KIND, 4, false>(E, phase, it, nsteps);
            }
        };
        switch (E.h->M.kind) {
        case 0: byP(std::integral_constant<int, 0>{}); break;
        case 1: byP(std::integral_constant<int, 1>{}); break;
        case 2: byP(std::integral_constant<int, 2>{}); break;
        case 3: byP(std::integral_constant<int, 3>{}); break;
        default: byP(std::integral_constant<int, 4>{});
        }
        return;
    }
    auto byP = [&](auto kindc) {
        constexpr int KIND = decltype(kindc)::value;
        switch (E.h->persist_P) {
        case 1: emu_persist_phase<KIND, 1>(E, phase, it, nsteps); break;
        case 2: emu_persist_phase<KIND, 2>(E, phase, it, nsteps); break;
        case 3: emu_persist_phase<KIND, 3>(E, phase, it, nsteps); break;
        default: emu_persist_phase<KIND, 4>(E, phase, it, nsteps);
        }
    };
    switch (E.h->M.kind) {
    case 0: byP(std::integral_constant<int, 0>{}); break;
    case 1: byP(std::integral_constant<int, 1>{}); break;
    case 3: byP(std::integral_constant<int, 3>{}); break;
    default: byP(std::integral_constant<int, 4>{});
    }
}

static size_t emu_rst_bytes(int P) {
    return P == 1 ? sizeof(BRSt<1>) : (P == 2 ? sizeof(BRSt<2>) : (P == 3 ? sizeof(BRSt<3>) : sizeof(BRSt<4>)));
}
static size_t emu_pst_bytes(int P) {
    return P == 1 ? sizeof(BBPst<1>) : (P == 2 ? sizeof(BBPst<2>) : (P == 3 ? sizeof(BBPst<3>) : sizeof(BBPst<4>)));
}

static int emu_run_group(bb_handle** hs, int n, long long nsteps) {
    std::vector<EmuPersist> es((size_t)n);
    for (int i = 0; i < n; ++i) {
        bb_handle* h = hs[i];
        es[i].h = h;
        es[i].A = make_args(h, h->step, 0, h->res_P ? h->o.samples_per_step : 1, true, false);
        if (h->res_P) { es[i].A.nblk = h->res_nblk; es[i].A.nbl = h->res_NBL; es[i].A.ng = h->res_ng; es[i].A.pf = h->res_pf; }
        es[i].lds.assign((size_t)std::max(h->nblk, h->res_nblk) * (h->lds_doubles_p + 64), 0.0);
        es[i].st.assign((size_t)std::max(h->nblk, h->res_nblk) * h->nthr * (h->res_stream ? sizeof(BSG) : (h->res_P ? emu_rst_bytes(h->res_P) : emu_pst_bytes(h->persist_P))), 0);
        emu_persist_dispatch(es[i], 0, 0, nsteps);
    }
    const long long NS = hs[0]->res_P ? std::max(hs[0]->o.samples_per_step, 1) : 1;
    for (long long it = 0; it < nsteps * NS; ++it)
        for (int phase = 1; phase <= 3; ++phase)
            for (int i = 0; i < n; ++i) emu_persist_dispatch(es[i], phase, it, nsteps);
    int rc = 0;
    for (int i = 0; i < n; ++i) {
        emu_persist_dispatch(es[i], 4, 0, nsteps);
        if (!es[i].ok) rc = bb_fail(BB_ERR_DEVICE, "emulated exchange found a missing row");
        hs[i]->step += nsteps;
    }
    return rc;
}

// test hook of the emulation build: the ranks of a sharded resident run, stepped in lock step in one process
extern "C" int bb_emu_run_group(bb_handle** hs, int32_t n, int64_t nsteps) {
    if (!hs || n < 1 || nsteps < 0) return bb_fail(BB_ERR_INVALID, "bad argument");
    for (int i = 0; i < n; ++i)
        if (!hs[i] || hs[i]->persist_P == 0) return bb_fail(BB_ERR_INVALID, "handle %d has no resident launch", i);
    for (int i = 0; i < n; ++i) hs[i]->req_steps += nsteps * std::max(hs[i]->o.samples_per_step, 1);
    int rc = emu_run_group(hs, n, nsteps);
    if (!rc) rc = theta_sync_local(hs, n);          // (genotype model: theta_g back from its owner, as bb_run does through RCCL)
    return rc;
}
#endif

static bool res_ms(const bb_handle* h) { return h->res_P > 0 && (h->o.samples_per_step != 1 || h->o.elbo_every != 0); }
static int launch_persistent(bb_handle* h, long long nsteps) {
    RunArgs A = make_args(h, h->step, 0, h->res_P ? h->o.samples_per_step : 1, true, false);
    int rc = 0;
    h->req_steps += nsteps * (h->res_P ? std::max(h->o.samples_per_step, 1) : 1);      // (exchanges asked: the rows' epochs count them)
#ifdef BB_EMU
    (void)A;
    if (h->p2p_on) return bb_fail(BB_ERR_UNSUPPORTED, "emulation: step the ranks of a sharded resident run with bb_emu_run_group");
    rc = emu_run_group(&h, 1, nsteps);
#else
    // No per-launch memsets: ready words carry base + step + 1 and only grow; the timeout word is sticky (a launch that finds it
    // set leaves at once, so a queue of launches behind a timed-out one neither runs nor skips steps); every launch takes its
    // first step from the device counter.
    bb_persist_kernel k = h->res_P ? nullptr : persist_kernel(h->M.kind, h->persist_P, h->nthr, h->p2p_on);
    bb_res_kernel kr = (h->res_P && !h->res_stream) ? res_kernel(h->M.kind, h->res_P, h->nthr, h->p2p_on, uniform_T(h->M), br_any_parity(h->M), res_ms(h)) : nullptr;
    if (h->res_P) { A.nblk = h->res_nblk; A.nbl = h->res_NBL; A.ng = h->res_ng; A.pf = h->res_pf; }
    if (h->p2p_first && nsteps > 0) { A.spin_limit = 1u << 25; h->p2p_first = false; }   // launch skew between the ranks' processes
    do {                                                  // (nsteps == 0: one launch that only loads and stores the state)
        const int n = (int)std::min<long long>(nsteps, 4096);
        if (++h->launch_seq == 0u) h->launch_seq = 1u;
        A.launch_tag = h->launch_seq;
        if (h->res_stream) hipLaunchKernelGGL(stream_kernel(h->M.kind, h->nthr, uniform_T(h->M), nullptr, res_ms(h)), dim3(h->res_nblk), dim3(h->nthr), h->lds_doubles_p * 8, h->stream,
                                              (const DevModel*)h->dM, (const DevState*)h->dS, (const BRLay*)h->dY, A, h->res_NB, n, h->res_P);
        else if (kr) hipLaunchKernelGGL(kr, dim3(h->res_nblk), dim3(h->nthr), h->lds_doubles_p * 8, h->stream, (const DevModel*)h->dM, (const DevState*)h->dS, (const BRLay*)h->dY, A, h->res_NB, n);
        else hipLaunchKernelGGL(k, dim3(h->nblk), dim3(h->nthr), h->lds_doubles_p * 8, h->stream, (const DevModel*)h->dM, (const DevState*)h->dS, (const BBLds*)h->dL, A, h->NB, n);
        rc = launch_check();
        h->step += n;
        nsteps -= n;
        A.spin_limit = 1u << 23;
    } while (nsteps > 0 && !rc);
#endif
    return rc;
}

// after the stream has drained: the launches' status words (host-mapped, no copy)
static int check_persistent(bb_handle* h) {
    if (!h->hstatus) return 0;
    volatile unsigned* st = h->hstatus;
    if (st[0] != 0) {
        st[0] = 0;
        int rc = 0;
#ifndef BB_EMU
        unsigned long long c[2];
        if (!d2h(c, h->S.ctr, sizeof c, h->stream)) h->step = (long long)c[0];
        rc = dzero(h->S.gbar, 32 * 10 * 4, h->stream);         // acknowledge: later launches may run again
        if (!rc) rc = dsync(h->stream);
#endif
        (void)rc;
        return bb_fail(BB_ERR_DEVICE, "an exchange of the resident launch timed out (not all %d workgroups resident -- device shared or masked? -- or a peer "
                                      "rank stalled); %lld steps completed; set launch_mode = 1 or re-initialise", h->nblk, h->step);
    }
    if (st[1] != 0) {
        st[1] = 0;
        return bb_fail(BB_ERR_NONFINITE, "the variational parameters went non-finite (NaN / Inf) during the run; %lld steps done", h->step);
    }
    return 0;
}

static int create_inner(const bb_model_desc* md, const bb_advi_opts* opts, bb_handle** out) {
    if (!md || !opts || !out) return bb_fail(BB_ERR_INVALID, "null argument");
    *out = nullptr;
    if (md->kind < 0 || md->kind > 4) return bb_fail(BB_ERR_INVALID, "unknown model kind %d", md->kind);
    if (md->n_rep < 1 || md->n_rep > BB_MAX_REP) return bb_fail(BB_ERR_INVALID, "n_rep must be in 1..%d", BB_MAX_REP);
    if (md->kind != BB_MODEL_REPLICATE && md->kind != BB_MODEL_MULTIENV_REPLICATE && md->n_rep != 1)
        return bb_fail(BB_ERR_INVALID, "only the replicate models take n_rep > 1");
    if (md->n_neutral < 1 || md->n_bc < 1) return bb_fail(BB_ERR_INVALID, "need at least one neutral and one mutant barcode");
    if (!md->n_time || !md->counts || !md->totals) return bb_fail(BB_ERR_INVALID, "n_time/counts/totals missing");
    if (opts->samples_per_step < 1) return bb_fail(BB_ERR_INVALID, "samples_per_step must be >= 1");
    if (opts->optimizer != BB_OPT_TRUNCATED_ADAGRAD && opts->optimizer != BB_OPT_DECAYED_ADAGRAD)
        return bb_fail(BB_ERR_INVALID, "unknown optimizer %d", opts->optimizer);
    if (opts->optimizer == BB_OPT_TRUNCATED_ADAGRAD && opts->window < 1) return bb_fail(BB_ERR_INVALID, "window must be >= 1");
    if (opts->world_size < 1 || opts->rank < 0 || opts->rank >= opts->world_size)
        return bb_fail(BB_ERR_INVALID, "bad rank/world_size %d/%d", opts->rank, opts->world_size);
    if (opts->n_devices > 1) return group_create(md, opts, out);

    bb_handle* h = new bb_handle();
    h->o = *opts;
    { const char* fr = getenv("BB_FORCE_ALLREDUCE"); h->force_reduce = fr && atoi(fr) > 0; }
    if (h->o.resum_every < 0) h->o.resum_every = 0;      // 0 = the default schedule (bb_slot_of)
    DevModel& M = h->M;
    M.kind = md->kind;
    M.R = md->n_rep;
    M.E = (md->kind == BB_MODEL_MULTIENV || md->kind == BB_MODEL_MULTIENV_REPLICATE) ? md->n_env : 1;
    M.G = md->kind == BB_MODEL_GENOTYPE ? md->n_geno : 0;
    M.nn = md->n_neutral;
    M.nb = md->n_bc;
    M.B = M.nn + M.nb;
    M.quirk = (md->kind == BB_MODEL_REPLICATE && (md->flags & BB_FLAG_RAGGED_METHOD)) ? 1 : 0;
    int rc = 0;
#define BB_TRY(x)                  \
    do {                           \
        rc = (x);                  \
        if (rc) { bb_destroy(h); return rc; } \
    } while (0)

#ifndef BB_EMU
    {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || opts->device < 0 || opts->device >= ndev) { delete h; return bb_fail(BB_ERR_DEVICE, "device %d: no such HIP device (%d visible)", opts->device, ndev); }
    }
    BB_ENTER(h);
    {
        hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete h; return bb_fail(BB_ERR_DEVICE, "hipStreamCreate: %s", hipGetErrorString(e)); }
        (void)hipEventCreate(&h->ev0);
        (void)hipEventCreate(&h->ev1);
    }
#endif

    // ---- shapes -----------------------------------------------------------------------------
    long long n_l = 0, cnt = 0;
    M.Ttot = 0; M.nt1 = 0; M.K = 0;
    for (int r = 0; r < M.R; ++r) {
        const int T = md->n_time[r];
        if (T < 2 || T > 255) { bb_destroy(h); return bb_fail(BB_ERR_INVALID, "n_time[%d] = %d outside 2..255", r, T); }
        M.T[r] = T;
        M.Tmagic[r] = (unsigned)(0x100000000ull / (unsigned)T) + 1u;
        M.Tmagic1[r] = T > 2 ? (unsigned)(0x100000000ull / (unsigned)(T - 1)) + 1u : 0u;   // T - 1 == 1: no division
        M.off_t[r] = M.nt1;
        M.cnt_off[r] = cnt;
        M.kq[r] = M.K;
        M.kqa[r] = M.K + T + 5 * (T - 1);
        M.tcum[r] = M.Ttot;
        M.Ttot += T;
        M.nt1 += T - 1;
        M.K += 6 * T - 5 + (M.quirk ? 2 * (T - 1) * (T - 1) : 0);
        n_l += (long long)T * M.B;
        cnt += (long long)T * M.B;
    }
    M.K += 2;
    const bool has_env = md->kind == BB_MODEL_MULTIENV || md->kind == BB_MODEL_MULTIENV_REPLICATE;
    if (has_env) {
        if (md->n_env < 1 || !md->env_idx) { bb_destroy(h); return bb_fail(BB_ERR_INVALID, "multienv models need n_env >= 1 and env_idx"); }
        for (int t = 0; t < M.Ttot; ++t)
            if (md->env_idx[t] < 0 || md->env_idx[t] >= md->n_env) { bb_destroy(h); return bb_fail(BB_ERR_INVALID, "env_idx[%d] out of range", t); }
    }
    if (md->kind == BB_MODEL_GENOTYPE) {
        if (md->n_geno < 1 || !md->geno_idx) { bb_destroy(h); return bb_fail(BB_ERR_INVALID, "genotype model needs n_geno >= 1 and geno_idx"); }
        for (long long m = 0; m < M.nb; ++m)
            if (md->geno_idx[m] < 0 || md->geno_idx[m] >= md->n_geno) { bb_destroy(h); return bb_fail(BB_ERR_INVALID, "geno_idx[%lld] out of range", m); }
    }

    // ---- flat layout, source order (model_*.jl `~` statements) -----------------
    long long off = 0;
    for (int k = 0; k < BK_COUNT; ++k) M.blk_lo[k] = M.blk_hi[k] = 0;
    add_block(h, "s_pop", BK_SPOP, M.nt1, &off);
    add_block(h, "logsigma_pop", BK_LSPOP, M.nt1, &off);
    if (M.kind == BB_MODEL_FITNESS || M.kind == BB_MODEL_MULTIENV) {
        add_block(h, "s_bc", BK_S, M.nb * M.E, &off);
        add_block(h, "logsigma_bc", BK_LS, M.nb * M.E, &off);
    } else {
        const long long E_ = M.kind == BB_MODEL_MULTIENV_REPLICATE ? M.E : 1;
        if (g_loglambda_first) add_block(h, "loglambda", BK_L, n_l, &off);          // (internal order only: bb_create)
        add_block(h, "theta", BK_S, M.kind == BB_MODEL_GENOTYPE ? M.G : M.nb * E_, &off);
        add_block(h, "theta_tilde", BK_TT, M.nb * M.R * E_, &off);
        add_block(h, "logtau", BK_LT, M.nb * M.R * E_, &off);
        add_block(h, "logsigma_bc", BK_LS, M.nb * M.R * E_, &off);
    }
    if (!(g_loglambda_first && M.kind >= BB_MODEL_GENOTYPE)) add_block(h, "loglambda", BK_L, n_l, &off);
    M.D = off;
    M.Dp = (off + 7) & ~7ll;
    for (int r = 0, o = 0; r < M.R; ++r) { M.off_l[r] = M.blk_lo[BK_L] + (long long)o * M.B; o += M.T[r]; }

    // ---- counts: validate totals == row sums (Multinomial support, Distributions.jl), to uint32
    std::vector<unsigned> c32((size_t)cnt);
    double sum_lgamma = 0.0;
    {
        long long co = 0, to = 0;
        for (int r = 0; r < M.R; ++r) {
            const int T = M.T[r];
            for (int t = 0; t < T; ++t) {
                long long s = 0;
                for (long long b = 0; b < M.B; ++b) {
                    const int64_t v = md->counts[co + b * T + t];
                    if (v < 0 || v > 0xFFFFFFFFll) { bb_destroy(h); return bb_fail(BB_ERR_INVALID, "count out of range at rep %d t %d barcode %lld", r, t, b); }
                    c32[(size_t)(co + b * T + t)] = (unsigned)v;
                    s += v;
                    sum_lgamma += lgamma((double)v + 1.0);
                }
                if (s != md->totals[to + t]) {
                    bb_destroy(h);
                    return bb_fail(BB_ERR_INVALID, "totals[rep %d, t %d] = %lld but the counts sum to %lld (the reference's Multinomial term is -Inf there)",
                                   r, t, (long long)md->totals[to + t], s);
                }
            }
            co += (long long)T * M.B;
            to += T;
        }
    }
    unsigned* dcounts = nullptr;
    BB_TRY(dalloc(h, &dcounts, (size_t)cnt));
    BB_TRY(h2d(dcounts, c32.data(), (size_t)cnt * 4, h->stream));
    M.counts = dcounts;

    if (has_env) {
        int* d = nullptr;
        BB_TRY(dalloc(h, &d, (size_t)M.Ttot));
        BB_TRY(h2d(d, md->env_idx, (size_t)M.Ttot * 4, h->stream));
        M.env_idx = d;
    }
    if (md->kind == BB_MODEL_GENOTYPE) {
        int *d = nullptr, *dp = nullptr, *dm = nullptr;
        BB_TRY(dalloc(h, &d, (size_t)M.nb));
        BB_TRY(h2d(d, md->geno_idx, (size_t)M.nb * 4, h->stream));
        M.geno_idx = d;
        std::vector<int> ptr((size_t)M.G + 1, 0), mem((size_t)M.nb);
        for (long long m = 0; m < M.nb; ++m) ptr[(size_t)md->geno_idx[m] + 1]++;
        for (int g = 0; g < M.G; ++g) ptr[(size_t)g + 1] += ptr[(size_t)g];
        std::vector<int> fill(ptr.begin(), ptr.end() - 1);
        for (long long m = 0; m < M.nb; ++m) mem[(size_t)fill[(size_t)md->geno_idx[m]]++] = (int)m;
        BB_TRY(dalloc(h, &dp, (size_t)M.G + 1));
        BB_TRY(dalloc(h, &dm, (size_t)M.nb));
        BB_TRY(h2d(dp, ptr.data(), ((size_t)M.G + 1) * 4, h->stream));
        BB_TRY(h2d(dm, mem.data(), (size_t)M.nb * 4, h->stream));
        M.geno_ptr = dp;
        M.geno_mem = dm;
        M.geno_sorted = 1;
        for (long long m = 1; m < M.nb; ++m) if (md->geno_idx[m] < md->geno_idx[m - 1]) { M.geno_sorted = 0; break; }
        h->geno_ptr_h = ptr;
    }

    // ---- priors (defaults: model_fitness_normal.jl:125-129, ..._genotypes.jl:162) -------------
    double sum_log_std = 0.0;
    BB_TRY(upload_prior(h, BK_SPOP, &md->s_pop_prior, 0.0, 2.0, "s_pop_prior", false, &sum_log_std));
    BB_TRY(upload_prior(h, BK_LSPOP, &md->logsigma_pop_prior, 0.0, 1.0, "logsigma_pop_prior", false, &sum_log_std));
    BB_TRY(upload_prior(h, BK_S, &md->s_bc_prior, 0.0, 2.0, "s_bc_prior", false, &sum_log_std));
    BB_TRY(upload_prior(h, BK_LS, &md->logsigma_bc_prior, 0.0, 1.0, "logsigma_bc_prior", false, &sum_log_std));
    BB_TRY(upload_prior(h, BK_L, &md->loglambda_prior, 3.0, 3.0, "loglambda_prior", false, &sum_log_std));
    if (M.kind >= BB_MODEL_GENOTYPE) {
        BB_TRY(upload_prior(h, BK_TT, nullptr, 0.0, 1.0, "theta_tilde", true, &sum_log_std));
        BB_TRY(upload_prior(h, BK_LT, &md->logtau_prior, -2.0, 1.0, "logtau_prior", true, &sum_log_std));
    }
    // constant part of the ELBO: prior normalisers, likelihood normalisers, lgamma terms, entropy constant
    {
        double nlik = 0.0;
        for (int r = 0; r < M.R; ++r) nlik += (double)(M.T[r] - 1) * (double)M.B;
        h->elbo_const = -sum_log_std - 0.5 * BB_LOG2PI * (double)M.D - sum_lgamma - 0.5 * BB_LOG2PI * nlik +
                        0.5 * (double)M.D * (1.0 + BB_LOG2PI);
    }

    // ---- shard + launch geometry -------------------------------------------------------------
    h->b_lo = M.B * opts->rank / opts->world_size;
    h->b_hi = M.B * (opts->rank + 1) / opts->world_size;
    h->g_lo = 0;
    h->g_hi = M.G;
    if (M.kind == BB_MODEL_GENOTYPE && M.geno_sorted && opts->world_size > 1) {
        // Genotypes in consecutive runs: cut the shards at genotype boundaries, so that every rank holds ALL mutants of the
        // genotypes it owns -- d/dtheta_g is then a rank-local sum.  The cut moves back to the first mutant
        // of the genotype it fell into; genotypes without mutants go with the one before.
        auto snap = [&](long long b, int* g) {
            if (b <= 0) { *g = 0; return (long long)0; }
            if (b >= M.B) { *g = M.G; return M.B; }
            if (b <= M.nn) { *g = 0; return b; }
            const int gg = md->geno_idx[b - M.nn];
            *g = gg;
            return M.nn + (long long)h->geno_ptr_h[(size_t)gg];
        };
        h->b_lo = snap(h->b_lo, &h->g_lo);
        h->b_hi = snap(h->b_hi, &h->g_hi);
        if (h->b_lo <= M.nn) h->g_lo = 0;          // (genotype ranges tile [0, G): whoever owns the first mutant also owns the empty ones before it)
    }
    {
        // One workgroup per CU (XCD-agnostic: every tile is independent), sized so that the whole
        // shard is resident at once: NB = ceil(barcodes / CUs) barcodes per tile, up to 1024 threads
        // (16 waves per CU) working a tile's ~NB*(T+2) latents.  BB_TUNE_* env vars override for experiments.
        int maxT = 0;
        for (int r = 0; r < M.R; ++r) maxT = std::max(maxT, M.T[r]);
        int cus = 256;
#ifndef BB_EMU
        { hipDeviceProp_t pr; if (hipGetDeviceProperties(&pr, opts->device) == hipSuccess && pr.multiProcessorCount > 0) cus = pr.multiProcessorCount; }
#endif
        h->cus = cus;
        const char* ev;
        int bpc = (ev = getenv("BB_TUNE_BLOCKS_PER_CU")) ? atoi(ev) : 1;
        if (bpc < 1) bpc = 1;
        const long long nbar = std::max<long long>(h->b_hi - h->b_lo, 1);
        long long target = (long long)cus * bpc;
        int NB = (int)std::max<long long>((nbar + target - 1) / target, 32);
        if ((ev = getenv("BB_TUNE_NB")) && atoi(ev) > 0) NB = atoi(ev);
        const size_t lds_cap = (size_t)160 * 1024 / (size_t)bpc;
        int nthr = 0;
        for (;;) {
            // one pair of latents per thread is the sweet spot; counted with the segments' rounding (tile_pairs_bound), the
            // number the resident launch sizes its per-thread state by (a tile of 257 pairs on 256 threads would need two)
            const long long pairs = tile_pairs_bound(M, NB);
            // > 1 pair per thread: 512 threads (256-VGPR budget, up to 4 pairs) beat 1024 threads with spills (C3: 28.8k vs 18.8k steps/s)
            // (768 threads x 2 pairs was tried for C3: 138 spills at 168 VGPRs, 23.0k vs 28.8k steps/s for 512 x 3)
            nthr = pairs > 2048 ? 1024 : (pairs > 1024 ? 512 : (pairs > 512 ? 1024 : (pairs > 256 ? 512 : 256)));
            if ((ev = getenv("BB_TUNE_NTHR")) && atoi(ev) >= 64) nthr = atoi(ev) / 64 * 64;
            while (nthr < maxT) nthr <<= 1;
            const size_t need = (size_t)bb_lds_layout(M.R, M.E, M.kind, M.Ttot, M.nt1, M.K, NB, nthr).total * 8;
            if ((need <= lds_cap && (long long)NB * maxT < 65536) || NB <= 8) break;
            NB = (NB + 1) / 2;
        }
        const size_t need = (size_t)bb_lds_layout(M.R, M.E, M.kind, M.Ttot, M.nt1, M.K, NB, nthr).total * 8;
        if (need > 160 * 1024 || nthr > 1024 || (long long)NB * maxT >= 65536) {
            bb_destroy(h);
            return bb_fail(BB_ERR_UNSUPPORTED, "a tile of %d barcodes needs %zu bytes of LDS / %d threads (n_time or n_rep too large for this build)", NB, need, nthr);
        }
        h->NB = NB;
        h->nthr = nthr;
        h->lds_doubles = need / 8;
        h->lds_doubles_p0 = h->lds_doubles_p = (size_t)bb_lds_layout(M.R, M.E, M.kind, M.Ttot, M.nt1, M.K, NB, nthr, 1).total;
        h->nblk = (int)((nbar + NB - 1) / NB);
        h->ngeno_blk = M.G > 0 ? (int)std::min<long long>(((M.G + 1) / 2 + 255) / 256, 64) : 0;
#ifndef BB_EMU
        if (need > 64 * 1024) {
            hipError_t e1 = hipFuncSetAttribute((const void*)sample_kernel(M.kind), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
            hipError_t e2 = hipFuncSetAttribute((const void*)update_kernel(M.kind), hipFuncAttributeMaxDynamicSharedMemorySize, (int)need);
            if (e1 != hipSuccess || e2 != hipSuccess) { bb_destroy(h); return bb_fail(BB_ERR_DEVICE, "cannot raise dynamic LDS to %zu bytes", need); }
        }
#endif
    }

    // ---- state ---------------------------------------------------------------------------------
    DevState& S = h->S;
    const size_t D = (size_t)M.D;
    BB_TRY(dalloc(h, &S.mu, D + 2));
    BB_TRY(dalloc(h, &S.om, D + 2));
    BB_TRY(dalloc(h, &S.acc_mu, D + 2));
    BB_TRY(dalloc(h, &S.acc_om, D + 2));
    BB_TRY(dalloc(h, &S.accl, 2 * D + 8));
    {
        double* oc = nullptr;
        const double v[8] = {opts->eta, opts->tau, opts->pre, opts->post, 0, 0, 0, 0};
        BB_TRY(dalloc(h, &oc, 8));
        BB_TRY(h2d(oc, v, sizeof v, h->stream));
        S.optc = oc;
    }
    // (zsv, asv, hsv, gacc_*, bak_*: per-sample scratch of the two-kernel step and of bb_elbo_grad -- ensure_scratch, on first use: a
    //  shard that only ever runs the resident launch never pays their 7 x 8 D bytes)
    hist_rows(h);
    if (opts->optimizer == BB_OPT_TRUNCATED_ADAGRAD) BB_TRY(dalloc(h, &S.hist, (size_t)opts->window * 2 * (size_t)M.Dh + 8));   // (+ 8: an edge pair's prefetch reads both halves)
    BB_TRY(dalloc(h, &S.partials, (size_t)M.K * (size_t)h->nblk));
    BB_TRY(dalloc(h, &S.totals, (size_t)M.K));
    BB_TRY(dalloc(h, &S.zg, (size_t)2 * M.nt1));
    BB_TRY(dalloc(h, &S.gbar, (size_t)32 * 10));
#ifdef BB_EMU
    h->hstatus = (unsigned*)calloc(16, sizeof(unsigned));
    S.hstatus = h->hstatus;
#else
    {
        void* hp = nullptr;
        void* dp = nullptr;
        if (hipHostMalloc(&hp, 16 * sizeof(unsigned), hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
            if (hp) (void)hipHostFree(hp);
            bb_destroy(h);
            return bb_fail(BB_ERR_DEVICE, "cannot allocate the host-mapped status words");
        }
        memset(hp, 0, 16 * sizeof(unsigned));
        h->hstatus = (unsigned*)hp;
        S.hstatus = (unsigned*)dp;
    }
#endif
    BB_TRY(dalloc(h, &S.prow, (size_t)(h->nblk + 8) * (M.K + 2 * M.nt1)));      // (+ 8: k_res's own tile map may need a few tiles more)
    BB_TRY(dalloc(h, &S.xrow, (size_t)2 * BB_NG_MAX * (M.K + 2 * M.nt1)));
    BB_TRY(dalloc(h, &S.grow, (size_t)(h->nblk + 8 + 16 * BB_NG_MAX) * bb_row_stride(M.K + 2 * M.nt1)));      // (+ 16 groups x 16: a leader's eight loads in flight run past its last member, bb_gran_poll8)
    BB_TRY(dalloc(h, &S.gxrow, (size_t)2 * BB_NG_MAX * (M.K + 2 * M.nt1)));
    BB_TRY(dalloc(h, &S.rdy, (size_t)32 * (h->nblk + 8 + 2 * BB_NG_MAX)));
    BB_TRY(dalloc(h, &S.xtab, (size_t)BB_NG_MAX));
    BB_TRY(dalloc(h, &S.xsel, (size_t)h->nblk + 8));
    BB_TRY(dalloc(h, &S.ztheta, (size_t)std::max(M.G, 1)));
    BB_TRY(dalloc(h, &S.gsum, (size_t)std::max(M.G, 1)));
    BB_TRY(dalloc(h, &S.ds, (size_t)M.nb));
    BB_TRY(dalloc(h, &S.geno_el, (size_t)std::max(h->ngeno_blk, 1)));
    BB_TRY(dalloc(h, &S.elbo_ring, (size_t)BB_ELBO_RING));
    BB_TRY(dalloc(h, &S.elbo_sample, (size_t)opts->samples_per_step + 64));
    BB_TRY(dalloc(h, &S.ctr, (size_t)2));
    BB_TRY(dalloc(h, &S.stamps, (size_t)(h->nblk + 8) * (32 + 64)));
    S.eps_in = nullptr;

    // algorithmic bytes per step on this shard: theta r+w, optimiser state r+w, counts
    {
        const long long nb_sh = h->b_hi - h->b_lo;
        double frac = (double)nb_sh / (double)M.B;
        const double Dsh = (double)M.D * frac;
        const double cnts = 4.0 * (double)cnt * frac;
        h->bytes_sample = (int64_t)(16.0 * Dsh + cnts);
        const double optb = opts->optimizer == BB_OPT_TRUNCATED_ADAGRAD ? 64.0 : 32.0;
        h->bytes_update = (int64_t)((16.0 + 16.0 + optb) * Dsh + cnts);
    }
    BB_TRY(setup_persistent(h));
    BB_TRY(sync_descriptors(h));
    BB_TRY(bb_init_meanfield(h));
    *out = h;
    return BB_OK;
}


// ---- caller's order <-> the handle's order (genotype model with geno_idx not in runs) ------------------------------------------
static void perm_gather(const bb_handle* h, const double* caller, double* internal) {
    const size_t D = h->cidx.size();
    for (size_t i = 0; i < D; ++i) internal[i] = caller[(size_t)h->cidx[i]];
}
static void perm_scatter(const bb_handle* h, const double* internal, double* caller) {
    const size_t D = h->cidx.size();
    for (size_t i = 0; i < D; ++i) caller[(size_t)h->cidx[i]] = internal[i];
}

// The reference hands barcodes over in order of appearance (utils.data_to_arrays, src/utils.jl:692-731), so a genotype's mutants
// are scattered; the resident launch and genotype-aligned shards need them in consecutive runs (a tile / shard owns whole
// genotypes and their theta).  The library groups them itself -- a stable sort of the mutants by genotype -- works in that order
// and presents the caller's at every entry point that takes or returns a latent vector (bb_get_params / posterior / set_params /
// elbo_grad / logdensity_grad / hier_fitness; bb_get_permutation tells the mapping).  The engine's normal stream is keyed by the
// INTERNAL index (bb_debug_normals likewise).
extern "C" int bb_create(const bb_model_desc* md, const bb_advi_opts* opts, bb_handle** out) {
    if (!md || !opts || !out) return bb_fail(BB_ERR_INVALID, "null argument");
    const bool geno_ok = md->kind == BB_MODEL_GENOTYPE && md->geno_idx && md->n_bc > 1 && md->n_geno >= 1 && md->n_neutral >= 1 && md->n_time &&
                         md->counts && md->n_rep == 1 && md->n_time[0] >= 2 && md->n_time[0] <= 255;
    bool regroup = geno_ok && !getenv("BB_NO_REGROUP");
    if (regroup) {
        bool sorted = true, valid = true;
        for (long long m = 0; m < md->n_bc && valid; ++m) {
            if (md->geno_idx[m] < 0 || md->geno_idx[m] >= md->n_geno) valid = false;
            else if (m > 0 && md->geno_idx[m] < md->geno_idx[m - 1]) sorted = false;
        }
        regroup = valid && !sorted;      // (anything invalid: create_inner says what)
    }
    // Round 4: the genotype model's flat vector s_pop | logsigma_pop | theta (G) | theta_tilde | logtau | logsigma_bc (n_bc each) | loglambda puts
    // loglambda at an ODD index whenever G + n_bc is odd; k_res's pairs (b, 2k), (b, 2k+1) are then not pairs (2q, 2q+1) of the flat index and
    // the any-parity instances ran (two Philox draws in divergent lanes, 8-byte accesses, 40 spilled registers: C5's rank shape 14.95 against
    // 12.8 us).  The library owns an internal order anyway: it lays loglambda out right behind the two global blocks (offset 2 (T - 1): even
    // for even T) and presents the reference's order at every entry point, as for the regrouped mutants.  BB_NO_REORDER=1: as handed over.
    const bool lfirst = geno_ok && !(md->n_time[0] & 1) && ((md->n_geno + md->n_bc) & 1) && !getenv("BB_NO_REORDER");
    if (!regroup && !lfirst) return create_inner(md, opts, out);
    const long long nn = md->n_neutral, nb = md->n_bc, B = nn + nb;
    const int T = md->n_time[0];
    std::vector<int> perm((size_t)nb);
    for (long long m = 0; m < nb; ++m) perm[(size_t)m] = (int)m;
    if (regroup) std::stable_sort(perm.begin(), perm.end(), [&](int a, int b) { return md->geno_idx[a] < md->geno_idx[b]; });
    auto src = [&](long long b) { return b < nn ? b : nn + perm[(size_t)(b - nn)]; };
    bb_model_desc md2 = *md;
    std::vector<int64_t> counts2;
    std::vector<int32_t> geno2;
    std::vector<double> lsm, lss, llm, lls;
    if (regroup) {
        counts2.resize((size_t)B * T);
        for (long long b = 0; b < B; ++b) memcpy(&counts2[(size_t)b * T], md->counts + src(b) * T, (size_t)T * sizeof(int64_t));
        geno2.resize((size_t)nb);
        for (long long m = 0; m < nb; ++m) geno2[(size_t)m] = md->geno_idx[perm[(size_t)m]];
        md2.counts = counts2.data();
        md2.geno_idx = geno2.data();
        // Matrix-form priors of the per-mutant and per-(time, barcode) blocks move with their barcodes
        if (md->logsigma_bc_prior.mean && md->logsigma_bc_prior.std && md->logsigma_bc_prior.n == nb && nb > 1) {
            lsm.resize((size_t)nb); lss.resize((size_t)nb);
            for (long long m = 0; m < nb; ++m) { lsm[(size_t)m] = md->logsigma_bc_prior.mean[perm[(size_t)m]]; lss[(size_t)m] = md->logsigma_bc_prior.std[perm[(size_t)m]]; }
            md2.logsigma_bc_prior.mean = lsm.data(); md2.logsigma_bc_prior.std = lss.data();
        }
        if (md->loglambda_prior.mean && md->loglambda_prior.std && md->loglambda_prior.n == (int64_t)B * T && B * T > 1) {
            llm.resize((size_t)B * T); lls.resize((size_t)B * T);
            for (long long b = 0; b < B; ++b)
                for (int t = 0; t < T; ++t) { llm[(size_t)b * T + t] = md->loglambda_prior.mean[src(b) * T + t]; lls[(size_t)b * T + t] = md->loglambda_prior.std[src(b) * T + t]; }
            md2.loglambda_prior.mean = llm.data(); md2.loglambda_prior.std = lls.data();
        }
    }
    g_loglambda_first = lfirst;
    int rc = create_inner(&md2, opts, out);
    g_loglambda_first = false;
    if (rc) return rc;
    bb_handle* h = *out;
    const DevModel& M = h->M;
    if (regroup) h->perm_m = perm;
    // the caller's layout: the reference's source order (what bb_get_layout reports), and the map internal -> caller
    const int order[BK_COUNT] = {BK_SPOP, BK_LSPOP, BK_S, BK_TT, BK_LT, BK_LS, BK_L};
    long long clo[BK_COUNT] = {0};
    {
        std::vector<bb_block_range> cb;
        long long off = 0;
        for (int k : order) {
            bb_block_range b;
            memset(&b, 0, sizeof b);
            for (const bb_block_range& ib : h->blocks) if (ib.lo == M.blk_lo[k] && ib.hi == M.blk_hi[k] && ib.hi > ib.lo) snprintf(b.name, sizeof b.name, "%s", ib.name);
            clo[k] = off;
            b.lo = off;
            b.hi = off + (M.blk_hi[k] - M.blk_lo[k]);
            off = b.hi;
            cb.push_back(b);
        }
        h->blocks = cb;
    }
    h->cidx.resize((size_t)M.D);
    for (int k : order)
        for (long long j = 0; j < M.blk_hi[k] - M.blk_lo[k]; ++j) h->cidx[(size_t)(M.blk_lo[k] + j)] = clo[k] + j;
    if (regroup) {
        for (int k : {BK_TT, BK_LT, BK_LS})
            for (long long m = 0; m < nb; ++m) h->cidx[(size_t)(M.blk_lo[k] + m)] = clo[k] + perm[(size_t)m];
        for (long long b = nn; b < B; ++b)
            for (int t = 0; t < T; ++t) h->cidx[(size_t)(M.blk_lo[BK_L] + b * T + t)] = clo[BK_L] + src(b) * T + t;
    }
    return BB_OK;
}

// caller indices of the latents THIS handle owns on a sharded run (its barcodes' latents; genotype model: theta of its own genotypes) -- what
// a gather of the ranks' posteriors takes from this rank.  The replicated global blocks are not in the list.  idx: [bb_num_latents(h)].
extern "C" int bb_get_owned(bb_handle* h, int64_t* idx, int64_t* n) {
    if (!h || !idx || !n) return bb_fail(BB_ERR_INVALID, "null argument");
    std::vector<std::pair<long long, long long>> rg;
    if (h->shards.empty()) owned_ranges(h, rg);
    else for (bb_handle* sh : h->shards) owned_ranges(sh, rg);
    int64_t c = 0;
    for (auto& r : rg)
        for (long long i = r.first; i < r.second; ++i) idx[c++] = h->cidx.empty() ? i : h->cidx[(size_t)i];
    *n = c;
    return BB_OK;
}

extern "C" int bb_get_permutation(bb_handle* h, int64_t* caller_index) {
    if (!h || !caller_index) return bb_fail(BB_ERR_INVALID, "null argument");
    for (long long i = 0; i < h->M.D; ++i) caller_index[i] = h->cidx.empty() ? i : h->cidx[(size_t)i];
    return BB_OK;
}

static void p2p_release(bb_handle* h);
static void group_destroy(bb_handle* g);
extern "C" void bb_destroy(bb_handle* h) {
    if (!h) return;
    if (!h->shards.empty()) { group_destroy(h); return; }
    BB_ENTER(h);
#ifndef BB_EMU
    (void)hipStreamSynchronize(h->stream);
    if (h->graph) (void)hipGraphExecDestroy(h->graph);
    if (h->comm && g_rccl.CommDestroy) g_rccl.CommDestroy(h->comm);
    if (h->ev0) (void)hipEventDestroy(h->ev0);
    if (h->ev1) (void)hipEventDestroy(h->ev1);
#endif
    p2p_release(h);
#ifdef BB_EMU
    free(h->hstatus);
#else
    if (h->hstatus) (void)hipHostFree(h->hstatus);
#endif
    for (void* p : h->owned) dfree(p);
    if (h->eps_buf) dfree(h->eps_buf);
    if (h->dbg_buf) dfree(h->dbg_buf);
#ifndef BB_EMU
    if (h->stream) (void)hipStreamDestroy(h->stream);
#endif
    delete h;
}

extern "C" int64_t bb_num_latents(const bb_handle* h) { return h ? h->M.D : 0; }

extern "C" int bb_get_layout(const bb_handle* h, bb_block_range* blocks, int32_t* n) {
    if (!h || !blocks || !n) return bb_fail(BB_ERR_INVALID, "null argument");
    *n = (int32_t)h->blocks.size();
    for (size_t i = 0; i < h->blocks.size(); ++i) blocks[i] = h->blocks[i];
    return BB_OK;
}

// ------------------------------------------------------------------------------------------------
// launches
// ------------------------------------------------------------------------------------------------
static RunArgs make_args(const bb_handle* h, long long step, int sample, int S, bool apply, bool with_elbo) {
    RunArgs A;
    memset(&A, 0, sizeof A);
    A.b_lo = h->b_lo;
    A.b_hi = h->b_hi;
    A.rank = h->o.rank;
    A.world = h->o.world_size;
    A.xepoch0 = h->epoch0;
    A.spin_limit = 1u << 23;                  // ~1 us per poll: seconds, not milliseconds
    { const char* ev = getenv("BB_TUNE_ROW_L2"); A.row_l2 = (ev && atoi(ev) == 0) ? 0 : 1; }
    A.nblk = h->nblk;
    A.nblk_alloc = h->nblk;
    A.ng = 8;
    A.par = (int)(step & 1);
    A.sample = sample;
    A.S = S;
    A.first_sample = sample == 0;
    A.last_sample = sample == S - 1;
    A.apply = apply ? 1 : 0;
    A.with_elbo = with_elbo ? 1 : 0;
    A.count_globals = h->o.rank == 0;
    A.opt = h->o.optimizer;
    A.W = h->o.window;
    A.resum_every = h->o.resum_every;
    A.elbo_every = h->o.elbo_every;
    A.eta = h->o.eta;
    A.tau = h->o.tau;
    A.pre = h->o.pre;
    A.post = h->o.post;
    A.seed = h->o.seed;
    A.elbo_const = h->elbo_const;
    if (h->use_reduce()) { A.red = h->S.totals; A.nred = 1; }
    else { A.red = h->S.partials; A.nred = h->nblk; }
    return A;
}

#ifdef BB_EMU
#define LAUNCH_CHECK() 0
#else
static int launch_check() {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return bb_fail(BB_ERR_DEVICE, "kernel launch: %s", hipGetErrorString(e));
    return 0;
}
#define LAUNCH_CHECK() launch_check()
#endif

// Per-sample scratch of the two-kernel step (z, eps sigmoid, sigmoid / softplus), the S > 1 / gradient-export accumulators and
// bb_elbo_grad's saved parameters: 7 arrays of D doubles, allocated on first use -- every entry point that can reach the
// two-kernel launchers calls this first (never from inside a stream capture); the resident launches use none of them.
static int ensure_scratch(bb_handle* h) {
    if (h->S.zsv) return BB_OK;
    // ONE slab carved into the seven arrays: a failed allocation leaves nothing behind (seven separate ones left the earlier arrays
    // in h->owned with stale pointers in S, and a retry allocated them all again -- ADVICE r03)
    const size_t D = (size_t)h->M.D, n = (D + 2 + 1) & ~(size_t)1;
    double* slab = nullptr;
    int rc = dalloc(h, &slab, 7 * n);
    if (rc) return rc;
    h->S.zsv = slab; h->S.asv = slab + n; h->S.hsv = slab + 2 * n; h->S.gacc_mu = slab + 3 * n; h->S.gacc_om = slab + 4 * n;
    h->bak_mu = slab + 5 * n; h->bak_om = slab + 6 * n;
    return h->dS ? h2d(h->dS, &h->S, sizeof(DevState), h->stream) : BB_OK;      // (kernels read the descriptor through its device copy)
}

static int launch_sample(bb_handle* h, const RunArgs& A) {
    if (!h->S.zsv) return bb_fail(BB_ERR_DEVICE, "internal: the two-kernel step's scratch arrays were not allocated (ensure_scratch)");
#ifdef BB_EMU
    emu_launch(h->nblk, h->nthr, h->lds_doubles, [&](BBCtx& cx) {
        switch (h->M.kind) {
        case 0: bb_block_sample<0>(cx, h->M, h->S, A, h->NB); break;
        case 1: bb_block_sample<1>(cx, h->M, h->S, A, h->NB); break;
        case 2: bb_block_sample<2>(cx, h->M, h->S, A, h->NB); break;
        case 3: bb_block_sample<3>(cx, h->M, h->S, A, h->NB); break;
        default: bb_block_sample<4>(cx, h->M, h->S, A, h->NB);
        }
    });
#else
    hipLaunchKernelGGL(sample_kernel(h->M.kind), dim3(h->nblk), dim3(h->nthr), h->lds_doubles * 8, h->stream, (const DevModel*)h->dM, (const DevState*)h->dS, A, h->NB);
#endif
    return LAUNCH_CHECK();
}
static int launch_update(bb_handle* h, const RunArgs& A) {
    if (!h->S.zsv) return bb_fail(BB_ERR_DEVICE, "internal: the two-kernel step's scratch arrays were not allocated (ensure_scratch)");
#ifdef BB_EMU
    emu_launch(h->nblk, h->nthr, h->lds_doubles, [&](BBCtx& cx) {
        switch (h->M.kind) {
        case 0: bb_block_update<0>(cx, h->M, h->S, A, h->NB); break;
        case 1: bb_block_update<1>(cx, h->M, h->S, A, h->NB); break;
        case 2: bb_block_update<2>(cx, h->M, h->S, A, h->NB); break;
        case 3: bb_block_update<3>(cx, h->M, h->S, A, h->NB); break;
        default: bb_block_update<4>(cx, h->M, h->S, A, h->NB);
        }
    });
#else
    hipLaunchKernelGGL(update_kernel(h->M.kind), dim3(h->nblk), dim3(h->nthr), h->lds_doubles * 8, h->stream, (const DevModel*)h->dM, (const DevState*)h->dS, A, h->NB);
#endif
    return LAUNCH_CHECK();
}
static int launch_reduce(bb_handle* h) {
#ifdef BB_EMU
    emu_launch(1, 256, (size_t)16 * h->M.K, [&](BBCtx& cx) { bb_block_reduce(cx, h->M, h->S, h->nblk, h->ngeno_blk); });
#else
    hipLaunchKernelGGL(k_reduce, dim3(1), dim3(256), (size_t)16 * h->M.K * 8, h->stream, h->M, h->S, h->nblk, h->ngeno_blk);
#endif
    return LAUNCH_CHECK();
}
static int launch_geno(bb_handle* h, const RunArgs& A, int do_update, int do_sample, int upd_par) {
    if (!h->S.zsv) return bb_fail(BB_ERR_DEVICE, "internal: the two-kernel step's scratch arrays were not allocated (ensure_scratch)");
#ifdef BB_EMU
    emu_launch(h->ngeno_blk, 256, 256 + 64, [&](BBCtx& cx) { bb_block_geno(cx, h->M, h->S, A, h->ngeno_blk, do_update, do_sample, upd_par); });
#else
    hipLaunchKernelGGL(k_geno, dim3(h->ngeno_blk), dim3(256), (256 + 64) * 8, h->stream, h->M, h->S, A, do_update, do_sample, upd_par);
#endif
    return LAUNCH_CHECK();
}
static int launch_geno_sum(bb_handle* h) {
    const long long m_lo = std::max(h->b_lo, h->M.nn) - h->M.nn, m_hi = std::max(h->b_hi, h->M.nn) - h->M.nn;
#ifdef BB_EMU
    const int gsb = (int)std::min<long long>((h->M.G + 31) / 32, 1024);      // 32 genotypes per 256-thread block
    emu_launch(gsb, 256, 256, [&](BBCtx& cx) { bb_block_geno_sum(cx, h->M, h->S, gsb, m_lo, m_hi); });
#else
    const int gsb = (int)std::min<long long>((h->M.G + 31) / 32, 1024);
    hipLaunchKernelGGL(k_geno_sum, dim3(gsb), dim3(256), 256 * 8, h->stream, h->M, h->S, m_lo, m_hi);
#endif
    return LAUNCH_CHECK();
}

static int allreduce(bb_handle* h, double* buf, size_t n) {
    if (h->o.world_size == 1 && !h->force_reduce) return 0;
#ifdef BB_EMU
    (void)buf; (void)n;
    return bb_fail(BB_ERR_COMM, "in-library collectives are not available in the emulation build");
#else
    if (!h->comm) return bb_fail(BB_ERR_COMM, "world_size = %d but bb_comm_init was not called (or use bb_step_moments/bb_step_apply)", h->o.world_size);
    int rc = g_rccl.AllReduce(buf, buf, n, /*ncclFloat64*/ 8, /*ncclSum*/ 0, h->comm, h->stream);
    if (rc) return bb_fail(BB_ERR_COMM, "ncclAllReduce: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
    return 0;
#endif
}

// Genotype model with the shards cut at genotype boundaries: a resident run updates theta_g on its owner only.  Gathering the
// owners' copies (parameters, optimiser accumulators, window rows) = summing rows that hold zeros for what a shard does not own.
static size_t theta_rows(const bb_handle* h) { return 6 + (h->o.optimizer == BB_OPT_TRUNCATED_ADAGRAD ? 2 * (size_t)h->o.window : 0); }
static bool theta_partial(const bb_handle* h) { return h->M.kind == BB_MODEL_GENOTYPE && (h->g_lo > 0 || h->g_hi < h->M.G); }
static int theta_pack(bb_handle* h, int unpack) {
    const size_t n = theta_rows(h) * (size_t)h->M.G;
    int rc;
    if (!h->theta_buf && (rc = dalloc(h, &h->theta_buf, n))) return rc;
    const int W = h->o.optimizer == BB_OPT_TRUNCATED_ADAGRAD ? h->o.window : 0;
    const int nb = (int)std::min<size_t>((n + 255) / 256, 512);
#ifdef BB_EMU
    emu_launch(nb, 256, 0, [&](BBCtx& cx) { bb_block_theta_pack(cx, h->M, h->S, h->theta_buf, h->g_lo, h->g_hi, W, unpack, nb); });
#else
    hipLaunchKernelGGL(k_theta_pack, dim3(nb), dim3(256), 0, h->stream, h->M, h->S, h->theta_buf, h->g_lo, h->g_hi, W, unpack);
#endif
    return LAUNCH_CHECK();
}
// ranks of a multi-process run: through the RCCL communicator (collective: every rank is here after the same bb_run)
static int theta_sync_comm(bb_handle* h) {
    int rc;
    if ((rc = theta_pack(h, 0))) return rc;
    if ((rc = allreduce(h, h->theta_buf, theta_rows(h) * (size_t)h->M.G))) return rc;
    if ((rc = theta_pack(h, 1))) return rc;
    h->theta_stale = false;
    return dsync(h->stream);
}
// handles of one process (the shards of a multi-device handle; the emulation's ranks): summed on the host
static int theta_sync_local(bb_handle* const* hs, int n) {
    if (n < 2 || !theta_partial(hs[0])) return 0;
    const size_t len = theta_rows(hs[0]) * (size_t)hs[0]->M.G;
    std::vector<double> part(len), total(len, 0.0);
    int rc;
    for (int i = 0; i < n; ++i) {
        BB_ENTER(hs[i]);
        if ((rc = theta_pack(hs[i], 0)) || (rc = d2h(part.data(), hs[i]->theta_buf, len * 8, hs[i]->stream))) return rc;
        for (size_t k = 0; k < len; ++k) total[k] += part[k];
    }
    for (int i = 0; i < n; ++i) {
        BB_ENTER(hs[i]);
        if ((rc = h2d(hs[i]->theta_buf, total.data(), len * 8, hs[i]->stream)) || (rc = theta_pack(hs[i], 1)) || (rc = dsync(hs[i]->stream))) return rc;
        hs[i]->theta_stale = false;
    }
    return 0;
}

// first half of one MC sample: draw + moments (+ reduce to totals when sharded / genotype)
static int sample_half(bb_handle* h, const RunArgs& A) {
    int rc;
    if (h->M.kind == BB_MODEL_GENOTYPE && (rc = launch_geno(h, A, 0, 1, 0))) return rc;
    if ((rc = launch_sample(h, A))) return rc;
    if (h->use_reduce() && (rc = launch_reduce(h))) return rc;
    return 0;
}
// second half: gradient + update (+ the genotype block's exchange and update)
static int update_half(bb_handle* h, const RunArgs& A) {
    int rc;
    if ((rc = launch_update(h, A))) return rc;
    if (h->M.kind == BB_MODEL_GENOTYPE) {
        if ((rc = launch_geno_sum(h))) return rc;
        if ((rc = allreduce(h, h->S.gsum, (size_t)h->M.G))) return rc;
        if ((rc = launch_geno(h, A, 1, 0, A.par))) return rc;
    }
    return 0;
}

static bool elbo_wanted(const bb_handle* h, long long step) {
    return h->o.elbo_every > 0 && step % h->o.elbo_every == 0;
}

static int enqueue_step(bb_handle* h, long long step) {
    const int S = h->o.samples_per_step;
    int rc;
    for (int s = 0; s < S; ++s) {
        RunArgs A = make_args(h, step, s, S, true, elbo_wanted(h, step));
        if ((rc = sample_half(h, A))) return rc;
        if ((rc = allreduce(h, h->S.totals, (size_t)h->M.K))) return rc;
        if ((rc = update_half(h, A))) return rc;
    }
    return 0;
}

static int set_step(bb_handle* h, long long step) {
    unsigned long long c[2] = {(unsigned long long)step, (unsigned long long)step};
    h->step = step;
    h->sample = 0;
    return h2d(h->S.ctr, c, sizeof c, h->stream);
}

static int reset_optimizer(bb_handle* h) {
    const size_t D = (size_t)h->M.D;
    int rc;
    if ((rc = dzero(h->S.accl, (2 * D + 8) * 4, h->stream))) return rc;
    if (h->o.optimizer == BB_OPT_TRUNCATED_ADAGRAD) {
        if ((rc = dzero(h->S.hist, (size_t)h->o.window * 2 * (size_t)h->M.Dh * 8, h->stream))) return rc;
        if ((rc = dzero(h->S.acc_mu, D * 8, h->stream))) return rc;
        if ((rc = dzero(h->S.acc_om, D * 8, h->stream))) return rc;
    } else {
        std::vector<double> a(D, 1e-8);   // AdvancedVI: acc = fill(1e-8, size(x))
        if ((rc = h2d(h->S.acc_mu, a.data(), D * 8, h->stream))) return rc;
        if ((rc = h2d(h->S.acc_om, a.data(), D * 8, h->stream))) return rc;
    }
    std::vector<double> nanv(BB_ELBO_RING, NAN);
    if ((rc = h2d(h->S.elbo_ring, nanv.data(), nanv.size() * 8, h->stream))) return rc;
    // ready / inbox words never repeat, also across restarts: the base moves by the steps ASKED of the resident launch since
    // the last restart -- the same number on every rank of a sharded run, wherever a timeout may have stopped each of them
    h->epoch0 += (unsigned)h->req_steps + 1u;
    h->req_steps = 0;
    return set_step(h, 0);
}


// ------------------------------------------------------------------------------------------------
// single-process multi-device handle (bb_advi_opts.n_devices > 1; SURVEY.md 8b: "multi-GPU is driven inside the library by
// one host thread").  The group handle owns one shard handle per device (rank i of n on device_ids[i]); their inboxes are wired
// in process (hipDeviceEnablePeerAccess + plain pointers, no IPC), the resident launches of all shards are enqueued before any
// is waited for.  Where the resident launch is not possible (a shard refuses: genotype model, S > 1, ELBO recording ...) the
// group falls back to the split-phase step with the K moments summed on the host -- correct everywhere, slow.
// ------------------------------------------------------------------------------------------------
static int p2p_alloc_inbox(bb_handle* h);
static int p2p_wire(bb_handle* h, void* const* bases);
static int p2p_probe_launch(bb_handle* h, unsigned** res);
static int p2p_probe_collect(bb_handle* h, unsigned* res, int32_t* ok);
static int run_enqueue(bb_handle* h, int64_t n_steps);
static int run_finish(bb_handle* h);
static void owned_ranges(const bb_handle* sh, std::vector<std::pair<long long, long long>>& out) {
    const DevModel& M = sh->M;
    const long long b_lo = sh->b_lo, b_hi = sh->b_hi;
    const long long m_lo = std::max(b_lo, M.nn) - M.nn, m_hi = std::max(b_hi, M.nn) - M.nn;
    for (int r = 0; r < M.R; ++r) out.push_back({M.off_l[r] + b_lo * M.T[r], M.off_l[r] + b_hi * M.T[r]});
    if (M.kind == BB_MODEL_GENOTYPE && sh->g_hi > sh->g_lo) out.push_back({M.blk_lo[BK_S] + sh->g_lo, M.blk_lo[BK_S] + sh->g_hi});   // theta of its own genotypes
    if (m_hi <= m_lo) return;
    if (M.kind == BB_MODEL_FITNESS || M.kind == BB_MODEL_MULTIENV) {
        out.push_back({M.blk_lo[BK_S] + m_lo * M.E, M.blk_lo[BK_S] + m_hi * M.E});
        out.push_back({M.blk_lo[BK_LS] + m_lo * M.E, M.blk_lo[BK_LS] + m_hi * M.E});
    } else if (M.kind == BB_MODEL_GENOTYPE) {      // (theta: above -- sharded by genotype where the cuts allow, else all shards hold all of it)
        for (int k : {BK_TT, BK_LT, BK_LS}) out.push_back({M.blk_lo[k] + m_lo, M.blk_lo[k] + m_hi});
    } else {
        const long long E_ = M.kind == BB_MODEL_MULTIENV_REPLICATE ? M.E : 1;
        out.push_back({M.blk_lo[BK_S] + m_lo * E_, M.blk_lo[BK_S] + m_hi * E_});
        for (int r = 0; r < M.R; ++r)
            for (int k : {BK_TT, BK_LT, BK_LS}) out.push_back({M.blk_lo[k] + (r * M.nb + m_lo) * E_, M.blk_lo[k] + (r * M.nb + m_hi) * E_});
    }
}

static void group_destroy(bb_handle* g) {
    for (bb_handle* sh : g->shards) bb_destroy(sh);
    g->shards.clear();
    delete g;
}

static int group_create(const bb_model_desc* md, const bb_advi_opts* opts, bb_handle** out) {
    const int n = opts->n_devices;
    if (n > BB_MAX_WORLD) return bb_fail(BB_ERR_UNSUPPORTED, "at most %d devices per handle", BB_MAX_WORLD);
    if (opts->world_size != 1 || opts->rank != 0) return bb_fail(BB_ERR_INVALID, "n_devices > 1 needs rank 0 / world_size 1 (the handle shards by itself)");
    bb_handle* g = new bb_handle();
    g->o = *opts;
    g->o.device = opts->device_ids ? opts->device_ids[0] : 0;
    int rc = 0;
    for (int i = 0; i < n && !rc; ++i) {
        bb_advi_opts o = *opts;
        o.n_devices = 1;
        o.device_ids = nullptr;
        o.device = opts->device_ids ? opts->device_ids[i] : i;
        o.rank = i;
        o.world_size = n;
        o.launch_mode = opts->launch_mode == 1 ? 1 : 0;      // (a shard alone cannot run resident before its inbox is wired: mode 2 is enforced below, on the group)
        bb_handle* sh = nullptr;
        rc = create_inner(md, &o, &sh);
        if (!rc) { sh->in_group = true; g->shards.push_back(sh); }
    }
    if (rc) { group_destroy(g); return rc; }
    g->M = g->shards[0]->M;                      // (host-side copies of shapes and block ranges; the device pointers inside are shard 0's)
    g->blocks = g->shards[0]->blocks;
    g->b_lo = 0;
    g->b_hi = g->M.B;
    // resident launches with in-process peer-mapped inboxes, if every shard can (and the caller did not ask for two kernels)
    bool ok = opts->launch_mode != 1;
#ifndef BB_EMU
    for (int i = 0; i < n && ok; ++i)
        for (int j = 0; j < n && ok; ++j) {
            const int di = g->shards[i]->o.device, dj = g->shards[j]->o.device;
            if (di == dj) continue;
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, di, dj) != hipSuccess || !can) { ok = false; break; }
            DevGuard guard(di);
            hipError_t e = hipDeviceEnablePeerAccess(dj, 0);
            if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) ok = false;
            (void)hipGetLastError();
        }
#endif
    void* bases[BB_MAX_WORLD] = {};
    for (int i = 0; i < n && ok; ++i) {
        BB_ENTER(g->shards[i]);
        ok = p2p_alloc_inbox(g->shards[i]) == BB_OK;
        bases[i] = g->shards[i]->p2p_inbox;
    }
    for (int i = 0; i < n && ok; ++i) { BB_ENTER(g->shards[i]); ok = p2p_wire(g->shards[i], bases) == BB_OK; }
    if (ok) {                                    // transport probe: all devices at once
        unsigned* res[BB_MAX_WORLD] = {};
        for (int i = 0; i < n; ++i) { BB_ENTER(g->shards[i]); if (p2p_probe_launch(g->shards[i], &res[i]) != BB_OK) ok = false; }
        for (int i = 0; i < n; ++i) {
            BB_ENTER(g->shards[i]);
            int32_t good = 0;
            if (p2p_probe_collect(g->shards[i], res[i], &good) != BB_OK || !good) ok = false;
        }
    }
    if (ok) for (int i = 0; i < n && ok; ++i) ok = bb_p2p_enable(g->shards[i], 1) == BB_OK;
    // the two resident kernels speak different inbox protocols (k_res: tagged entries, k_persist: rows + ready words): all shards the same one
    if (ok) for (int i = 1; i < n && ok; ++i) ok = (g->shards[i]->res_P > 0) == (g->shards[0]->res_P > 0);
    if (!ok) for (bb_handle* sh : g->shards) if (sh->p2p_ready) (void)bb_p2p_enable(sh, 0);
    g->group_resident = ok;
    if (!ok && opts->launch_mode == 2) {
        const std::string why(g_err);                 // (bb_fail formats INTO g_err: the message must not be its own argument)
        group_destroy(g);
        return bb_fail(BB_ERR_UNSUPPORTED, "launch_mode = 2: the shards cannot run resident launches with peer-mapped inboxes: %s", why.c_str());
    }
    *out = g;
    return BB_OK;
}

static int group_run(bb_handle* g, int64_t n_steps) {
    int rc = 0;
    if (g->group_resident) {
#ifdef BB_EMU
        for (bb_handle* sh : g->shards) sh->req_steps += n_steps * std::max(sh->o.samples_per_step, 1);
        rc = emu_run_group(g->shards.data(), (int)g->shards.size(), n_steps);      // the emulation steps the shards in lock step
        if (!rc) rc = theta_sync_local(g->shards.data(), (int)g->shards.size());
        g->step = g->shards[0]->step;
        return rc;
#endif
        for (bb_handle* sh : g->shards) { BB_ENTER(sh); if ((rc = run_enqueue(sh, n_steps))) break; }
        int rc2 = 0;                              // (wait for whatever was launched, also after an error)
        for (bb_handle* sh : g->shards) { BB_ENTER(sh); const int r = run_finish(sh); if (r && !rc2) rc2 = r; }
        if (!rc) rc = rc2;
        if (!rc) rc = theta_sync_local(g->shards.data(), (int)g->shards.size());     // (genotype model: theta_g back from its owner)
    } else {
        // the two-kernel step of every shard with the exchanges (K moments; per-genotype gradient sums of the genotype model)
        // summed on the host: enqueue_step with host reductions in place of the RCCL all-reduces
        const int S = g->o.samples_per_step < 1 ? 1 : g->o.samples_per_step;
        const size_t K = (size_t)g->shards[0]->M.K, G = (size_t)g->shards[0]->M.G;
        const bool geno = g->shards[0]->M.kind == BB_MODEL_GENOTYPE;
        std::vector<double> part(std::max(K, G)), total(std::max(K, G));
        for (bb_handle* sh : g->shards) { BB_ENTER(sh); if ((rc = ensure_scratch(sh))) return rc; }
        auto exchange = [&](size_t n, double* DevState::*buf) -> int {
            std::fill(total.begin(), total.begin() + n, 0.0);
            for (bb_handle* sh : g->shards) {
                BB_ENTER(sh);
                int r = d2h(part.data(), sh->S.*buf, n * 8, sh->stream);
                if (r) return r;
                for (size_t k = 0; k < n; ++k) total[k] += part[k];
            }
            for (bb_handle* sh : g->shards) { BB_ENTER(sh); int r = h2d(sh->S.*buf, total.data(), n * 8, sh->stream); if (r) return r; }
            return 0;
        };
        for (int64_t it = 0; it < n_steps && !rc; ++it) {
            for (int smp = 0; smp < S && !rc; ++smp) {
                std::vector<RunArgs> As;
                for (bb_handle* sh : g->shards) As.push_back(make_args(sh, sh->step, smp, S, true, elbo_wanted(sh, sh->step)));
                for (size_t i = 0; i < g->shards.size() && !rc; ++i) { BB_ENTER(g->shards[i]); rc = sample_half(g->shards[i], As[i]); }
                if (!rc) rc = exchange(K, &DevState::totals);
                for (size_t i = 0; i < g->shards.size() && !rc; ++i) {
                    BB_ENTER(g->shards[i]);
                    rc = launch_update(g->shards[i], As[i]);
                    if (!rc && geno) rc = launch_geno_sum(g->shards[i]);
                }
                if (!rc && geno) rc = exchange(G, &DevState::gsum);
                for (size_t i = 0; i < g->shards.size() && !rc && geno; ++i) { BB_ENTER(g->shards[i]); rc = launch_geno(g->shards[i], As[i], 1, 0, As[i].par); }
            }
            for (bb_handle* sh : g->shards) sh->step++;
        }
        for (bb_handle* sh : g->shards) { BB_ENTER(sh); const int r = dsync(sh->stream); if (r && !rc) rc = r; }
    }
    g->step = g->shards[0]->step;
    return rc;
}

static int group_get_params(bb_handle* g, double* mu, double* omega) {
    const size_t D = (size_t)g->M.D;
    std::vector<double> tm(D), to(D);
    int rc = bb_get_params(g->shards[0], mu, omega);      // replicated blocks (and its own shard) from shard 0
    for (size_t i = 1; i < g->shards.size() && !rc; ++i) {
        if ((rc = bb_get_params(g->shards[i], tm.data(), to.data()))) break;
        std::vector<std::pair<long long, long long>> rg;
        owned_ranges(g->shards[i], rg);
        for (auto& r : rg) {
            std::copy(tm.begin() + r.first, tm.begin() + r.second, mu + r.first);
            std::copy(to.begin() + r.first, to.begin() + r.second, omega + r.first);
        }
    }
    return rc;
}

#define BB_GROUP_UNSUPPORTED(h, what) \
    do { if (!(h)->shards.empty()) return bb_fail(BB_ERR_UNSUPPORTED, what " is not available on a multi-device handle (n_devices > 1)"); } while (0)

extern "C" int bb_init_meanfield(bb_handle* h) {
    if (!h) return bb_fail(BB_ERR_INVALID, "null handle");
    if (!h->shards.empty()) { int rc = 0; for (bb_handle* sh : h->shards) if (!rc) rc = bb_init_meanfield(sh); h->step = 0; return rc; }
    BB_ENTER(h);
    h->theta_stale = false;
    const int nb = (int)std::min<long long>(((h->M.D + 1) / 2 + 255) / 256, 1024);
#ifdef BB_EMU
    emu_launch(nb, 256, 0, [&](BBCtx& cx) { bb_block_init(cx, h->M, h->S, h->o.seed, nb); });
#else
    hipLaunchKernelGGL(k_init, dim3(nb), dim3(256), 0, h->stream, h->M, h->S, (unsigned long long)h->o.seed);
    int rc = launch_check();
    if (rc) return rc;
#endif
    return reset_optimizer(h);
}

extern "C" int bb_set_params(bb_handle* h, const double* mu, const double* omega) {
    if (!h || !mu || !omega) return bb_fail(BB_ERR_INVALID, "null argument");
    std::vector<double> pm, po;
    if (!h->cidx.empty()) {          // the caller's order -> the handle's
        pm.resize(h->cidx.size()); po.resize(h->cidx.size());
        perm_gather(h, mu, pm.data()); perm_gather(h, omega, po.data());
        mu = pm.data(); omega = po.data();
    }
    if (!h->shards.empty()) { int rc = 0; for (bb_handle* sh : h->shards) if (!rc) rc = bb_set_params(sh, mu, omega); h->step = 0; return rc; }
    BB_ENTER(h);
    int rc;
    h->theta_stale = false;
    if ((rc = h2d(h->S.mu, mu, (size_t)h->M.D * 8, h->stream))) return rc;
    if ((rc = h2d(h->S.om, omega, (size_t)h->M.D * 8, h->stream))) return rc;
    return reset_optimizer(h);
}

static int get_params_raw(bb_handle* h, double* mu, double* omega) {
    if (!h->shards.empty()) return group_get_params(h, mu, omega);
    BB_ENTER(h);
    int rc;
    if ((rc = dsync(h->stream))) return rc;
    if ((rc = d2h(mu, h->S.mu, (size_t)h->M.D * 8, h->stream))) return rc;
    return d2h(omega, h->S.om, (size_t)h->M.D * 8, h->stream);
}
extern "C" int bb_get_params(bb_handle* h, double* mu, double* omega) {
    if (!h || !mu || !omega) return bb_fail(BB_ERR_INVALID, "null argument");
    if (h->cidx.empty()) return get_params_raw(h, mu, omega);
    std::vector<double> a(h->cidx.size()), b(h->cidx.size());
    int rc = get_params_raw(h, a.data(), b.data());
    if (rc) return rc;
    perm_scatter(h, a.data(), mu);
    perm_scatter(h, b.data(), omega);
    return BB_OK;
}

extern "C" int bb_get_posterior(bb_handle* h, double* mean, double* sigma) {
    if (!h || !mean || !sigma) return bb_fail(BB_ERR_INVALID, "null argument");
    BB_ENTER(h);
    int rc = bb_get_params(h, mean, sigma);
    if (rc) return rc;
    for (long long i = 0; i < h->M.D; ++i) {   // sigma = softplus(omega), O(D) once at the end
        const double om = sigma[i];
        sigma[i] = std::max(om, 0.0) + log1p(exp(-fabs(om)));
    }
    return BB_OK;
}

#ifndef BB_EMU
static int build_graph(bb_handle* h, int steps) {
    if (h->graph && h->graph_steps == steps) return 0;
    if (h->graph) { (void)hipGraphExecDestroy(h->graph); h->graph = nullptr; }
    hipGraph_t g = nullptr;
    BB_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    int rc = 0;
    for (int i = 0; i < steps && !rc; ++i) rc = enqueue_step(h, i);   // parity of i == parity of the real step (even start)
    hipError_t e = hipStreamEndCapture(h->stream, &g);
    if (rc || e != hipSuccess) {
        if (g) (void)hipGraphDestroy(g);
        (void)hipGetLastError();
        h->graph_failed = true;      // not an error: bb_run launches eagerly instead
        return 0;
    }
    e = hipGraphInstantiate(&h->graph, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) { h->graph = nullptr; (void)hipGetLastError(); h->graph_failed = true; return 0; }
    h->graph_steps = steps;
    return 0;
}
#endif

// bb_run in two halves (a multi-device handle enqueues on all its shards before it waits for any)
static int run_enqueue(bb_handle* h, int64_t n_steps) {
    if (h->sample != 0) return bb_fail(BB_ERR_INVALID, "a split-phase step is in flight");
    int rc = 0;
    int64_t done = 0;
    if (!(h->persist_P > 0) && (rc = ensure_scratch(h))) return rc;
    if (h->hstatus) h->hstatus[1] = 0;          // divergence flag of THIS run (nothing of this handle is in flight here)
#ifndef BB_EMU
    BB_HIP(hipEventRecord(h->ev0, h->stream));
#endif
    h->launches_last_run = 0;
    if (h->persist_P > 0 && n_steps > 0) {
        if ((rc = launch_persistent(h, n_steps))) return rc;
        done = n_steps;
        h->launches_last_run = (int)((n_steps + 4095) / 4096);
        if (theta_partial(h)) h->theta_stale = true;         // only the owner's theta_g moved (bb_run / group_run gather it afterwards)
    } else if (h->theta_stale && n_steps > 0 && (rc = theta_sync_comm(h))) return rc;
#ifndef BB_EMU
    // graphs: whole steps only, starting on an even step (static ping-pong parity), elbo_every
    // pattern must repeat with the graph -> only when ELBO recording is off; no collectives inside.
    int gs = h->o.steps_per_graph == 0 ? 50 : h->o.steps_per_graph;
    // A sharded step (with its RCCL all-reduce) can be captured too (BB_GRAPH_COLLECTIVE=1; equal results, no gain
    // measured: the step is GPU-bound), but multi-rank capture could not be exercised on the one-GPU boxes this was
    // developed on, so sharded runs launch eagerly by default.
    const bool graph_ok = gs > 0 && h->o.elbo_every == 0 && !h->graph_failed && !getenv("BB_NO_GRAPH") &&
                          ((h->o.world_size == 1 && !h->force_reduce) || getenv("BB_GRAPH_COLLECTIVE"));
    if (graph_ok) {
        gs &= ~1;
        if (gs < 2) gs = 2;
        if ((h->step & 1) && done < n_steps) { if ((rc = enqueue_step(h, h->step))) return rc; h->step++; done++; }
        if (n_steps - done >= gs) {
            if ((rc = build_graph(h, gs))) return rc;
            while (h->graph && n_steps - done >= gs) {
                BB_HIP(hipGraphLaunch(h->graph, h->stream));
                h->step += gs;
                done += gs;
            }
        }
    }
#endif
    for (; done < n_steps; ++done) {
        if ((rc = enqueue_step(h, h->step))) return rc;
        h->step++;
    }
#ifndef BB_EMU
    BB_HIP(hipEventRecord(h->ev1, h->stream));
#endif
    return BB_OK;
}
static int run_finish(bb_handle* h) {
#ifndef BB_EMU
    BB_HIP(hipStreamSynchronize(h->stream));
    float ms = 0;
    BB_HIP(hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->last_run_ms = ms;
#endif
    return check_persistent(h);
}

extern "C" int bb_run(bb_handle* h, int64_t n_steps) {
    if (!h || n_steps < 0) return bb_fail(BB_ERR_INVALID, "bad argument");
    if (!h->shards.empty()) return group_run(h, n_steps);
#ifdef BB_HOST_TIMES          // diagnostics (tools/launch_probe.py): where the host's share of a run goes
    timespec ht0, ht1, ht2, ht3;
    clock_gettime(CLOCK_MONOTONIC, &ht0);
#endif
    BB_ENTER(h);
#ifdef BB_HOST_TIMES
    clock_gettime(CLOCK_MONOTONIC, &ht1);
#endif
    int rc = run_enqueue(h, n_steps);
    if (rc) return rc;
#ifdef BB_HOST_TIMES
    clock_gettime(CLOCK_MONOTONIC, &ht2);
#endif
    rc = run_finish(h);
#ifdef BB_HOST_TIMES
    clock_gettime(CLOCK_MONOTONIC, &ht3);
    auto us = [](const timespec& a, const timespec& b) { return (b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) * 1e-3; };
    fprintf(stderr, "[bb_run %lld] device guard %.1f us, enqueue %.1f us, wait + status %.1f us, events %.1f us\n", (long long)n_steps, us(ht0, ht1), us(ht1, ht2), us(ht2, ht3), h->last_run_ms * 1e3);
#endif
#ifndef BB_EMU
    if (h->theta_stale && h->comm) {
        // Collective, so EVERY rank takes it whatever its own launch reported: the timeout word and the non-finite flag run_finish
        // looks at are this rank's alone, and a rank that returned early here would leave the clean ranks inside ncclAllReduce.
        // The rank's own error is reported afterwards (the gathered theta rows of a failed run mean nothing, but nobody hangs).
        const std::string why(g_err);
        const int rc2 = theta_sync_comm(h);
        if (rc) snprintf(g_err, sizeof g_err, "%s", why.c_str());
        else rc = rc2;
    }
#endif
    return rc;
}

extern "C" int bb_run_profiled(bb_handle* h, int64_t n_steps) {
    if (!h || n_steps < 0) return bb_fail(BB_ERR_INVALID, "bad argument");
    BB_GROUP_UNSUPPORTED(h, "bb_run_profiled");
    BB_ENTER(h);
#ifdef BB_EMU
    return bb_run(h, n_steps);
#else
    if (h->use_reduce()) return bb_fail(BB_ERR_UNSUPPORTED, "bb_run_profiled covers the single-GPU two-kernel step only");
    { const int rcs = ensure_scratch(h); if (rcs) return rcs; }
    const int S = h->o.samples_per_step;
    const size_t nl = (size_t)n_steps * S;
    std::vector<hipEvent_t> ev(3 * nl);
    for (auto& e : ev) BB_HIP(hipEventCreate(&e));
    int rc = 0;
    for (int64_t i = 0; i < n_steps && !rc; ++i) {
        for (int s = 0; s < S && !rc; ++s) {
            RunArgs A = make_args(h, h->step, s, S, true, elbo_wanted(h, h->step));
            const size_t k = 3 * ((size_t)i * S + s);
            BB_HIP(hipEventRecord(ev[k], h->stream));
            rc = launch_sample(h, A);
            BB_HIP(hipEventRecord(ev[k + 1], h->stream));
            if (!rc) rc = launch_update(h, A);
            BB_HIP(hipEventRecord(ev[k + 2], h->stream));
        }
        h->step++;
    }
    BB_HIP(hipStreamSynchronize(h->stream));
    double ts = 0, tu = 0;
    for (size_t k = 0; k < nl; ++k) {
        float a = 0, b = 0;
        BB_HIP(hipEventElapsedTime(&a, ev[3 * k], ev[3 * k + 1]));
        BB_HIP(hipEventElapsedTime(&b, ev[3 * k + 1], ev[3 * k + 2]));
        ts += a;
        tu += b;
    }
    for (auto& e : ev) (void)hipEventDestroy(e);
    if (nl) { h->avg_sample_ms = ts / nl; h->avg_update_ms = tu / nl; }
    return rc;
#endif
}

static int elbo_grad_raw(bb_handle* h, const double* mu, const double* omega, const double* eps, int32_t S,
                         double* elbo, double* grad_mu, double* grad_omega);
extern "C" int bb_elbo_grad(bb_handle* h, const double* mu, const double* omega, const double* eps, int32_t S,
                            double* elbo, double* grad_mu, double* grad_omega) {
    if (!h || !mu || !omega || S < 1) return bb_fail(BB_ERR_INVALID, "bad argument");
    if (h->cidx.empty()) return elbo_grad_raw(h, mu, omega, eps, S, elbo, grad_mu, grad_omega);
    const size_t D = h->cidx.size();
    std::vector<double> pm(D), po(D), pe, gm(D), go(D);
    perm_gather(h, mu, pm.data());
    perm_gather(h, omega, po.data());
    if (eps) { pe.resize((size_t)S * D); for (int s = 0; s < S; ++s) perm_gather(h, eps + (size_t)s * D, pe.data() + (size_t)s * D); }
    int rc = elbo_grad_raw(h, pm.data(), po.data(), eps ? pe.data() : nullptr, S, elbo, grad_mu ? gm.data() : nullptr, grad_omega ? go.data() : nullptr);
    if (rc) return rc;
    if (grad_mu) perm_scatter(h, gm.data(), grad_mu);
    if (grad_omega) perm_scatter(h, go.data(), grad_omega);
    return BB_OK;
}
static int elbo_grad_raw(bb_handle* h, const double* mu, const double* omega, const double* eps, int32_t S,
                         double* elbo, double* grad_mu, double* grad_omega) {
    if (!h || !mu || !omega || S < 1) return bb_fail(BB_ERR_INVALID, "bad argument");
    BB_GROUP_UNSUPPORTED(h, "bb_elbo_grad");
    BB_ENTER(h);
    if (h->o.world_size > 1 && !eps) return bb_fail(BB_ERR_UNSUPPORTED, "bb_elbo_grad on a sharded handle needs explicit eps");
    const size_t D = (size_t)h->M.D;
    int rc;
    if ((rc = ensure_scratch(h))) return rc;
    if ((rc = d2d(h->bak_mu, h->S.mu, D * 8, h->stream))) return rc;
    if ((rc = d2d(h->bak_om, h->S.om, D * 8, h->stream))) return rc;
    if ((rc = h2d(h->S.mu, mu, D * 8, h->stream))) return rc;
    if ((rc = h2d(h->S.om, omega, D * 8, h->stream))) return rc;
    if (eps) {
        if (h->eps_cap < (size_t)S * D) {
            if (h->eps_buf) dfree(h->eps_buf);
            h->eps_buf = nullptr;
            void* p = nullptr;
            if ((rc = dmalloc(&p, (size_t)S * D * 8))) return rc;
            h->eps_buf = (double*)p;
            h->eps_cap = (size_t)S * D;
        }
        if ((rc = h2d(h->eps_buf, eps, (size_t)S * D * 8, h->stream))) return rc;
        h->S.eps_in = h->eps_buf;
    }
    double* es = nullptr;   // per-sample ELBO values
    if (S > h->o.samples_per_step + 64) {
        void* p = nullptr;
        if ((rc = dmalloc(&p, (size_t)S * 8))) return rc;
        es = (double*)p;
    }
    double* saved_es = h->S.elbo_sample;
    if (es) h->S.elbo_sample = es;
    if ((rc = sync_descriptors(h))) return rc;
    for (int s = 0; s < S && !rc; ++s) {
        RunArgs A = make_args(h, h->step, s, S, false, true);
        rc = sample_half(h, A);
        if (!rc) rc = allreduce(h, h->S.totals, (size_t)h->M.K);
        if (!rc) rc = update_half(h, A);
    }
    std::vector<double> ev((size_t)S);
    if (!rc) rc = d2h(ev.data(), h->S.elbo_sample, (size_t)S * 8, h->stream);
    h->S.elbo_sample = saved_es;
    if (es) dfree(es);
    h->S.eps_in = nullptr;
    { int rcs = sync_descriptors(h); if (!rc) rc = rcs; }
    if (!rc && grad_mu) rc = d2h(grad_mu, h->S.gacc_mu, D * 8, h->stream);
    if (!rc && grad_omega) rc = d2h(grad_omega, h->S.gacc_om, D * 8, h->stream);
    int rc2 = d2d(h->S.mu, h->bak_mu, D * 8, h->stream);
    int rc3 = d2d(h->S.om, h->bak_om, D * 8, h->stream);
    int rc4 = dsync(h->stream);
    if (rc) return rc;
    if (rc2 || rc3 || rc4) return rc2 ? rc2 : (rc3 ? rc3 : rc4);
    if (elbo) {
        double v = 0;
        for (int s = 0; s < S; ++s) v += ev[(size_t)s];
        *elbo = v / S;
    }
    return BB_OK;
}

// log-joint density and its gradient at a point of the unconstrained latent space: the ELBO machinery with the
// draw pinned to the mean (eps = 0) and sigma = softplus(omega) = 1, minus the entropy terms -- what an HMC / NUTS
// sampler asks of a model (`LogDensityProblems.logdensity_and_gradient`; the reference's src/mcmc.jl:86-160 path)
extern "C" int bb_logdensity_grad(bb_handle* h, const double* z, double* logp, double* grad) {
    if (!h || !z) return bb_fail(BB_ERR_INVALID, "null argument");
    BB_ENTER(h);
    const size_t D = (size_t)h->M.D;
    const double om1 = 0.54132485461291810;                  // log(e - 1): softplus = 1
    if (h->ld_omega.size() != D) { h->ld_omega.assign(D, om1); h->ld_zero.assign(D, 0.0); }
    double elbo = 0.0;
    int rc = bb_elbo_grad(h, z, h->ld_omega.data(), h->ld_zero.data(), 1, &elbo, grad, nullptr);
    if (rc) return rc;
    if (logp) *logp = elbo - 0.5 * (double)D * (1.0 + BB_LOG2PI) - (double)D * log(log1p(exp(om1)));
    return BB_OK;
}

extern "C" int bb_get_elbo_trace(bb_handle* h, int64_t first_step, int64_t n, double* out) {
    if (!h || !out || n < 0) return bb_fail(BB_ERR_INVALID, "bad argument");
    if (!h->shards.empty()) return bb_get_elbo_trace(h->shards[0], first_step, n, out);      // (every shard forms the same estimate from the same totals)
    BB_ENTER(h);
    if (h->o.elbo_every <= 0) return bb_fail(BB_ERR_INVALID, "ELBO recording is off (elbo_every = 0)");
    std::vector<double> ring(BB_ELBO_RING);
    int rc = dsync(h->stream);
    if (rc) return rc;
    if ((rc = d2h(ring.data(), h->S.elbo_ring, ring.size() * 8, h->stream))) return rc;
    const int64_t ev = h->o.elbo_every;
    const int64_t newest = h->step > 0 ? (h->step - 1) / ev : -1;
    for (int64_t k = 0; k < n; ++k) {
        const int64_t st = first_step + k * ev;
        const int64_t idx = st / ev;
        const bool have = st % ev == 0 && st >= 0 && st < h->step && idx > newest - BB_ELBO_RING;
        out[k] = have ? ring[(size_t)(idx % BB_ELBO_RING)] : NAN;
    }
    return BB_OK;
}

extern "C" int bb_debug_normals(bb_handle* h, int64_t step, uint32_t stream, int64_t lo, int64_t hi, double* out) {
    if (!h || !out || lo < 0 || hi < lo) return bb_fail(BB_ERR_INVALID, "bad argument");
    if (!h->shards.empty()) return bb_debug_normals(h->shards[0], step, stream, lo, hi, out);
    BB_ENTER(h);
    const size_t n = (size_t)(hi - lo);
    if (n == 0) return BB_OK;
    int rc;
    if (h->dbg_cap < n) {
        if (h->dbg_buf) dfree(h->dbg_buf);
        h->dbg_buf = nullptr;
        void* p = nullptr;
        if ((rc = dmalloc(&p, n * 8))) return rc;
        h->dbg_buf = (double*)p;
        h->dbg_cap = n;
    }
    const int nb = (int)std::min<size_t>((n / 2 + 256) / 256, 1024);
#ifdef BB_EMU
    emu_launch(nb, 256, 0, [&](BBCtx& cx) { bb_block_normals(cx, h->o.seed, (unsigned)step, stream, lo, hi, h->dbg_buf, nb); });
#else
    hipLaunchKernelGGL(k_normals, dim3(nb), dim3(256), 0, h->stream, (unsigned long long)h->o.seed, (unsigned)step, stream,
                       (long long)lo, (long long)hi, h->dbg_buf);
    if ((rc = launch_check())) return rc;
#endif
    return d2h(out, h->dbg_buf, n * 8, h->stream);
}

extern "C" int bb_debug_stamps(bb_handle* h, uint64_t* out, int64_t n) {
    if (!h || !out) return bb_fail(BB_ERR_INVALID, "bad argument");
    if (!h->shards.empty()) return bb_debug_stamps(h->shards[0], out, n);
    if (n < 0) { out[0] = (uint64_t)(h->nblk + 8); return BB_OK; }     // rows of the block-stamp area (the per-wave area follows it)
    BB_ENTER(h);
    const int64_t have = (int64_t)(h->nblk + 8) * (32 + 64);
    int rc = dsync(h->stream);
    if (rc) return rc;
    return d2h(out, h->S.stamps, (size_t)std::min(n, have) * 8, h->stream);
}

// ------------------------------------------------------------------------------------------------
// cross-GPU leg of the resident launch: inbox, IPC handles, transport probe, switch
// ------------------------------------------------------------------------------------------------
static size_t p2p_rows_bytes(const bb_handle* h) {
    const size_t n = (size_t)2 * h->o.world_size * 8 * (size_t)(h->M.K + 2 * h->M.nt1) * 8;
    return (n + 255) & ~(size_t)255;
}
// ready words: [2][world][8] lines of 128 B, then the probe's [world] lines
static size_t p2p_probe_words_off(const bb_handle* h) { return (size_t)32 * 2 * h->o.world_size * 8; }

static void p2p_release(bb_handle* h) {
#ifndef BB_EMU
    for (int r = 0; r < BB_MAX_WORLD; ++r)
        if (h->p2p_peer[r] && r != h->o.rank && !h->in_group) (void)hipIpcCloseMemHandle(h->p2p_peer[r]);
    if (h->p2p_inbox) (void)hipFree(h->p2p_inbox);
#else
    if (h->p2p_inbox) free(h->p2p_inbox);
#endif
    h->p2p_inbox = nullptr;
    h->p2p_ready = h->p2p_on = false;
}

// this rank's inbox: fine-grained device memory (remote stores and local polls must meet in memory, not in either side's L2)
static int p2p_alloc_inbox(bb_handle* h) {
    // (BB_P2P_SELF=1, diagnostics: a whole-problem handle runs the inbox protocol against its own inbox -- tools/xg_self.py)
    if (h->o.world_size < 2 && !(getenv("BB_P2P_SELF") && atoi(getenv("BB_P2P_SELF")) > 0)) return bb_fail(BB_ERR_INVALID, "the cross-GPU leg needs a sharded handle (world_size > 1)");
    if (h->o.world_size > BB_MAX_WORLD) return bb_fail(BB_ERR_UNSUPPORTED, "at most %d ranks", BB_MAX_WORLD);
    if (h->M.kind == BB_MODEL_GENOTYPE && !h->M.geno_sorted)
        return bb_fail(BB_ERR_UNSUPPORTED, "the genotype model's resident launch needs geno_idx in consecutive runs (shards must own whole genotypes)");
    if (h->p2p_inbox) return BB_OK;
    h->p2p_rows_bytes = p2p_rows_bytes(h);
    h->p2p_gran_off = (h->p2p_rows_bytes + (p2p_probe_words_off(h) + (size_t)32 * h->o.world_size) * 4 + 255) & ~(size_t)255;
    h->p2p_bytes = h->p2p_gran_off + 2 * h->p2p_rows_bytes + 16 * 16 * (size_t)(h->M.K + 2 * h->M.nt1);      // (tagged rows: 16 B per entry; + the polls' eight-rows-in-flight slack)
#ifdef BB_EMU
    h->p2p_inbox = calloc(1, h->p2p_bytes);
    if (!h->p2p_inbox) return bb_fail(BB_ERR_DEVICE, "out of memory");
#else
    BB_HIP(hipExtMallocWithFlags(&h->p2p_inbox, h->p2p_bytes, hipDeviceMallocFinegrained));
    BB_HIP(hipMemsetAsync(h->p2p_inbox, 0, h->p2p_bytes, h->stream));
    BB_HIP(hipStreamSynchronize(h->stream));
#endif
    return BB_OK;
}

// every rank's inbox as seen from this rank (bases[own rank] = the local inbox)
static int p2p_wire(bb_handle* h, void* const* bases) {
    for (int r = 0; r < h->o.world_size; ++r) {
        if (!bases[r]) return bb_fail(BB_ERR_COMM, "rank %d's inbox did not map", r);
        h->p2p_peer[r] = bases[r];
        h->S.xout[r] = (double*)bases[r];
        h->S.xout_rdy[r] = (unsigned*)((char*)bases[r] + h->p2p_rows_bytes);
        h->S.xgr[r] = (bb_gran*)((char*)bases[r] + h->p2p_gran_off);
    }
    h->p2p_ready = true;
    return sync_descriptors(h);
}

extern "C" int bb_p2p_export(bb_handle* h, void* handle_out) {
    if (!h || !handle_out) return bb_fail(BB_ERR_INVALID, "null argument");
    if (!h->shards.empty()) return bb_fail(BB_ERR_UNSUPPORTED, "a multi-device handle wires its shards itself");
    BB_ENTER(h);
    memset(handle_out, 0, BB_P2P_HANDLE_BYTES);
    int rc = p2p_alloc_inbox(h);
    if (rc) return rc;
#ifdef BB_EMU
    memcpy(handle_out, &h->p2p_inbox, sizeof(void*));
#else
    static_assert(sizeof(hipIpcMemHandle_t) <= BB_P2P_HANDLE_BYTES, "IPC handle does not fit");
    hipIpcMemHandle_t hnd;
    BB_HIP(hipIpcGetMemHandle(&hnd, h->p2p_inbox));
    memcpy(handle_out, &hnd, sizeof hnd);
#endif
    return BB_OK;
}

extern "C" int bb_p2p_import(bb_handle* h, const void* handles) {
    if (!h || !handles) return bb_fail(BB_ERR_INVALID, "null argument");
    if (!h->shards.empty()) return bb_fail(BB_ERR_UNSUPPORTED, "a multi-device handle wires its shards itself");
    BB_ENTER(h);
    if (!h->p2p_inbox) return bb_fail(BB_ERR_INVALID, "bb_p2p_export comes first");
    const int W = h->o.world_size;
    void* bases[BB_MAX_WORLD] = {};
    for (int r = 0; r < W; ++r) {
        void* base = nullptr;
        if (r == h->o.rank) base = h->p2p_inbox;
        else {
            const char* src = (const char*)handles + (size_t)r * BB_P2P_HANDLE_BYTES;
#ifdef BB_EMU
            memcpy(&base, src, sizeof(void*));
#else
            if (!h->p2p_peer[r]) {
                hipIpcMemHandle_t hnd;
                memcpy(&hnd, src, sizeof hnd);
                BB_HIP(hipIpcOpenMemHandle(&base, hnd, hipIpcMemLazyEnablePeerAccess));
            } else base = h->p2p_peer[r];
#endif
        }
        bases[r] = base;
    }
    return p2p_wire(h, bases);
}

static void dfree_probe(unsigned* res) {
#ifndef BB_EMU
    if (res) (void)hipFree(res);
#else
    (void)res;
#endif
}

// transport probe in two halves, so that one host thread can run it on several devices at once: every rank's kernel waits for
// the tokens of all the others
static int p2p_probe_launch(bb_handle* h, unsigned** res) {
    *res = nullptr;
    ++h->p2p_seq;
#ifndef BB_EMU
    BB_HIP(hipMalloc((void**)res, 64 * 4));
    BB_HIP(hipMemsetAsync(*res, 0, 64 * 4, h->stream));
    hipLaunchKernelGGL(k_p2p_probe_seq, dim3(1), dim3(64), 0, h->stream, h->S, h->o.rank, h->o.world_size, p2p_probe_words_off(h), h->p2p_seq, *res);
    return launch_check();
#else
    return BB_OK;
#endif
}
static int p2p_probe_collect(bb_handle* h, unsigned* res, int32_t* ok) {
    const int W = h->o.world_size;
    *ok = 0;
#ifdef BB_EMU
    (void)res;
    // single address space: the "transport" is a pointer; check that every rank's inbox is distinct and writable
    for (int r = 0; r < W; ++r) {
        if (!h->S.xout_rdy[r]) return BB_OK;
        for (int q = 0; q < r; ++q) if (h->S.xout_rdy[q] == h->S.xout_rdy[r]) return BB_OK;
    }
    *ok = 1;
    return BB_OK;
#else
    unsigned host[64] = {0};
    int rc = d2h(host, res, sizeof host, h->stream);
    (void)hipFree(res);
    if (rc) return rc;
    int good = 1;
    for (int r = 0; r < W; ++r) good &= host[r] == 1u;
    *ok = good;
    return BB_OK;
#endif
}

extern "C" int bb_p2p_selftest(bb_handle* h, int32_t* ok) {
    if (!h || !ok) return bb_fail(BB_ERR_INVALID, "null argument");
    if (!h->shards.empty()) return bb_fail(BB_ERR_UNSUPPORTED, "a multi-device handle wires its shards itself");
    BB_ENTER(h);
    *ok = 0;
    if (!h->p2p_ready) return bb_fail(BB_ERR_INVALID, "bb_p2p_import comes first");
    // Every rank's token carries the same sequence number only if all ranks call this the same number of times
    // (they do: the caller votes on the outcome), so a peer's token is predictable: replace the low byte.
    unsigned* res = nullptr;
    int rc = p2p_probe_launch(h, &res);
    if (rc) { dfree_probe(res); return rc; }
    return p2p_probe_collect(h, res, ok);
}

extern "C" int bb_p2p_enable(bb_handle* h, int32_t on) {
    if (!h) return bb_fail(BB_ERR_INVALID, "null argument");
    BB_GROUP_UNSUPPORTED(h, "bb_p2p_enable");
    BB_ENTER(h);
    if (on && !h->p2p_ready) return bb_fail(BB_ERR_INVALID, "bb_p2p_import comes first");
    h->p2p_on = on != 0;
    const int saved_mode = h->o.launch_mode;
    if (h->p2p_on) h->o.launch_mode = 2;      // ask setup_persistent to say why not
    int rc = setup_persistent(h);
    h->o.launch_mode = saved_mode;
    if (rc || (h->p2p_on && h->persist_P == 0)) {
        h->p2p_on = false;
        (void)setup_persistent(h);
        (void)sync_descriptors(h);
        return rc ? rc : bb_fail(BB_ERR_UNSUPPORTED, "resident launch not possible on this shard");
    }
    if ((rc = sync_descriptors(h))) return rc;
#ifndef BB_EMU
    if (h->p2p_on) {
        // a zero-step launch loads the kernel's code object now (seconds on a cold process), not while the peers already poll;
        // the first real launch still gets a longer poll limit (launch skew between the ranks' processes)
        h->p2p_first = true;
        const long long keep = h->req_steps;
        rc = launch_persistent(h, 0);
        h->req_steps = keep;
        if (!rc) rc = dsync(h->stream);
        if (rc) return rc;
    }
#endif
    return BB_OK;
}

extern "C" int64_t bb_hier_units(const bb_handle* h) {
    if (!h || h->M.kind < BB_MODEL_GENOTYPE) return 0;
    return h->M.blk_hi[BK_TT] - h->M.blk_lo[BK_TT];
}

static int hier_fitness_raw(bb_handle* h, int32_t n_samples, uint64_t seed, double* median, double* stdv);
extern "C" int bb_hier_fitness(bb_handle* h, int32_t n_samples, uint64_t seed, double* median, double* stdv) {
    if (!h || !median || !stdv) return bb_fail(BB_ERR_INVALID, "null argument");
    if (h->perm_m.empty()) return hier_fitness_raw(h, n_samples, seed, median, stdv);
    const size_t n = h->perm_m.size();          // (genotype model: one unit per mutant)
    std::vector<double> a(n), b(n);
    int rc = hier_fitness_raw(h, n_samples, seed, a.data(), b.data());
    if (rc) return rc;
    for (size_t m = 0; m < n; ++m) { median[(size_t)h->perm_m[m]] = a[m]; stdv[(size_t)h->perm_m[m]] = b[m]; }
    return BB_OK;
}
static int hier_fitness_raw(bb_handle* h, int32_t n_samples, uint64_t seed, double* median, double* stdv) {
    if (!h || !median || !stdv) return bb_fail(BB_ERR_INVALID, "null argument");
    if (!h->shards.empty()) {
        // the whole posterior onto shard 0 (entries it does not own are dead weight there: never read by its tiles), then its sampler
        bb_handle* s0 = h->shards[0];
        const size_t D = (size_t)h->M.D;
        std::vector<double> mu(D), om(D);
        int rc = group_get_params(h, mu.data(), om.data());
        BB_ENTER(s0);
        if (!rc) rc = h2d(s0->S.mu, mu.data(), D * 8, s0->stream);
        if (!rc) rc = h2d(s0->S.om, om.data(), D * 8, s0->stream);
        if (rc) return rc;
        const int ws = s0->o.world_size;
        s0->o.world_size = 1;
        rc = hier_fitness_raw(s0, n_samples, seed, median, stdv);
        s0->o.world_size = ws;
        return rc;
    }
    BB_ENTER(h);
    if (h->M.kind < BB_MODEL_GENOTYPE) return bb_fail(BB_ERR_INVALID, "bb_hier_fitness applies to the hierarchical models only");
    if (n_samples < 2 || n_samples > 16384) return bb_fail(BB_ERR_UNSUPPORTED, "n_samples must be in 2..16384");
    // (a shard of a sharded run keeps full-length parameter arrays but only its own barcodes' entries are current: the caller makes them
    //  whole first -- bb_set_params with the gathered vector, as barbay.jl_amd.vi does -- the draws are then those of a whole-problem handle)
    const long long n = bb_hier_units(h);
    const size_t D = (size_t)h->M.D;
    int rc;
    if ((rc = ensure_scratch(h))) return rc;
    // posterior sigma = softplus(omega) into the (free between steps) z scratch array
    std::vector<double> om(D);
    if ((rc = dsync(h->stream)) || (rc = d2h(om.data(), h->S.om, D * 8, h->stream))) return rc;
    for (size_t i = 0; i < D; ++i) om[i] = std::max(om[i], 0.0) + log1p(exp(-fabs(om[i])));
    if ((rc = h2d(h->S.zsv, om.data(), D * 8, h->stream))) return rc;
    HierArgs H;
    memset(&H, 0, sizeof H);
    H.mean = h->S.mu;
    H.sigma = h->S.zsv;
    H.median_out = h->S.asv;           // n <= D: scratch arrays are free between steps
    H.std_out = h->S.hsv;
    H.n_units = n;
    H.lo_theta = h->M.blk_lo[BK_S];
    H.lo_tt = h->M.blk_lo[BK_TT];
    H.lo_lt = h->M.blk_lo[BK_LT];
    H.theta_mod = h->M.kind == BB_MODEL_GENOTYPE ? 0 : (h->M.blk_hi[BK_S] - h->M.blk_lo[BK_S]);
    H.geno_idx = h->M.geno_idx;
    H.n_samples = n_samples;
    H.n_pad = 2;
    while (H.n_pad < n_samples) H.n_pad <<= 1;
    H.seed = seed;
    const int nthr = H.n_pad >= 2048 ? 1024 : 256;
    const size_t lds = (size_t)H.n_pad + nthr + 8;
    const int nb = (int)std::min<long long>(n, 2048);
#ifdef BB_EMU
    emu_launch(nb, nthr, lds, [&](BBCtx& cx) { bb_block_hier(cx, H, nb); });
#else
    if (lds * 8 > 64 * 1024 && hipFuncSetAttribute((const void*)k_hier, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds * 8)) != hipSuccess)
        return bb_fail(BB_ERR_DEVICE, "cannot raise dynamic LDS to %zu bytes", lds * 8);
    hipLaunchKernelGGL(k_hier, dim3(nb), dim3(nthr), lds * 8, h->stream, H);
    if ((rc = launch_check())) return rc;
#endif
    if ((rc = d2h(median, H.median_out, (size_t)n * 8, h->stream))) return rc;
    return d2h(stdv, H.std_out, (size_t)n * 8, h->stream);
}

extern "C" int bb_get_stats(bb_handle* h, bb_stats* s) {
    if (!h || !s) return bb_fail(BB_ERR_INVALID, "null argument");
    BB_ENTER(h);
    if (!h->shards.empty()) {       // shard 0's figures; what adds up over the shards is added up
        int rc = bb_get_stats(h->shards[0], s);
        for (size_t i = 1; i < h->shards.size() && !rc; ++i) {
            bb_stats t;
            if ((rc = bb_get_stats(h->shards[i], &t))) break;
            s->bytes_per_step += t.bytes_per_step; s->bytes_sample += t.bytes_sample; s->bytes_update += t.bytes_update;
            s->n_blocks += t.n_blocks;
            s->rows_same_xcd += t.rows_same_xcd;
            s->device_bytes += t.device_bytes;
            s->last_run_ms = std::max(s->last_run_ms, t.last_run_ms);
            s->persistent_pairs = std::min(s->persistent_pairs, t.persistent_pairs);
            s->resident_kernel = std::min(s->resident_kernel, t.resident_kernel);
        }
        s->shard_lo = 0;
        s->shard_hi = h->M.B;
        s->geno_lo = 0;
        s->geno_hi = h->M.G;
        return rc;
    }
    memset(s, 0, sizeof *s);
    s->n_latents = h->M.D;
    s->n_moments = h->M.K;
    s->steps_done = h->step;
    s->shard_lo = h->b_lo;
    s->shard_hi = h->b_hi;
    s->geno_lo = h->g_lo;
    s->geno_hi = h->g_hi;
    s->device_bytes = h->dev_bytes;
    s->window_row = h->M.Dh;
    s->bytes_sample = h->bytes_sample;
    s->bytes_update = h->bytes_update;
    s->bytes_per_step = h->bytes_update;   // theta read once when the two sweeps are fused (96 D + 4 TBR)
    s->last_run_ms = h->last_run_ms;
    s->avg_sample_ms = h->avg_sample_ms;
    s->avg_update_ms = h->avg_update_ms;
    s->n_blocks = h->res_P ? h->res_nblk : h->nblk;
    s->block_threads = h->nthr;
    s->lds_bytes = (int32_t)((h->persist_P > 0 ? h->lds_doubles_p : h->lds_doubles) * 8);
    s->persistent_pairs = h->persist_P;
    s->launches_last_run = h->launches_last_run;
    s->resident_kernel = h->res_P > 0 ? (h->res_stream ? 3 : 2) : (h->persist_P > 0 ? 1 : 0);
    if (h->res_P > 0 && h->res_nblk > 0) {          // (k_res / k_stream: what the tiles of the last launch decided about their row stores)
        std::vector<int> sel((size_t)h->res_nblk);
        int rc = d2h(sel.data(), h->S.xsel, sel.size() * sizeof(int), h->stream);
        if (rc) return rc;
        for (int v : sel) s->rows_same_xcd += v > 0 ? 1 : 0;
    }
    return BB_OK;
}

// The kernel bb_run launches, as text: the template instance in declaration order -- k_res<KIND, P, NT, XG, TT, AP, MS>,
// k_stream<KIND, NT, TT>, k_persist<KIND, P, NT[, XG]> -- or "k_sample + k_update" for the two-kernel step.  Tests and bench.py
// print / assert THIS instead of reconstructing the instance from bb_stats.
extern "C" int bb_kernel_name(bb_handle* h, char* buf, int64_t len) {
    if (!h || !buf || len < 2) return bb_fail(BB_ERR_INVALID, "null argument / no room");
    if (!h->shards.empty()) return bb_kernel_name(h->shards[0], buf, len);
    const char* nm = "";
    char tmp[96];
#ifdef BB_EMU
    // (the emulation runs the block programs as host functions: it names the launch, the compile-time T / AP / MS are the product's)
    if (h->res_P && h->res_stream) snprintf(tmp, sizeof tmp, "emu:k_stream<%d,%d,%d%s>", h->M.kind, h->nthr, uniform_T(h->M), (h->o.samples_per_step != 1 || h->o.elbo_every != 0) ? ",true" : "");
    else if (h->res_P) snprintf(tmp, sizeof tmp, "emu:k_res<%d,%d,%d,%s,*,%s,%s>", h->M.kind, h->res_P, h->nthr, h->p2p_on ? "true" : "false",
                                br_any_parity(h->M) ? "true" : "false", (h->o.samples_per_step != 1 || h->o.elbo_every != 0) ? "true" : "false");
    else if (h->persist_P) snprintf(tmp, sizeof tmp, "emu:k_persist<%d,%d,%d>", h->M.kind, h->persist_P, h->nthr);
    else snprintf(tmp, sizeof tmp, "emu:k_sample + k_update");
    nm = tmp;
#else
    if (h->res_P && h->res_stream) (void)stream_kernel(h->M.kind, h->nthr, uniform_T(h->M), &nm, res_ms(h));
    else if (h->res_P) (void)res_kernel(h->M.kind, h->res_P, h->nthr, h->p2p_on, uniform_T(h->M), br_any_parity(h->M), res_ms(h), &nm);
    else if (h->persist_P) (void)persist_kernel(h->M.kind, h->persist_P, h->nthr, h->p2p_on, &nm);
    else { snprintf(tmp, sizeof tmp, "k_sample<%d> + k_update<%d>", h->M.kind, h->M.kind); nm = tmp; }
#endif
    snprintf(buf, (size_t)len, "%s", nm);
    return BB_OK;
}

// ------------------------------------------------------------------------------------------------
// sharded execution
// ------------------------------------------------------------------------------------------------
extern "C" int bb_comm_make_id(void* id_out) {
    if (!id_out) return bb_fail(BB_ERR_INVALID, "null argument");
#ifdef BB_EMU
    return bb_fail(BB_ERR_COMM, "no RCCL in the emulation build");
#else
    int rc = rccl_load();
    if (rc) return rc;
    bb_ncclUniqueId id;
    rc = g_rccl.GetUniqueId(&id);
    if (rc) return bb_fail(BB_ERR_COMM, "ncclGetUniqueId: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error");
    memcpy(id_out, &id, sizeof id);
    return BB_OK;
#endif
}

extern "C" int bb_comm_init(bb_handle* h, const void* id_in) {
    if (!h || !id_in) return bb_fail(BB_ERR_INVALID, "null argument");
    BB_GROUP_UNSUPPORTED(h, "bb_comm_init");
    BB_ENTER(h);
#ifdef BB_EMU
    return bb_fail(BB_ERR_COMM, "no RCCL in the emulation build");
#else
    int rc = rccl_load();
    if (rc) return rc;
    BB_HIP(hipSetDevice(h->o.device));
    bb_ncclUniqueId id;
    memcpy(&id, id_in, sizeof id);
    rc = g_rccl.CommInitRank(&h->comm, h->o.world_size, id, h->o.rank);
    if (rc) { h->comm = nullptr; return bb_fail(BB_ERR_COMM, "ncclCommInitRank: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "error"); }
    return BB_OK;
#endif
}

extern "C" int bb_step_moments(bb_handle* h, double* partial) {
    if (!h || !partial) return bb_fail(BB_ERR_INVALID, "null argument");
    BB_GROUP_UNSUPPORTED(h, "bb_step_moments");
    BB_ENTER(h);
    if (h->M.kind == BB_MODEL_GENOTYPE && h->o.world_size > 1)
        return bb_fail(BB_ERR_UNSUPPORTED, "split-phase stepping of the sharded genotype model needs a second exchange; use bb_comm_init + bb_run");
    const int S = h->o.samples_per_step;
    int rc;
    if ((rc = ensure_scratch(h))) return rc;
    RunArgs A = make_args(h, h->step, h->sample, S, true, elbo_wanted(h, h->step));
    A.red = h->S.totals;   // the caller-reduced totals come back through bb_step_apply
    A.nred = 1;
    if (h->M.kind == BB_MODEL_GENOTYPE && (rc = launch_geno(h, A, 0, 1, 0))) return rc;
    if ((rc = launch_sample(h, A))) return rc;
    if ((rc = launch_reduce(h))) return rc;
    return d2h(partial, h->S.totals, (size_t)h->M.K * 8, h->stream);
}

extern "C" int bb_step_apply(bb_handle* h, const double* total) {
    if (!h || !total) return bb_fail(BB_ERR_INVALID, "null argument");
    BB_GROUP_UNSUPPORTED(h, "bb_step_apply");
    BB_ENTER(h);
    const int S = h->o.samples_per_step;
    RunArgs A = make_args(h, h->step, h->sample, S, true, elbo_wanted(h, h->step));
    A.red = h->S.totals;
    A.nred = 1;
    int rc;
    if ((rc = ensure_scratch(h))) return rc;
    if ((rc = h2d(h->S.totals, total, (size_t)h->M.K * 8, h->stream))) return rc;
    if ((rc = update_half(h, A))) return rc;
    if (++h->sample == S) { h->sample = 0; h->step++; }
    return dsync(h->stream);
}
