// Diagnostic micro-benchmark (not part of the product): in-launch all-reduce protocols among the 256 resident workgroups of
// one MI355X, at the row size of the ADVI step's exchange (KK = 59 doubles per tile and step).
//   V0  what k_res runs today: rows stored sc1, drain, workgroup barrier, ready word; 16 group leaders poll their 16 members'
//       ready words, read the rows, publish a group row the same way; every tile polls 16 group words and reads 16 group rows
//   V2  the same two hops with SELF-VALIDATING rows: every double travels as one 16-byte store {lo32, tag, hi32, tag} -- two
//       8-byte granules that each carry the step's tag -- so a producer neither drains nor meets nor stores a ready word, and a
//       leader's poll of its members' rows IS the read.  V2a: every tile polls the 16 group rows whole; V2b: 16 lanes poll one
//       sentinel granule per group row, then one sweep reads (and checks) the rows
//   V1  one hop through memory-side integer atomics: every value as four 40-bit limbs of a fixed-point number, added into one
//       of R replicas of an accumulator that only ever grows (a reader keeps the previous raw value per parity), arrival
//       counters per replica; integer adds commute, so the totals are bit-reproducible
// Each step is preceded by `work` dependent fp64 FMAs per thread (+ a per-tile, per-step jitter) standing in for the S / M / G
// passes.  Prints us per step of every variant and checks every tile's totals of the last step.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/xchg_probe.hip -o gpurun_out/xchg_probe && gpurun_out/xchg_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s -> %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define KK 59
#define NG 16
#define NLIMB 4
#define SPIN_LIMIT (1u << 22)

typedef unsigned v4u __attribute__((ext_vector_type(4)));
struct Prm {
    double* prow; double* xrow; unsigned* rdy;            // V0
    v4u* grow; v4u* gxrow; v4u* grow2;                // V2, V4
    unsigned long long* acc; unsigned* cnt;               // V1: [2][R][KK * 4], [2][R] (one 128-B line each)
    double* out; unsigned* tmo; unsigned long long* stamps;
    int nsteps, nblk, work, jitter, R, epoch0;
    double* hist; int hmode, W;       // window-slot traffic beside the exchange: 0 none, 1 loads after the publish (k_res today), 2 loads before the step's work
};

__device__ __forceinline__ double rowval(int b, int step, int k) {
    const unsigned h = (unsigned)(b * 131 + k * 17 + step * 7) % 1000u;
    double v = (double)h * 1e-3 + (double)k * (k & 1 ? -0.37 : 1.25) + 1e6 * (k % 7 == 0);
    return v;
}
__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

__device__ __forceinline__ void st_sc1(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ld_sc1(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_w(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_w(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool wait_w(const unsigned* p, unsigned want, unsigned* tmo) {
    for (unsigned spins = 0; ld_w(p) != want; ++spins) {
        __builtin_amdgcn_s_sleep(1);
        if ((spins & 1023u) == 1023u && (ld_w(tmo) != 0u || spins > SPIN_LIMIT)) { st_w(tmo, 1u); return false; }
    }
    return true;
}
// 16-byte sc1 store / load of one tagged value.  The loads are inline asm (no builtin gives a 16-byte sc1 load): the compiler
// does not know they are asynchronous, so the wait that follows them takes the destination registers as operands -- nothing
// that reads them can be scheduled in front of it.
__device__ __forceinline__ void st16(v4u* p, double v, unsigned tag) {
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    v4u g = {lo, tag, hi, tag};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(g) : "memory");
}
__device__ __forceinline__ void st16_plain(v4u* p, double v, unsigned tag) {      // stays in the storing XCD's L2 (dirty): a same-XCD reader's L2 hit
    const unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    v4u g = {lo, tag, hi, tag};
    asm volatile("global_store_dwordx4 %0, %1, off" :: "v"(p), "v"(g) : "memory");
}
#define LD16(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(dst) : "v"(ptr) : "memory")
#define WAIT8(a, o) asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[o]), "+v"(a[o + 1]), "+v"(a[o + 2]), "+v"(a[o + 3]), "+v"(a[o + 4]), "+v"(a[o + 5]), "+v"(a[o + 6]), "+v"(a[o + 7]) :: "memory")
__device__ __forceinline__ void vm0() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ double unpack(v4u g) { return __hiloint2double((int)g.z, (int)g.x); }

#define HLOAD1() do { if (P.hmode == 1) { hm = *(const d2*)(P.hist + hoff); ho = *(const d2*)(P.hist + hoff + (long long)P.nblk * 2048); } } while (0)
template <int V>
__global__ void __launch_bounds__(1024) k_xchg(Prm P) {
    extern __shared__ double lds[];
    double* tot = lds;            // [KK]
    double* red = lds + 64;       // [256]
    int* ok = (int*)(lds + 400);
    const int tid = threadIdx.x, b = blockIdx.x;
    if (tid == 0) *ok = 1;
    unsigned long long prev[2] = {0ull, 0ull};
    unsigned nretry = 0;
    double x = 1.0 + tid * 1e-6;
    __syncthreads();
    long long t_x = 0;
    for (int step = 0; step < P.nsteps; ++step) {
        // ---- stand-in for the step's arithmetic
        // stand-in for the TruncatedADAGrad window slot: 32 B per thread read per step from a window far larger than the caches, and written back
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 hm = {0.0, 0.0}, ho = {0.0, 0.0};
        const long long hoff = ((long long)(step % P.W) * 2 * P.nblk + b) * 2048 + 2 * tid;
        if (P.hmode == 2) { hm = *(const d2*)(P.hist + hoff); ho = *(const d2*)(P.hist + hoff + (long long)P.nblk * 2048); }
        const int n = P.work + (P.jitter ? (int)(hash32((unsigned)(b * 7919 + step * 104729)) % (unsigned)(P.jitter + 1)) : 0);
        for (int i = 0; i < n; ++i) x = fma(x, 0.9999999, 1e-7);
        if (x == 12345.678) lds[500] = x;
        const unsigned epoch = (unsigned)P.epoch0 + (unsigned)step + 1u;
        const int par = step & 1;
        const double v = tid < KK ? rowval(b, step, tid) : 0.0;
        __syncthreads();                               // (the tile's row needs every wave's contributions: barrier 2 of the M pass)
        const long long t0 = __builtin_amdgcn_s_memtime();
        if (V == 0) {
            if (tid < KK) st_sc1(P.prow + (long long)b * KK + tid, v);
            vm0();
            __syncthreads();
            if (tid == 0) st_w(P.rdy + 32 * b, epoch);
            HLOAD1();
            if (b < NG) {
                const int members = (P.nblk - b + NG - 1) / NG;
                if (tid < members && !wait_w(P.rdy + 32 * (b + tid * NG), epoch, P.tmo)) *ok = 0;
                __syncthreads();
                if (tid < KK) {
                    double w[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) w[i] = i < members ? ld_sc1(P.prow + (long long)(b + i * NG) * KK + tid) : 0.0;
                    double s = 0.0;
#pragma unroll
                    for (int i = 0; i < 16; ++i) s += w[i];
                    st_sc1(P.xrow + ((long long)par * NG + b) * KK + tid, s);
                }
                vm0();
                __syncthreads();
                if (tid == 0) st_w(P.rdy + 32 * (P.nblk + par * NG + b), epoch);
            }
            const int half = tid >> 6, k = tid & 63;
            if (half < 2 && k < 8 && !wait_w(P.rdy + 32 * (P.nblk + par * NG + 8 * half + k), epoch, P.tmo)) *ok = 0;
            if (tid < 128) vm0();
            double s0 = 0.0;
            if (half < 2 && k < KK) {
                double w[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) w[g] = ld_sc1(P.xrow + ((long long)par * NG + 8 * half + g) * KK + k);
#pragma unroll
                for (int g = 0; g < 8; ++g) s0 += w[g];
                if (half == 1) red[k] = s0;
            }
            __syncthreads();
            if (tid < KK) tot[tid] = s0 + red[tid];
            __syncthreads();
        } else if (V == 2 || V == 3) {
            const unsigned tag = epoch;
            if (tid < KK) st16(P.grow + (long long)b * KK + tid, v, tag);
            HLOAD1();
            if (b < NG && tid < 64) {
                const int members = (P.nblk - b + NG - 1) / NG;
                const int k = tid < KK ? tid : KK - 1;
                v4u g[16];
                unsigned spins = 0;
                for (;;) {
#pragma unroll
                    for (int i = 0; i < 16; ++i) LD16(g[i], P.grow + (long long)(b + (i < members ? i : 0) * NG) * KK + k);
                    WAIT8(g, 0);
                    WAIT8(g, 8);
                    bool good = true;
#pragma unroll
                    for (int i = 0; i < 16; ++i) good = good && g[i].y == tag && g[i].w == tag;
                    if (__builtin_amdgcn_ballot_w64(!good) == 0ull) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 255u) == 255u && (ld_w(P.tmo) != 0u || spins > SPIN_LIMIT)) { st_w(P.tmo, 1u); *ok = 0; break; }
                }
                double s = 0.0;
#pragma unroll
                for (int i = 0; i < 16; ++i) s += i < members ? unpack(g[i]) : 0.0;
                if (tid < KK) st16(P.gxrow + ((long long)par * NG + b) * KK + tid, s, tag);
            }
            const int half = tid >> 6, kk = tid & 63;
            double s0 = 0.0;
            if (half < 2) {
                const int k = kk < KK ? kk : KK - 1;
                const v4u* base = P.gxrow + ((long long)par * NG + 8 * half) * KK;
                if (V == 3) {
                    // sentinel: lanes 0 .. 7 poll granule 0 of their group row
                    unsigned spins = 0;
                    for (;;) {
                        bool good = true;
                        if (kk < 8) { v4u g1; LD16(g1, base + (long long)kk * KK); asm volatile("s_waitcnt vmcnt(0)" : "+v"(g1) :: "memory"); good = g1.y == tag && g1.w == tag; }
                        if (__builtin_amdgcn_ballot_w64(!good) == 0ull) break;
                        __builtin_amdgcn_s_sleep(1);
                        if ((++spins & 255u) == 255u && (ld_w(P.tmo) != 0u || spins > SPIN_LIMIT)) { st_w(P.tmo, 1u); *ok = 0; break; }
                    }
                }
                v4u g[8];
                unsigned spins = 0;
                for (;;) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) LD16(g[i], base + (long long)i * KK + k);
                    WAIT8(g, 0);
                    bool good = true;
#pragma unroll
                    for (int i = 0; i < 8; ++i) good = good && g[i].y == tag && g[i].w == tag;
                    if (__builtin_amdgcn_ballot_w64(!good) == 0ull) break;
                    __builtin_amdgcn_s_sleep(1);
                    if ((++spins & 255u) == 255u && (ld_w(P.tmo) != 0u || spins > SPIN_LIMIT)) { st_w(P.tmo, 1u); *ok = 0; break; }
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) s0 += unpack(g[i]);
                if (half == 1 && kk < KK) red[kk] = s0;
            }
            __syncthreads();
            if (tid < KK) tot[tid] = s0 + red[tid];
            __syncthreads();
        } else if (V == 4 || V == 5 || V == 6) {
            // XCD-local first hop: tiles b with equal b % 8 are ASSUMED to share an XCD (round-robin dispatch; checked by the host from
            // XCC_ID) -- speed only: every entry is self-validating (V5: a copy stored sc1 backs the local one, the leader polls both).
            // V4 / V5: tiles -> leader b % 8 through the XCD's L2 (plain stores), then every tile polls the 8 group rows (sc1).
            // V6: three hops -- local, the 8 leaders among themselves (sc1), local broadcast of the totals row.
            const unsigned tag = epoch;
            const int xg = b & 7, nm = (P.nblk - xg + 7) / 8;          // my group, its members b = xg + 8 m
            if (tid < KK) { st16_plain(P.grow + (long long)b * KK + tid, v, tag); if (V == 5) st16(P.grow2 + (long long)b * KK + tid, v, tag); }
            if (tid < 64) {
                const int k = tid < KK ? tid : KK - 1;
                v4u g[8];
                if (b < 8) {
                    double xs = 0.0;
                    for (int m0 = 0; m0 < nm; m0 += 8) {
                        unsigned spins = 0;
                        for (;;) {
#pragma unroll
                            for (int i = 0; i < 8; ++i) LD16(g[i], ((V == 5 && (spins & 1)) ? P.grow2 : P.grow) + (long long)(xg + 8 * (m0 + i < nm ? m0 + i : m0)) * KK + k);
                            WAIT8(g, 0);
                            bool good = true;
#pragma unroll
                            for (int i = 0; i < 8; ++i) good = good && g[i].y == tag && g[i].w == tag;
                            if (__builtin_amdgcn_ballot_w64(!good) == 0ull) break;
                            __builtin_amdgcn_s_sleep(1);
                            if ((++spins & 255u) == 255u && (ld_w(P.tmo) != 0u || spins > SPIN_LIMIT)) { st_w(P.tmo, 1u); *ok = 0; break; }
                        }
                        if (tid == 0) nretry += spins;
#pragma unroll
                        for (int i = 0; i < 8; ++i) xs += m0 + i < nm ? unpack(g[i]) : 0.0;
                    }
                    if (tid < KK) st16(P.gxrow + ((long long)par * NG + b) * KK + tid, xs, tag);       // the group's row, cross-XCD
                }
                double s0 = 0.0;
                if (V != 6 || b < 8) {
                    unsigned spins = 0;
                    for (;;) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) LD16(g[i], P.gxrow + ((long long)par * NG + i) * KK + k);
                        WAIT8(g, 0);
                        bool good = true;
#pragma unroll
                        for (int i = 0; i < 8; ++i) good = good && g[i].y == tag && g[i].w == tag;
                        if (__builtin_amdgcn_ballot_w64(!good) == 0ull) break;
                        __builtin_amdgcn_s_sleep(1);
                        if ((++spins & 255u) == 255u && (ld_w(P.tmo) != 0u || spins > SPIN_LIMIT)) { st_w(P.tmo, 1u); *ok = 0; break; }
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) s0 += unpack(g[i]);
                    if (V == 6 && tid < KK) st16_plain(P.gxrow + ((long long)(2 + par) * NG + b) * KK + tid, s0, tag);   // totals, for my XCD's tiles
                }
                if (V == 6 && b >= 8) {
                    unsigned spins = 0;
                    for (;;) {
                        LD16(g[0], P.gxrow + ((long long)(2 + par) * NG + xg) * KK + k);
                        asm volatile("s_waitcnt vmcnt(0)" : "+v"(g[0]) :: "memory");
                        if (__builtin_amdgcn_ballot_w64(!(g[0].y == tag && g[0].w == tag)) == 0ull) break;
                        __builtin_amdgcn_s_sleep(1);
                        if ((++spins & 255u) == 255u && (ld_w(P.tmo) != 0u || spins > SPIN_LIMIT)) { st_w(P.tmo, 1u); *ok = 0; break; }
                    }
                    s0 = unpack(g[0]);
                }
                if (tid < KK) tot[tid] = s0;
            }
            __syncthreads();
        } else if (V == 1) {
            const int R = P.R, r = b % R;
            const int k = tid >> 2, j = tid & 3;
            // value k as fixed point with the last bit at 2^-72, limb j = bits [40 j, 40 j + 40)
            if (tid < KK * NLIMB) {
                const double vk = rowval(b, step, k);
                const unsigned long long bits = (unsigned long long)__double_as_longlong(vk);
                const int e = (int)((bits >> 52) & 0x7ff);
                unsigned long long m = (bits & 0xfffffffffffffull) | (e ? 0x10000000000000ull : 0ull);
                const int p = (e ? e : 1) - 1023 - 52 + 72;           // position of the mantissa's last bit
                const int s = 40 * j - p;
                unsigned long long piece;
                if (s >= 0) piece = s < 53 ? (m >> s) : 0ull;
                else piece = -s < 40 ? (m << -s) : 0ull;
                piece &= 0xffffffffffull;
                if ((long long)bits < 0) piece = 0ull - piece;
                if (piece) __hip_atomic_fetch_add(P.acc + ((long long)par * R + r) * (KK * NLIMB) + tid, piece, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            vm0();
            __syncthreads();
            if (tid == 0) __hip_atomic_fetch_add(P.cnt + 32 * (par * R + r), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            HLOAD1();
            if (tid < R) {
                const unsigned members = (unsigned)((P.nblk - tid + R - 1) / R);
                if (!wait_w(P.cnt + 32 * (par * R + tid), members * (unsigned)(step / 2 + 1), P.tmo)) *ok = 0;
            }
            __syncthreads();
            if (tid < KK * NLIMB) {
                unsigned long long raw = 0ull;
                for (int r0 = 0; r0 < R; r0 += 8) {
                    unsigned long long w[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        w[i] = r0 + i < R ? __hip_atomic_load(P.acc + ((long long)par * R + r0 + i) * (KK * NLIMB) + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
#pragma unroll
                    for (int i = 0; i < 8; ++i) raw += w[i];
                }
                const long long d = (long long)(raw - prev[par]);
                prev[par] = raw;
                red[tid] = ldexp((double)d, 40 * j - 72);
            }
            __syncthreads();
            if (tid < KK) tot[tid] = ((red[4 * tid + 3] + red[4 * tid + 2]) + red[4 * tid + 1]) + red[4 * tid];
            __syncthreads();
        } else {
            HLOAD1();
            if (tid < KK) tot[tid] = v;
            __syncthreads();
        }
        t_x += __builtin_amdgcn_s_memtime() - t0;
        if (*ok == 0) break;
        x += tot[tid % KK] * 1e-30;                    // the next step's work depends on the totals
        if (P.hmode) {
            x += (hm.x + ho.y) * 1e-30;
            hm.y += x * 1e-300; ho.x += x * 1e-300;
            *(d2*)(P.hist + hoff) = hm;
            *(d2*)(P.hist + hoff + (long long)P.nblk * 2048) = ho;
        }
    }
    if (tid < KK) P.out[(long long)b * KK + tid] = tot[tid];
    if (tid == 0) { P.stamps[b] = (unsigned long long)t_x; P.stamps[P.nblk + b] = ((unsigned long long)nretry << 8) | (unsigned)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15); if (x == 3.0) P.out[0] = x; }
}

template <int V>
static double run(Prm P, int nblk, const char* name, bool check) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const size_t lds = 100 * 1024;                    // one workgroup per CU, as the resident launch
    CK(hipFuncSetAttribute((const void*)k_xchg<V>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    Prm W = P;
    W.nsteps = 50;
    if (V == 1) { CK(hipMemset(P.acc, 0, (size_t)2 * 64 * KK * NLIMB * 8)); CK(hipMemset(P.cnt, 0, (size_t)2 * 64 * 32 * 4)); }
    hipLaunchKernelGGL(k_xchg<V>, dim3(nblk), dim3(1024), lds, 0, W);      // warm-up (code object, caches)
    CK(hipDeviceSynchronize());
    P.epoch0 += 1000;
    // (V1's accumulators and counters only ever grow: the timed launch starts from fresh ones)
    if (V == 1) { CK(hipMemset(P.acc, 0, (size_t)2 * 64 * KK * NLIMB * 8)); CK(hipMemset(P.cnt, 0, (size_t)2 * 64 * 32 * 4)); }
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_xchg<V>, dim3(nblk), dim3(1024), lds, 0, P);
    CK(hipEventRecord(e1, 0));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    unsigned tmo = 0;
    CK(hipMemcpy(&tmo, P.tmo, 4, hipMemcpyDeviceToHost));
    std::vector<double> out((size_t)nblk * KK);
    std::vector<unsigned long long> st((size_t)2 * nblk);
    CK(hipMemcpy(out.data(), P.out, out.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(st.data(), P.stamps, st.size() * 8, hipMemcpyDeviceToHost));
    double maxerr = 0.0;
    int diff = 0;
    if (check) {
        const int step = P.nsteps - 1;
        for (int k = 0; k < KK; ++k) {
            long double ex = 0;
            for (int b = 0; b < nblk; ++b) {
                const unsigned h = (unsigned)(b * 131 + k * 17 + step * 7) % 1000u;
                ex += (long double)((double)h * 1e-3 + (double)k * (k & 1 ? -0.37 : 1.25) + 1e6 * (k % 7 == 0));
            }
            for (int b = 0; b < nblk; ++b) {
                const double got = out[(size_t)b * KK + k];
                maxerr = fmax(maxerr, fabs((double)(got - ex)) / fmax(1.0, fabs((double)ex)));
                if (memcmp(&got, &out[k], 8)) ++diff;
            }
        }
    }
    std::vector<unsigned long long> s2(st.begin(), st.begin() + nblk);
    int misplaced = 0; unsigned long long retries = 0;
    for (int b = 0; b < nblk; ++b) { misplaced += (st[nblk + b] & 15) != (st[nblk + (b & 7)] & 15); retries += st[nblk + b] >> 8; }
    std::sort(s2.begin(), s2.end());
    const double us = ms * 1e3 / P.nsteps;
    printf("%-34s work %5d jitter %5d  %8.3f us/step   exchange cycles/step (median tile) %7.0f   %s", name, P.work, P.jitter, us,
           (double)s2[s2.size() / 2] / P.nsteps, tmo ? "TIMEOUT " : "");
    if (check) printf("max rel err %.2e, tiles differing from tile 0: %d", maxerr, diff);
    if (V == 4 || V == 5 || V == 6) printf("; tiles NOT on the XCD of tile b %% 8: %d; local polls that needed a retry (all tiles, all steps): %llu", misplaced, retries);
    printf("\n");
    fflush(stdout);
    if (tmo) { unsigned z = 0; CK(hipMemcpy(P.tmo, &z, 4, hipMemcpyHostToDevice)); }
    return us;
}

#include <algorithm>
int main(int argc, char** argv) {
    const int nblk = 256, nsteps = argc > 1 ? atoi(argv[1]) : 2000;
    Prm P;
    memset(&P, 0, sizeof P);
    CK(hipMalloc(&P.prow, (size_t)nblk * KK * 8));
    CK(hipMalloc(&P.xrow, (size_t)2 * NG * KK * 8));
    CK(hipMalloc(&P.rdy, (size_t)32 * (nblk + 2 * NG) * 4));
    CK(hipMalloc(&P.grow, (size_t)nblk * KK * 16));
    CK(hipMalloc(&P.gxrow, (size_t)4 * NG * KK * 16));
    CK(hipMalloc(&P.acc, (size_t)2 * 64 * KK * NLIMB * 8));
    CK(hipMalloc(&P.cnt, (size_t)2 * 64 * 32 * 4));
    CK(hipMalloc(&P.out, (size_t)nblk * KK * 8));
    CK(hipMalloc(&P.tmo, 128));
    CK(hipMalloc(&P.stamps, (size_t)2 * nblk * 8));
    CK(hipMalloc(&P.grow2, (size_t)nblk * KK * 16));
    CK(hipMemset(P.grow2, 0, (size_t)nblk * KK * 16));
    CK(hipMemset(P.prow, 0, (size_t)nblk * KK * 8));
    CK(hipMemset(P.xrow, 0, (size_t)2 * NG * KK * 8));
    CK(hipMemset(P.rdy, 0, (size_t)32 * (nblk + 2 * NG) * 4));
    CK(hipMemset(P.grow, 0, (size_t)nblk * KK * 16));
    CK(hipMemset(P.gxrow, 0, (size_t)4 * NG * KK * 16));
    CK(hipMemset(P.tmo, 0, 128));
    P.nsteps = nsteps;
    P.nblk = nblk;
    P.epoch0 = 1;
    P.W = 100;
    CK(hipMalloc(&P.hist, (size_t)P.W * 2 * nblk * 2048 * 8));
    CK(hipMemset(P.hist, 0, (size_t)P.W * 2 * nblk * 2048 * 8));
    const int works[5][3] = {{0, 0, 0}, {0, 0, 1}, {600, 0, 0}, {600, 0, 1}, {600, 0, 2}};
    for (auto& wj : works) {
        P.work = wj[0];
        P.jitter = wj[1];
        P.hmode = wj[2];
        printf("---- window-slot traffic mode %d\n", P.hmode);
        run<9>(P, nblk, "no exchange", false);
        P.epoch0 += 100000; run<0>(P, nblk, "V0 flags, 2 hops (today)", true);
        P.epoch0 += 100000; run<2>(P, nblk, "V2a tagged rows, all tiles poll rows", true);
        P.epoch0 += 100000; run<3>(P, nblk, "V2b tagged rows, sentinel poll", true);
        P.epoch0 += 100000; run<4>(P, nblk, "V4 XCD-local gather + 8 group rows", true);
        P.epoch0 += 100000; run<5>(P, nblk, "V5 same, sc1 backup copy polled too", true);
        P.epoch0 += 100000; run<6>(P, nblk, "V6 local, 8 leaders, local broadcast", true);
        for (int R : {16}) {
            P.R = R;
            char nm[64];
            snprintf(nm, sizeof nm, "V1 atomics, 1 hop, R = %d", R);
            run<1>(P, nblk, nm, true);
        }
    }
    return 0;
}
