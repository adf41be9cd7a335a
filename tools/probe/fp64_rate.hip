// Diagnostic micro-benchmark: issue cost of fp64 VALU instructions on gfx950, per wave and per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/fp64_rate.hip -o tools/probe/fp64_rate && tools/probe/fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS, int MODE>
__global__ void k(double* out, long long* cyc, int iters, double seed) {
    double x[CHAINS];
    for (int c = 0; c < CHAINS; ++c) x[c] = seed + c * 1e-3 + threadIdx.x * 1e-6;
    const double a = 0.999999, b = 1e-9;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (MODE == 0) x[c] = fma(x[c], a, b);                       // v_fma_f64, constants in registers
                else if (MODE == 1) x[c] = __builtin_amdgcn_rcp(x[c]);       // v_rcp_f64
                else if (MODE == 2) {                                        // s_mov x2 + v_fma_f64 (the BB_FMAK block)
                    double r;
                    asm volatile("s_mov_b32 s92, 0x11111111\n\ts_mov_b32 s93, 0x3f811111\n\tv_fma_f64 %0, %1, %2, s[92:93]" : "=v"(r) : "v"(x[c]), "v"(a) : "s92", "s93");
                    x[c] = r;
                } else if (MODE == 3) x[c] = ldexp(x[c], 1) * 0.5;           // v_ldexp_f64 + v_mul_f64
                else if (MODE == 4) { float f = (float)x[c]; f = fmaf(f, 0.999f, 1e-6f); x[c] = f; }   // cvt + f32 fma + cvt
                else if (MODE == 5) x[c] = x[c] + b;                          // v_add_f64
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int c = 0; c < CHAINS; ++c) s += x[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int CHAINS, int MODE>
void run(const char* name, int threads) {
    double* out; long long* cyc;
    const int blocks = 256, iters = 2000;
    hipMalloc(&out, sizeof(double) * blocks * threads);
    hipMalloc(&cyc, sizeof(long long) * blocks);
    k<CHAINS, MODE><<<blocks, threads>>>(out, cyc, iters, 1.0);
    k<CHAINS, MODE><<<blocks, threads>>>(out, cyc, iters, 1.0);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= blocks;
    const double per_wave_instr = avg / (iters * 8.0 * CHAINS);
    const int waves_per_simd = threads / 256 > 0 ? threads / 256 : 1;
    printf("%-28s chains %d waves/SIMD %d: %6.2f cycles per instr per wave, %6.2f per SIMD-instr\n", name, CHAINS, waves_per_simd, per_wave_instr,
           per_wave_instr / waves_per_simd);
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int thr : {256, 512, 1024}) {
        run<1, 0>("v_fma_f64 dependent", thr);
        run<2, 0>("v_fma_f64", thr);
        run<4, 0>("v_fma_f64", thr);
        run<8, 0>("v_fma_f64", thr);
        run<1, 2>("s_mov x2 + v_fma_f64 (asm)", thr);
        run<2, 2>("s_mov x2 + v_fma_f64 (asm)", thr);
        run<4, 2>("s_mov x2 + v_fma_f64 (asm)", thr);
        run<1, 1>("v_rcp_f64", thr);
        run<4, 1>("v_rcp_f64", thr);
        run<4, 3>("v_ldexp_f64 + v_mul_f64", thr);
        run<4, 5>("v_add_f64", thr);
        run<4, 4>("cvt + v_fma_f32 + cvt", thr);
    }
    return 0;
}
