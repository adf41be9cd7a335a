// Diagnostic: relative error of the v_rcp_f64 / v_rsq_f64 hardware seeds on gfx950, and of bb_rcp / bb_sqrt after one and two
// refinement steps (bb_math.h uses two).   hipcc --offload-arch=gfx950 -O3 tools/probe/seed_accuracy.hip -o build/seed_accuracy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, double* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = x[i];
    const double r0 = __builtin_amdgcn_rcp(v);
    const double r1 = fma(fma(-v, r0, 1.0), r0, r0);
    const double r2 = fma(fma(-v, r1, 1.0), r1, r1);
    const double y0 = __builtin_amdgcn_rsq(v);
    double g = v * y0, h = 0.5 * y0;
    double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    const double s1 = fma(fma(-g, g, v), h, g);          // one coupled iteration + residual correction
    r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    const double s2 = fma(fma(-g, g, v), h, g);          // two (bb_sqrt)
    out[7 * i + 0] = r0; out[7 * i + 1] = r1; out[7 * i + 2] = r2; out[7 * i + 3] = y0; out[7 * i + 4] = s1; out[7 * i + 5] = s2; out[7 * i + 6] = v;
}
int main() {
    const int n = 1 << 22;
    std::vector<double> x(n);
    unsigned long long st = 88172645463325252ull;
    for (int i = 0; i < n; ++i) {
        st ^= st << 13; st ^= st >> 7; st ^= st << 17;
        const double u = (double)(st >> 11) * 0x1.0p-53;
        x[i] = i % 2 ? 1.0 + u : exp((u - 0.5) * 200.0);       // mantissa sweep in [1, 2) and a wide range
    }
    double *dx, *dout;
    hipMalloc(&dx, n * 8); hipMalloc(&dout, (size_t)n * 7 * 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
    std::vector<double> o((size_t)n * 7);
    hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost);
    double e[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const long double v = o[7 * i + 6], rc = 1.0L / v, rs = 1.0L / sqrtl(v), sq = sqrtl(v);
        const long double w[6] = {rc, rc, rc, rs, sq, sq};
        for (int j = 0; j < 6; ++j) e[j] = fmax(e[j], (double)fabsl(((long double)o[7 * i + j] - w[j]) / w[j]));
    }
    printf("max relative error: v_rcp_f64 %.3e (2^%.1f)  + 1 Newton %.3e  + 2 Newton %.3e | v_rsq_f64 %.3e (2^%.1f)  sqrt after 1 iteration + correction %.3e  after 2 %.3e\n",
           e[0], log2(e[0]), e[1], e[2], e[3], log2(e[3]), e[4], e[5]);
    return 0;
}
