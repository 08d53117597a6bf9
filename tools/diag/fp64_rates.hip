// Issue rates of fp64 vector instructions on gfx950: cycles per wave-instruction per SIMD for fma / mul / add / rsq / rcp / sqrt
// (eight independent chains per lane, eight waves per SIMD: throughput, not latency).
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/diag/fp64_rates.hip -o /tmp/fp64_rates && /tmp/fp64_rates
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(double *out, int iters, double seed) {
    double x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = seed + 0.001 * (threadIdx.x + 64 * i);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) x[i] = __builtin_fma(x[i], 0.999999, 1e-7);
            if (OP == 1) x[i] = x[i] * 0.9999999;
            if (OP == 2) x[i] = x[i] + 1e-9;
            if (OP == 3) x[i] = __builtin_amdgcn_rsq(x[i]) + 0.5;          // v_rsq_f64 + v_add_f64
            if (OP == 4) x[i] = __builtin_amdgcn_rcp(x[i]) + 0.5;          // v_rcp_f64 + v_add_f64
            if (OP == 5) x[i] = __builtin_amdgcn_sqrt(x[i]) + 0.5;         // v_sqrt_f64 + v_add_f64
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
static double run(double *out, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    const int blocks = 256 * 8;                     // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, 16, 1.5);
    hipEventRecord(a, 0);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    double *out;
    hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
    const int iters = 4096 * 16;
    const char *names[6] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rsq_f64 (+ add)", "v_rcp_f64 (+ add)", "v_sqrt_f64 (+ add)"};
    double ms[6] = {run<0>(out, iters), run<1>(out, iters), run<2>(out, iters), run<3>(out, iters), run<4>(out, iters), run<5>(out, iters)};
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate / 1e6;
    for (int i = 0; i < 6; ++i) {
        // wave-instructions per SIMD: 8 waves x iters x 8 chains (x 2 for the ops that carry an add)
        const double insts = 8.0 * iters * 8 * (i >= 3 ? 2 : 1);
        const double cycles = ms[i] * 1e-3 * ghz * 1e9;
        printf("%-20s %8.3f ms  %6.2f cycles per wave-instruction per SIMD at %.2f GHz%s\n", names[i], ms[i], cycles / insts, ghz,
               i >= 3 ? "  (pair: transcendental + add)" : "");
    }
    return 0;
}
