// Issue rates of fp32 vector FMA forms on gfx950: cycles per wave-instruction per SIMD for v_fma_f32 and v_pk_fma_f32 (VGPR operands,
// and with one SGPR-pair operand as the sparse first layer uses it), eight independent chains per lane, eight waves per SIMD.
// build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/diag/fp32_rates.hip -o /tmp/fp32_rates && /tmp/fp32_rates
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(float *out, int iters, float seed, const float *sc) {
    f2 x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = f2{seed + 0.001f * (threadIdx.x + 64 * i), seed - 0.001f * threadIdx.x};
    const f2 m = f2{0.999999f, 0.9999f}, c = f2{1e-7f, 2e-7f};
    f2 sv = f2{sc[0], sc[1]};                       // uniform: lives in an SGPR pair
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) x[i].x = __builtin_fmaf(x[i].x, m.x, c.x);                               // v_fma_f32
            if (OP == 1) x[i] = __builtin_elementwise_fma(x[i], m, c);                            // v_pk_fma_f32
            if (OP == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(m), "s"(sv));   // accumulate: acc += w * s
            if (OP == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(x[i]) : "v"(m), "v"(c));
            if (OP == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(x[i].x) : "v"(m.x), "v"(c.x));
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
static double run(float *out, int iters, const float *sc) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int blocks = 256 * 8;
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, 16, 1.5f, sc);
    (void)hipEventRecord(a, 0);
    hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.5f, sc);
    (void)hipEventRecord(b, 0);
    (void)hipEventSynchronize(b);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main() {
    float *out, *sc;
    (void)hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    (void)hipMalloc(&sc, 2 * sizeof(float));
    const float h[2] = {1e-7f, 2e-7f};
    (void)hipMemcpy(sc, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 8192 * 16;
    const char *names[5] = {"v_fma_f32 (compiler)", "v_pk_fma_f32 (compiler)", "v_pk_fma_f32 v,v,s,v", "v_pk_fma_f32 v,v,v,v", "v_fma_f32 v,v,v,v"};
    double ms[5] = {run<0>(out, iters, sc), run<1>(out, iters, sc), run<2>(out, iters, sc), run<3>(out, iters, sc), run<4>(out, iters, sc)};
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const double ghz = p.clockRate / 1e6;
    for (int i = 0; i < 5; ++i) {
        const double insts = 8.0 * iters * 8;
        const double cycles = ms[i] * 1e-3 * ghz * 1e9;
        const double flops_per_inst = (i == 0 || i == 4) ? 128.0 : 256.0;
        const double tf = insts * 1024 * flops_per_inst / (ms[i] * 1e-3) / 1e12;
        printf("%-26s %8.3f ms  %6.2f cycles per wave-instruction per SIMD at %.2f GHz  = %6.1f TFLOP/s on the chip\n", names[i], ms[i],
               cycles / insts, ghz, tf);
    }
    return 0;
}
