// Calibration / ablation probes of the fp32 matrix pipe (measurement only; NOT part of libtrajnet_hip.so).
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -I trajnetplusplusbaselines_amd/csrc tools/experiments/mfma_probe.hip -o tools/experiments/libtnp_probe.so
#include "tnp_internal.h"
#undef TNP_FAIL
#define TNP_FAIL(code, ...) do { fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); return (code); } while (0)
#undef TNP_HIP
#define TNP_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { fprintf(stderr, "%s\n", hipGetErrorString(e_)); return -2; } } while (0)

// ---------------------------------------------------------------------------------------------------------
// Calibration probe: a pure v_mfma_f32_32x32x2_f32 stream (no memory traffic) on every SIMD of the chip.
// bench.py / tools use it to report what the matrix pipe sustains on THIS box (clock, power state) next to
// the datasheet peak, so that roofline fractions can be read against both.
// ---------------------------------------------------------------------------------------------------------
namespace tnp {
typedef float pf32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void __launch_bounds__(512) mfma_probe_kernel(float *out, int iters) {
    pf32x16 acc[NACC];
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.0f;
    float x = (float)(threadIdx.x & 7) * 0.25f, y = (float)(threadIdx.x & 3) * 0.5f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, acc[a], 0, 0, 0);
    }
    float s = 0.0f;
#pragma unroll
    for (int a = 0; a < NACC; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[a][r];
    if (s == 12345.678f) out[0] = s;  // keep the chain alive
}
}  // namespace tnp

extern "C" TNP_API int tnp_mfma_probe(int waves_per_wg, int n_acc, int iters, int blocks, float *scratch, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (waves_per_wg < 1 || waves_per_wg > 8) TNP_FAIL(-1, "waves_per_wg must be 1..8");
    dim3 grid(blocks), block(64 * waves_per_wg);
    if (n_acc == 1) hipLaunchKernelGGL(tnp::mfma_probe_kernel<1>, grid, block, 0, s, scratch, iters);
    else if (n_acc == 2) hipLaunchKernelGGL(tnp::mfma_probe_kernel<2>, grid, block, 0, s, scratch, iters);
    else if (n_acc == 4) hipLaunchKernelGGL(tnp::mfma_probe_kernel<4>, grid, block, 0, s, scratch, iters);
    else TNP_FAIL(-1, "n_acc must be 1, 2 or 4");
    TNP_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// Ablation probe: the same MFMA stream with the GEMM's side traffic added one ingredient at a time
// (bit 0: fragments via ds_read_b128, bit 1: a barrier every 16 MFMAs, bit 2: global loads + ds_write per 16).
// ---------------------------------------------------------------------------------------------------------
namespace tnp {
typedef float af32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(512) mfma_ablate_kernel(const float *src, float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float asm_[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 16384; i += blockDim.x) asm_[i] = (iters < 0) ? src[(blockIdx.x * 16384 + i) & 0xFFFFF] : (float)(i & 15) * 0.125f;
    if (iters < 0) iters = -iters;
    __syncthreads();
    pf32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    af32x4 a4 = {0.5f, 0.25f, 0.125f, 1.0f}, b4 = {1.0f, 0.5f, 0.25f, 0.125f};
    const float *lbase = asm_ + ((tid >> 6) * 1024) + (lane & 31) * 36 + (lane >> 5) * 4;
    const float *gp = src + (size_t)blockIdx.x * 8192 + tid * 4;
    af32x4 st[3];
    for (int i = 0; i < iters; ++i) {
        if (MODE & 4) {
#pragma unroll
            for (int c = 0; c < 3; ++c) st[c] = *reinterpret_cast<const af32x4 *>(gp + ((i * 3 + c) & 3) * 2048);
        }
#pragma unroll
        for (int k8 = 0; k8 < 4; ++k8) {
            if (MODE & 1) {
                a4 = *reinterpret_cast<const af32x4 *>(lbase + k8 * 8);
                b4 = *reinterpret_cast<const af32x4 *>(lbase + 4608 + k8 * 8);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q], b4[q], acc, 0, 0, 0);
        }
        if (MODE & 4) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                *reinterpret_cast<af32x4 *>(asm_ + 8192 + ((i & 1) * 4096) + (tid * 4 + c * 2048) % 4096) = st[c];
        }
        if (MODE & 2) __syncthreads();
    }
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) out[0] = s;
}
}  // namespace tnp

extern "C" TNP_API int tnp_mfma_ablate(int mode, int iters, int blocks, const float *src, float *scratch, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(blocks), block(512);
    const size_t smem = 16384 * sizeof(float);
    static bool set = false;
#define TNP_ABL(M) case M: { if (!set) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(tnp::mfma_ablate_kernel<M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); } \
        hipLaunchKernelGGL(tnp::mfma_ablate_kernel<M>, grid, block, smem, s, src, scratch, iters); break; }
    switch (mode) {
        TNP_ABL(0) TNP_ABL(1) TNP_ABL(2) TNP_ABL(3) TNP_ABL(4) TNP_ABL(5) TNP_ABL(6) TNP_ABL(7)
        default: TNP_FAIL(-1, "mode 0..7");
    }
    TNP_HIP(hipGetLastError());
    return 0;
}
