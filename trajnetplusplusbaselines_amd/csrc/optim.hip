// Adam update of the reference's trainer (lstm/trainer.py:497: torch.optim.Adam(lr, weight_decay)), gfx950.
//
// torch's default ("foreach") implementation walks the parameter list once per elementwise operation -- nine
// multi_tensor_apply launches of 15-25 us each per optimisation step at BASELINE config 2 (4.9 M parameters) -- and reads /
// writes every state tensor several times.  Here one launch updates all tensors: a table of up to ADAM_MAX_TENSORS tensor
// descriptors travels in the kernel arguments, a block owns a 4096-element chunk of one tensor, every element is read once
// (p, g, m, v) and written once (p, m, v): 7 x 19.7 MB = 138 MB per step, HBM-bound.
// Arithmetic: the single-tensor formulas of torch.optim.Adam (amsgrad = False, maximize = False), one rounding per
// operation as the foreach kernels do (no contraction across torch's operation boundaries):
//   g' = g + wd * p;  m = m + (g' - m) * (1 - b1);  v = v * b2 + (1 - b2) * g' * g';
//   p = p - (lr / (1 - b1^t)) * (m / (sqrt(v) / sqrt(1 - b2^t) + eps))
#include "tnp_internal.h"

#include <math.h>

namespace tnp {

#define ADAM_MAX_TENSORS 24
#define ADAM_CHUNK 4096

struct AdamTable {
    float *p[ADAM_MAX_TENSORS];
    const float *g[ADAM_MAX_TENSORS];
    float *m[ADAM_MAX_TENSORS];
    float *v[ADAM_MAX_TENSORS];
    int first_block[ADAM_MAX_TENSORS + 1];   // prefix sums of the tensors' chunk counts
    long long n[ADAM_MAX_TENSORS];
    int count;
    float wd, one_minus_b1, b2, one_minus_b2, step_size, inv_bc2_sqrt_den, eps;
};

__device__ __forceinline__ void adam_one(float &p, float g, float &m, float &v, const AdamTable &t) {
    g = __fadd_rn(g, __fmul_rn(t.wd, p));                                            // _foreach_add(grads, params, alpha=wd)
    m = __fadd_rn(m, __fmul_rn(__fsub_rn(g, m), t.one_minus_b1));                    // lerp_(exp_avg, grad, 1 - beta1)
    v = __fadd_rn(__fmul_rn(v, t.b2), __fmul_rn(__fmul_rn(t.one_minus_b2, g), g));   // mul_(beta2).addcmul_(g, g, 1 - beta2)
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), t.inv_bc2_sqrt_den), t.eps);   // sqrt / bias_correction2_sqrt + eps
    p = __fadd_rn(p, __fmul_rn(t.step_size, __fdiv_rn(m, denom)));                   // addcdiv_(exp_avg, denom, value=-step_size)
}

__global__ void __launch_bounds__(256) adam_step_kernel(const AdamTable t) {
    // which tensor owns this block: the table has at most 24 entries, a linear scan of SGPR-resident prefix sums
    int ti = 0;
    while (ti + 1 < t.count && (int)blockIdx.x >= t.first_block[ti + 1]) ++ti;
    const long long base = (long long)((int)blockIdx.x - t.first_block[ti]) * ADAM_CHUNK;
    const long long n = t.n[ti];
    float *p = t.p[ti]; const float *g = t.g[ti]; float *m = t.m[ti]; float *v = t.v[ti];
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
#pragma unroll
    for (int it = 0; it < ADAM_CHUNK / (256 * 4); ++it) {
        const long long i = base + (long long)(it * 256 + threadIdx.x) * 4;
        if (i >= n) break;
        if (vec && i + 4 <= n) {
            float4 pp = *reinterpret_cast<const float4 *>(p + i), gg = *reinterpret_cast<const float4 *>(g + i);
            float4 mm = *reinterpret_cast<const float4 *>(m + i), vv = *reinterpret_cast<const float4 *>(v + i);
            adam_one(pp.x, gg.x, mm.x, vv.x, t); adam_one(pp.y, gg.y, mm.y, vv.y, t);
            adam_one(pp.z, gg.z, mm.z, vv.z, t); adam_one(pp.w, gg.w, mm.w, vv.w, t);
            *reinterpret_cast<float4 *>(p + i) = pp; *reinterpret_cast<float4 *>(m + i) = mm; *reinterpret_cast<float4 *>(v + i) = vv;
        } else {
            for (long long k = i; k < n && k < i + 4; ++k) {
                float pp = p[k], mm = m[k], vv = v[k];
                adam_one(pp, g[k], mm, vv, t);
                p[k] = pp; m[k] = mm; v[k] = vv;
            }
        }
    }
}

}  // namespace tnp

extern "C" TNP_API int tnp_adam_step(const tnp_adam_tensor *tensors, int n_tensors, int step, double lr, double beta1,
                                     double beta2, double eps, double weight_decay, void *stream) {
    using namespace tnp;
    if (n_tensors <= 0) return 0;
    if (tensors == nullptr || step < 1) TNP_FAIL(-1, "tnp_adam_step: tensors == NULL or step < 1");
    // host-side scalars exactly as torch.optim.Adam computes them (python floats = doubles, handed to the kernels as fp32)
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    int done = 0;
    while (done < n_tensors) {
        AdamTable t;
        t.count = 0;
        t.first_block[0] = 0;
        for (; done < n_tensors && t.count < ADAM_MAX_TENSORS; ++done) {
            const tnp_adam_tensor &x = tensors[done];
            if (x.n <= 0) continue;
            if (!x.param || !x.grad || !x.exp_avg || !x.exp_avg_sq) TNP_FAIL(-1, "tnp_adam_step: tensor %d has a NULL pointer", done);
            const long long chunks = (x.n + ADAM_CHUNK - 1) / ADAM_CHUNK;
            if (chunks + t.first_block[t.count] > 0x3fffffff) TNP_FAIL(-1, "tnp_adam_step: too many elements");
            t.p[t.count] = x.param; t.g[t.count] = x.grad; t.m[t.count] = x.exp_avg; t.v[t.count] = x.exp_avg_sq;
            t.n[t.count] = x.n;
            t.first_block[t.count + 1] = t.first_block[t.count] + (int)chunks;
            ++t.count;
        }
        if (t.count == 0) break;
        t.wd = (float)weight_decay; t.one_minus_b1 = (float)(1.0 - beta1); t.b2 = (float)beta2; t.one_minus_b2 = (float)(1.0 - beta2);
        t.step_size = (float)(-(lr / bc1));
        t.inv_bc2_sqrt_den = (float)sqrt(bc2);
        t.eps = (float)eps;
        hipLaunchKernelGGL(adam_step_kernel, dim3(t.first_block[t.count]), dim3(256), 0, (hipStream_t)stream, t);
        TNP_HIP(hipGetLastError());
    }
    return 0;
}
