// Losses of the reference's lstm/loss.py on the primaries of a batch (gfx950): PredictionLoss (2-D Gaussian NLL with
// a flat background, :6-91), L2Loss (:93-135) and CollisionLoss (:138-162).  One lane per (time step, scene);
// reductions are two-stage, fixed order (deterministic).
#include "tnp_internal.h"

namespace tnp {

// values[t*B + s] = per-element loss of primary s at step t;  mode 0 = PredictionLoss, 1 = L2 (sum of the 2 squared errors)
__global__ void loss_values_kernel(int mode, const float *inputs, const float *targets, const int32_t *scene_start,
                                   int B, int T, int M, float bg, float *values) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * B) return;
    const int t = idx / B, s = idx - t * B;
    const int m = scene_start[s];
    const float *in = inputs + ((size_t)t * M + m) * 5;
    const float *tg = targets + ((size_t)t * M + m) * 2;
    values[idx] = primary_loss_value(mode, in[0], in[1], in[2], in[3], in[4], tg[0], tg[1], bg);
}

// d(out)/d(inputs) of tnp_primary_loss_forward times the upstream gradient: one lane per (step, scene) writes the five
// derivatives of its primary (all other rows of d_inputs stay zero).  Analytic derivatives of
//   -log(0.01 + b N(x | mu, 3, 3, 0) + (0.99 - b) N(x | mu, s1, s2, rho))   with a = n1/s1, c = n2/s2, q = 1 - rho^2:
//   dlnN/dmu1 = (a - rho c)/(q s1),  dlnN/ds1 = (a^2 - rho a c)/(q s1) - 1/s1,  dlnN/drho = (a c q - rho z)/q^2 + rho/q
__global__ void loss_backward_kernel(int mode, const float *inputs, const float *targets, const int32_t *scene_start, int B,
                                     int T, int M, float bg, int per_scene, float scale, const float *grad_out,
                                     float *d_inputs) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * B) return;
    const int t = idx / B, s = idx - t * B;
    const int m = scene_start[s];
    const float *in = inputs + ((size_t)t * M + m) * 5;
    const float *tg = targets + ((size_t)t * M + m) * 2;
    const float w = per_scene ? grad_out[s] * scale / (float)T : grad_out[0] * scale / (float)(T * B);
    float d[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (mode == 0) {
        const float n1 = tg[0] - in[0], n2 = tg[1] - in[1];
        const float s1 = in[2], s2 = in[3], r = in[4];
        const float g1 = gaussian_2d_dev(in[0], in[1], 3.0f, 3.0f, 0.0f, tg[0], tg[1]);
        const float g2 = gaussian_2d_dev(in[0], in[1], s1, s2, r, tg[0], tg[1]);
        const float D = 0.01f + bg * g1 + (0.99f - bg) * g2;
        const float a = n1 / s1, c = n2 / s2, q = 1.0f - r * r;
        const float z = a * a + c * c - 2.0f * r * a * c;
        const float k1 = bg * g1 / D, k2 = (0.99f - bg) * g2 / D;
        d[0] = -(k1 * n1 / 9.0f + k2 * (a - r * c) / (q * s1));
        d[1] = -(k1 * n2 / 9.0f + k2 * (c - r * a) / (q * s2));
        d[2] = -k2 * ((a * a - r * a * c) / (q * s1) - 1.0f / s1);
        d[3] = -k2 * ((c * c - r * a * c) / (q * s2) - 1.0f / s2);
        d[4] = -k2 * ((a * c * q - r * z) / (q * q) + r / q);
    } else {
        d[0] = 2.0f * (in[0] - tg[0]);
        d[1] = 2.0f * (in[1] - tg[1]);
    }
    float *o = d_inputs + ((size_t)t * M + m) * 5;
#pragma unroll
    for (int k = 0; k < 5; ++k) o[k] = w * d[k];
}

// out[s] = scale * mean_t values[t*B+s]   (keep_batch_dim)   or   out[0] = scale * mean over everything
__global__ void __launch_bounds__(256) loss_reduce_kernel(const float *values, int B, int T, int per_scene, float scale,
                                                          float *out) {
    __shared__ float sh[256];
    const int tid = threadIdx.x;
    if (per_scene) {
        const int s = blockIdx.x;
        float a = 0.0f;
        for (int t = tid; t < T; t += 256) a += values[(size_t)t * B + s];
        sh[tid] = a;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) { if (tid < w) sh[tid] += sh[tid + w]; __syncthreads(); }
        if (tid == 0) out[s] = scale * (sh[0] / (float)T);
    } else {
        float a = 0.0f;
        for (int q = tid; q < T * B; q += 256) a += values[q];
        sh[tid] = a;
        __syncthreads();
        for (int w = 128; w > 0; w >>= 1) { if (tid < w) sh[tid] += sh[tid + w]; __syncthreads(); }
        if (tid == 0) out[0] = scale * (sh[0] / (float)(T * B));
    }
}

// CollisionLoss, lstm/loss.py:138-162: one workgroup per scene, partial[s] = col_wt * sum (1 - d/col_distance)
__global__ void __launch_bounds__(256) collision_scene_kernel(const float *pred, int ld, const int32_t *scene_start, int T,
                                                              int M, float col_wt, float col_distance, float *partial) {
    __shared__ float sh[256];
    const int s = blockIdx.x, tid = threadIdx.x;
    const int lo = scene_start[s], hi = scene_start[s + 1];
    const int nn = hi - lo - 1;
    float a = 0.0f;
    for (int q = tid; q < T * nn; q += 256) {
        const int t = q / nn, j = lo + 1 + (q - t * nn);
        const float *pp = pred + ((size_t)t * M + lo) * ld, *pn = pred + ((size_t)t * M + j) * ld;
        float px = pp[0], py = pp[1], nx = pn[0], ny = pn[1];
        if (px != px) px = -1000.0f;   // :148 NaN -> -1000 (elementwise)
        if (py != py) py = -1000.0f;
        if (nx != nx) nx = -1000.0f;
        if (ny != ny) ny = -1000.0f;
        const float dx = px - nx, dy = py - ny;
        const float d = sqrtf(dx * dx + dy * dy);
        if (d <= col_distance) a += 1.0f - d / col_distance;
    }
    sh[tid] = a;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (tid < w) sh[tid] += sh[tid + w]; __syncthreads(); }
    if (tid == 0) partial[s] = col_wt * sh[0];
}

// Backward of CollisionLoss with respect to the primaries' predicted positions (autograd through lstm/loss.py:148-161:
// the neighbours are detached, NaN coordinates were overwritten in place with -1000 so they carry no gradient, and
// torch.norm's subgradient at distance 0 is 0):
//   d loss / d p = -col_wt / col_distance * sum over colliding neighbours of (p - n) / |p - n|
// One wave per (scene, frame): lanes stride over the neighbours, fixed-order butterfly reduction (bit-reproducible).
__global__ void __launch_bounds__(64) collision_backward_kernel(const float *pred, int ld, const int32_t *scene_start, int T,
                                                                int M, float col_wt, float col_distance,
                                                                const float *grad_out, float *d_pred) {
    const int s = blockIdx.x, t = blockIdx.y, lane = threadIdx.x;
    const int lo = scene_start[s], hi = scene_start[s + 1];
    if (hi <= lo) return;
    const float *pp = pred + ((size_t)t * M + lo) * ld;
    float px = pp[0], py = pp[1];
    const bool live_x = px == px, live_y = py == py;
    if (!live_x) px = -1000.0f;
    if (!live_y) py = -1000.0f;
    float gx = 0.0f, gy = 0.0f;
    for (int j = lo + 1 + lane; j < hi; j += 64) {
        const float *pn = pred + ((size_t)t * M + j) * ld;
        float nx = pn[0], ny = pn[1];
        if (nx != nx) nx = -1000.0f;
        if (ny != ny) ny = -1000.0f;
        const float dx = px - nx, dy = py - ny;
        const float d = sqrtf(dx * dx + dy * dy);
        if (d <= col_distance && d > 0.0f) { gx += dx / d; gy += dy / d; }
    }
    for (int off = 32; off > 0; off >>= 1) { gx += __shfl_xor(gx, off); gy += __shfl_xor(gy, off); }
    if (lane == 0) {
        const float k = -grad_out[0] * col_wt / col_distance;
        float *dp = d_pred + ((size_t)t * M + lo) * ld;
        dp[0] = live_x ? k * gx : 0.0f;
        dp[1] = live_y ? k * gy : 0.0f;
    }
}

__global__ void __launch_bounds__(256) sum_kernel(const float *x, int n, float *out) {
    __shared__ float sh[256];
    const int tid = threadIdx.x;
    float a = 0.0f;
    for (int q = tid; q < n; q += 256) a += x[q];
    sh[tid] = a;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (tid < w) sh[tid] += sh[tid + w]; __syncthreads(); }
    if (tid == 0) out[0] = sh[0];
}

}  // namespace tnp

extern "C" TNP_API int tnp_primary_loss_forward(int mode, const float *inputs, const float *targets,
                                                const int32_t *scene_start, int B, int T, int M, float background_rate,
                                                int keep_batch_dim, float scale, float *values_ws, float *out,
                                                void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (mode != 0 && mode != 1) TNP_FAIL(-1, "tnp_primary_loss_forward: mode must be 0 (NLL) or 1 (L2)");
    if (B <= 0 || T <= 0) return 0;
    const int n = T * B;
    hipLaunchKernelGGL(tnp::loss_values_kernel, dim3((n + 255) / 256), dim3(256), 0, s, mode, inputs, targets, scene_start,
                       B, T, M, background_rate, values_ws);
    TNP_HIP(hipGetLastError());
    hipLaunchKernelGGL(tnp::loss_reduce_kernel, dim3(keep_batch_dim ? B : 1), dim3(256), 0, s, values_ws, B, T,
                       keep_batch_dim, scale, out);
    TNP_HIP(hipGetLastError());
    return 0;
}

// values laid out per ROW ([T][ld], the primaries' entries written by the sequence driver's fused loss): gather the
// primaries' columns into [T][B], then the same fixed-order reduction
namespace tnp {
__global__ void loss_gather_kernel(const float *rows, int ld, const int32_t *scene_start, int B, int T, float *values) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * B) return;
    const int t = idx / B, s = idx - t * B;
    values[idx] = rows[(size_t)t * ld + scene_start[s]];
}
}  // namespace tnp

extern "C" TNP_API int tnp_primary_loss_reduce(const float *row_values, int ld, const int32_t *scene_start, int B, int T,
                                               int keep_batch_dim, float scale, float *values_ws, float *out, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (B <= 0 || T <= 0) return 0;
    const int n = T * B;
    hipLaunchKernelGGL(tnp::loss_gather_kernel, dim3((n + 255) / 256), dim3(256), 0, s, row_values, ld, scene_start, B, T, values_ws);
    TNP_HIP(hipGetLastError());
    hipLaunchKernelGGL(tnp::loss_reduce_kernel, dim3(keep_batch_dim ? B : 1), dim3(256), 0, s, values_ws, B, T, keep_batch_dim,
                       scale, out);
    TNP_HIP(hipGetLastError());
    return 0;
}

extern "C" TNP_API int tnp_primary_loss_backward(int mode, const float *inputs, const float *targets,
                                                 const int32_t *scene_start, int B, int T, int M, float background_rate,
                                                 int keep_batch_dim, float scale, const float *grad_out, float *d_inputs,
                                                 void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (mode != 0 && mode != 1) TNP_FAIL(-1, "tnp_primary_loss_backward: mode must be 0 (NLL) or 1 (L2)");
    if (T <= 0 || M <= 0) return 0;
    TNP_HIP(hipMemsetAsync(d_inputs, 0, (size_t)T * M * 5 * sizeof(float), s));
    if (B <= 0) return 0;
    const int n = T * B;
    hipLaunchKernelGGL(tnp::loss_backward_kernel, dim3((n + 255) / 256), dim3(256), 0, s, mode, inputs, targets, scene_start, B,
                       T, M, background_rate, keep_batch_dim, scale, grad_out, d_inputs);
    TNP_HIP(hipGetLastError());
    return 0;
}

extern "C" TNP_API int tnp_collision_loss_forward(const float *predictions, int ld, const int32_t *scene_start, int B,
                                                  int T, int M, float col_wt, float col_distance, float *partial_ws,
                                                  float *out, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (B <= 0) return 0;
    hipLaunchKernelGGL(tnp::collision_scene_kernel, dim3(B), dim3(256), 0, s, predictions, ld, scene_start, T, M, col_wt,
                       col_distance, partial_ws);
    TNP_HIP(hipGetLastError());
    hipLaunchKernelGGL(tnp::sum_kernel, dim3(1), dim3(256), 0, s, partial_ws, B, out);
    TNP_HIP(hipGetLastError());
    return 0;
}

extern "C" TNP_API int tnp_collision_loss_backward(const float *predictions, int ld, const int32_t *scene_start, int B,
                                                   int T, int M, float col_wt, float col_distance, const float *grad_out,
                                                   float *d_predictions, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    if (T <= 0 || M <= 0) return 0;
    if (ld < 2) TNP_FAIL(-1, "tnp_collision_loss_backward: ld %d < 2", ld);
    TNP_HIP(hipMemsetAsync(d_predictions, 0, (size_t)T * M * ld * sizeof(float), s));
    if (B <= 0) return 0;
    hipLaunchKernelGGL(tnp::collision_backward_kernel, dim3(B, T), dim3(64), 0, s, predictions, ld, scene_start, T, M, col_wt,
                       col_distance, grad_out, d_predictions);
    TNP_HIP(hipGetLastError());
    return 0;
}
