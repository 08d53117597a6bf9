// Pointwise / gather kernels of the training backward sweep (reference: autograd through LSTM.forward,
// lstm/trainer.py:229-269).  All contractions of the sweep are tnp_linear_forward calls; what is left are the
// derivative expressions of Hidden2Normal and LSTMCell, ReLU masks and the backward of the social grid's scatter.
// HBM-bound elementwise work over [M, H] / [M, 4H] arrays; one launch each instead of ~25 eager tensor expressions.
#include "tnp_internal.h"

namespace tnp {

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }

// one wave per track: lanes stride over the H hidden units
__global__ void __launch_bounds__(64) h2n_backward_kernel(const float *__restrict__ h_out, const float *__restrict__ Wn,
                                                          const float *__restrict__ bn, const float *__restrict__ d_normal,
                                                          const float *__restrict__ d_pos, const float *__restrict__ obs1,
                                                          const float *__restrict__ obs2, const float *__restrict__ dh_in,
                                                          int M, int H, float *__restrict__ dlin,
                                                          float *__restrict__ dh_tot) {
    const int m = blockIdx.x, lane = threadIdx.x;
    if (m >= M) return;
    const float a = obs1[2 * m], b = obs2[2 * m];
    const bool present = (a == a) && (b == b);                       // lstm/lstm.py:118
    float dl[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (present) {
        // Linear output of Hidden2Normal (needed for the sigmoid derivatives): 5 dot products over H
        float lin[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        for (int k = lane; k < H; k += 64) {
            const float hv = h_out[(size_t)m * H + k];
#pragma unroll
            for (int q = 0; q < 5; ++q) lin[q] = fmaf(hv, Wn[q * H + k], lin[q]);
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) lin[q] += __shfl_xor(lin[q], off, 64);
            lin[q] += bn[q];
        }
        float dn[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            float v = d_normal ? d_normal[(size_t)m * 5 + q] : 0.0f;
            if (v != v) v = 0.0f;                                    // NaN rows of absent tracks carry no gradient
            dn[q] = v;
        }
        if (d_pos) {                                                // positions = obs2 + normal[:, :2]
            float px = d_pos[2 * m], py = d_pos[2 * m + 1];
            dn[0] += (px == px) ? px : 0.0f;
            dn[1] += (py == py) ? py : 0.0f;
        }
        const float s2 = sigm(lin[2]), s3 = sigm(lin[3]), s4 = sigm(lin[4]);
        dl[0] = dn[0]; dl[1] = dn[1];
        dl[2] = dn[2] * 0.2f * s2 * (1.0f - s2);
        dl[3] = dn[3] * 0.2f * s3 * (1.0f - s3);
        dl[4] = dn[4] * 0.7f * s4 * (1.0f - s4);
    }
    if (lane < 5) dlin[(size_t)m * 5 + lane] = dl[lane];
    for (int k = lane; k < H; k += 64) {
        float acc = dh_in[(size_t)m * H + k];
#pragma unroll
        for (int q = 0; q < 5; ++q) acc = fmaf(dl[q], Wn[q * H + k], acc);
        dh_tot[(size_t)m * H + k] = acc;
    }
}

__global__ void __launch_bounds__(256) lstm_cell_backward_kernel(const float *__restrict__ gates, const float *__restrict__ c_prev,
                                                                 const float *__restrict__ dh_tot, const float *__restrict__ dc,
                                                                 const float *__restrict__ obs1, const float *__restrict__ obs2,
                                                                 int M, int H, float *__restrict__ dG,
                                                                 float *__restrict__ dc_prev, float *__restrict__ dh_pass) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (long)M * H) return;
    const int m = (int)(q / H), k = (int)(q - (long)m * H);
    const float a = obs1[2 * m], b = obs2[2 * m];
    const bool present = (a == a) && (b == b);
    float *g = dG + (size_t)m * 4 * H + k;
    const float dh = dh_tot[q], dcv = dc[q];
    if (present) {
        const float *gs = gates + (size_t)m * 4 * H + k;
        const float gi = gs[0], gf = gs[H], gg = gs[2 * H], go = gs[3 * H];
        const float cp = c_prev[q];
        const float cn = gf * cp + gi * gg;
        const float tc = tanhf(cn);
        const float d_o = dh * tc;
        const float dct = dcv + dh * go * (1.0f - tc * tc);
        g[0] = dct * gg * gi * (1.0f - gi);
        g[H] = dct * cp * gf * (1.0f - gf);
        g[2 * H] = dct * gi * (1.0f - gg * gg);
        g[3 * H] = d_o * go * (1.0f - go);
        dc_prev[q] = dct * gf;
        dh_pass[q] = 0.0f;
    } else {  // state copied through: the gradient bypasses the cell
        g[0] = 0.0f; g[H] = 0.0f; g[2 * H] = 0.0f; g[3 * H] = 0.0f;
        dc_prev[q] = dcv;
        dh_pass[q] = dh;
    }
}

__global__ void __launch_bounds__(256) relu_mask_kernel(const float *__restrict__ dy, int ld_dy, const float *__restrict__ act,
                                                        int ld_act, int M, int N, float *__restrict__ out, int ld_out) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (long)M * N) return;
    const int m = (int)(q / N), n = (int)(q - (long)m * N);
    out[(size_t)m * ld_out + n] = act[(size_t)m * ld_act + n] > 0.0f ? dy[(size_t)m * ld_dy + n] : 0.0f;
}

// thread <-> (track j, channel ch): fixed summation order over the egos of j's scene (deterministic, no atomics)
__global__ void __launch_bounds__(256) social_scatter_backward_kernel(const float *__restrict__ dgrid, int ldg,
                                                                      const int32_t *__restrict__ cells,
                                                                      const int32_t *__restrict__ row_base,
                                                                      const int32_t *__restrict__ row_count, int M,
                                                                      int n_max, int C, int ncell,
                                                                      float *__restrict__ denc) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (long)M * C) return;
    const int j = (int)(q / C), ch = (int)(q - (long)j * C);
    const int lo = row_base[j], ns = row_count[j], jj = j - lo;
    float acc = 0.0f;
    for (int i = lo; i < lo + ns; ++i) {
        const int c = cells[(size_t)i * n_max + jj];
        if (c >= 0) acc += dgrid[(size_t)i * ldg + (size_t)ch * ncell + c];
    }
    denc[q] = acc;
}

// Backward of the directional grid's values with respect to the tracks' velocities (reference gridbased_pooling.py:
// 118-143: value(i, j) = nan_to_num(v_j - v_i) scattered to cell(i, j); autograd gives every in-range pair its cell's
// gradient and nothing through the integer cell index).  thread <-> track t:
//   dvel[t] = sum_i dgrid[i, :, cell(i,t)]  (t as neighbour of ego i)  -  sum_j dgrid[t, :, cell(t,j)]  (t as ego)
// over the pairs of t's scene whose two velocities are finite.
__global__ void __launch_bounds__(256) directional_scatter_backward_kernel(const float *__restrict__ dgrid, int ldg,
                                                                           const int32_t *__restrict__ cells,
                                                                           const int32_t *__restrict__ row_base,
                                                                           const int32_t *__restrict__ row_count,
                                                                           const float *__restrict__ obs1,
                                                                           const float *__restrict__ obs2, int M, int n_max,
                                                                           int ncell, float *__restrict__ dvel) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M) return;
    const int lo = row_base[t], ns = row_count[t], tt = t - lo;
    auto finite_vel = [&](int r) {
        const float a = obs1[2 * r], b = obs1[2 * r + 1], c = obs2[2 * r], d = obs2[2 * r + 1];
        return (a == a) && (b == b) && (c == c) && (d == d);
    };
    float ax = 0.0f, ay = 0.0f;
    if (finite_vel(t)) {
        for (int i = lo; i < lo + ns; ++i) {
            if (i == t || !finite_vel(i)) continue;
            const int c1 = cells[(size_t)i * n_max + tt];
            if (c1 >= 0) { ax += dgrid[(size_t)i * ldg + c1]; ay += dgrid[(size_t)i * ldg + ncell + c1]; }
            const int c2 = cells[(size_t)t * n_max + (i - lo)];
            if (c2 >= 0) { ax -= dgrid[(size_t)t * ldg + c2]; ay -= dgrid[(size_t)t * ldg + ncell + c2]; }
        }
    }
    dvel[2 * t] = ax;
    dvel[2 * t + 1] = ay;
}

// out[c][r] = in[r][c]: 64x64 tiles through LDS (row stride 65 floats: conflict-free on both sides), coalesced 256-byte
// reads and writes.  The weight-gradient GEMMs contract over the tracks, so both operands are needed K-major; the
// stacked per-step operands are hundreds of MB and a strided copy kernel would run far below HBM speed.
__global__ void __launch_bounds__(256) transpose_kernel(const float *__restrict__ in, int ld_in, int R, int C,
                                                        float *__restrict__ out, int ld_out) {
    __shared__ float tile[64][65];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;          // 4 rows of 64 lanes
    for (int k = ty; k < 64; k += 4) {
        const int r = r0 + k, c = c0 + tx;
        tile[k][tx] = (r < R && c < C) ? in[(size_t)r * ld_in + c] : 0.0f;
    }
    __syncthreads();
    for (int k = ty; k < 64; k += 4) {
        const int c = c0 + k, r = r0 + tx;
        if (c < C && r < R) out[(size_t)c * ld_out + r] = tile[tx][k];
    }
}

}  // namespace tnp

extern "C" TNP_API int tnp_transpose(const float *in, int ld_in, int rows, int cols, float *out, int ld_out, void *stream) {
    if (rows <= 0 || cols <= 0) return 0;
    hipLaunchKernelGGL(tnp::transpose_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream, in,
                       ld_in, rows, cols, out, ld_out);
    TNP_HIP(hipGetLastError());
    return 0;
}

extern "C" TNP_API int tnp_directional_scatter_backward(const float *dgrid, int ldg, const int32_t *cells,
                                                        const int32_t *row_base, const int32_t *row_count,
                                                        const float *obs1, const float *obs2, int M, int n_max, int ncell,
                                                        float *dvel, void *stream) {
    if (M <= 0) return 0;
    hipLaunchKernelGGL(tnp::directional_scatter_backward_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                       dgrid, ldg, cells, row_base, row_count, obs1, obs2, M, n_max, ncell, dvel);
    TNP_HIP(hipGetLastError());
    return 0;
}

extern "C" TNP_API int tnp_h2n_backward(const float *h_out, const float *Wn, const float *bn, const float *d_normal,
                                        const float *d_pos, const float *obs1, const float *obs2, const float *dh_in, int M,
                                        int H, float *dlin, float *dh_tot, void *stream) {
    if (M <= 0) return 0;
    hipLaunchKernelGGL(tnp::h2n_backward_kernel, dim3(M), dim3(64), 0, (hipStream_t)stream, h_out, Wn, bn, d_normal, d_pos,
                       obs1, obs2, dh_in, M, H, dlin, dh_tot);
    TNP_HIP(hipGetLastError());
    return 0;
}

extern "C" TNP_API int tnp_lstm_cell_backward(const float *gates, const float *c_prev, const float *dh_tot, const float *dc,
                                              const float *obs1, const float *obs2, int M, int H, float *dG,
                                              float *dc_prev, float *dh_pass, void *stream) {
    if (M <= 0) return 0;
    const long tot = (long)M * H;
    hipLaunchKernelGGL(tnp::lstm_cell_backward_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       gates, c_prev, dh_tot, dc, obs1, obs2, M, H, dG, dc_prev, dh_pass);
    TNP_HIP(hipGetLastError());
    return 0;
}

extern "C" TNP_API int tnp_relu_mask(const float *dy, int ld_dy, const float *act, int ld_act, int M, int N, float *out,
                                     int ld_out, void *stream) {
    if (M <= 0 || N <= 0) return 0;
    const long tot = (long)M * N;
    hipLaunchKernelGGL(tnp::relu_mask_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy,
                       ld_dy, act, ld_act, M, N, out, ld_out);
    TNP_HIP(hipGetLastError());
    return 0;
}

extern "C" TNP_API int tnp_social_scatter_backward(const float *dgrid, int ldg, const int32_t *cells, const int32_t *row_base,
                                                   const int32_t *row_count, int M, int n_max, int C, int ncell,
                                                   float *denc, void *stream) {
    if (M <= 0) return 0;
    const long tot = (long)M * C;
    hipLaunchKernelGGL(tnp::social_scatter_backward_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, dgrid, ldg, cells, row_base, row_count, M, n_max, C, ncell, denc);
    TNP_HIP(hipGetLastError());
    return 0;
}
