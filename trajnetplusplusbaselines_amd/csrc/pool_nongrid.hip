// Non-grid interaction modules (reference lstm/non_gridbased_pooling.py), gfx950.
//
//   pool_nn_kernel         NearestNeighborMLP.forward (:98-147): per ego the n nearest other tracks of its scene
//                          (distance NaN -> 1000, ascending, first index wins exact ties), attributes
//                          [rel pos | rel vel] with NaN -> 0, Linear(C -> d) + ReLU per neighbour, concatenated.
//   pool_hiddenmlp_kernel  HiddenStateMLPPooling.forward (:196-239) up to the max-pool: per (ego, slot) incl. the
//                          ego itself ReLU(Linear) of the relative position, of the slot's hidden state and of 4x the
//                          relative velocity (fill -100 where NaN, :53-61), max over the slots.  The projection
//                          Linear(mlp_dim -> out_dim) runs on the fp32 MFMA GEMM (tnp_linear_forward).
//
// Tracks are the concatenated rows of the batch (scene s = rows scene_start[s] .. scene_start[s+1]); the padded
// slots of the reference's [B, N, *] tensors only ever contribute absent neighbours / -100 fill values, which
// never win against a present slot, so they are simply not visited.  Both kernels are latency-bound O(N^2) pair
// loops with a few flops per pair (A*(A-1) pairs per scene: 992 at config 2); HBM traffic is the positions and
// the [M, out] result.
#include "tnp_internal.h"

namespace tnp {

constexpr int NN_MAX_SEL = 8;

__global__ void __launch_bounds__(64) pool_nn_kernel(const float *__restrict__ obs1, const float *__restrict__ obs2,
                                                     const int32_t *__restrict__ scene_start, int n_sel, int in_dim,
                                                     const float *__restrict__ W, const float *__restrict__ bias, int d,
                                                     float *__restrict__ out, int ldo, float *__restrict__ attrs) {
    const int lo = scene_start[blockIdx.x], hi = scene_start[blockIdx.x + 1];
    for (int i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const float xi = obs2[2 * i], yi = obs2[2 * i + 1];
        const float vxi = xi - obs1[2 * i], vyi = yi - obs1[2 * i + 1];
        float bd[NN_MAX_SEL];
        int bj[NN_MAX_SEL];
        int cnt = 0;
        for (int j = lo; j < hi; ++j) {
            if (j == i) continue;
            const float dx = obs2[2 * j] - xi, dy = obs2[2 * j + 1] - yi;
            float dd = sqrtf(dx * dx + dy * dy);
            if (dd != dd) dd = 1000.0f;                    // absent neighbour (or absent ego): high dummy distance
            // insertion into the ascending list of the cnt <= n_sel best so far (fully unrolled: static register
            // indices); the new entry goes AFTER equal distances, so the earlier index wins exact ties
            int pos = 0;
#pragma unroll
            for (int k = 0; k < NN_MAX_SEL; ++k)
                if (k < cnt && bd[k] <= dd) pos = k + 1;
            if (pos < n_sel) {
#pragma unroll
                for (int k = NN_MAX_SEL - 1; k > 0; --k)
                    if (k > pos && k <= cnt && k < n_sel) { bd[k] = bd[k - 1]; bj[k] = bj[k - 1]; }
#pragma unroll
                for (int k = 0; k < NN_MAX_SEL; ++k)
                    if (k == pos) { bd[k] = dd; bj[k] = j; }
                if (cnt < n_sel) ++cnt;
            }
        }
        float *o = out + (size_t)i * ldo;
        for (int k = 0; k < n_sel; ++k) {
            float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            int j = -1;
#pragma unroll
            for (int q = 0; q < NN_MAX_SEL; ++q)
                if (q == k && k < cnt) j = bj[q];
            if (j >= 0) {
                float v;
                v = obs2[2 * j] - xi;                       a[0] = (v == v) ? v : 0.0f;
                v = obs2[2 * j + 1] - yi;                   a[1] = (v == v) ? v : 0.0f;
                if (in_dim == 4) {
                    v = (obs2[2 * j] - obs1[2 * j]) - vxi;          a[2] = (v == v) ? v : 0.0f;
                    v = (obs2[2 * j + 1] - obs1[2 * j + 1]) - vyi;  a[3] = (v == v) ? v : 0.0f;
                }
            }
            if (attrs)   // training: the gathered (NaN -> 0) attributes are the embedding's input in the backward pass
                for (int c = 0; c < in_dim; ++c) attrs[((size_t)i * n_sel + k) * in_dim + c] = a[c];
            for (int q = 0; q < d; ++q) {
                float acc = bias[q];
                for (int c = 0; c < in_dim; ++c) acc = fmaf(a[c], W[q * in_dim + c], acc);
                o[k * d + q] = acc > 0.0f ? acc : 0.0f;
            }
        }
    }
}

// One wave per ego (blockIdx.x = scene, blockIdx.y x 4 waves stride over its egos), lanes <-> neighbours: the distances of up
// to 64 NN_WAVE_T neighbours in registers, then n_sel rounds of a wave-level arg-min on (distance, index) -- the same order as
// pool_nn_kernel's insertion (ascending distance, the earlier index wins exact ties) -- and lanes <-> (slot, output unit) for
// the embedding.  pool_nn_kernel runs one LANE per ego (one wave per scene: 64 waves at config 2, 27 us per step).
constexpr int NN_WAVE_T = 8;             // scenes up to 512 tracks
__global__ void __launch_bounds__(256) pool_nn_wave_kernel(const float *__restrict__ obs1, const float *__restrict__ obs2,
                                                           const int32_t *__restrict__ scene_start, int n_sel, int in_dim,
                                                           const float *__restrict__ W, const float *__restrict__ bias, int d,
                                                           float *__restrict__ out, int ldo, float *__restrict__ attrs) {
    const int lo = scene_start[blockIdx.x], hi = scene_start[blockIdx.x + 1], ns = hi - lo;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = lo + blockIdx.y * 4 + wave; i < hi; i += gridDim.y * 4) {
        const float xi = obs2[2 * i], yi = obs2[2 * i + 1];
        const float vxi = xi - obs1[2 * i], vyi = yi - obs1[2 * i + 1];
        float dist[NN_WAVE_T];
#pragma unroll
        for (int t = 0; t < NN_WAVE_T; ++t) {
            const int j = lo + lane + 64 * t;
            dist[t] = INFINITY;                                  // not a candidate: beyond the scene, or the ego itself
            if (64 * t < ns) {
                const int jc = j < hi ? j : hi - 1;
                const float dx = obs2[2 * jc] - xi, dy = obs2[2 * jc + 1] - yi;
                float dd = sqrtf(dx * dx + dy * dy);
                if (dd != dd) dd = 1000.0f;                      // absent neighbour (or absent ego): high dummy distance
                if (j < hi && j != i) dist[t] = dd;
            }
        }
        int sel[NN_MAX_SEL];
#pragma unroll
        for (int k = 0; k < NN_MAX_SEL; ++k) {
            sel[k] = -1;
            if (k >= n_sel) continue;
            // this lane's best remaining candidate, then the wave's: smaller distance, then smaller index
            float bd = INFINITY;
            int bj = 0x7fffffff;
#pragma unroll
            for (int t = 0; t < NN_WAVE_T; ++t)
                if (dist[t] < bd) { bd = dist[t]; bj = lo + lane + 64 * t; }     // ascending t = ascending index: first wins ties
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const float od = __shfl_xor(bd, off, 64);
                const int oj = __shfl_xor(bj, off, 64);
                if (od < bd || (od == bd && oj < bj)) { bd = od; bj = oj; }
            }
            if (bd < INFINITY) {
                sel[k] = bj;
#pragma unroll
                for (int t = 0; t < NN_WAVE_T; ++t)
                    if (lo + lane + 64 * t == bj) dist[t] = INFINITY;            // taken
            }
        }
        // embedding: lane <-> (slot k, unit q)
        float *o = out + (size_t)i * ldo;
        for (int e = lane; e < n_sel * d; e += 64) {
            const int k = e / d, q = e - k * d;
            int j = -1;
#pragma unroll
            for (int kk = 0; kk < NN_MAX_SEL; ++kk)
                if (kk == k) j = sel[kk];
            float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            if (j >= 0) {
                float v;
                v = obs2[2 * j] - xi;                       a[0] = (v == v) ? v : 0.0f;
                v = obs2[2 * j + 1] - yi;                   a[1] = (v == v) ? v : 0.0f;
                if (in_dim == 4) {
                    v = (obs2[2 * j] - obs1[2 * j]) - vxi;          a[2] = (v == v) ? v : 0.0f;
                    v = (obs2[2 * j + 1] - obs1[2 * j + 1]) - vyi;  a[3] = (v == v) ? v : 0.0f;
                }
            }
            if (attrs && q < in_dim) attrs[((size_t)i * n_sel + k) * in_dim + q] = a[q];     // (d >= in_dim: checked by the launcher)
            float acc = bias[q];
            for (int c = 0; c < in_dim; ++c) acc = fmaf(a[c], W[q * in_dim + c], acc);
            o[k * d + q] = acc > 0.0f ? acc : 0.0f;
        }
    }
}

// blockIdx.x = scene, blockIdx.y strides over its egos; thread <-> one of the D = ms + mh + mv pooled dimensions
__global__ void __launch_bounds__(256) pool_hiddenmlp_kernel(const float *__restrict__ obs1, const float *__restrict__ obs2,
                                                             const float *__restrict__ henc, int ldh, int henc_relu,
                                                             const int32_t *__restrict__ scene_start, int ms, int mv,
                                                             int mh, const float *__restrict__ Ws,
                                                             const float *__restrict__ bs, const float *__restrict__ Wv,
                                                             const float *__restrict__ bv, float *__restrict__ pooled,
                                                             int ldp) {
    const int lo = scene_start[blockIdx.x], hi = scene_start[blockIdx.x + 1];
    const int D = ms + mh + mv;
    for (int k = threadIdx.x; k < D; k += blockDim.x) {
        const int part = k < ms ? 0 : (k < ms + mh ? 1 : 2);
        float w0 = 0.0f, w1 = 0.0f, b0 = 0.0f;
        if (part == 0) { w0 = Ws[2 * k]; w1 = Ws[2 * k + 1]; b0 = bs[k]; }
        if (part == 2) { const int q = k - ms - mh; w0 = Wv[2 * q]; w1 = Wv[2 * q + 1]; b0 = bv[q]; }
        for (int i = lo + blockIdx.y; i < hi; i += gridDim.y) {
            const float xi = obs2[2 * i], yi = obs2[2 * i + 1];
            const float vxi = xi - obs1[2 * i], vyi = yi - obs1[2 * i + 1];
            float best = -INFINITY;
            for (int j = lo; j < hi; ++j) {
                float e;
                if (part == 0) {
                    const float rx = obs2[2 * j] - xi, ry = obs2[2 * j + 1] - yi;
                    if (rx != rx || ry != ry) e = -100.0f;
                    else { e = fmaf(ry, w1, fmaf(rx, w0, b0)); e = e > 0.0f ? e : 0.0f; }
                } else if (part == 1) {
                    e = henc[(size_t)j * ldh + (k - ms)];
                    if (henc_relu) e = e > 0.0f ? e : 0.0f;
                } else {
                    const float rx = ((obs2[2 * j] - obs1[2 * j]) - vxi) * 4.0f;
                    const float ry = ((obs2[2 * j + 1] - obs1[2 * j + 1]) - vyi) * 4.0f;
                    if (rx != rx || ry != ry) e = -100.0f;
                    else { e = fmaf(ry, w1, fmaf(rx, w0, b0)); e = e > 0.0f ? e : 0.0f; }
                }
                best = e > best ? e : best;
            }
            pooled[(size_t)i * ldp + k] = best;
        }
    }
}

// ---- backward of HiddenStateMLPPooling's max-pool (training) -----------------------------------------------------
// torch.max routes the gradient of pooled[i, k] to the slot that attained the maximum; the ReLU in front of it passes
// it on only where the winner's pre-activation is positive.  One thread per pooled dimension k recomputes the forward
// of its (ego, k) pairs and writes
//   spatial / velocity parts: G[i, k'] = routed gradient, R[i, k', 0:2] = the winning slot's input (relative position /
//                             4 x relative velocity), from which the Linear(2 -> dim) gradients are column sums;
//   hidden part:              widx[i, k'] = scene-local winner (or -1: inactive), consumed by the gather kernel below.
__global__ void __launch_bounds__(256) pool_hiddenmlp_backward_kernel(const float *__restrict__ obs1, const float *__restrict__ obs2,
                                                                      const float *__restrict__ henc, int ldh,
                                                                      const int32_t *__restrict__ scene_start, int ms, int mv, int mh,
                                                                      const float *__restrict__ Ws, const float *__restrict__ bs,
                                                                      const float *__restrict__ Wv, const float *__restrict__ bv,
                                                                      const float *__restrict__ dpool, int ldp,
                                                                      float *__restrict__ G, float *__restrict__ R,
                                                                      int32_t *__restrict__ widx, int32_t *__restrict__ wslot) {
    const int lo = scene_start[blockIdx.x], hi = scene_start[blockIdx.x + 1];
    const int D = ms + mh + mv, GD = ms + mv;
    for (int k = threadIdx.x; k < D; k += blockDim.x) {
        const int part = k < ms ? 0 : (k < ms + mh ? 1 : 2);
        float w0 = 0.0f, w1 = 0.0f, b0 = 0.0f;
        if (part == 0) { w0 = Ws[2 * k]; w1 = Ws[2 * k + 1]; b0 = bs[k]; }
        if (part == 2) { const int q = k - ms - mh; w0 = Wv[2 * q]; w1 = Wv[2 * q + 1]; b0 = bv[q]; }
        for (int i = lo + blockIdx.y; i < hi; i += gridDim.y) {
            const float xi = obs2[2 * i], yi = obs2[2 * i + 1];
            const float vxi = xi - obs1[2 * i], vyi = yi - obs1[2 * i + 1];
            float best = -INFINITY, brx = 0.0f, bry = 0.0f;
            int bj = -1;
            for (int j = lo; j < hi; ++j) {
                float e, rx = 0.0f, ry = 0.0f;
                if (part == 0) {
                    rx = obs2[2 * j] - xi; ry = obs2[2 * j + 1] - yi;
                    if (rx != rx || ry != ry) e = -100.0f;
                    else { e = fmaf(ry, w1, fmaf(rx, w0, b0)); e = e > 0.0f ? e : 0.0f; }
                } else if (part == 1) {
                    e = henc[(size_t)j * ldh + (k - ms)];
                    e = e > 0.0f ? e : 0.0f;
                } else {
                    rx = ((obs2[2 * j] - obs1[2 * j]) - vxi) * 4.0f;
                    ry = ((obs2[2 * j + 1] - obs1[2 * j + 1]) - vyi) * 4.0f;
                    if (rx != rx || ry != ry) e = -100.0f;
                    else { e = fmaf(ry, w1, fmaf(rx, w0, b0)); e = e > 0.0f ? e : 0.0f; }
                }
                if (e > best) { best = e; bj = j; brx = rx; bry = ry; }   // first maximal slot, as torch.max
            }
            const bool active = best > 0.0f;     // ReLU output 0 (or the -100 fill): nothing flows to the parameters
            const float g = active ? dpool[(size_t)i * ldp + k] : 0.0f;
            if (part == 1) {
                widx[(size_t)i * mh + (k - ms)] = active ? (bj - lo) : -1;
            } else {
                const int kk = part == 0 ? k : (k - mh);
                G[(size_t)i * GD + kk] = g;
                R[((size_t)i * GD + kk) * 2] = active ? brx : 0.0f;
                R[((size_t)i * GD + kk) * 2 + 1] = active ? bry : 0.0f;
                if (wslot) wslot[(size_t)i * GD + kk] = active ? bj : -1;     // position gradients: which slot the gradient is routed to
            }
        }
    }
}

// d(hidden embedding pre-activation)[j, k'] = sum over the egos i of j's scene that picked j for dimension k' of dpool[i, ms+k']
// one wave per track j: lanes = (ego slice, k'), fixed combination order
__global__ void __launch_bounds__(256) pool_hiddenmlp_gather_kernel(const int32_t *__restrict__ widx, const float *__restrict__ dpool,
                                                                    int ldp, const int32_t *__restrict__ row_base,
                                                                    const int32_t *__restrict__ row_count, int M, int ms, int mh,
                                                                    float *__restrict__ denc) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= M) return;
    const int lo = row_base[j], ns = row_count[j], jj = j - lo;
    for (int k = lane; k < mh; k += 64) {
        float acc = 0.0f;
        // eight egos per batch: winners and gradients requested together, added in ego order (one ego at a time the loop was a
        // chain of ns round trips: 19 us per step)
        for (int i0 = lo; i0 < lo + ns; i0 += 8) {
            int w[8];
            float g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int i = i0 + u < lo + ns ? i0 + u : lo + ns - 1;
                w[u] = widx[(size_t)i * mh + k];
                g[u] = dpool[(size_t)i * ldp + ms + k];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u < lo + ns && w[u] == jj) acc += g[u];
        }
        denc[(size_t)j * mh + k] = acc;
    }
}

// dW[k, c] = sum_rows G[row, k] * R[row, k, c] (c = 0, 1), db[k] = sum_rows G[row, k]: gradients of a Linear(2 -> cols)
// whose input differs per output unit (the max-pool's winner).  Two deterministic stages: row chunks, then chunks.
// Row chunks of 256: at config 2 (38 912 rows, 64 columns) chunks of 2048 rows were 19 workgroups, each lane a chain of 512
// dependent round trips, and the second stage one more chain over the chunks -- 216 us for 30 MB.  Now 152 workgroups per column
// block with eight rows in flight per lane, and a second stage that gives every output 64 lanes (chunks strided over the
// lanes, fixed-order shuffle tree): deterministic as before.
constexpr int CS_ROWS = 256;
__global__ void __launch_bounds__(256) colsum_prod_kernel(const float *__restrict__ G, const float *__restrict__ R, long rows, int cols,
                                                          float *__restrict__ part) {
    __shared__ float red[4][64][3];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), sub = threadIdx.x >> 6;
    const long r0 = (long)blockIdx.y * CS_ROWS, r1 = min(rows, r0 + CS_ROWS);
    float a0 = 0.0f, a1 = 0.0f, ab = 0.0f;
    if (c < cols) {
        constexpr int U = 8;
        for (long rb = r0 + sub; rb < r1; rb += 4 * U) {
            float g[U], x0[U], x1[U], x2[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long r = rb + 4 * u, rc = r < r1 ? r : r1 - 1;      // past the chunk: its last row, weighted 0 below
                if (G) {
                    g[u] = G[rc * cols + c];
                    x0[u] = R[(rc * cols + c) * 2]; x1[u] = R[(rc * cols + c) * 2 + 1]; x2[u] = 0.0f;
                } else {   // R = [rows, cols, 3]: per-row sums already formed (attention: every slot contributes)
                    const float *p = R + (rc * cols + c) * 3;
                    g[u] = 1.0f; x0[u] = p[0]; x1[u] = p[1]; x2[u] = p[2];
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (rb + 4 * u >= r1) continue;
                if (G) { a0 = fmaf(g[u], x0[u], a0); a1 = fmaf(g[u], x1[u], a1); ab += g[u]; }
                else { a0 += x0[u]; a1 += x1[u]; ab += x2[u]; }
            }
        }
    }
    red[sub][threadIdx.x & 63][0] = a0; red[sub][threadIdx.x & 63][1] = a1; red[sub][threadIdx.x & 63][2] = ab;
    __syncthreads();
    if (sub == 0 && c < cols) {
        const int l = threadIdx.x;
        for (int q = 0; q < 3; ++q)
            part[((size_t)blockIdx.y * cols + c) * 3 + q] = (red[0][l][q] + red[1][l][q]) + (red[2][l][q] + red[3][l][q]);
    }
}
// one wave per output element: lane l adds chunks l, l + 64, ... in ascending order, then a fixed shuffle tree
__global__ void __launch_bounds__(256) colsum_reduce_kernel(const float *__restrict__ part, int nchunks, int cols, float *__restrict__ dW,
                                                            float *__restrict__ db) {
    const int e = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (e >= cols * 3) return;
    float acc = 0.0f;
    for (int ch = lane; ch < nchunks; ch += 64) acc += part[(size_t)ch * cols * 3 + e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
        const int c = e / 3, q = e - c * 3;
        if (q < 2) dW[c * 2 + q] = acc; else db[c] = acc;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Gradients with respect to the POSITIONS through the non-grid interaction modules (needed when the frames fed to the
// sequence carry gradient: the S-GAN discriminator scoring the generator's prediction, sgan/sgan.py:512-576 under autograd).
// Rows are the stacked steps (row r = step * M + track); row_base / row_count give a row's scene.
// ---------------------------------------------------------------------------------------------------------
// NearestNeighborMLP features (lstm/non_gridbased_pooling.py:98-147): the selection of the n nearest neighbours carries no
// gradient (indices), the gathered attributes [rel pos | rel vel] do, NaN components excluded (nan_to_num).  One wave per
// ego: the selection of pool_nn_wave_kernel recomputed (same order: ascending distance, the earlier index wins ties), then
// per slot k   ga[c] = sum_q dpre[i, k d + q] W[q, c]   masked like the attribute was.  Records: sel [R, n] (row or -1),
// ga [R, n, 4].
__global__ void __launch_bounds__(256) nn_pos_pairs_kernel(const float *__restrict__ obs1, const float *__restrict__ obs2,
                                                           const int32_t *__restrict__ row_base, const int32_t *__restrict__ row_count,
                                                           int R, int n_sel, int in_dim, const float *__restrict__ W, int d,
                                                           const float *__restrict__ dpre, int ldd, int32_t *__restrict__ sel_out,
                                                           float *__restrict__ ga_out) {
    const int lane = threadIdx.x & 63;
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= R) return;
    const int lo = row_base[i], ns = row_count[i], hi = lo + ns;
    const float xi = obs2[2 * i], yi = obs2[2 * i + 1];
    const float vxi = xi - obs1[2 * i], vyi = yi - obs1[2 * i + 1];
    float dist[NN_WAVE_T];
#pragma unroll
    for (int t = 0; t < NN_WAVE_T; ++t) {
        const int j = lo + lane + 64 * t;
        dist[t] = INFINITY;
        if (64 * t < ns) {
            const int jc = j < hi ? j : hi - 1;
            const float dx = obs2[2 * jc] - xi, dy = obs2[2 * jc + 1] - yi;
            float dd = sqrtf(dx * dx + dy * dy);
            if (dd != dd) dd = 1000.0f;
            if (j < hi && j != i) dist[t] = dd;
        }
    }
    int sel[NN_MAX_SEL];
#pragma unroll
    for (int k = 0; k < NN_MAX_SEL; ++k) {
        sel[k] = -1;
        if (k >= n_sel) continue;
        float bd = INFINITY;
        int bj = 0x7fffffff;
#pragma unroll
        for (int t = 0; t < NN_WAVE_T; ++t)
            if (dist[t] < bd) { bd = dist[t]; bj = lo + lane + 64 * t; }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float od = __shfl_xor(bd, off, 64);
            const int oj = __shfl_xor(bj, off, 64);
            if (od < bd || (od == bd && oj < bj)) { bd = od; bj = oj; }
        }
        if (bd < INFINITY) {
            sel[k] = bj;
#pragma unroll
            for (int t = 0; t < NN_WAVE_T; ++t)
                if (lo + lane + 64 * t == bj) dist[t] = INFINITY;
        }
    }
    // lane <-> (slot k, attribute c)
    for (int e = lane; e < n_sel * 4; e += 64) {
        const int k = e >> 2, c = e & 3;
        int j = -1;
#pragma unroll
        for (int kk = 0; kk < NN_MAX_SEL; ++kk)
            if (kk == k) j = sel[kk];
        float g = 0.0f;
        if (j >= 0 && c < in_dim) {
            float v;
            if (c == 0) v = obs2[2 * j] - xi;
            else if (c == 1) v = obs2[2 * j + 1] - yi;
            else if (c == 2) v = (obs2[2 * j] - obs1[2 * j]) - vxi;
            else v = (obs2[2 * j + 1] - obs1[2 * j + 1]) - vyi;
            if (v == v) {                                            // nan_to_num: a NaN attribute passes nothing
                for (int q = 0; q < d; ++q) g = fmaf(dpre[(size_t)i * ldd + k * d + q], W[q * in_dim + c], g);
            }
        }
        ga_out[((size_t)i * n_sel + k) * 4 + c] = g;
        if (c == 0) sel_out[(size_t)i * n_sel + k] = j;
    }
}

// d obs2[t] = sum over the pairs (i, k) that selected t of (ga.pos + ga.vel)  -  sum over t's own slots of the same;
// d obs1[t] = -(sum of ga.vel as neighbour) + (sum of ga.vel as ego).  One wave per row t, fixed shuffle tree.
__global__ void __launch_bounds__(256) nn_pos_gather_kernel(const int32_t *__restrict__ sel, const float *__restrict__ ga,
                                                            const int32_t *__restrict__ row_base, const int32_t *__restrict__ row_count,
                                                            int R, int n_sel, float *__restrict__ d1, float *__restrict__ d2) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= R) return;
    const int lo = row_base[t], ns = row_count[t];
    float px = 0.0f, py = 0.0f, vx = 0.0f, vy = 0.0f;
    for (int e = lane; e < ns * n_sel; e += 64) {
        const int i = lo + e / n_sel, k = e - (e / n_sel) * n_sel;
        const int j = sel[(size_t)i * n_sel + k];
        if (j < 0) continue;
        const float *g = ga + ((size_t)i * n_sel + k) * 4;
        float sgn = 0.0f;
        if (j == t) sgn += 1.0f;
        if (i == t) sgn -= 1.0f;
        if (sgn != 0.0f) { px = fmaf(sgn, g[0], px); py = fmaf(sgn, g[1], py); vx = fmaf(sgn, g[2], vx); vy = fmaf(sgn, g[3], vy); }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        px += __shfl_xor(px, off); py += __shfl_xor(py, off); vx += __shfl_xor(vx, off); vy += __shfl_xor(vy, off);
    }
    if (lane == 0) {
        d2[2 * t] = px + vx; d2[2 * t + 1] = py + vy;
        d1[2 * t] = -vx; d1[2 * t + 1] = -vy;
    }
}

// Per-pair records rec [R, n_max, 4] = (d rel pos x, y, d rel vel x, y) of (ego row i, slot j of its scene) -- AttentionMLPPooling's
// pair kernel leaves them -- gathered per row t: + as the neighbour (slot t - lo of every ego i), - as the ego (all its slots).
__global__ void __launch_bounds__(256) pair_pos_gather_kernel(const float *__restrict__ rec, const int32_t *__restrict__ row_base,
                                                              const int32_t *__restrict__ row_count, int R, int n_max,
                                                              float *__restrict__ d1, float *__restrict__ d2) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= R) return;
    const int lo = row_base[t], ns = row_count[t], tt = t - lo;
    float px = 0.0f, py = 0.0f, vx = 0.0f, vy = 0.0f;
    for (int q = lane; q < ns; q += 64) {
        const float4 a = *reinterpret_cast<const float4 *>(rec + ((size_t)(lo + q) * n_max + tt) * 4);     // ego lo + q sees t
        const float4 b = *reinterpret_cast<const float4 *>(rec + ((size_t)t * n_max + q) * 4);             // t sees slot q
        px += a.x - b.x; py += a.y - b.y; vx += a.z - b.z; vy += a.w - b.w;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        px += __shfl_xor(px, off); py += __shfl_xor(py, off); vx += __shfl_xor(vx, off); vy += __shfl_xor(vy, off);
    }
    if (lane == 0) {
        d2[2 * t] = px + vx; d2[2 * t + 1] = py + vy;
        d1[2 * t] = -vx; d1[2 * t + 1] = -vy;
    }
}

// HiddenStateMLPPooling (lstm/non_gridbased_pooling.py:196-239): the max-pool routes the gradient of pooled dimension kk to
// ONE slot (tnp_pool_hiddenmlp_backward: G [R, ms+mv] routed gradient of the pre-activation, wslot [R, ms+mv] the winning
// row or -1); its input is the relative position (spatial part) or 4 x the relative velocity (velocity part), so
//   d obs2[winner] += G W[kk, :] (x4 for the velocity part), d obs2[ego] -= the same, d obs1 the velocity part negated.
// One wave per row t, lanes over (ego of the scene, dimension).
__global__ void __launch_bounds__(256) hiddenmlp_pos_gather_kernel(const float *__restrict__ G, const int32_t *__restrict__ wslot,
                                                                   const float *__restrict__ Ws, const float *__restrict__ Wv,
                                                                   const int32_t *__restrict__ row_base, const int32_t *__restrict__ row_count,
                                                                   int R, int M_step, int ms, int mv, float *__restrict__ d1,
                                                                   float *__restrict__ d2) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= R) return;
    const int lo = row_base[t], ns = row_count[t], GD = ms + mv;
    const int off = M_step > 0 ? (t / M_step) * M_step : 0;          // winner rows are relative to their own step
    float px = 0.0f, py = 0.0f, vx = 0.0f, vy = 0.0f;
    for (int e = lane; e < ns * GD; e += 64) {
        const int i = lo + e / GD, kk = e - (e / GD) * GD;
        const int j = wslot[(size_t)i * GD + kk];
        if (j < 0) continue;
        float sgn = 0.0f;
        if (j + off == t) sgn += 1.0f;
        if (i == t) sgn -= 1.0f;
        if (sgn == 0.0f) continue;
        const float g = sgn * G[(size_t)i * GD + kk];
        if (kk < ms) { px = fmaf(g, Ws[2 * kk], px); py = fmaf(g, Ws[2 * kk + 1], py); }
        else { const int q = kk - ms; vx = fmaf(g * 4.0f, Wv[2 * q], vx); vy = fmaf(g * 4.0f, Wv[2 * q + 1], vy); }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        px += __shfl_xor(px, off); py += __shfl_xor(py, off); vx += __shfl_xor(vx, off); vy += __shfl_xor(vy, off);
    }
    if (lane == 0) {
        d2[2 * t] = px + vx; d2[2 * t + 1] = py + vy;
        d1[2 * t] = -vx; d1[2 * t + 1] = -vy;
    }
}

int launch_pool_nn(const float *obs1, const float *obs2, const int32_t *scene_start, int B, int n_sel, int in_dim,
                   const float *W, const float *bias, int d, float *out, int ldo, hipStream_t s, float *attrs, int n_max_hint) {
    if (B <= 0) return 0;
    if (n_sel < 1 || n_sel > NN_MAX_SEL) TNP_FAIL(-1, "NearestNeighborMLP: n = %d not in 1..%d", n_sel, NN_MAX_SEL);
    if (in_dim != 2 && in_dim != 4) TNP_FAIL(-1, "NearestNeighborMLP: input_dim %d not 2 or 4", in_dim);
    if (d >= in_dim && n_max_hint > 0 && n_max_hint <= 64 * NN_WAVE_T)     // 27 -> 7 us per step at config 2
        hipLaunchKernelGGL(pool_nn_wave_kernel, dim3(B, 8), dim3(256), 0, s, obs1, obs2, scene_start, n_sel, in_dim, W, bias, d, out,
                           ldo, attrs);
    else
        hipLaunchKernelGGL(pool_nn_kernel, dim3(B), dim3(64), 0, s, obs1, obs2, scene_start, n_sel, in_dim, W, bias, d, out, ldo,
                           attrs);
    TNP_HIP(hipGetLastError());
    return 0;
}

int launch_pool_hiddenmlp(const float *obs1, const float *obs2, const float *henc, int ldh, int henc_relu,
                          const int32_t *scene_start, int B, int ms, int mv, int mh, const float *Ws, const float *bs,
                          const float *Wv, const float *bv, float *pooled, int ldp, hipStream_t s) {
    if (B <= 0) return 0;
    if (ms <= 0) TNP_FAIL(-1, "HiddenStateMLPPooling: mlp_dim_spatial must be positive");
    if (mh > 0 && !henc) TNP_FAIL(-1, "HiddenStateMLPPooling: hidden embedding missing");
    const int D = ms + mh + mv;
    const int threads = D <= 64 ? 64 : (D <= 128 ? 128 : 256);
    hipLaunchKernelGGL(pool_hiddenmlp_kernel, dim3(B, 32), dim3(threads), 0, s, obs1, obs2, henc, ldh, henc_relu,
                       scene_start, ms, mv, mh, Ws, bs, Wv, bv, pooled, ldp);
    TNP_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// AttentionMLPPooling (reference :242-351).  The reference runs a single-head torch.nn.MultiheadAttention over the
// N slots of every ego's padded scene and keeps only the output at the ego's own position.  All maps around the
// softmax are linear, so they are folded on the host (tnp_lstm_model.Wx): with q_i = Wq e_ii + bq and
// u_i = [Wk^T q_i ; bk . q_i] the score of slot j is (u_i[0:D] . e_ij + u_i[D]) / sqrt(D), and because the
// attention weights sum to one the output is Wfin (sum_j a_ij e_ij) + bfin.  What is left for this file is the
// O(N^2) part: the pair embeddings e_ij, the scores, the softmax and the weighted sum of the embeddings.
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float attn_embed(int k, int ms, int mh, float fill, float xi, float yi, float vxi,
                                            float vyi, float xj, float yj, float vxj, float vyj, float henc_jk,
                                            float w0, float w1, float b0) {
    if (k < ms) {
        const float rx = xj - xi, ry = yj - yi;
        if (rx != rx || ry != ry) return fill;
        const float e = fmaf(ry, w1, fmaf(rx, w0, b0));
        return e > 0.0f ? e : 0.0f;
    }
    if (k < ms + mh) return henc_jk;
    const float rx = (vxj - vxi) * 4.0f, ry = (vyj - vyi) * 4.0f;
    if (rx != rx || ry != ry) return fill;
    const float e = fmaf(ry, w1, fmaf(rx, w0, b0));
    return e > 0.0f ? e : 0.0f;
}

__global__ void __launch_bounds__(256) pool_attn_self_kernel(const float *__restrict__ obs1, const float *__restrict__ obs2,
                                                             const float *__restrict__ henc, int ldh, int henc_relu, int M,
                                                             int ms, int mv, int mh, const float *__restrict__ bs,
                                                             const float *__restrict__ bv, float fill,
                                                             float *__restrict__ e_self, int lde) {
    const int D = ms + mh + mv;
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= M * D) return;
    const int i = q / D, k = q - i * D;
    const float xi = obs2[2 * i], yi = obs2[2 * i + 1];
    const float vxi = xi - obs1[2 * i], vyi = yi - obs1[2 * i + 1];
    float h = 0.0f, b0 = 0.0f;
    if (k < ms) b0 = bs[k];
    else if (k < ms + mh) { h = henc[(size_t)i * ldh + (k - ms)]; if (henc_relu) h = h > 0.0f ? h : 0.0f; }
    else b0 = bv[k - ms - mh];
    e_self[(size_t)i * lde + k] = attn_embed(k, ms, mh, fill, xi, yi, vxi, vyi, xi, yi, vxi, vyi, h, 0.0f, 0.0f, b0);
}

constexpr int ATT_MAXD_PER_LANE = 4;   // D <= 256
constexpr size_t ATT_MAX_LDS = (size_t)160 * 1024;   // a scene's tracks are staged in LDS (the whole CU's when needed)

// one wave per ego (blockIdx.x = scene, blockIdx.y strides over its egos); lane <-> dims lane, lane + 64, ...
__global__ void __launch_bounds__(64) pool_attn_pair_kernel(const float *__restrict__ obs1, const float *__restrict__ obs2,
                                                            const float *__restrict__ henc, int ldh, int henc_relu,
                                                            const int32_t *__restrict__ scene_start, int n_max,
                                                            const int32_t *__restrict__ scene_slots, int ms,
                                                            int mv, int mh, const float *__restrict__ Ws,
                                                            const float *__restrict__ bs, const float *__restrict__ Wv,
                                                            const float *__restrict__ bv, float fill,
                                                            const float *__restrict__ u, int ldu,
                                                            float *__restrict__ ebar, int lde) {
    extern __shared__ float att_sc[];                      // [n_max] scores of the current ego | [ns][4] x, y, vx, vy | [ns][mh]
    const int lo = scene_start[blockIdx.x], hi = scene_start[blockIdx.x + 1];
    const int D = ms + mh + mv, lane = threadIdx.x;
    const float scale = 1.0f / sqrtf((float)D);
    // the scene's tracks once into LDS: the two passes over the neighbours below were a chain of 2 x ns global round trips
    // per ego (87 us per step at config 2, one wave per SIMD)
    float *s_xy = att_sc + ((n_max + 3) & ~3), *s_h = s_xy + 4 * (hi - lo);        // (16-byte aligned: float4 reads)
    for (int j = lane; j < hi - lo; j += 64) {
        const float x = obs2[2 * (lo + j)], y = obs2[2 * (lo + j) + 1];
        s_xy[4 * j] = x; s_xy[4 * j + 1] = y; s_xy[4 * j + 2] = x - obs1[2 * (lo + j)]; s_xy[4 * j + 3] = y - obs1[2 * (lo + j) + 1];
    }
    for (int q = lane; q < (hi - lo) * mh; q += 64) {
        const int j = q / mh, k = q - j * mh;
        float h = henc[(size_t)(lo + j) * ldh + k];
        if (henc_relu) h = h > 0.0f ? h : 0.0f;
        s_h[q] = h;
    }
    __syncthreads();
    const int npad = (scene_slots ? scene_slots[blockIdx.x] : n_max) - (hi - lo);   // virtual padded slots: [fill.., 0.., fill..]
    float w0[ATT_MAXD_PER_LANE], w1[ATT_MAXD_PER_LANE], b0[ATT_MAXD_PER_LANE];
#pragma unroll
    for (int t = 0; t < ATT_MAXD_PER_LANE; ++t) {
        const int k = lane + 64 * t;
        w0[t] = w1[t] = b0[t] = 0.0f;
        if (k < ms) { w0[t] = Ws[2 * k]; w1[t] = Ws[2 * k + 1]; b0[t] = bs[k]; }
        else if (k >= ms + mh && k < D) { const int q = k - ms - mh; w0[t] = Wv[2 * q]; w1[t] = Wv[2 * q + 1]; b0[t] = bv[q]; }
    }
    for (int i = lo + blockIdx.y; i < hi; i += gridDim.y) {
        const float xi = obs2[2 * i], yi = obs2[2 * i + 1];
        const float vxi = xi - obs1[2 * i], vyi = yi - obs1[2 * i + 1];
        float ui[ATT_MAXD_PER_LANE];
#pragma unroll
        for (int t = 0; t < ATT_MAXD_PER_LANE; ++t) { const int k = lane + 64 * t; ui[t] = k < D ? u[(size_t)i * ldu + k] : 0.0f; }
        const float ci = u[(size_t)i * ldu + D];
        auto embed = [&](int j, float (&e)[ATT_MAXD_PER_LANE]) {
            const float4 pj = *reinterpret_cast<const float4 *>(s_xy + 4 * (j - lo));
            const float xj = pj.x, yj = pj.y, vxj = pj.z, vyj = pj.w;
#pragma unroll
            for (int t = 0; t < ATT_MAXD_PER_LANE; ++t) {
                const int k = lane + 64 * t;
                float h = 0.0f;
                if (k >= ms && k < ms + mh) h = s_h[(j - lo) * mh + (k - ms)];
                e[t] = k < D ? attn_embed(k, ms, mh, fill, xi, yi, vxi, vyi, xj, yj, vxj, vyj, h, w0[t], w1[t], b0[t]) : 0.0f;
            }
        };
        auto wave_sum = [&](float v) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            return v;
        };
        // pass 1: scores
        float mx = -INFINITY;
        for (int j = lo; j < hi; ++j) {
            float e[ATT_MAXD_PER_LANE];
            embed(j, e);
            float part = 0.0f;
#pragma unroll
            for (int t = 0; t < ATT_MAXD_PER_LANE; ++t) part = fmaf(ui[t], e[t], part);
            const float sj = (wave_sum(part) + ci) * scale;
            if (lane == 0) att_sc[j - lo] = sj;
            mx = fmaxf(mx, sj);
        }
        float s_pad = 0.0f;
        if (npad > 0) {
            float part = 0.0f;
#pragma unroll
            for (int t = 0; t < ATT_MAXD_PER_LANE; ++t) {
                const int k = lane + 64 * t;
                const float ep = (k < D && !(k >= ms && k < ms + mh)) ? fill : 0.0f;
                part = fmaf(ui[t], ep, part);
            }
            s_pad = (wave_sum(part) + ci) * scale;
            mx = fmaxf(mx, s_pad);
        }
        __syncthreads();
        // pass 2: softmax weights and the weighted sum of the embeddings
        float acc[ATT_MAXD_PER_LANE] = {0.0f, 0.0f, 0.0f, 0.0f};
        float den = 0.0f;
        for (int j = lo; j < hi; ++j) {
            float e[ATT_MAXD_PER_LANE];
            embed(j, e);
            const float a = expf(att_sc[j - lo] - mx);
            den += a;
#pragma unroll
            for (int t = 0; t < ATT_MAXD_PER_LANE; ++t) acc[t] = fmaf(a, e[t], acc[t]);
        }
        if (npad > 0) {
            const float a = expf(s_pad - mx) * (float)npad;
            den += a;
#pragma unroll
            for (int t = 0; t < ATT_MAXD_PER_LANE; ++t) {
                const int k = lane + 64 * t;
                const float ep = (k < D && !(k >= ms && k < ms + mh)) ? fill : 0.0f;
                acc[t] = fmaf(a, ep, acc[t]);
            }
        }
#pragma unroll
        for (int t = 0; t < ATT_MAXD_PER_LANE; ++t) {
            const int k = lane + 64 * t;
            if (k < D) ebar[(size_t)i * lde + k] = acc[t] / den;
        }
        __syncthreads();
    }
}

// ---- backward of the attention pair kernel (training) --------------------------------------------------------------
// Given debar_i = d(loss)/d(sum_j a_ij e_ij) and u_i, one wave per ego recomputes the scores and softmax weights and
// propagates:  da_j = debar . e_ij,  dscore_j = a_j (da_j - sum_l a_l da_l),  de_ij = a_j debar + dscore_j/sqrt(D) u_i[0:D],
//   du_i[0:D] = sum_j dscore_j/sqrt(D) e_ij,  du_i[D] = sum_j dscore_j/sqrt(D)   (padded slots included: constants).
// de_ij is consumed on the spot: the spatial / velocity units (Linear(2 -> dim) + ReLU on the pair's relative position /
// 4 x relative velocity) accumulate A[i, k] = (sum_j g r_x, sum_j g r_y, sum_j g) with g = de_ij[k] where the unit is
// active; the hidden-state units write dEh[i, slot j, k'] for the per-track gather.  ebar_i is written again (stacked
// operand of the folded value / output projections' weight gradients).
__global__ void __launch_bounds__(64) pool_attn_pair_backward_kernel(const float *__restrict__ obs1, const float *__restrict__ obs2,
                                                                     const float *__restrict__ henc, int ldh,
                                                                     const int32_t *__restrict__ scene_start, int n_max,
                                                                     const int32_t *__restrict__ scene_slots, int ms,
                                                                     int mv, int mh, const float *__restrict__ Ws,
                                                                     const float *__restrict__ bs, const float *__restrict__ Wv,
                                                                     const float *__restrict__ bv, float fill,
                                                                     const float *__restrict__ u, int ldu,
                                                                     const float *__restrict__ debar, int ldd,
                                                                     float *__restrict__ du, float *__restrict__ A3,
                                                                     float *__restrict__ dEh, float *__restrict__ ebar, int lde,
                                                                     float *__restrict__ pos_rec) {
    extern __shared__ float att_sc[];                      // [2][n_max]: softmax weights, then da | [ns][4] x, y, vx, vy | [ns][mh]
    const int lo = scene_start[blockIdx.x], hi = scene_start[blockIdx.x + 1], ns = hi - lo;
    float *att_a = att_sc, *att_da = att_sc + n_max;
    const int D = ms + mh + mv, GD = ms + mv, lane = threadIdx.x;
    // the scene's tracks once into LDS (see pool_attn_pair_kernel)
    float *s_xy = att_sc + ((2 * n_max + 3) & ~3), *s_h = s_xy + 4 * ns;
    for (int j = lane; j < ns; j += 64) {
        const float x = obs2[2 * (lo + j)], y = obs2[2 * (lo + j) + 1];
        s_xy[4 * j] = x; s_xy[4 * j + 1] = y; s_xy[4 * j + 2] = x - obs1[2 * (lo + j)]; s_xy[4 * j + 3] = y - obs1[2 * (lo + j) + 1];
    }
    for (int q = lane; q < ns * mh; q += 64) {
        const int j = q / mh, k = q - j * mh;
        const float h = henc[(size_t)(lo + j) * ldh + k];
        s_h[q] = h > 0.0f ? h : 0.0f;
    }
    __syncthreads();
    const float scale = 1.0f / sqrtf((float)D);
    const int npad = (scene_slots ? scene_slots[blockIdx.x] : n_max) - ns;
    float w0[ATT_MAXD_PER_LANE], w1[ATT_MAXD_PER_LANE], b0[ATT_MAXD_PER_LANE];
#pragma unroll
    for (int t = 0; t < ATT_MAXD_PER_LANE; ++t) {
        const int k = lane + 64 * t;
        w0[t] = w1[t] = b0[t] = 0.0f;
        if (k < ms) { w0[t] = Ws[2 * k]; w1[t] = Ws[2 * k + 1]; b0[t] = bs[k]; }
        else if (k >= ms + mh && k < D) { const int q = k - ms - mh; w0[t] = Wv[2 * q]; w1[t] = Wv[2 * q + 1]; b0[t] = bv[q]; }
    }
    auto wave_sum = [&](float v) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        return v;
    };
    for (int i = lo + blockIdx.y; i < hi; i += gridDim.y) {
        const float xi = obs2[2 * i], yi = obs2[2 * i + 1];
        const float vxi = xi - obs1[2 * i], vyi = yi - obs1[2 * i + 1];
        float ui[ATT_MAXD_PER_LANE], db[ATT_MAXD_PER_LANE], ep[ATT_MAXD_PER_LANE];
#pragma unroll
        for (int t = 0; t < ATT_MAXD_PER_LANE; ++t) {
            const int k = lane + 64 * t;
            ui[t] = k < D ? u[(size_t)i * ldu + k] : 0.0f;
            db[t] = k < D ? debar[(size_t)i * ldd + k] : 0.0f;
            ep[t] = (k < D && !(k >= ms && k < ms + mh)) ? fill : 0.0f;     // embedding of a padded slot
        }
        const float ci = u[(size_t)i * ldu + D];
        // embedding of slot j and, for the spatial / velocity units, the inputs and whether the unit passes a gradient
        auto embed = [&](int j, float (&e)[ATT_MAXD_PER_LANE], float (&rx)[ATT_MAXD_PER_LANE], float (&ry)[ATT_MAXD_PER_LANE]) {
            const float4 pj = *reinterpret_cast<const float4 *>(s_xy + 4 * (j - lo));
            const float xj = pj.x, yj = pj.y, vxj = pj.z, vyj = pj.w;
#pragma unroll
            for (int t = 0; t < ATT_MAXD_PER_LANE; ++t) {
                const int k = lane + 64 * t;
                float h = 0.0f;
                rx[t] = 0.0f; ry[t] = 0.0f;
                if (k >= ms && k < ms + mh) h = s_h[(j - lo) * mh + (k - ms)];
                else if (k < ms) { rx[t] = xj - xi; ry[t] = yj - yi; }
                else if (k < D) { rx[t] = (vxj - vxi) * 4.0f; ry[t] = (vyj - vyi) * 4.0f; }
                e[t] = k < D ? attn_embed(k, ms, mh, fill, xi, yi, vxi, vyi, xj, yj, vxj, vyj, h, w0[t], w1[t], b0[t]) : 0.0f;
            }
        };
        // pass 1: scores -> softmax weights
        float mx = -INFINITY;
        for (int j = lo; j < hi; ++j) {
            float e[ATT_MAXD_PER_LANE], rx[ATT_MAXD_PER_LANE], ry[ATT_MAXD_PER_LANE];
            embed(j, e, rx, ry);
            float part = 0.0f, pd = 0.0f;
#pragma unroll
            for (int t = 0; t < ATT_MAXD_PER_LANE; ++t) { part = fmaf(ui[t], e[t], part); pd = fmaf(db[t], e[t], pd); }
            const float sj = (wave_sum(part) + ci) * scale;
            const float daj = wave_sum(pd);
            if (lane == 0) { att_a[j - lo] = sj; att_da[j - lo] = daj; }
            mx = fmaxf(mx, sj);
        }
        float s_pad = 0.0f, da_pad = 0.0f;
        if (npad > 0) {
            float part = 0.0f, pd = 0.0f;
#pragma unroll
            for (int t = 0; t < ATT_MAXD_PER_LANE; ++t) { part = fmaf(ui[t], ep[t], part); pd = fmaf(db[t], ep[t], pd); }
            s_pad = (wave_sum(part) + ci) * scale;
            da_pad = wave_sum(pd);
            mx = fmaxf(mx, s_pad);
        }
        __syncthreads();
        float den = 0.0f, dot = 0.0f;
        for (int j = 0; j < ns; ++j) { const float a = expf(att_a[j] - mx); den += a; dot = fmaf(a, att_da[j], dot); }
        float a_pad = 0.0f;
        if (npad > 0) { a_pad = expf(s_pad - mx); den += a_pad * (float)npad; dot = fmaf(a_pad * (float)npad, da_pad, dot); }
        const float inv = 1.0f / den;
        dot *= inv;                                         // sum_l a_l da_l
        // pass 2: propagate
        float dul[ATT_MAXD_PER_LANE] = {0.0f, 0.0f, 0.0f, 0.0f}, eb[ATT_MAXD_PER_LANE] = {0.0f, 0.0f, 0.0f, 0.0f};
        float a0[ATT_MAXD_PER_LANE] = {0.0f, 0.0f, 0.0f, 0.0f}, a1[ATT_MAXD_PER_LANE] = {0.0f, 0.0f, 0.0f, 0.0f};
        float ab[ATT_MAXD_PER_LANE] = {0.0f, 0.0f, 0.0f, 0.0f};
        float du_c = 0.0f;
        for (int j = lo; j < hi; ++j) {
            float e[ATT_MAXD_PER_LANE], rx[ATT_MAXD_PER_LANE], ry[ATT_MAXD_PER_LANE];
            embed(j, e, rx, ry);
            const float a = expf(att_a[j - lo] - mx) * inv;
            const float ds = a * (att_da[j - lo] - dot) * scale;
            du_c += ds;
#pragma unroll
            for (int t = 0; t < ATT_MAXD_PER_LANE; ++t) {
                const int k = lane + 64 * t;
                if (k >= D) continue;
                eb[t] = fmaf(a, e[t], eb[t]);
                dul[t] = fmaf(ds, e[t], dul[t]);
                const float de = fmaf(a, db[t], ds * ui[t]);
                if (k >= ms && k < ms + mh) {
                    dEh[((size_t)i * n_max + (j - lo)) * mh + (k - ms)] = de;
                } else if (e[t] > 0.0f && rx[t] == rx[t] && ry[t] == ry[t]) {   // active unit of a present pair (fill <= 0)
                    a0[t] = fmaf(de, rx[t], a0[t]);
                    a1[t] = fmaf(de, ry[t], a1[t]);
                    ab[t] += de;
                }
            }
            if (pos_rec) {
                // position gradients (the frames carry gradient: S-GAN discriminator in a generator step): the pair's
                // embedding inputs are the relative position (spatial units) and 4 x the relative velocity (velocity units)
                float gpx = 0.0f, gpy = 0.0f, gvx = 0.0f, gvy = 0.0f;
#pragma unroll
                for (int t = 0; t < ATT_MAXD_PER_LANE; ++t) {
                    const int k = lane + 64 * t;
                    if (k >= D || (k >= ms && k < ms + mh)) continue;
                    if (!(e[t] > 0.0f && rx[t] == rx[t] && ry[t] == ry[t])) continue;
                    const float de = fmaf(a, db[t], ds * ui[t]);
                    if (k < ms) { gpx = fmaf(de, w0[t], gpx); gpy = fmaf(de, w1[t], gpy); }
                    else { gvx = fmaf(de * 4.0f, w0[t], gvx); gvy = fmaf(de * 4.0f, w1[t], gvy); }
                }
                gpx = wave_sum(gpx); gpy = wave_sum(gpy); gvx = wave_sum(gvx); gvy = wave_sum(gvy);
                if (lane == 0)
                    *reinterpret_cast<float4 *>(pos_rec + ((size_t)i * n_max + (j - lo)) * 4) = make_float4(gpx, gpy, gvx, gvy);
            }
        }
        if (npad > 0) {
            const float a = a_pad * inv;
            const float ds = a * (da_pad - dot) * scale * (float)npad;
            du_c += ds;
#pragma unroll
            for (int t = 0; t < ATT_MAXD_PER_LANE; ++t) { eb[t] = fmaf(a * (float)npad, ep[t], eb[t]); dul[t] = fmaf(ds, ep[t], dul[t]); }
        }
#pragma unroll
        for (int t = 0; t < ATT_MAXD_PER_LANE; ++t) {
            const int k = lane + 64 * t;
            if (k >= D) continue;
            du[(size_t)i * ldu + k] = dul[t];
            ebar[(size_t)i * lde + k] = eb[t];
            if (!(k >= ms && k < ms + mh)) {
                const int kk = k < ms ? k : k - mh;
                float *o = A3 + ((size_t)i * GD + kk) * 3;
                o[0] = a0[t]; o[1] = a1[t]; o[2] = ab[t];
            }
        }
        if (lane == 0) {
            du[(size_t)i * ldu + D] = du_c;
            for (int q = D + 1; q < ldu; ++q) du[(size_t)i * ldu + q] = 0.0f;
        }
        __syncthreads();
    }
}

// the ego's own slot also feeds the query: de_self = Wq^T dq is routed through e_ii (relative position / velocity 0 ->
// only the biases of the spatial / velocity units, and the ego's own hidden embedding)
__global__ void __launch_bounds__(256) pool_attn_self_backward_kernel(const float *__restrict__ obs1, const float *__restrict__ obs2,
                                                                      int M, int ms, int mv, int mh, const float *__restrict__ bs,
                                                                      const float *__restrict__ bv, const float *__restrict__ de_self,
                                                                      int ldd, float *__restrict__ A3, float *__restrict__ dself_h) {
    const int D = ms + mh + mv, GD = ms + mv;
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (long)M * D) return;
    const int i = (int)(q / D), k = (int)(q - (long)i * D);
    const float g = de_self[(size_t)i * ldd + k];
    if (k >= ms && k < ms + mh) { dself_h[(size_t)i * mh + (k - ms)] = g; return; }
    const float xi = obs2[2 * i], yi = obs2[2 * i + 1];
    const float vxi = xi - obs1[2 * i], vyi = yi - obs1[2 * i + 1];
    bool ok;
    float b;
    if (k < ms) { ok = (xi == xi) && (yi == yi); b = bs[k]; }
    else { ok = (vxi == vxi) && (vyi == vyi); b = bv[k - ms - mh]; }
    if (ok && b > 0.0f) A3[((size_t)i * GD + (k < ms ? k : k - mh)) * 3 + 2] += g;
}

// d(hidden embedding pre-activation)[j, k'] = relu'( henc_pre[j, k'] ) * ( dself_h[j, k'] + sum over the egos i of j's
// scene of dEh[i, slot of j, k'] ); one wave per track, ascending ego order
__global__ void __launch_bounds__(256) pool_attn_gather_kernel(const float *__restrict__ dEh, const float *__restrict__ dself_h,
                                                               const float *__restrict__ henc_pre, int ldh,
                                                               const int32_t *__restrict__ row_base, const int32_t *__restrict__ row_count,
                                                               int M, int n_max, int mh, float *__restrict__ denc) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= M) return;
    const int lo = row_base[j], ns = row_count[j], jj = j - lo;
    for (int k = lane; k < mh; k += 64) {
        float acc = dself_h[(size_t)j * mh + k];
        const float pre = henc_pre[(size_t)j * ldh + k];
        for (int i0 = lo; i0 < lo + ns; i0 += 8) {          // eight egos per batch in flight, added in ascending ego order
            float g[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) g[u] = dEh[((size_t)(i0 + u < lo + ns ? i0 + u : lo + ns - 1) * n_max + jj) * mh + k];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (i0 + u < lo + ns) acc += g[u];
        }
        denc[(size_t)j * mh + k] = pre > 0.0f ? acc : 0.0f;
    }
}

int launch_pool_attn_self(const float *obs1, const float *obs2, const float *henc, int ldh, int henc_relu, int M, int ms,
                          int mv, int mh, const float *bs, const float *bv, float fill, float *e_self, int lde,
                          hipStream_t s) {
    if (M <= 0) return 0;
    const int D = ms + mh + mv;
    const long total = (long)M * D;
    hipLaunchKernelGGL(pool_attn_self_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, obs1, obs2, henc, ldh,
                       henc_relu, M, ms, mv, mh, bs, bv, fill, e_self, lde);
    TNP_HIP(hipGetLastError());
    return 0;
}

int launch_pool_attn_pair(const float *obs1, const float *obs2, const float *henc, int ldh, int henc_relu,
                          const int32_t *scene_start, int B, int n_max, const int32_t *scene_slots, int ms, int mv, int mh,
                          const float *Ws, const float *bs, const float *Wv, const float *bv, float fill, const float *u,
                          int ldu, float *ebar, int lde, hipStream_t s) {
    if (B <= 0) return 0;
    const int D = ms + mh + mv;
    if (D > 64 * ATT_MAXD_PER_LANE) TNP_FAIL(-1, "AttentionMLPPooling: mlp_dim %d > %d", D, 64 * ATT_MAXD_PER_LANE);
    const size_t lds = ((size_t)((n_max + 3) & ~3) + (size_t)n_max * (4 + (mh > 0 ? mh : 0))) * sizeof(float);   // scores + the scene's tracks (n_scene <= n_max)
    // (a scene's tracks are staged in LDS: up to the CU's whole 160 KB, i.e. ~400 tracks per scene at mlp_dim_hidden = 96)
    if (n_max < 1 || lds > ATT_MAX_LDS) TNP_FAIL(-1, "AttentionMLPPooling: a scene of %d slots (x mlp_dim_hidden %d) needs %zu bytes of LDS, limit %zu", n_max, mh, lds, ATT_MAX_LDS);
    if (lds > 60000) {
        static bool raised = false;     // idempotent: a race only repeats the call
        if (!raised) { TNP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(pool_attn_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATT_MAX_LDS)); raised = true; }
    }
    // one wave per ego up to 64 egos per scene (two egos per wave left one wave per SIMD: 59 -> 35 us at config 2)
    hipLaunchKernelGGL(pool_attn_pair_kernel, dim3(B, n_max < 64 ? n_max : 64), dim3(64), lds, s, obs1, obs2, henc, ldh,
                       henc_relu, scene_start, n_max, scene_slots, ms, mv, mh, Ws, bs, Wv, bv, fill, u, ldu, ebar, lde);
    TNP_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// TrajectronPooling features (reference :513-529): per visible track [pos, vel] ++ the sum of [pos, vel] over all
// OTHER visible tracks of the WHOLE batch (the reference's one_cold loop is not per scene), Linear(8 -> P) + ReLU;
// invisible tracks get a zero row.  The batch total is one deterministic fp64 tree reduction by a single workgroup;
// "total - own" in fp64 is then closer to the exact sum than the reference's own fp32 summation.
// ---------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) traj_total_kernel(const float *__restrict__ obs1, const float *__restrict__ obs2, int M,
                                                         double *__restrict__ total) {
    __shared__ double red[256][4];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < M; i += 256) {
        const float x = obs2[2 * i], y = obs2[2 * i + 1];
        const float vx = x - obs1[2 * i], vy = y - obs1[2 * i + 1];
        if (x == x && y == y && vx == vx && vy == vy) { acc[0] += x; acc[1] += y; acc[2] += vx; acc[3] += vy; }
    }
    for (int q = 0; q < 4; ++q) red[threadIdx.x][q] = acc[q];
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {
        if (threadIdx.x < d)
            for (int q = 0; q < 4; ++q) red[threadIdx.x][q] += red[threadIdx.x + d][q];
        __syncthreads();
    }
    if (threadIdx.x < 4) total[threadIdx.x] = red[0][threadIdx.x];
}

__global__ void __launch_bounds__(256) traj_feature_kernel(const float *__restrict__ obs1, const float *__restrict__ obs2,
                                                           int M, const double *__restrict__ total,
                                                           const float *__restrict__ W, const float *__restrict__ bias,
                                                           int P, float *__restrict__ out, int ldo,
                                                           float *__restrict__ inputs) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= M * P) return;
    const int i = q / P, o = q - i * P;
    const float x = obs2[2 * i], y = obs2[2 * i + 1];
    const float vx = x - obs1[2 * i], vy = y - obs1[2 * i + 1];
    float r = 0.0f;
    float in[8] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (x == x && y == y && vx == vx && vy == vy) {
        in[0] = x; in[1] = y; in[2] = vx; in[3] = vy;
        in[4] = (float)(total[0] - (double)x); in[5] = (float)(total[1] - (double)y);
        in[6] = (float)(total[2] - (double)vx); in[7] = (float)(total[3] - (double)vy);
        float acc = bias[o];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc = fmaf(in[c], W[o * 8 + c], acc);
        r = acc > 0.0f ? acc : 0.0f;
    }
    out[(size_t)i * ldo + o] = r;
    if (inputs && o == 0)   // training: the embedding's input rows (zero for invisible tracks, whose feature row is zero)
        for (int c = 0; c < 8; ++c) inputs[(size_t)i * 8 + c] = in[c];
}

int launch_pool_traj(const float *obs1, const float *obs2, int M, const float *W, const float *bias, int P, float *out,
                     int ldo, double *scratch4, hipStream_t s, float *inputs) {
    if (M <= 0) return 0;
    if (!scratch4) TNP_FAIL(-1, "TrajectronPooling: scratch (4 doubles) missing");
    hipLaunchKernelGGL(traj_total_kernel, dim3(1), dim3(256), 0, s, obs1, obs2, M, scratch4);
    const long tot = (long)M * P;
    hipLaunchKernelGGL(traj_feature_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, obs1, obs2, M, scratch4, W,
                       bias, P, out, ldo, inputs);
    TNP_HIP(hipGetLastError());
    return 0;
}

}  // namespace tnp

extern "C" TNP_API int tnp_pool_nn_forward(const float *obs1, const float *obs2, const int32_t *scene_start, int B,
                                           int n_sel, int in_dim, const float *W, const float *bias, int d,
                                           float *out, int ldo, void *stream) {
    return tnp::launch_pool_nn(obs1, obs2, scene_start, B, n_sel, in_dim, W, bias, d, out, ldo, (hipStream_t)stream, nullptr);
}

extern "C" TNP_API int tnp_pool_hiddenmlp_forward(const float *obs1, const float *obs2, const float *hidden_emb, int ldh,
                                                  int hidden_emb_relu, const int32_t *scene_start, int B, int ms,
                                                  int mv, int mh, const float *W_spatial, const float *b_spatial,
                                                  const float *W_vel, const float *b_vel, float *pooled, int ldp,
                                                  void *stream) {
    return tnp::launch_pool_hiddenmlp(obs1, obs2, hidden_emb, ldh, hidden_emb_relu, scene_start, B, ms, mv, mh,
                                      W_spatial, b_spatial, W_vel, b_vel, pooled, ldp, (hipStream_t)stream);
}

extern "C" TNP_API int tnp_pool_attn_self(const float *obs1, const float *obs2, const float *hidden_emb, int ldh,
                                          int hidden_emb_relu, int M, int ms, int mv, int mh, const float *b_spatial,
                                          const float *b_vel, float fill, float *e_self, int lde, void *stream) {
    return tnp::launch_pool_attn_self(obs1, obs2, hidden_emb, ldh, hidden_emb_relu, M, ms, mv, mh, b_spatial, b_vel, fill,
                                      e_self, lde, (hipStream_t)stream);
}

extern "C" TNP_API int tnp_pool_attn_pair(const float *obs1, const float *obs2, const float *hidden_emb, int ldh,
                                          int hidden_emb_relu, const int32_t *scene_start, int B, int n_max,
                                          const int32_t *scene_slots, int ms, int mv, int mh, const float *W_spatial,
                                          const float *b_spatial,
                                          const float *W_vel, const float *b_vel, float fill, const float *u, int ldu,
                                          float *ebar, int lde, void *stream) {
    return tnp::launch_pool_attn_pair(obs1, obs2, hidden_emb, ldh, hidden_emb_relu, scene_start, B, n_max, scene_slots, ms, mv, mh,
                                      W_spatial, b_spatial, W_vel, b_vel, fill, u, ldu, ebar, lde, (hipStream_t)stream);
}

extern "C" TNP_API int tnp_pool_traj_forward(const float *obs1, const float *obs2, int M, const float *W, const float *bias,
                                             int P, float *out, int ldo, double *scratch4, void *stream) {
    return tnp::launch_pool_traj(obs1, obs2, M, W, bias, P, out, ldo, scratch4, (hipStream_t)stream);
}

extern "C" TNP_API int tnp_pool_hiddenmlp_backward(const float *obs1, const float *obs2, const float *hidden_emb_pre, int ldh,
                                                   const int32_t *scene_start, const int32_t *row_base,
                                                   const int32_t *row_count, int B, int M, int ms, int mv, int mh,
                                                   const float *W_spatial, const float *b_spatial, const float *W_vel,
                                                   const float *b_vel, const float *d_pooled, int ldp, float *G, float *R,
                                                   float *d_hidden_emb_pre, int32_t *winner_scratch, int32_t *winner_slots,
                                                   void *stream) {
    if (B <= 0 || M <= 0) return 0;
    if (mh > 0 && (!hidden_emb_pre || !d_hidden_emb_pre || !winner_scratch))
        TNP_FAIL(-1, "tnp_pool_hiddenmlp_backward: hidden-embedding buffers missing");
    const int D = ms + mh + mv;
    const int threads = D <= 64 ? 64 : (D <= 128 ? 128 : 256);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tnp::pool_hiddenmlp_backward_kernel, dim3(B, 32), dim3(threads), 0, s, obs1, obs2, hidden_emb_pre, ldh,
                       scene_start, ms, mv, mh, W_spatial, b_spatial, W_vel, b_vel, d_pooled, ldp, G, R, winner_scratch, winner_slots);
    TNP_HIP(hipGetLastError());
    if (mh > 0) {
        hipLaunchKernelGGL(tnp::pool_hiddenmlp_gather_kernel, dim3((M + 3) / 4), dim3(256), 0, s, winner_scratch, d_pooled, ldp,
                           row_base, row_count, M, ms, mh, d_hidden_emb_pre);
        TNP_HIP(hipGetLastError());
    }
    return 0;
}

extern "C" TNP_API int tnp_pool_nn_pos_backward(const float *obs1, const float *obs2, const int32_t *row_base,
                                                const int32_t *row_count, int R, int n_max, int n_sel, int in_dim, const float *W,
                                                int d, const float *d_pre, int ldd, int32_t *sel_scratch, float *ga_scratch,
                                                float *d_obs1, float *d_obs2, void *stream) {
    if (R <= 0) return 0;
    if (n_sel < 1 || n_sel > tnp::NN_MAX_SEL || (in_dim != 2 && in_dim != 4)) TNP_FAIL(-1, "tnp_pool_nn_pos_backward: n = %d, input_dim = %d", n_sel, in_dim);
    if (n_max > 64 * tnp::NN_WAVE_T) TNP_FAIL(-1, "tnp_pool_nn_pos_backward: scenes of more than %d tracks", 64 * tnp::NN_WAVE_T);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tnp::nn_pos_pairs_kernel, dim3((R + 3) / 4), dim3(256), 0, s, obs1, obs2, row_base, row_count, R, n_sel, in_dim, W,
                       d, d_pre, ldd, sel_scratch, ga_scratch);
    TNP_HIP(hipGetLastError());
    hipLaunchKernelGGL(tnp::nn_pos_gather_kernel, dim3((R + 3) / 4), dim3(256), 0, s, sel_scratch, ga_scratch, row_base, row_count, R,
                       n_sel, d_obs1, d_obs2);
    TNP_HIP(hipGetLastError());
    return 0;
}

extern "C" TNP_API int tnp_pool_pair_pos_gather(const float *pair_records, const int32_t *row_base, const int32_t *row_count, int R,
                                                int n_max, float *d_obs1, float *d_obs2, void *stream) {
    if (R <= 0) return 0;
    hipLaunchKernelGGL(tnp::pair_pos_gather_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, pair_records, row_base,
                       row_count, R, n_max, d_obs1, d_obs2);
    TNP_HIP(hipGetLastError());
    return 0;
}

extern "C" TNP_API int tnp_pool_hiddenmlp_pos_backward(const float *G, const int32_t *winner_slots, const float *W_spatial,
                                                       const float *W_vel, const int32_t *row_base, const int32_t *row_count, int R,
                                                       int M_step, int ms, int mv, float *d_obs1, float *d_obs2, void *stream) {
    if (R <= 0) return 0;
    hipLaunchKernelGGL(tnp::hiddenmlp_pos_gather_kernel, dim3((R + 3) / 4), dim3(256), 0, (hipStream_t)stream, G, winner_slots, W_spatial,
                       W_vel, row_base, row_count, R, M_step, ms, mv, d_obs1, d_obs2);
    TNP_HIP(hipGetLastError());
    return 0;
}

extern "C" TNP_API size_t tnp_colsum_prod_workspace_bytes(long rows, int cols) {
    if (rows <= 0 || cols <= 0) return 0;
    return (size_t)((rows + tnp::CS_ROWS - 1) / tnp::CS_ROWS) * cols * 3 * sizeof(float);
}

extern "C" TNP_API int tnp_colsum_prod(const float *G, const float *R, long rows, int cols, float *dW, float *db, void *workspace,
                                       size_t workspace_bytes, void *stream) {
    if (rows <= 0 || cols <= 0) return 0;
    const int nchunks = (int)((rows + tnp::CS_ROWS - 1) / tnp::CS_ROWS);
    if (!workspace || workspace_bytes < (size_t)nchunks * cols * 3 * sizeof(float)) TNP_FAIL(-1, "tnp_colsum_prod: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tnp::colsum_prod_kernel, dim3((cols + 63) / 64, nchunks), dim3(256), 0, s, G, R, rows, cols, (float *)workspace);
    TNP_HIP(hipGetLastError());
    hipLaunchKernelGGL(tnp::colsum_reduce_kernel, dim3((cols * 3 + 3) / 4), dim3(256), 0, s, (const float *)workspace, nchunks, cols,
                       dW, db);
    TNP_HIP(hipGetLastError());
    return 0;
}

extern "C" TNP_API int tnp_pool_attn_pair_backward(const float *obs1, const float *obs2, const float *hidden_emb_pre, int ldh,
                                                   const int32_t *scene_start, int B, int n_max, const int32_t *scene_slots,
                                                   int ms, int mv, int mh,
                                                   const float *W_spatial, const float *b_spatial, const float *W_vel,
                                                   const float *b_vel, float fill, const float *u, int ldu, const float *d_ebar,
                                                   int ldd, float *du, float *A3, float *dEh, float *ebar, int lde, float *pos_rec,
                                                   void *stream) {
    if (B <= 0) return 0;
    const int D = ms + mh + mv;
    if (D > 64 * tnp::ATT_MAXD_PER_LANE) TNP_FAIL(-1, "AttentionMLPPooling: mlp_dim %d > %d", D, 64 * tnp::ATT_MAXD_PER_LANE);
    const size_t lds = ((size_t)((2 * n_max + 3) & ~3) + (size_t)n_max * (4 + (mh > 0 ? mh : 0))) * sizeof(float);
    if (n_max < 1 || lds > tnp::ATT_MAX_LDS) TNP_FAIL(-1, "AttentionMLPPooling: a scene of %d slots (x mlp_dim_hidden %d) needs %zu bytes of LDS, limit %zu", n_max, mh, lds, tnp::ATT_MAX_LDS);
    if (lds > 60000) {
        static bool raised = false;
        if (!raised) { TNP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tnp::pool_attn_pair_backward_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)tnp::ATT_MAX_LDS)); raised = true; }
    }
    if (ldu < D + 1) TNP_FAIL(-1, "tnp_pool_attn_pair_backward: ldu %d < mlp_dim + 1", ldu);
    hipLaunchKernelGGL(tnp::pool_attn_pair_backward_kernel, dim3(B, n_max < 64 ? n_max : 64), dim3(64), lds,
                       (hipStream_t)stream, obs1, obs2, hidden_emb_pre, ldh, scene_start, n_max, scene_slots, ms, mv, mh, W_spatial, b_spatial,
                       W_vel, b_vel, fill, u, ldu, d_ebar, ldd, du, A3, dEh, ebar, lde, pos_rec);
    TNP_HIP(hipGetLastError());
    return 0;
}

extern "C" TNP_API int tnp_pool_attn_self_backward(const float *obs1, const float *obs2, const float *hidden_emb_pre, int ldh,
                                                   const int32_t *row_base, const int32_t *row_count, int M, int n_max, int ms,
                                                   int mv, int mh, const float *b_spatial, const float *b_vel,
                                                   const float *de_self, int ldd, const float *dEh, float *A3, float *dself_scratch,
                                                   float *d_hidden_emb_pre, void *stream) {
    if (M <= 0) return 0;
    const int D = ms + mh + mv;
    const long total = (long)M * D;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tnp::pool_attn_self_backward_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, obs1, obs2, M, ms,
                       mv, mh, b_spatial, b_vel, de_self, ldd, A3, dself_scratch);
    TNP_HIP(hipGetLastError());
    if (mh > 0) {
        hipLaunchKernelGGL(tnp::pool_attn_gather_kernel, dim3((M + 3) / 4), dim3(256), 0, s, dEh, dself_scratch, hidden_emb_pre, ldh,
                           row_base, row_count, M, n_max, mh, d_hidden_emb_pre);
        TNP_HIP(hipGetLastError());
    }
    return 0;
}
