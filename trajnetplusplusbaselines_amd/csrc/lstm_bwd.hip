// Pointwise / gather kernels of the training backward sweep (reference: autograd through LSTM.forward,
// lstm/trainer.py:229-269).  All contractions of the sweep are tnp_linear_forward calls; what is left are the
// derivative expressions of Hidden2Normal and LSTMCell, ReLU masks and the backward of the social grid's scatter.
// HBM-bound elementwise work over [M, H] / [M, 4H] arrays; one launch each instead of ~25 eager tensor expressions.
#include "tnp_internal.h"
#include <stdlib.h>

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

// h2n_backward_kernel + lstm_cell_backward_kernel in one launch: the wave that forms dh_tot[m, :] applies the LSTMCell
// derivatives to it right away (same expressions as the two kernels, no dh_tot round trip)
__global__ void __launch_bounds__(64) h2n_cell_backward_kernel(const float *__restrict__ h_out, const float *__restrict__ Wn,
                                                          const float *__restrict__ bn, const float *__restrict__ d_normal,
                                                          const float *__restrict__ d_pos, const float *__restrict__ obs1,
                                                          const float *__restrict__ obs2, const float *__restrict__ dh_in,
                                                          int M, int H, float *__restrict__ dlin,
                                                          const float *__restrict__ gates, const float *__restrict__ c_prev,
                                                          const float *__restrict__ dc, float *__restrict__ dG,
                                                          float *__restrict__ dc_prev, float *__restrict__ dh_pass) {
    const int m = blockIdx.x, lane = threadIdx.x;
    if (m >= M) return;
    // Every operand is fetched before anything is used (one global round trip for the whole kernel instead of one per
    // phase: presence test -> Hidden2Normal dot products -> loss gradients -> cell derivatives); lanes own elements
    // k = lane, lane + 64 (H <= 128 runs entirely from these registers, larger H falls back to loops for the rest).
    constexpr int KP = 2;
    const float a = obs1[2 * m], b = obs2[2 * m];
    float pbn[5], pdn[5], ppx = 0.0f, ppy = 0.0f;
#pragma unroll
    for (int q = 0; q < 5; ++q) { pbn[q] = bn[q]; pdn[q] = d_normal ? d_normal[(size_t)m * 5 + q] : 0.0f; }
    if (d_pos) { ppx = d_pos[2 * m]; ppy = d_pos[2 * m + 1]; }
    float ph[KP], pw[KP][5], pdh[KP], pdc[KP], pg[KP][4], pcp[KP];
#pragma unroll
    for (int i = 0; i < KP; ++i) {
        const int k = lane + 64 * i, kc = k < H ? k : H - 1;
        const size_t q = (size_t)m * H + kc;
        ph[i] = h_out[q]; pdh[i] = dh_in[q]; pdc[i] = dc[q]; pcp[i] = c_prev[q];
#pragma unroll
        for (int w = 0; w < 5; ++w) pw[i][w] = Wn[w * H + kc];
        const float *gs = gates + (size_t)m * 4 * H + kc;
        pg[i][0] = gs[0]; pg[i][1] = gs[H]; pg[i][2] = gs[2 * H]; pg[i][3] = gs[3 * H];
    }
    const bool present = (a == a) && (b == b);                       // lstm/lstm.py:118
    float dl[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (present) {
        // Linear output of Hidden2Normal (needed for the sigmoid derivatives): 5 dot products over H
        float lin[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < KP; ++i)
            if (lane + 64 * i < H) {
#pragma unroll
                for (int q = 0; q < 5; ++q) lin[q] = fmaf(ph[i], pw[i][q], lin[q]);
            }
        for (int k = lane + 64 * KP; k < H; k += 64) {
            const float hv = h_out[(size_t)m * H + k];
#pragma unroll
            for (int q = 0; q < 5; ++q) lin[q] = fmaf(hv, Wn[q * H + k], lin[q]);
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) {
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) lin[q] += __shfl_xor(lin[q], off, 64);
            lin[q] += pbn[q];
        }
        float dn[5];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            float v = pdn[q];
            if (v != v) v = 0.0f;                                    // NaN rows of absent tracks carry no gradient
            dn[q] = v;
        }
        if (d_pos) {                                                // positions = obs2 + normal[:, :2]
            dn[0] += (ppx == ppx) ? ppx : 0.0f;
            dn[1] += (ppy == ppy) ? ppy : 0.0f;
        }
        const float s2 = sigm(lin[2]), s3 = sigm(lin[3]), s4 = sigm(lin[4]);
        dl[0] = dn[0]; dl[1] = dn[1];
        dl[2] = dn[2] * 0.2f * s2 * (1.0f - s2);
        dl[3] = dn[3] * 0.2f * s3 * (1.0f - s3);
        dl[4] = dn[4] * 0.7f * s4 * (1.0f - s4);
    }
    if (lane < 5) dlin[(size_t)m * 5 + lane] = dl[lane];
    auto cell = [&](int k, float acc, float dcv, float gi, float gf, float gg, float go, float cp) {
        const size_t q = (size_t)m * H + k;
        float *g = dG + (size_t)m * 4 * H + k;
        if (present) {
            const float cn = gf * cp + gi * gg;
            const float tc = tanhf(cn);
            const float d_o = acc * tc;
            const float dct = dcv + acc * go * (1.0f - tc * tc);
            g[0] = dct * gg * gi * (1.0f - gi);
            g[H] = dct * cp * gf * (1.0f - gf);
            g[2 * H] = dct * gi * (1.0f - gg * gg);
            g[3 * H] = d_o * go * (1.0f - go);
            dc_prev[q] = dct * gf;
            dh_pass[q] = 0.0f;
        } else {  // state copied through: the gradient bypasses the cell
            g[0] = 0.0f; g[H] = 0.0f; g[2 * H] = 0.0f; g[3 * H] = 0.0f;
            dc_prev[q] = dcv;
            dh_pass[q] = acc;
        }
    };
#pragma unroll
    for (int i = 0; i < KP; ++i) {
        const int k = lane + 64 * i;
        if (k >= H) continue;
        float acc = pdh[i];
#pragma unroll
        for (int q = 0; q < 5; ++q) acc = fmaf(dl[q], pw[i][q], acc);
        cell(k, acc, pdc[i], pg[i][0], pg[i][1], pg[i][2], pg[i][3], pcp[i]);
    }
    for (int k = lane + 64 * KP; k < H; k += 64) {
        float acc = dh_in[(size_t)m * H + k];
#pragma unroll
        for (int q = 0; q < 5; ++q) acc = fmaf(dl[q], Wn[q * H + k], acc);
        const float *gs = gates + (size_t)m * 4 * H + k;
        cell(k, acc, dc[(size_t)m * H + k], gs[0], gs[H], gs[2 * H], gs[3 * H], c_prev[(size_t)m * H + k]);
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
// over the pairs of t's scene whose two velocities are finite.  `cells` comes from tnp_pool_pair_cells_autograd (a cell
// whose value is the constant 0 passes nothing); with its `winner` table the second consequence of lp_pool2d's zero
// derivative at 0 is applied per channel: the cell's value is its WINNER's nan_to_num(v_w - v_i), and where that is exactly
// 0 (a winner without a finite velocity: both channels) no pair of the cell receives that channel's gradient.
__global__ void __launch_bounds__(256) directional_scatter_backward_kernel(const float *__restrict__ dgrid, int ldg,
                                                                           const int32_t *__restrict__ cells,
                                                                           const int32_t *__restrict__ winner,
                                                                           const int32_t *__restrict__ row_base,
                                                                           const int32_t *__restrict__ row_count,
                                                                           const float *__restrict__ obs1,
                                                                           const float *__restrict__ obs2, int M, int n_max,
                                                                           int ncell, float *__restrict__ dvel) {
    // one wave per track t: the lanes take the partners of t's scene, then a fixed shuffle tree (deterministic)
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (t >= M) return;
    const int lo = row_base[t], ns = row_count[t], tt = t - lo;
    auto finite_vel = [&](int r) {
        const float a = obs1[2 * r], b = obs1[2 * r + 1], c = obs2[2 * r], d = obs2[2 * r + 1];
        return (a == a) && (b == b) && (c == c) && (d == d);
    };
    // channel mask of ego row `e`'s cell that pair slot `slot` stands in: 1 where the winner's value is non-zero
    auto pass = [&](int e, int slot, bool &px, bool &py) {
        px = py = true;
        if (!winner) return;
        const int w = winner[(size_t)e * n_max + slot];
        if (w < 0) return;                                           // -2: the cell holds a non-zero constant
        const int wr = row_base[e] + w;
        float vx = (obs2[2 * wr] - obs1[2 * wr]) - (obs2[2 * e] - obs1[2 * e]);
        float vy = (obs2[2 * wr + 1] - obs1[2 * wr + 1]) - (obs2[2 * e + 1] - obs1[2 * e + 1]);
        if (vx != vx) vx = 0.0f;                                     // nan_to_num, gridbased_pooling.py:140
        if (vy != vy) vy = 0.0f;
        px = vx != 0.0f; py = vy != 0.0f;
    };
    float ax = 0.0f, ay = 0.0f;
    if (finite_vel(t)) {
        for (int i = lo + lane; i < lo + ns; i += 64) {
            if (i == t || !finite_vel(i)) continue;
            bool px, py;
            const int c1 = cells[(size_t)i * n_max + tt];
            if (c1 >= 0) {
                pass(i, tt, px, py);
                if (px) ax += dgrid[(size_t)i * ldg + c1];
                if (py) ay += dgrid[(size_t)i * ldg + ncell + c1];
            }
            const int c2 = cells[(size_t)t * n_max + (i - lo)];
            if (c2 >= 0) {
                pass(t, i - lo, px, py);
                if (px) ax -= dgrid[(size_t)t * ldg + c2];
                if (py) ay -= dgrid[(size_t)t * ldg + ncell + c2];
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { ax += __shfl_xor(ax, off); ay += __shfl_xor(ay, off); }
    if (lane == 0) { dvel[2 * t] = ax; dvel[2 * t + 1] = ay; }
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

#define TR_MAX_PROBLEMS 32
struct TransposeGroup {
    const float *in[TR_MAX_PROBLEMS];
    float *out[TR_MAX_PROBLEMS];
    int ld_in[TR_MAX_PROBLEMS], ld_out[TR_MAX_PROBLEMS], R[TR_MAX_PROBLEMS], C[TR_MAX_PROBLEMS], tiles_x[TR_MAX_PROBLEMS];
    int first_block[TR_MAX_PROBLEMS + 1];
    int count;
};

// transpose_kernel for several matrices: block -> (problem, 64 x 64 tile)
__global__ void __launch_bounds__(256) transpose_group_kernel(const TransposeGroup g) {
    __shared__ float tile[64][65];
    int p = 0;
    while (p + 1 < g.count && (int)blockIdx.x >= g.first_block[p + 1]) ++p;
    const int local = (int)blockIdx.x - g.first_block[p];
    const int c0 = (local % g.tiles_x[p]) * 64, r0 = (local / g.tiles_x[p]) * 64;
    const float *__restrict__ in = g.in[p];
    float *__restrict__ out = g.out[p];
    const int R = g.R[p], C = g.C[p], ld_in = g.ld_in[p], ld_out = g.ld_out[p];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
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

// W [N1][C ncell] -> cell-major W'[c][ch][o] and (optional) quad-major W''[c][o/64][ch/4][o%64][ch%4] (tnp_lstm_model).
// Workgroup (cell block of 32, column block of 64 outputs, channel quad q): the 4 x 64 x 32 values go through LDS once; reads
// run along the cells (contiguous in W), writes along the outputs (contiguous in both copies).
__global__ void __launch_bounds__(256) weight_layouts_kernel(const float *__restrict__ W, int ldw, int N1, int C, int ncell,
                                                            float *__restrict__ Wc, float *__restrict__ Wq) {
    __shared__ float t[4][64][33];                                   // [ch % 4][o % 64][cell % 32]
    const int c0 = blockIdx.x * 32, o0 = blockIdx.y * 64, q = blockIdx.z;
    const int tid = threadIdx.x, lx = tid & 31, ly = tid >> 5;       // 8 rows of 32 lanes
    for (int k = 0; k < 4; ++k) {
        const int ch = 4 * q + k;
        for (int r = ly; r < 64; r += 8) {
            const int o = o0 + r, c = c0 + lx;
            t[k][r][lx] = (ch < C && o < N1 && c < ncell) ? W[(size_t)o * ldw + (size_t)ch * ncell + c] : 0.0f;
        }
    }
    __syncthreads();
    const int ox = tid & 63, cy = tid >> 6;                          // 4 cells per pass, 64 outputs per cell
    for (int cc = cy; cc < 32; cc += 4) {
        const int c = c0 + cc, o = o0 + ox;
        if (c >= ncell || o >= N1) continue;
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = t[k][ox][cc];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (4 * q + k < C) Wc[((size_t)c * C + 4 * q + k) * N1 + o] = v[k];
        if (Wq)       // N1 % 64 == 0, C % 4 == 0 (checked by the launcher): a lane's four channels are 16 contiguous bytes
            *reinterpret_cast<float4 *>(Wq + ((((size_t)c * (N1 >> 6) + (o >> 6)) * (C >> 2) + q) * 64 + (o & 63)) * 4) =
                make_float4(v[0], v[1], v[2], v[3]);
    }
}

// ---- sparse backward of the first grid-embedding layer (social pooling) ------------------------------------------
// Forward (pool_embed_sparse.hip): y1[i, :] = sum over the occupied cells c of ego i of  W'[c][ch][:] * enc[winner(i,c), ch]
// with W' the cell-major copy of pool.embedding[0].weight.  A scene of 32 agents occupies a few of the 256 cells, so
// both gradients touch ~3 % of the dense [M, C*ncell] grid; the dense forms (one [M,N1]x[N1,C*ncell] GEMM per step
// and one [N1, S*M] x [S*M, C*ncell] GEMM per sweep) were a third of the optimisation step.

// Gradient of the occupied part of the grid, grouped by cell: for cell c and the egos r that have an in-range
// neighbour in c (ego lists from pair_occupancy + hits_transpose + hits_compact),
//     dcell[r][c][ch] = dy1[r, :] . W'[c][ch][:]
// is a gathered [egos x N1] x [N1 x C] product per cell, so the 64 KB weight block of a cell is read once per workgroup
// instead of once per (ego, neighbour) pair (the per-pair form moved 2.6 GB through L2 per step).  16 egos x 16 channels
// per v_mfma_f32_16x16x4_f32; the four waves of a workgroup split N1 and their partial tiles are summed through LDS in a
// fixed order.  The cell's weights sit in LDS in the lanes' operand order (wave-private, linear ds_read_b128).
typedef float floatx4 __attribute__((ext_vector_type(4)));

template <int NB>
__global__ void __launch_bounds__(256) dgrid_cells_kernel(const float *__restrict__ dy, int ldy, const float *__restrict__ Wc,
                                                          const int2 *__restrict__ list, const int32_t *__restrict__ count,
                                                          int R, int nseg, int seg, int M, int C, int ncell, int N1,
                                                          float *__restrict__ dcell) {
    extern __shared__ __attribute__((aligned(16))) float4 cell_lds[];
    const int c = blockIdx.x;
    const int cnt = count[c * nseg + seg];
    const int ngroups = (cnt + 15) >> 4;
    if ((int)blockIdx.y >= ngroups) return;
    const int2 *L = list + (size_t)c * R + (size_t)seg * M;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = lane & 15, kq = lane >> 4;
    const int KQ = N1 >> 2, T = KQ >> 4, k0 = wave * KQ;
    float4 *Bw = cell_lds + (size_t)wave * T * NB * 64;
    float *red = reinterpret_cast<float *>(cell_lds + (size_t)4 * T * NB * 64);   // [4 waves][NB][16 egos][16 ch]
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int ch = nb * 16 + row;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ch < C) v = *reinterpret_cast<const float4 *>(Wc + ((size_t)c * C + ch) * N1 + k0 + 16 * t + 4 * kq);
            Bw[(t * NB + nb) * 64 + lane] = v;
        }
    const int row_off = seg * M;
    for (int g = blockIdx.y; g < ngroups; g += gridDim.y) {
        const int idx = g * 16 + row;
        const int rl = L[idx < cnt ? idx : 0].x - row_off;
        const float *src = dy + (size_t)rl * ldy + k0 + 4 * kq;
        floatx4 acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = floatx4{0.f, 0.f, 0.f, 0.f};
        for (int t0 = 0; t0 < T; t0 += 16) {
            float4 a[16];
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (t0 + u < T) a[u] = *reinterpret_cast<const float4 *>(src + 16 * (t0 + u));
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (t0 + u < T) {
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) {
                        const float4 b = Bw[((t0 + u) * NB + nb) * 64 + lane];
                        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b.x, acc[nb], 0, 0, 0);
                        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b.y, acc[nb], 0, 0, 0);
                        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b.z, acc[nb], 0, 0, 0);
                        acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b.w, acc[nb], 0, 0, 0);
                    }
                }
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int v = 0; v < 4; ++v) red[((wave * NB + nb) * 16 + 4 * kq + v) * 16 + row] = acc[nb][v];
        __syncthreads();
        for (int o = tid; o < NB * 256; o += 256) {
            const int nb = o >> 8, i = (o >> 4) & 15, ch = nb * 16 + (o & 15);
            const int at = (nb * 16 + i) * 16 + (o & 15);
            const float sum = (red[at] + red[NB * 256 + at]) + (red[2 * NB * 256 + at] + red[3 * NB * 256 + at]);
            const int id2 = g * 16 + i;
            if (id2 < cnt && ch < C) dcell[((size_t)(L[id2].x - row_off) * ncell + c) * C + ch] = sum;
        }
        __syncthreads();
    }
}

// occ[r][c] = 1 when ego row r has an in-range neighbour in cell c (cells from pair_cells_kernel; occ zeroed before)
__global__ void __launch_bounds__(256) pair_occupancy_kernel(const int32_t *__restrict__ cells, long tot, int n_max, int ncell,
                                                             uint8_t *__restrict__ occ) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q >= tot) return;
    const int c = cells[q];
    if (c >= 0) occ[(q / n_max) * ncell + c] = 1;
}

// d(enc)[j, ch] = sum over the egos i of j's scene with cell(i,j) >= 0 of dcell[i][cell(i,j)][ch]   (the scatter's
// autograd gives every in-range neighbour its cell's gradient); fixed summation order
__global__ void __launch_bounds__(256) social_scatter_backward_cells_kernel(const float *__restrict__ dcell,
                                                                            const int32_t *__restrict__ cells,
                                                                            const int32_t *__restrict__ row_base,
                                                                            const int32_t *__restrict__ row_count, int M,
                                                                            int n_max, int C, int ncell,
                                                                            float *__restrict__ denc) {
    // one wave per neighbour track j: lanes = (ego slice, channel); the four ego slices are combined in a fixed order
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, sub = lane >> 4;
    if (j >= M) return;
    const int lo = row_base[j], ns = row_count[j], jj = j - lo;
    for (int ch = lane & 15; ch < ((C + 15) & ~15); ch += 16) {
        float acc = 0.0f;
#pragma unroll 4
        for (int i = lo + sub; i < lo + ns; i += 4) {
            const int c = cells[(size_t)i * n_max + jj];
            if (c >= 0 && ch < C) acc += dcell[((size_t)i * ncell + c) * C + ch];
        }
        acc += __shfl_xor(acc, 16);
        acc += __shfl_xor(acc, 32);
        if (sub == 0 && ch < C) denc[(size_t)j * C + ch] = acc;
    }
}

// hit_t[c][r] = global row (step*M + track) of the encoding stored in cell c of ego row r, or -1: the winner tables of
// all steps, transposed so that a cell's hits over the whole sweep are contiguous.  64 x 64 tiles through LDS.
// OCC: the table is the 0/1 occupancy of pair_occupancy_kernel and the entry is the ego row itself.
template <typename T, bool OCC>
__global__ void __launch_bounds__(256) hits_transpose_kernel(const T *__restrict__ table, const int32_t *__restrict__ row_base,
                                                             int R, int M, int ncell, int32_t *__restrict__ hit_t) {
    __shared__ int32_t tile[64][65];
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64, tid = threadIdx.x;
    // read: four table entries per load (8 bytes of int16 / 4 bytes of uint8), 16 threads per row, four rows per thread, all
    // four loads (and their row_base entries) in flight together -- entry by entry this kernel ran at 1 TB/s.  The vector
    // loads need the row segment aligned: ncell a multiple of 4 and a 16-byte aligned table (else entry by entry).
    const bool vec = (ncell & 3) == 0 && (reinterpret_cast<uintptr_t>(table) & 15) == 0;
    const int q = tid & 15, rr = tid >> 4;                           // column quad, row within a group of 16
    T e[4][4];
    int base[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + rr + 16 * k, c = c0 + 4 * q;
        const int rc = r < R ? r : R - 1;
        base[k] = 0;
        if (!OCC) { const int i = rc % M; base[k] = rc - i + row_base[i]; }
        if (vec && c + 3 < ncell) {
            if (sizeof(T) == 2) {
                const uint2 v = *reinterpret_cast<const uint2 *>(table + (size_t)rc * ncell + c);
                e[k][0] = (T)(v.x & 0xffff); e[k][1] = (T)(v.x >> 16); e[k][2] = (T)(v.y & 0xffff); e[k][3] = (T)(v.y >> 16);
            } else {
                const uint32_t v = *reinterpret_cast<const uint32_t *>(table + (size_t)rc * ncell + c);
                e[k][0] = (T)(v & 0xff); e[k][1] = (T)((v >> 8) & 0xff); e[k][2] = (T)((v >> 16) & 0xff); e[k][3] = (T)(v >> 24);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) e[k][j] = table[(size_t)rc * ncell + (c + j < ncell ? c + j : ncell - 1)];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + rr + 16 * k;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + 4 * q + j;
            int v = -1;
            if (r < R && c < ncell) {
                const int w = (int)e[k][j];
                if (OCC) { if (w) v = r; }
                else if (w >= 0) v = base[k] + w;
            }
            tile[rr + 16 * k][4 * q + j] = v;
        }
    }
    __syncthreads();
    const int tx = tid & 63, ty = tid >> 6;
    for (int k = ty; k < 64; k += 4) {
        const int c = c0 + k, r = r0 + tx;
        if (r < R && c < ncell) hit_t[(size_t)c * R + r] = tile[tx][k];
    }
}

// compact list of a cell's hits in ascending row order, per segment of `seg` rows (one segment = the whole sweep for the
// weight gradient, one step for the ego lists): list[c][y*seg + k] = (ego row, entry), count[c*nseg + y]
template <int NW, int KEEP>
__global__ void __launch_bounds__(64 * NW) hits_compact_kernel(const int32_t *__restrict__ hit_t, int R, int seg,
                                                               int2 *__restrict__ list, int32_t *__restrict__ count) {
    // Two passes without a barrier inside the loops (the first version synchronised the workgroup twice per 256 rows: 152
    // round trips for the whole-sweep list): each of the NW waves owns a contiguous slice of the segment, counts its hits,
    // the counts are exchanged once, and the second pass writes the wave's hits behind those of the waves before it.
    // NW = 16 for the whole-sweep list (one segment of 19 * M rows per cell: 42 -> 14 us with four times the waves per
    // cell), 4 for the per-step ego lists (19 short segments per cell).
    __shared__ int wcnt[NW];
    const int c = blockIdx.x, y = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t *src = hit_t + (size_t)c * R + (size_t)y * seg;
    int2 *dst = list + (size_t)c * R + (size_t)y * seg;
    const int per = ((seg + 64 * NW - 1) / (64 * NW)) * 64;   // rows per wave, a multiple of 64
    const int k0 = wave * per, k1 = min(seg, k0 + per);
    constexpr int UN = 8;                                     // rows are fetched UN x 64 at a time: the loads of a batch are
    int mine = 0;                                             // independent, so one memory round trip serves 512 rows
    if (KEEP > 0 && per <= 64 * KEEP) {
        // the wave's whole slice fits in registers (whole-sweep list of config 2: 38 rows per lane): ONE read, every load in
        // flight at once, the second pass works on the registers (two passes over memory in batches of eight were ~ten
        // dependent round trips per wave: 23 us for 22 MB)
        constexpr int KK = KEEP > 0 ? KEEP : 1;
        int e[KK];
#pragma unroll
        for (int u = 0; u < KK; ++u) {
            const int k = k0 + 64 * u + lane;
            const int v = src[k < k1 ? k : (k1 > k0 ? k1 - 1 : 0)];
            e[u] = k < k1 ? v : -1;
        }
#pragma unroll
        for (int u = 0; u < KK; ++u) mine += __popcll(__ballot(e[u] >= 0));
        if (lane == 0) wcnt[wave] = mine;
        __syncthreads();
        int off = 0, total = 0;
#pragma unroll
        for (int q = 0; q < NW; ++q) { const int w = wcnt[q]; if (q < wave) off += w; total += w; }
#pragma unroll
        for (int u = 0; u < KK; ++u) {
            const unsigned long long m = __ballot(e[u] >= 0);
            if (e[u] >= 0) dst[off + __popcll(m & ((1ull << lane) - 1ull))] = make_int2(y * seg + k0 + 64 * u + lane, e[u]);
            off += __popcll(m);
        }
        if (tid == 0) count[c * gridDim.y + y] = total;
        return;
    }
    for (int base = k0; base < k1; base += 64 * UN) {
        int e[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) { const int k = base + 64 * u + lane; e[u] = (k < k1) ? src[k] : -1; }
#pragma unroll
        for (int u = 0; u < UN; ++u) mine += __popcll(__ballot(e[u] >= 0));
    }
    if (lane == 0) wcnt[wave] = mine;
    __syncthreads();
    int off = 0, total = 0;
#pragma unroll
    for (int q = 0; q < NW; ++q) { const int w = wcnt[q]; if (q < wave) off += w; total += w; }
    for (int base = k0; base < k1; base += 64 * UN) {
        int e[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) { const int k = base + 64 * u + lane; e[u] = (k < k1) ? src[k] : -1; }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const unsigned long long m = __ballot(e[u] >= 0);
            if (e[u] >= 0) dst[off + __popcll(m & ((1ull << lane) - 1ull))] = make_int2(y * seg + base + 64 * u + lane, e[u]);
            off += __popcll(m);
        }
    }
    if (tid == 0) count[c * gridDim.y + y] = total;
}

// dW'[c][ch][n] = sum over the hits (r, e) of cell c of  dy1[r, n] * enc[e, ch]   (all steps of the sweep at once)
// One workgroup per (cell, 64 output columns); its four waves take the hits k = 4*g + wave of every group of four
// batches (a wave's hit loop is a chain of dependent scalar + vector loads, so the list is split four ways) and the four
// partial tiles are combined through LDS in a fixed order -> deterministic.  The hit list and the encodings are
// wave-uniform (scalar loads), the dy1 rows are 256-byte coalesced reads.
template <int C>
__global__ void __launch_bounds__(256) sparse_wgrad_kernel(const float *__restrict__ dy, int ldy,
                                                           const float *__restrict__ enc, int lde,
                                                           const int2 *__restrict__ list, const int32_t *__restrict__ count,
                                                           int R, int N1, int ncell, float *__restrict__ dWc) {
    __shared__ float red[3][C][64];
    // Workgroups b, b+8, ... share an XCD (and its L2).  A dy row is needed by every cell its ego occupies (~8), in the
    // same 64-column chunk: all cells of a column chunk go to ONE XCD, so the row segment is fetched from HBM once per
    // chunk instead of once per cell (the hit lists are sorted by row, the cells sweep the rows roughly together).
    int c, chunk;
    {
        const int nchunk = N1 >> 6, bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
        if ((nchunk & 7) == 0) { chunk = xcd + 8 * (slot / ncell); c = slot - (slot / ncell) * ncell; }
        else { chunk = bid / ncell; c = bid - chunk * ncell; }
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = chunk * 64 + lane;
    const int cnt = count[c];
    const int2 *L = list + (size_t)c * R;
    // even / odd channels in the two halves of a packed register: C/2 v_pk_fma_f32 per hit; every channel still accumulates
    // its hits in list order
    typedef float f2 __attribute__((ext_vector_type(2)));
    static_assert(C % 2 == 0, "channel pairs");
    f2 acc[C / 2];
#pragma unroll
    for (int ch = 0; ch < C / 2; ++ch) acc[ch] = f2{0.0f, 0.0f};
    int k = wave * 4;
    for (; k + 4 <= cnt; k += 16) {
        const int2 h0 = L[k], h1 = L[k + 1], h2 = L[k + 2], h3 = L[k + 3];
        const float d0 = dy[(size_t)h0.x * ldy + n], d1 = dy[(size_t)h1.x * ldy + n];
        const float d2 = dy[(size_t)h2.x * ldy + n], d3 = dy[(size_t)h3.x * ldy + n];
        const float *e0 = enc + (size_t)h0.y * lde, *e1 = enc + (size_t)h1.y * lde;
        const float *e2 = enc + (size_t)h2.y * lde, *e3 = enc + (size_t)h3.y * lde;
        const f2 dd0 = {d0, d0}, dd1 = {d1, d1}, dd2 = {d2, d2}, dd3 = {d3, d3};
#pragma unroll
        for (int ch = 0; ch < C / 2; ++ch) {
            f2 t = __builtin_elementwise_fma(dd0, f2{e0[2 * ch], e0[2 * ch + 1]}, acc[ch]);
            t = __builtin_elementwise_fma(dd1, f2{e1[2 * ch], e1[2 * ch + 1]}, t);
            t = __builtin_elementwise_fma(dd2, f2{e2[2 * ch], e2[2 * ch + 1]}, t);
            acc[ch] = __builtin_elementwise_fma(dd3, f2{e3[2 * ch], e3[2 * ch + 1]}, t);
        }
    }
    for (int kk = k; kk < cnt && kk < k + 4; ++kk) {   // ragged last batch (if it falls to this wave)
        const int2 h = L[kk];
        const float d = dy[(size_t)h.x * ldy + n];
        const float *e = enc + (size_t)h.y * lde;
        const f2 dd = {d, d};
#pragma unroll
        for (int ch = 0; ch < C / 2; ++ch) acc[ch] = __builtin_elementwise_fma(dd, f2{e[2 * ch], e[2 * ch + 1]}, acc[ch]);
    }
    if (wave > 0) {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) red[wave - 1][ch][lane] = acc[ch / 2][ch & 1];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int ch = 0; ch < C; ++ch)
            dWc[((size_t)c * C + ch) * N1 + n] = (acc[ch / 2][ch & 1] + red[0][ch][lane]) + (red[1][ch][lane] + red[2][ch][lane]);
    }
}

static int launch_hit_lists(bool occ, const void *table, const int32_t *row_base, int R, int M, int ncell, int seg,
                            int32_t *hit_t, int32_t *list, int32_t *count, hipStream_t s) {
    const dim3 tg((R + 63) / 64, (ncell + 63) / 64);
    if (occ)
        hipLaunchKernelGGL((hits_transpose_kernel<uint8_t, true>), tg, dim3(256), 0, s, (const uint8_t *)table, row_base, R, M,
                           ncell, hit_t);
    else
        hipLaunchKernelGGL((hits_transpose_kernel<int16_t, false>), tg, dim3(256), 0, s, (const int16_t *)table, row_base, R, M,
                           ncell, hit_t);
    TNP_HIP(hipGetLastError());
    if (seg > 8192)
        hipLaunchKernelGGL((hits_compact_kernel<16, 40>), dim3(ncell, R / seg), dim3(1024), 0, s, hit_t, R, seg,
                           reinterpret_cast<int2 *>(list), count);
    else
        hipLaunchKernelGGL((hits_compact_kernel<4, 8>), dim3(ncell, R / seg), dim3(256), 0, s, hit_t, R, seg,
                           reinterpret_cast<int2 *>(list), count);
    TNP_HIP(hipGetLastError());
    return 0;
}

template <int NB>
static int launch_dgrid_cells(const float *dy, int ldy, const float *Wc, const int2 *list, const int32_t *count, int R, int nseg,
                              int seg, int M, int C, int ncell, int N1, float *dcell, hipStream_t s) {
    const size_t lds = (size_t)N1 * NB * 64 + (size_t)NB * 4096;
    if (lds > 160 * 1024) TNP_FAIL(-1, "tnp_social_dgrid_cells: N1 = %d, C = %d needs %zu bytes of LDS", N1, C, lds);
    static bool configured = false;
    if (!configured) {
        TNP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(dgrid_cells_kernel<NB>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured = true;
    }
    // workgroups per cell (each stages the cell's weight block once and takes every ysplit-th ego group):
    // 2 measured best at config 2 (1: 43.6, 2: 31.9, 3: 38.8, 4: 35.8 us per launch)
    const int ysplit = 2;
    hipLaunchKernelGGL(dgrid_cells_kernel<NB>, dim3(ncell, ysplit), dim3(256), lds, s, dy, ldy, Wc, list, count, R, nseg, seg, M, C,
                       ncell, N1, dcell);
    TNP_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------------
// K-slice-per-XCD form of the same product (C <= 16, N1 % 128 == 0): dgrid_cells_kernel gathers every dy row for each of
// the ~12 cells its ego occupies, from workgroups on all eight XCDs, right after another kernel produced dy on yet other
// XCDs -- it moves ~136 MB per launch at 4.9 TB/s and that, not the matrix pipe, is its 31 us (tools/experiments/README.md).
// Here workgroup (cell c, slice x) takes columns [x N1/8, (x+1) N1/8) of the contraction only, and x = blockIdx.x & 7 puts
// all workgroups of a slice on ONE XCD: that XCD's L2 then holds an eighth of the step's dy (1 MB at config 2) and an eighth
// of the weights (2 MB), both re-read from it.  The eight partial results go to eight copies of `dcell`; the scatter that
// consumes them (social_scatter_backward_cells8_kernel) adds the copies in slice order, so nothing is atomic and the order
// is fixed.  A wave owns whole 16-ego groups (g = wave, wave + 4, ...): no barrier and no cross-wave reduction in the loop.
// ---------------------------------------------------------------------------------------------------------
template <int TT>                                                             // TT = N1 / 128 when that is 8 (straight-line code), 0 = any
__global__ void __launch_bounds__(256) dgrid_cells_xcd_kernel(const float *__restrict__ dy, int ldy, const float *__restrict__ Wc,
                                                              const int2 *__restrict__ list, const int32_t *__restrict__ count,
                                                              int R, int nseg, int seg, int M, int C, int ncell, int N1,
                                                              float *__restrict__ dcell8) {
    extern __shared__ __attribute__((aligned(16))) float4 xcd_lds[];          // [T][64] weights of the slice, B-operand order
    const int x = blockIdx.x & 7, c = blockIdx.x >> 3;
    const int cnt = count[c * nseg + seg];
    if (cnt <= 0) return;
    const int ngroups = (cnt + 15) >> 4;
    const int2 *L = list + (size_t)c * R + (size_t)seg * M;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, row = lane & 15, kq = lane >> 4;
    const int ncol = N1 >> 3, T = TT ? TT : ncol >> 4, k0 = x * ncol;
    const int row_off = seg * M;
    float *out = dcell8 + (size_t)x * M * ncell * C;
    // software pipeline over the wave's groups: list entries two groups ahead, dy operands one group ahead (a group is
    // otherwise a chain of two global round trips in front of 32 MFMAs, and the busiest cells have five groups per wave)
    auto rows_of = [&](int g, int &rl, int (&ro)[4]) {
        const int idx = g * 16 + row;
        rl = L[(g < ngroups && idx < cnt) ? idx : 0].x - row_off;
#pragma unroll
        for (int v = 0; v < 4; ++v) { const int id2 = g * 16 + 4 * kq + v; ro[v] = L[(g < ngroups && id2 < cnt) ? id2 : 0].x - row_off; }
    };
    auto load_a = [&](float4 (&a)[8], int rl, int t0) {
        const float *src = dy + (size_t)rl * ldy + k0 + 4 * kq;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (t0 + u < T) a[u] = *reinterpret_cast<const float4 *>(src + 16 * (t0 + u));
    };
    // T <= 8 for N1 <= 1024: one or two 16-byte loads per thread, both in flight before the first is stored (channels
    // >= C read channel 0 and store zeros: a load under `if (ch < C)` is waited for where the branch joins)
    for (int q0 = tid; q0 < T * 64; q0 += 512) {
        float4 v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = q0 + 256 * h, t = (q < T * 64 ? q : q0) >> 6, l = q & 63, ch = l & 15, kk = l >> 4;
            v[h] = *reinterpret_cast<const float4 *>(Wc + ((size_t)c * C + (ch < C ? ch : 0)) * N1 + k0 + 16 * t + 4 * kk);
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int q = q0 + 256 * h;
            if (q < T * 64) xcd_lds[q] = ((q & 15) < C) ? v[h] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    __syncthreads();
    int g = wave;
    if (g >= ngroups) return;
    int rl0, ro0[4], rl1, ro1[4];
    rows_of(g, rl0, ro0);
    rows_of(g + 4, rl1, ro1);
    float4 a0[8], a1[8];
    load_a(a0, rl0, 0);
    for (; g < ngroups; g += 4) {
        int rl2, ro2[4];
        rows_of(g + 8, rl2, ro2);
        if (g + 4 < ngroups) load_a(a1, rl1, 0);
        floatx4 acc = floatx4{0.f, 0.f, 0.f, 0.f};
        int bl = lane;
        asm volatile("" : "+v"(bl));                                  // re-read the weights per group: hoisted out of the loop they
                                                                      // cost 32 registers = one wave per SIMD (3 instead of 4)
        for (int t0 = 0; t0 < T; t0 += 8) {
            if (t0 > 0) load_a(a0, rl0, t0);                          // N1 > 1024: further K batches of this group, not prefetched
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (t0 + u < T) {
                    const float4 b = xcd_lds[(t0 + u) * 64 + bl];
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u].x, b.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u].y, b.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u].z, b.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u].w, b.w, acc, 0, 0, 0);
                }
        }
        // lane (row = channel, kq) holds D[ego 4 kq + v][channel row]
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (g * 16 + 4 * kq + v < cnt && row < C) out[((size_t)ro0[v] * ncell + c) * C + row] = acc[v];
#pragma unroll
        for (int u = 0; u < 8; ++u) a0[u] = a1[u];
        rl0 = rl1; rl1 = rl2;
#pragma unroll
        for (int v = 0; v < 4; ++v) { ro0[v] = ro1[v]; ro1[v] = ro2[v]; }
    }
}

// social_scatter_backward_cells_kernel over the eight slice copies: the copies of an (ego, cell) entry are added in slice order
__global__ void __launch_bounds__(256) social_scatter_backward_cells8_kernel(const float *__restrict__ dcell8,
                                                                             const int32_t *__restrict__ cells,
                                                                             const int32_t *__restrict__ row_base,
                                                                             const int32_t *__restrict__ row_count, int M,
                                                                             int n_max, int C, int ncell,
                                                                             float *__restrict__ denc) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, sub = lane >> 4;
    if (j >= M) return;
    const int lo = row_base[j], ns = row_count[j], jj = j - lo;
    const size_t slice = (size_t)M * ncell * C;
    for (int ch = lane & 15; ch < ((C + 15) & ~15); ch += 16) {
        float acc = 0.0f;
        // four egos per batch: their cell indices first, then the 32 slice entries, then the adds (ego order, slice order)
        for (int i0 = lo + sub; i0 < lo + ns; i0 += 16) {
            int cc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) { const int i = i0 + 4 * u; cc[u] = i < lo + ns ? cells[(size_t)i * n_max + jj] : -1; }
            float v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = cc[u] >= 0 && ch < C;
                const float *p = dcell8 + ((size_t)(ok ? i0 + 4 * u : lo) * ncell + (ok ? cc[u] : 0)) * C + (ch < C ? ch : 0);
#pragma unroll
                for (int xx = 0; xx < 8; ++xx) v[u][xx] = ok ? p[xx * slice] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (cc[u] >= 0 && ch < C) {
                    float sum = v[u][0];
#pragma unroll
                    for (int xx = 1; xx < 8; ++xx) sum += v[u][xx];
                    acc += sum;
                }
        }
        acc += __shfl_xor(acc, 16);
        acc += __shfl_xor(acc, 32);
        if (sub == 0 && ch < C) denc[(size_t)j * C + ch] = acc;
    }
}

// social_scatter_backward_cells8_kernel + state_grad_combine_kernel in one launch (C <= 16, C % 4 == 0, H = 64 HK): the
// wave that forms denc[j][:] also forms dh[j][:] = dxh[j][:] + pass[j][:] + denc[j][:] . Wh -- the operands of the second
// half are requested before the gathers of the first, so the launch costs what the scatter alone did (7.8 us; the combine
// kernel was another 4.9 us per step).  Both halves keep the arithmetic and the order of the kernels they replace.
template <int HK>
__global__ void __launch_bounds__(256) social_scatter_combine_kernel(const float *__restrict__ dcell8, const int32_t *__restrict__ cells,
                                                                     const int32_t *__restrict__ row_base,
                                                                     const int32_t *__restrict__ row_count, int M, int n_max, int C,
                                                                     int ncell, float *__restrict__ denc,
                                                                     const float *__restrict__ dxh, int ld,
                                                                     const float *__restrict__ pass, const float *__restrict__ whT,
                                                                     float *__restrict__ out) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, sub = lane >> 4, ch = lane & 15;
    if (j >= M) return;
    constexpr int H = 64 * HK;
    float v0[HK], v1[HK];
    float4 wv[HK][4];
#pragma unroll
    for (int t = 0; t < HK; ++t) {
        const int k = lane + 64 * t;
        v0[t] = dxh[(size_t)j * ld + k]; v1[t] = pass[(size_t)j * H + k];
        const float4 *w4 = reinterpret_cast<const float4 *>(whT + (size_t)k * C);
#pragma unroll
        for (int c = 0; c < 4; ++c) wv[t][c] = w4[c < C / 4 ? c : 0];
    }
    const int lo = row_base[j], ns = row_count[j], jj = j - lo;
    const size_t slice = (size_t)M * ncell * C;
    float acc = 0.0f;
    // the cell indices of trip t + 1 are requested before the gathers of trip t (cells -> dcell8 is a chain of two round
    // trips per trip otherwise; a 39-agent scene has three trips)
    auto cells_of = [&](int i0, int (&cc)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = i0 + 4 * u; cc[u] = i < lo + ns ? cells[(size_t)i * n_max + jj] : -1; }
    };
    int cc[4];
    cells_of(lo + sub, cc);
    for (int i0 = lo + sub; i0 < lo + ns; i0 += 16) {
        int cn[4];
        cells_of(i0 + 16, cn);
        float v[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = cc[u] >= 0 && ch < C;
            const float *p = dcell8 + ((size_t)(ok ? i0 + 4 * u : lo) * ncell + (ok ? cc[u] : 0)) * C + (ch < C ? ch : 0);
#pragma unroll
            for (int xx = 0; xx < 8; ++xx) v[u][xx] = ok ? p[xx * slice] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (cc[u] >= 0 && ch < C) {
                float sum = v[u][0];
#pragma unroll
                for (int xx = 1; xx < 8; ++xx) sum += v[u][xx];
                acc += sum;
            }
#pragma unroll
        for (int u = 0; u < 4; ++u) cc[u] = cn[u];
    }
    acc += __shfl_xor(acc, 16);
    acc += __shfl_xor(acc, 32);
    if (sub == 0 && ch < C) denc[(size_t)j * C + ch] = acc;
    float d[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) d[c] = __shfl(acc, c);                           // lane c (sub 0) holds denc[j][c]
#pragma unroll
    for (int t = 0; t < HK; ++t) {
        float a = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < C / 4) {
                a = fmaf(d[4 * c], wv[t][c].x, a); a = fmaf(d[4 * c + 1], wv[t][c].y, a);
                a = fmaf(d[4 * c + 2], wv[t][c].z, a); a = fmaf(d[4 * c + 3], wv[t][c].w, a);
            }
        out[(size_t)j * H + lane + 64 * t] = (v0[t] + v1[t]) + a;
    }
}

static bool dgrid_xcd_ok(int C, int N1) { return C <= 16 && N1 % 128 == 0; }

static int launch_dgrid_cells_xcd(const float *dy, int ldy, const float *Wc, const int2 *list, const int32_t *count, int R, int nseg,
                                  int seg, int M, int C, int ncell, int N1, float *dcell8, hipStream_t s) {
    const size_t lds = (size_t)(N1 / 8 / 16) * 64 * sizeof(float4);
    if (N1 == 1024)
        hipLaunchKernelGGL(dgrid_cells_xcd_kernel<8>, dim3(ncell * 8), dim3(256), lds, s, dy, ldy, Wc, list, count, R, nseg, seg, M, C, ncell,
                           N1, dcell8);
    else
        hipLaunchKernelGGL(dgrid_cells_xcd_kernel<0>, dim3(ncell * 8), dim3(256), lds, s, dy, ldy, Wc, list, count, R, nseg, seg, M, C, ncell,
                           N1, dcell8);
    TNP_HIP(hipGetLastError());
    return 0;
}

// Matrix-core form for C <= 16: dW'[c][ch][n] = sum over the cell's hits of enc[hit][ch] * dy[hit][n] is a [16 x hits] x
// [hits x 64] product per (cell, 64-column chunk).  v_mfma_f32_16x16x4_f32 takes four hits per instruction: lane (g = l >> 4,
// i = l & 15) holds A[ch i][hit g] = enc[hit g][i] and B[hit g][col] from one 16-byte load of dy (columns 4 i .. 4 i + 3 of
// the chunk feed four MFMAs).  Everything comes through the vector memory path in 64-byte (enc) / 256-byte (dy) pieces; the
// VALU form above needs the encoding row of every hit in SGPRs, and those 4.7 M scalar loads of 2.5 MB of randomly addressed
// rows (scalar-cache misses) are what held it at 313 us against a 68 us arithmetic floor.  Hits are taken in list order,
// four per MFMA, batches b = wave, wave + 4, ... per wave, the four waves' tiles added in fixed order: deterministic.
// 64 zeros: what a past-the-end hit of sparse_wgrad_mfma_kernel loads.  A zero-initialised device global of the code object:
// present on every device the library is loaded on, no allocation, no fill on any stream, nothing to order a first launch
// against (a lazily hipMalloc'ed + NULL-stream hipMemset buffer was not ordered before a launch on a non-blocking stream,
// not thread-safe and not capturable -- ADVICE round 3).
__device__ __attribute__((aligned(256))) float g_swg_zeros[64] = {};
#define TNP_SWG_SMALL_PLAN (16 * 8 + 1)          // batch_size 8: 4x1 72.9, 4x2 72.2, 8x1 55.8, 8x2 61.0, 16x1 59.1, 16x2 65.5 us (tools/diag/small_train_ab.sh)

template <int C, int NW, int U>                                     // waves per workgroup, batches per trip (see below)
__global__ void __launch_bounds__(64 * NW) sparse_wgrad_mfma_kernel(const float *__restrict__ dy, int ldy,
                                                                const float *__restrict__ enc, int lde,
                                                                const int2 *__restrict__ list, const int32_t *__restrict__ count,
                                                                int R, int N1, int ncell, float *__restrict__ dWc) {
    const float *swg_zeros = g_swg_zeros;
    static_assert(C <= 16, "one 16-channel A tile");
    typedef float f4 __attribute__((ext_vector_type(4)));
    static_assert((NW & (NW - 1)) == 0, "pairwise reduction over the waves");
    __shared__ float red[NW - 1][16][64];
    int c, chunk;
    {
        const int nchunk = N1 >> 6, bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
        if ((nchunk & 7) == 0) { chunk = xcd + 8 * (slot / ncell); c = slot - (slot / ncell) * ncell; }
        else { chunk = bid / ncell; c = bid - chunk * ncell; }
    }
    const int lane = threadIdx.x & 63, g = lane >> 4, i = lane & 15;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cnt = count[c];
    const int2 *L = list + (size_t)c * R;
    const float *dyc = dy + chunk * 64 + 4 * i;
    f4 acc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = f4{0.0f, 0.0f, 0.0f, 0.0f};
    const int nb = (cnt + 3) >> 2;                                  // batches of four hits
    // Every load is unconditional: past the end of the list the entry index is 0 and the operands come from a row of
    // zeros (g_swg_zeros, a zero-initialised device global), chosen by ADDRESS.  A `valid ? load : 0` select is turned back
    // into a branch around the load by the compiler, and a load under a branch is waited for at the join: round 2's kernel
    // did that and took 286 us, this one 235 (profiles/archive/round3_p_sparse_wgrad_sweep.md).
    auto entry = [&](int b) -> int2 { const int k = 4 * b + g; return L[(b < nb && k < cnt) ? k : 0]; };
    auto valid = [&](int b) -> bool { return b < nb && 4 * b + g < cnt; };
    // (as element offsets from the operand bases, so that every lane forms its address with the same arithmetic)
    const long za = swg_zeros - enc, zb = (swg_zeros + 4 * i) - dyc;
    auto fetch = [&](int2 e, bool ok, float &av, f4 &bv) {
        const long oa = (ok && i < C) ? (long)e.y * lde + i : za;
        const long ob = ok ? (long)e.x * ldy : zb;
        av = enc[oa];
        bv = *reinterpret_cast<const f4 *>(dyc + ob);
    };
    // U batches per trip, three trips in flight: list entries of trip t + 2 and operands of trip t + 1 while trip t runs on
    // the matrix pipe.  The order in which a wave adds its batches is b = wave, wave + NW, ... whatever U is.  Measured
    // (same file): U = 2 / 4 / 8 -> 311 / 380 / 335 us, 8 / 16 waves -> 239 / 309 us, a block map that ignores the XCDs 310 us:
    // the kernel is not short of loads in flight, it is bound by what the XCD's L2 has to fetch again (the 288 workgroups of
    // an XCD sweep the 38912 rows at the pace of their own cell's hit density), and more in flight evicts more.
    int b = wave;
    int2 e1[U];
    float a0[U]; f4 b0[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { fetch(entry(b + NW * u), valid(b + NW * u), a0[u], b0[u]); e1[u] = entry(b + NW * (U + u)); }
    for (; b < nb; b += NW * U) {
        int2 e2[U];
        float a1[U]; f4 b1[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { e2[u] = entry(b + NW * (2 * U + u)); fetch(e1[u], valid(b + NW * (U + u)), a1[u], b1[u]); }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u], b0[u][q], acc[q], 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) { a0[u] = a1[u]; b0[u] = b1[u]; e1[u] = e2[u]; }
    }
    // acc[q][r]: channel 4 g + r, column 4 i + q of the chunk
    if (wave > 0) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[wave - 1][4 * g + r][4 * i + q] = acc[q][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int ch = 4 * g + r;
            if (ch >= C) continue;
            f4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                   // pairwise in wave order: (w0 + w1) + (w2 + w3) ...
                float v[NW];
                v[0] = acc[q][r];
#pragma unroll
                for (int w = 1; w < NW; ++w) v[w] = red[w - 1][ch][4 * i + q];
#pragma unroll
                for (int st = 1; st < NW; st *= 2)
#pragma unroll
                    for (int j = 0; j + st < NW; j += 2 * st) v[j] += v[j + st];
                o[q] = v[0];
            }
            *reinterpret_cast<f4 *>(dWc + ((size_t)c * C + ch) * N1 + chunk * 64 + 4 * i) = o;
        }
    }
}

template <int C>
static int launch_sparse_wgrad(const float *dy, int ldy, const float *enc, int lde, const int2 *list, const int32_t *count, int R,
                               int ncell, int N1, float *dWc, hipStream_t s) {
    if constexpr (C <= 16) {
        if (ldy % 4 == 0 && (reinterpret_cast<uintptr_t>(dy) & 15) == 0 && (reinterpret_cast<uintptr_t>(dWc) & 15) == 0) {
            // Waves per (cell, 64-column chunk): four at training batches (more evict each other's rows from the XCD's L2, see
            // the kernel); at small ones (batch_size 8: ~275 hits per cell, 24 MB of dy) the launch is the longest cell's chain
            // of dependent gathers, and more waves / more batches per trip shorten exactly that.
            int plan = tuning().sparse_wgrad_plan;                                // 16 * waves + batches per trip, 0 = by size
            if (plan == 0) plan = R <= 16384 ? TNP_SWG_SMALL_PLAN : 16 * 4 + 1;
            const dim3 grid(ncell * (N1 / 64));
#define TNP_SWG(nw, u) case 16 * nw + u: hipLaunchKernelGGL((sparse_wgrad_mfma_kernel<C, nw, u>), grid, dim3(64 * nw), 0, s, dy, ldy, enc, lde, \
                                                              list, count, R, N1, ncell, dWc); break
            switch (plan) {
                TNP_SWG(4, 1); TNP_SWG(4, 2); TNP_SWG(8, 1); TNP_SWG(8, 2); TNP_SWG(16, 1); TNP_SWG(16, 2);
                default: TNP_FAIL(-1, "tnp_sparse_wgrad: plan %d (16 * waves + batches per trip: waves 4 / 8 / 16, batches 1 / 2)", plan);
            }
#undef TNP_SWG
            TNP_HIP(hipGetLastError());
            return 0;
        }
    }
    hipLaunchKernelGGL(sparse_wgrad_kernel<C>, dim3(ncell * (N1 / 64)), dim3(256), 0, s, dy, ldy, enc, lde, list, count, R, N1, ncell, dWc);
    TNP_HIP(hipGetLastError());
    return 0;
}

// gradient of the previous hidden state: the W_hh part of the fused data-gradient GEMM + what bypassed the cell for
// absent rows (+ the social encoding's contribution, either precomputed in `extra` or formed here as
// denc[m, :] . Wh[:, k] with whT [H, C] -- a C-term dot per element instead of a GEMM launch)
__global__ void __launch_bounds__(256) state_grad_combine_kernel(const float *__restrict__ dxh, int ld, const float *__restrict__ pass,
                                                                 const float *__restrict__ extra, int M, int H,
                                                                 float *__restrict__ out, const float *__restrict__ denc = nullptr,
                                                                 const float *__restrict__ whT = nullptr, int C = 0) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= (long)M * H) return;
    const int m = (int)(q / H), k = (int)(q - (long)m * H);
    // all operands in flight before the first use (the C-term loop otherwise pays a round trip per iteration)
    const float v0 = dxh[(size_t)m * ld + k], v1 = pass[q], v2 = extra ? extra[q] : 0.0f;
    float v = v0 + v1;
    if (extra) v += v2;
    if (denc && (C & 3) == 0) {
        // 16-byte loads: a lane's whT row is C contiguous floats, scalar loads of it are 64 accesses 4 C bytes apart per
        // instruction (this kernel was bound by exactly that)
        float acc = 0.0f;
        const float4 *d4 = reinterpret_cast<const float4 *>(denc + (size_t)m * C), *w4 = reinterpret_cast<const float4 *>(whT + (size_t)k * C);
        constexpr int QB = 4;
        for (int c0 = 0; c0 < C / 4; c0 += QB) {
            float4 dv[QB], wv[QB];
#pragma unroll
            for (int i = 0; i < QB; ++i) { const int c = c0 + i < C / 4 ? c0 + i : C / 4 - 1; dv[i] = d4[c]; wv[i] = w4[c]; }
#pragma unroll
            for (int i = 0; i < QB; ++i)
                if (c0 + i < C / 4) {
                    acc = fmaf(dv[i].x, wv[i].x, acc); acc = fmaf(dv[i].y, wv[i].y, acc);
                    acc = fmaf(dv[i].z, wv[i].z, acc); acc = fmaf(dv[i].w, wv[i].w, acc);
                }
        }
        v += acc;
    } else if (denc) {
        float acc = 0.0f;
        constexpr int CB = 16;
        for (int c0 = 0; c0 < C; c0 += CB) {
            float dv[CB], wv[CB];
#pragma unroll
            for (int i = 0; i < CB; ++i) {
                const int c = c0 + i < C ? c0 + i : C - 1;
                dv[i] = denc[(size_t)m * C + c]; wv[i] = whT[(size_t)k * C + c];
            }
#pragma unroll
            for (int i = 0; i < CB; ++i)
                if (c0 + i < C) acc = fmaf(dv[i], wv[i], acc);
        }
        v += acc;
    }
    out[q] = v;
}

struct SweepScratch {
    float *dh_tot, *dh_pass, *dxh, *dc_alt, *d_in, *tmp_h, *dgrid, *dcell, *d_pooled, *at_u, *at_deh, *at_dself;
    float *st_tot, *st_pass, *st_dxh, *st_dc_alt;
    int32_t *cells, *widx;
    size_t bytes;
};

static void plan_sweep(const tnp_bwd_sweep *a, void *base, SweepScratch &w) {
    const tnp_lstm_model *md = a->model;
    const size_t M = (size_t)a->M, H = md->H;
    const bool grid = md->pool_type >= TNP_POOL_OCCUPANCY && md->pool_type <= TNP_POOL_SOCIAL && !a->nn_pool;
    const bool social = grid && md->pool_type == TNP_POOL_SOCIAL;
    const bool to_hidden = md->pool_type != TNP_POOL_NONE && ((md->variant >> 17) & 1);
    const size_t I = (size_t)md->E + (md->goal_flag ? md->goal_dim : 0) + ((md->pool_type != TNP_POOL_NONE && !to_hidden) ? md->P : 0);
    size_t off = 0;
    auto take = [&](size_t bytes) {
        void *p = base ? (char *)base + off : nullptr;
        off += (bytes + 255) & ~(size_t)255;
        return p;
    };
    w.dh_tot = (float *)take(M * H * 4);
    w.dh_pass = (float *)take(M * H * 4);
    w.dxh = (float *)take(M * (I + H) * 4);
    w.dc_alt = (float *)take(M * H * 4);
    size_t maxw = 0;
    if (grid) for (int l = 1; l < md->n_layers; ++l) maxw = maxw > (size_t)md->dims[l] ? maxw : (size_t)md->dims[l];
    w.d_in = (float *)take(M * maxw * 4);
    const bool hm = a->hidden_mlp != 0;
    const bool at = a->attention != 0;
    w.tmp_h = (float *)take((social || hm || at) ? M * H * 4 : 0);
    w.d_pooled = (float *)take((hm || at) ? M * (size_t)(md->dims[0] + md->dims[1] + md->dims[2]) * 4 : 0);
    w.at_u = (float *)take(at ? M * (size_t)(md->dims[0] + md->dims[1] + md->dims[2] + 4) * 4 : 0);
    w.at_deh = (float *)take(at ? M * (size_t)a->n_max * md->dims[2] * 4 : 0);
    w.at_dself = (float *)take(at ? M * (size_t)md->dims[2] * 4 : 0);
    const bool stf = a->stateful != 0;
    const size_t Hp = stf ? (size_t)md->dims[0] : 0;
    w.st_tot = (float *)take(M * Hp * 4);
    w.st_pass = (float *)take(M * Hp * 4);
    w.st_dxh = (float *)take(stf ? M * ((size_t)md->P + Hp) * 4 : 0);
    w.st_dc_alt = (float *)take(M * Hp * 4);
    w.widx = (int32_t *)take(hm ? M * (size_t)md->dims[2] * 4 : 0);
    const bool dense0 = grid && ((social && !a->social_sparse) || a->directional_in);
    w.dgrid = (float *)take(dense0 ? M * (size_t)md->dims[0] * 4 : 0);
    w.cells = (int32_t *)take(dense0 ? M * (size_t)a->n_max * 4 : 0);
    // eight slice copies for the K-slice-per-XCD data gradient of the first layer
    const size_t dcell_copies = (social && a->social_sparse && dgrid_xcd_ok(md->C, md->dims[1])) ? 8 : 1;
    w.dcell = (float *)take((social && a->social_sparse) ? dcell_copies * M * (size_t)md->dims[0] * 4 : 0);
    w.bytes = off;
}

}  // namespace tnp

extern "C" TNP_API int tnp_pool_embed_weight_layouts(const float *W, int ldw, int N1, int C, int ncell, float *w_cell_major,
                                                     float *w_quad_major, void *stream) {
    if (N1 <= 0 || C <= 0 || ncell <= 0) return 0;
    if (!W || !w_cell_major) TNP_FAIL(-1, "tnp_pool_embed_weight_layouts: NULL pointer");
    if (ldw < C * ncell) TNP_FAIL(-1, "tnp_pool_embed_weight_layouts: ldw %d < C n n = %d", ldw, C * ncell);
    if (w_quad_major && (N1 % 64 != 0 || C % 4 != 0 || (reinterpret_cast<uintptr_t>(w_quad_major) & 15) != 0))
        TNP_FAIL(-1, "tnp_pool_embed_weight_layouts: the quad-major copy needs N1 %% 64 == 0, C %% 4 == 0 and a 16-byte aligned "
                     "buffer (N1 %d, C %d)", N1, C);
    hipLaunchKernelGGL(tnp::weight_layouts_kernel, dim3((ncell + 31) / 32, (N1 + 63) / 64, (C + 3) / 4), dim3(256), 0,
                       (hipStream_t)stream, W, ldw, N1, C, ncell, w_cell_major, w_quad_major);
    TNP_HIP(hipGetLastError());
    return 0;
}

extern "C" TNP_API int tnp_transpose_grouped(const tnp_transpose_problem *problems, int n, void *stream) {
    if (n <= 0) return 0;
    if (!problems) TNP_FAIL(-1, "tnp_transpose_grouped: problems == NULL");
    int done = 0;
    while (done < n) {
        tnp::TransposeGroup g;
        g.count = 0; g.first_block[0] = 0;
        for (; done < n && g.count < TR_MAX_PROBLEMS; ++done) {
            const tnp_transpose_problem &q = problems[done];
            if (q.rows <= 0 || q.cols <= 0) continue;
            if (!q.in || !q.out) TNP_FAIL(-1, "tnp_transpose_grouped: problem %d: NULL pointer", done);
            const int c = g.count++;
            g.in[c] = q.in; g.out[c] = q.out; g.ld_in[c] = q.ld_in; g.ld_out[c] = q.ld_out; g.R[c] = q.rows; g.C[c] = q.cols;
            g.tiles_x[c] = (q.cols + 63) / 64;
            g.first_block[c + 1] = g.first_block[c] + g.tiles_x[c] * ((q.rows + 63) / 64);
        }
        if (g.count == 0) break;
        hipLaunchKernelGGL(tnp::transpose_group_kernel, dim3(g.first_block[g.count]), dim3(256), 0, (hipStream_t)stream, g);
        TNP_HIP(hipGetLastError());
    }
    return 0;
}

extern "C" TNP_API int tnp_transpose(const float *in, int ld_in, int rows, int cols, float *out, int ld_out, void *stream) {
    if (rows <= 0 || cols <= 0) return 0;
    hipLaunchKernelGGL(tnp::transpose_kernel, dim3((cols + 63) / 64, (rows + 63) / 64), dim3(256), 0, (hipStream_t)stream, in,
                       ld_in, rows, cols, out, ld_out);
    TNP_HIP(hipGetLastError());
    return 0;
}

extern "C" TNP_API int tnp_directional_scatter_backward(const float *dgrid, int ldg, const int32_t *cells, const int32_t *winner,
                                                        const int32_t *row_base, const int32_t *row_count,
                                                        const float *obs1, const float *obs2, int M, int n_max, int ncell,
                                                        float *dvel, void *stream) {
    if (M <= 0) return 0;
    hipLaunchKernelGGL(tnp::directional_scatter_backward_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       dgrid, ldg, cells, winner, row_base, row_count, obs1, obs2, M, n_max, ncell, dvel);
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

extern "C" TNP_API int tnp_h2n_cell_backward(const float *h_out, const float *Wn, const float *bn, const float *d_normal,
                                             const float *d_pos, const float *obs1, const float *obs2, const float *dh_in,
                                             const float *gates, const float *c_prev, const float *dc, int M, int H,
                                             float *dlin, float *dG, float *dc_prev, float *dh_pass, void *stream) {
    if (M <= 0) return 0;
    hipLaunchKernelGGL(tnp::h2n_cell_backward_kernel, dim3(M), dim3(64), 0, (hipStream_t)stream, h_out, Wn, bn, d_normal, d_pos,
                       obs1, obs2, dh_in, M, H, dlin, gates, c_prev, dc, dG, dc_prev, dh_pass);
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

// out = nan_to_num(a - b) * scale (torch.nan_to_num's defaults: NaN -> 0, +-inf -> +-FLT_MAX), one launch for the three ATen
// launches the input embedding's operand `nan_to_num(obs2 - obs1) * 4` cost behind every backward sweep
namespace tnp {
__global__ void __launch_bounds__(256) scaled_diff_kernel(const float *__restrict__ a, const float *__restrict__ b, long n, float scale,
                                                          float *__restrict__ out) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float d = a[i] - b[i];
    if (d != d) d = 0.0f;
    else if (d > 3.402823466e+38f) d = 3.402823466e+38f;
    else if (d < -3.402823466e+38f) d = -3.402823466e+38f;
    out[i] = d * scale;
}
}  // namespace tnp

extern "C" TNP_API int tnp_scaled_diff(const float *a, const float *b, long n, float scale, float *out, void *stream) {
    if (n <= 0) return 0;
    if (!a || !b || !out) TNP_FAIL(-1, "tnp_scaled_diff: NULL pointer");
    hipLaunchKernelGGL(tnp::scaled_diff_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, n, scale, out);
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

extern "C" TNP_API int tnp_pair_ego_lists(const int32_t *cells, int R, int M, int n_max, int ncell, uint8_t *occ,
                                          int32_t *occ_t, int32_t *list, int32_t *count, void *stream) {
    if (R <= 0 || ncell <= 0 || n_max <= 0) return 0;
    if (M <= 0 || R % M != 0) TNP_FAIL(-1, "tnp_pair_ego_lists: R = %d is not a multiple of M = %d", R, M);
    hipStream_t s = (hipStream_t)stream;
    TNP_HIP(hipMemsetAsync(occ, 0, (size_t)R * ncell, s));
    const long tot = (long)R * n_max;
    hipLaunchKernelGGL(tnp::pair_occupancy_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, cells, tot, n_max, ncell,
                       occ);
    TNP_HIP(hipGetLastError());
    return tnp::launch_hit_lists(true, occ, nullptr, R, M, ncell, M, occ_t, list, count, s);
}

extern "C" TNP_API int tnp_social_dgrid_cells(const float *dy, int ldy, const float *w_cell_major, const int32_t *list,
                                              const int32_t *count, int R, int step, int M, int C, int ncell, int N1,
                                              float *dcell, void *stream) {
    if (M <= 0 || ncell <= 0) return 0;
    if (N1 % 64 != 0 || ldy % 4 != 0) TNP_FAIL(-1, "tnp_social_dgrid_cells: N1 = %d must be a multiple of 64 (ldy %d of 4)", N1, ldy);
    if (R % M != 0 || step < 0 || step >= R / M) TNP_FAIL(-1, "tnp_social_dgrid_cells: step %d outside the %d-row lists", step, R);
    if (C < 1 || C > 32) TNP_FAIL(-1, "tnp_social_dgrid_cells: C = %d not in 1..32", C);
    const int2 *l2 = reinterpret_cast<const int2 *>(list);
    if (C <= 16) return tnp::launch_dgrid_cells<1>(dy, ldy, w_cell_major, l2, count, R, R / M, step, M, C, ncell, N1, dcell, (hipStream_t)stream);
    return tnp::launch_dgrid_cells<2>(dy, ldy, w_cell_major, l2, count, R, R / M, step, M, C, ncell, N1, dcell, (hipStream_t)stream);
}

extern "C" TNP_API int tnp_social_scatter_backward_cells(const float *dcell, const int32_t *cells, const int32_t *row_base,
                                                         const int32_t *row_count, int M, int n_max, int C, int ncell,
                                                         float *denc, void *stream) {
    if (M <= 0) return 0;
    hipLaunchKernelGGL(tnp::social_scatter_backward_cells_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0,
                       (hipStream_t)stream, dcell, cells, row_base, row_count, M, n_max, C, ncell, denc);
    TNP_HIP(hipGetLastError());
    return 0;
}

extern "C" TNP_API int tnp_sparse_hits_build(const int16_t *winners, const int32_t *row_base, int R, int M, int ncell,
                                             int32_t *hit_t, int32_t *list, int32_t *count, void *stream) {
    if (R <= 0 || ncell <= 0) return 0;
    if (M <= 0 || R % M != 0) TNP_FAIL(-1, "tnp_sparse_hits_build: R = %d is not a multiple of M = %d", R, M);
    return tnp::launch_hit_lists(false, winners, row_base, R, M, ncell, R, hit_t, list, count, (hipStream_t)stream);
}

extern "C" TNP_API int tnp_sparse_wgrad(const float *dy, int ldy, const float *enc, int lde, const int32_t *list,
                                        const int32_t *count, int R, int C, int ncell, int N1, float *dw_cell_major,
                                        void *stream) {
    if (R <= 0 || ncell <= 0) return 0;
    if (N1 % 64 != 0) TNP_FAIL(-1, "tnp_sparse_wgrad: N1 = %d must be a multiple of 64", N1);
    const int2 *l2 = reinterpret_cast<const int2 *>(list);
    hipStream_t s = (hipStream_t)stream;
    switch (C) {
        case 4: return tnp::launch_sparse_wgrad<4>(dy, ldy, enc, lde, l2, count, R, ncell, N1, dw_cell_major, s);
        case 8: return tnp::launch_sparse_wgrad<8>(dy, ldy, enc, lde, l2, count, R, ncell, N1, dw_cell_major, s);
        case 16: return tnp::launch_sparse_wgrad<16>(dy, ldy, enc, lde, l2, count, R, ncell, N1, dw_cell_major, s);
        case 32: return tnp::launch_sparse_wgrad<32>(dy, ldy, enc, lde, l2, count, R, ncell, N1, dw_cell_major, s);
        default: TNP_FAIL(-1, "tnp_sparse_wgrad: C = %d not in {4, 8, 16, 32}", C);
    }
}

extern "C" TNP_API size_t tnp_lstm_backward_scratch_bytes(const tnp_bwd_sweep *a) {
    if (!a || !a->model || a->M <= 0) return 0;
    tnp::SweepScratch w;
    tnp::plan_sweep(a, nullptr, w);
    return w.bytes;
}

#define TNP_RC(expr) do { int rc_ = (expr); if (rc_) return rc_; } while (0)

extern "C" TNP_API int tnp_lstm_backward_sweep(const tnp_bwd_sweep *a, int s_hi, int s_lo, void *scratch,
                                               size_t scratch_bytes, void *stream) {
    if (!a || !a->model || !a->saves) TNP_FAIL(-1, "tnp_lstm_backward_sweep: NULL arguments");
    const tnp_lstm_model *md = a->model;
    const tnp_train_saves *sv = a->saves;
    const int M = a->M, H = md->H, E = md->E, S = a->S;
    if (M <= 0 || S <= 0) return 0;
    if (s_hi >= S || s_lo < 0 || s_lo > s_hi) TNP_FAIL(-1, "tnp_lstm_backward_sweep: steps %d..%d outside 0..%d", s_hi, s_lo, S - 1);
    const bool grid = md->pool_type >= TNP_POOL_OCCUPANCY && md->pool_type <= TNP_POOL_SOCIAL && !a->nn_pool;
    const bool social = grid && md->pool_type == TNP_POOL_SOCIAL;
    const bool hm = a->hidden_mlp != 0 && md->pool_type == TNP_POOL_HIDDENMLP;
    const bool at = a->attention != 0 && md->pool_type == TNP_POOL_ATTNMLP;
    const bool stf = a->stateful != 0 && (md->pool_type == TNP_POOL_NNLSTM || md->pool_type == TNP_POOL_TRAJ);
    if (md->pool_type != TNP_POOL_NONE && !grid && !a->nn_pool && !hm && !at && !stf)
        TNP_FAIL(-1, "tnp_lstm_backward_sweep: pool type %d has no backward", md->pool_type);
    const bool to_hidden = md->pool_type != TNP_POOL_NONE && ((md->variant >> 17) & 1);   // LSTM(pool_to_input=False)
    if (to_hidden && !sv->pvec_all) TNP_FAIL(-1, "tnp_lstm_backward_sweep: pool_to_input=False needs saves->pvec_all");
    if ((social || (grid && a->directional_in)) && !a->cells_all)
        TNP_FAIL(-1, "tnp_lstm_backward_sweep: cells_all (tnp_pool_pair_cells_autograd over the stacked steps) is required for the "
                     "social scatter's backward and for directional input gradients");
    tnp::SweepScratch w;
    tnp::plan_sweep(a, scratch, w);
    if (!scratch || scratch_bytes < w.bytes)
        TNP_FAIL(-1, "tnp_lstm_backward_sweep: scratch too small (need %zu bytes, got %zu)", w.bytes, scratch_bytes);
    hipStream_t s = (hipStream_t)stream;
    const int GD = md->goal_flag ? md->goal_dim : 0;
    const int Pw = md->pool_type != TNP_POOL_NONE ? md->P : 0;
    const int I = E + GD + (to_hidden ? 0 : Pw), LDX = I + H, P0 = E + GD;
    const size_t MH = (size_t)M * H;
    const int nl = md->n_layers, ncell = md->n * md->n, C = md->C;
    float *dc_cur = a->dc, *dc_nxt = w.dc_alt;
    float *dpc_cur = a->st_dpc, *dpc_nxt = w.st_dc_alt;
    for (int st = s_hi; st >= s_lo; --st) {
        const size_t r = (size_t)st * M;
        const float *o1 = sv->obs1_all + r * 2, *o2 = sv->obs2_all + r * 2;
        const float *h_out = (st == a->h_override_step) ? a->h_override : sv->h_all + (size_t)(st + 1) * MH;
        // ---- Hidden2Normal backward + gradient of the new hidden state ----
        // ---- Hidden2Normal backward + LSTMCell backward (absent rows pass the state gradient through) ----
        float *dG = a->dG_all + r * 4 * H;
        if (md->Wn) {
            TNP_RC(tnp_h2n_cell_backward(h_out, md->Wn, md->bn, a->d_rel ? a->d_rel + r * 5 : nullptr,
                                         a->d_pred ? a->d_pred + (r + (size_t)a->pos_offset * M) * 2 : nullptr, o1, o2, a->dh,
                                         sv->gates_all + r * 4 * H, sv->c_all + (size_t)st * MH, dc_cur, M, H, a->dlin_all + r * 5,
                                         dG, dc_nxt, w.dh_pass, stream));
        } else {   // no output head (S-GAN discriminator): the state gradient goes straight into the cell
            TNP_RC(tnp_lstm_cell_backward(sv->gates_all + r * 4 * H, sv->c_all + (size_t)st * MH, a->dh, dc_cur, o1, o2, M, H, dG,
                                          dc_nxt, w.dh_pass, stream));
        }
        const float *wT = st >= a->n_enc ? a->wT_dec : a->wT_enc;
        if (!wT) TNP_FAIL(-1, "tnp_lstm_backward_sweep: transposed cell weights missing for step %d", st);
        // ---- [dG.W_ih | dG.W_hh] -> dxh, and on the same launch's epilogue the input / goal embedding backward (X holds
        // the ReLU outputs) and the ReLU mask of a ReLU-output interaction vector: three masked copies of column ranges of
        // dxh (GemmArgs::seg; a relu_mask3 launch per step before) ----
        const float *Xs = sv->X_all + r * I;
        // gradient and forward value of the interaction vector: pooled columns of X, or (pool_to_input=False) the vector
        // that was added to the hidden operand -- its gradient is the hidden operand's
        const float *pgrad = to_hidden ? w.dxh + I : w.dxh + P0;
        const float *pact = to_hidden ? sv->pvec_all + r * H : Xs + P0;
        const int pact_ld = to_hidden ? H : I;
        {
            tnp::GemmArgs g = {};
            g.A1 = dG; g.lda1 = 4 * H; g.K1 = 4 * H; g.B1 = wT; g.ldb1 = 4 * H;
            g.M = M; g.N = LDX; g.C = w.dxh; g.ldc = LDX;
            int ns = 0;
            g.seg[ns++] = {0, E - 2, Xs, I, a->de_all + r * (E - 2), E - 2};
            if (GD) g.seg[ns++] = {E, GD - 2, Xs + E, I, a->dgoal_all + r * (GD - 2), GD - 2};
            float *pout = a->nn_pool ? a->dnn_all + r * Pw : (grid ? a->dy_all[md->n_layers - 1] + r * Pw : nullptr);
            if (pout) g.seg[ns++] = {to_hidden ? I : P0, Pw, pact, pact_ld, pout, Pw};
            g.nseg = ns;
            TNP_RC(tnp::launch_linear(g, 0, s));
        }
        // ---- grid embedding MLP + scatter + social encoding backward ----
        const float *extra = nullptr, *social_denc = nullptr;
        bool combined = false;                  // a->dh of this step already formed (social_scatter_combine_kernel)
        if (grid) {
            if (a->grid_all) {   // the dense grid is the only intermediate that is recomputed
                TNP_RC(tnp_pool_grid_forward(md->pool_type, o1, o2, social ? sv->enc_all + r * C : nullptr, C, a->scene_start, a->B,
                                             a->n_max, a->scene_slots, md->n, C, md->cell, md->half_x, md->half_y, md->constant,
                                             a->grid_all + r * md->dims[0], md->dims[0], nullptr, stream));
            }
            // (last layer: its ReLU output is the pooled part of X, masked above)
            for (int l = nl - 1; l >= 1; --l) {
                const int n_out = md->dims[l + 1], n_in = md->dims[l];
                // d(act_{l-1}) = (dy_l . W_l) masked by act_{l-1} > 0: the ReLU backward rides on the GEMM's epilogue
                tnp::GemmArgs g = {};
                g.A1 = a->dy_all[l] + r * n_out; g.lda1 = n_out; g.K1 = n_out; g.B1 = a->layT[l]; g.ldb1 = n_out;
                g.M = M; g.N = n_in; g.C = a->dy_all[l - 1] + r * n_in; g.ldc = n_in;
                g.mask_act = sv->act_all[l - 1] + r * n_in; g.ld_mask = n_in;
                TNP_RC(tnp::launch_linear(g, 0, s));
            }
            const int N1 = md->dims[1];
            const float *dy0 = a->dy_all[0] + r * N1;
            if (a->directional_in) {
                TNP_RC(tnp_linear_forward(dy0, N1, a->layT[0], N1, nullptr, w.dgrid, md->dims[0], M, md->dims[0], N1, 0, 0, stream));
                // pair cells as autograd sees them, all steps at once by the caller (tnp_pool_pair_cells_autograd)
                TNP_RC(tnp_directional_scatter_backward(w.dgrid, md->dims[0], a->cells_all + r * a->n_max,
                                                        a->cellwin_all ? a->cellwin_all + r * a->n_max : nullptr, a->row_base,
                                                        a->row_count, o1, o2, M, a->n_max, ncell, a->dvel_pool_all + r * 2, stream));
            }
            if (social) {
                float *denc = a->denc_all + r * C;
                if (a->social_sparse) {   // only the cells that hold a neighbour carry a gradient
                    if (tnp::dgrid_xcd_ok(C, N1) && (reinterpret_cast<uintptr_t>(dy0) & 15) == 0) {
                        TNP_RC(tnp::launch_dgrid_cells_xcd(dy0, N1, a->w_cell_major, reinterpret_cast<const int2 *>(a->ego_list), a->ego_count,
                                                           S * M, S, st, M, C, ncell, N1, w.dcell, s));
                        if ((H == 64 || H == 128) && (C & 3) == 0 && !extra) {      // scatter + state-gradient combine in one launch
                            auto k = H == 64 ? tnp::social_scatter_combine_kernel<1> : tnp::social_scatter_combine_kernel<2>;
                            hipLaunchKernelGGL(k, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, w.dcell, a->cells_all + r * a->n_max,
                                               a->row_base, a->row_count, M, a->n_max, C, ncell, denc, w.dxh + I, LDX, w.dh_pass, a->whT,
                                               a->dh);
                            TNP_HIP(hipGetLastError());
                            combined = true;
                        } else {
                            hipLaunchKernelGGL(tnp::social_scatter_backward_cells8_kernel, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s,
                                               w.dcell, a->cells_all + r * a->n_max, a->row_base, a->row_count, M, a->n_max, C, ncell, denc);
                            TNP_HIP(hipGetLastError());
                        }
                    } else {
                    TNP_RC(tnp_social_dgrid_cells(dy0, N1, a->w_cell_major, a->ego_list, a->ego_count, S * M, st, M, C, ncell, N1,
                                                  w.dcell, stream));
                    TNP_RC(tnp_social_scatter_backward_cells(w.dcell, a->cells_all + r * a->n_max, a->row_base, a->row_count, M,
                                                             a->n_max, C, ncell, denc, stream));
                    }
                } else {
                    TNP_RC(tnp_linear_forward(dy0, N1, a->layT[0], N1, nullptr, w.dgrid, md->dims[0], M, md->dims[0], N1, 0, 0,
                                              stream));
                    TNP_RC(tnp_social_scatter_backward(w.dgrid, md->dims[0], a->cells_all + r * a->n_max, a->row_base, a->row_count,
                                                       M, a->n_max, C, ncell, denc, stream));
                }
                social_denc = denc;     // its contribution to dh (denc . Wh) is formed inside the combine kernel
            }
        }
        if (hm) {   // HiddenStateMLPPooling: out_projection (linear) <- max-pool routing <- embeddings
            const int ms = md->dims[0], mv = md->dims[1], mh = md->dims[2], D = ms + mh + mv, GDm = ms + mv;
            float *dP = a->dy_all[0] + r * Pw;
            TNP_HIP(hipMemcpy2DAsync(dP, (size_t)Pw * 4, pgrad, (size_t)LDX * 4, (size_t)Pw * 4, M, hipMemcpyDeviceToDevice, s));
            TNP_RC(tnp_linear_forward(dP, Pw, a->layT[0], Pw, nullptr, w.d_pooled, D, M, D, Pw, 0, 0, stream));
            float *denc = mh > 0 ? a->denc_all + r * mh : nullptr;
            TNP_RC(tnp_pool_hiddenmlp_backward(o1, o2, mh > 0 ? sv->enc_all + r * mh : nullptr, mh, a->scene_start, a->row_base,
                                               a->row_count, a->B, M, ms, mv, mh, md->Wp[0], md->bp[0], md->Wp[1], md->bp[1],
                                               w.d_pooled, D, a->hm_G_all + r * GDm, a->hm_R_all + r * GDm * 2, denc, w.widx,
                                               a->hm_wslot_all ? a->hm_wslot_all + r * GDm : nullptr, stream));
            if (mh > 0) {
                TNP_RC(tnp_linear_forward(denc, mh, a->whT, mh, nullptr, w.tmp_h, H, M, H, mh, 0, 0, stream));
                extra = w.tmp_h;
            }
        }
        if (stf) {   // NearestNeighborLSTM / TrajectronPooling: hidden2pool (linear) <- pool_lstm (BPTT) <- embedding
            const int Hp = md->dims[0], LP = Pw + Hp;
            const size_t MHp = (size_t)M * Hp;
            float *dP = a->dy_all[0] + r * Pw;
            TNP_HIP(hipMemcpy2DAsync(dP, (size_t)Pw * 4, pgrad, (size_t)LDX * 4, (size_t)Pw * 4, M, hipMemcpyDeviceToDevice, s));
            // gradient of the encoder's new hidden state: through hidden2pool + what the next step handed back
            TNP_RC(tnp_linear_forward(dP, Pw, a->st_h2pT, Pw, nullptr, w.st_pass, Hp, M, Hp, Pw, 0, 0, stream));
            hipLaunchKernelGGL(tnp::state_grad_combine_kernel, dim3((unsigned)((MHp + 255) / 256)), dim3(256), 0, s, w.st_pass, Hp,
                               a->st_dph, nullptr, M, Hp, w.st_tot);
            TNP_HIP(hipGetLastError());
            float *dGp = a->st_dG_all + r * 4 * Hp;
            TNP_RC(tnp_lstm_cell_backward(sv->pgates_all + r * 4 * Hp, sv->pc_all + (size_t)st * MHp, w.st_tot, dpc_cur, a->st_zeros,
                                          a->st_zeros, M, Hp, dGp, dpc_nxt, w.st_pass, stream));
            TNP_RC(tnp_linear_forward(dGp, 4 * Hp, a->st_pwT, 4 * Hp, nullptr, w.st_dxh, LP, M, LP, 4 * Hp, 0, 0, stream));
            TNP_RC(tnp_relu_mask(w.st_dxh, LP, sv->act_all[0] + r * Pw, Pw, M, Pw, a->st_dfeat_all + r * Pw, Pw, stream));
            hipLaunchKernelGGL(tnp::state_grad_combine_kernel, dim3((unsigned)((MHp + 255) / 256)), dim3(256), 0, s, w.st_dxh + Pw, LP,
                               w.st_pass, nullptr, M, Hp, a->st_dph);       // st_pass == 0: every track is updated
            TNP_HIP(hipGetLastError());
            float *t2 = dpc_cur; dpc_cur = dpc_nxt; dpc_nxt = t2;
        }
        if (at) {   // AttentionMLPPooling: Wfin (linear) <- softmax attention over the slots <- embeddings, query path
            const int ms = md->dims[0], mv = md->dims[1], mh = md->dims[2], D = ms + mh + mv, GDm = ms + mv, LU = D + 4;
            float *dP = a->dy_all[0] + r * Pw;
            TNP_HIP(hipMemcpy2DAsync(dP, (size_t)Pw * 4, pgrad, (size_t)LDX * 4, (size_t)Pw * 4, M, hipMemcpyDeviceToDevice, s));
            TNP_RC(tnp_linear_forward(dP, Pw, a->layT[0], Pw, nullptr, w.d_pooled, D, M, D, Pw, 0, 0, stream));       // d ebar
            const float *hpre = mh > 0 ? sv->enc_all + r * mh : nullptr;
            float *es = a->at_eself_all + r * D, *qs = a->at_q_all + r * D;
            TNP_RC(tnp_pool_attn_self(o1, o2, hpre, mh, 1, M, ms, mv, mh, md->bp[0], md->bp[1], md->constant, es, D, stream));
            TNP_RC(tnp_linear_forward(es, D, md->Wx[0], D, md->bx[0], qs, D, M, D, D, 0, 0, stream));                  // q
            TNP_RC(tnp_linear_forward(qs, D, md->Wx[1], D, nullptr, w.at_u, LU, M, LU, D, 0, 0, stream));              // u
            float *dus = a->at_du_all + r * LU;
            TNP_RC(tnp_pool_attn_pair_backward(o1, o2, hpre, mh, a->scene_start, a->B, a->n_max, a->scene_slots, ms, mv, mh, md->Wp[0], md->bp[0],
                                               md->Wp[1], md->bp[1], md->constant, w.at_u, LU, w.d_pooled, D, dus,
                                               a->at_A_all + r * GDm * 3, w.at_deh, a->at_ebar_all + r * D, D,
                                               a->at_posrec_all ? a->at_posrec_all + r * a->n_max * 4 : nullptr, stream));
            float *dqs = a->at_dq_all + r * D;
            TNP_RC(tnp_linear_forward(dus, LU, a->at_WuT, LU, nullptr, dqs, D, M, D, LU, 0, 0, stream));               // dq = Wu^T du
            TNP_RC(tnp_linear_forward(dqs, D, a->at_WqT, D, nullptr, w.d_pooled, D, M, D, D, 0, 0, stream));           // de_self
            float *denc = mh > 0 ? a->denc_all + r * mh : nullptr;
            TNP_RC(tnp_pool_attn_self_backward(o1, o2, hpre, mh, a->row_base, a->row_count, M, a->n_max, ms, mv, mh, md->bp[0],
                                               md->bp[1], w.d_pooled, D, w.at_deh, a->at_A_all + r * GDm * 3, w.at_dself, denc,
                                               stream));
            if (mh > 0) {
                TNP_RC(tnp_linear_forward(denc, mh, a->whT, mh, nullptr, w.tmp_h, H, M, H, mh, 0, 0, stream));
                extra = w.tmp_h;
            }
        }
        if (!combined) {
            hipLaunchKernelGGL(tnp::state_grad_combine_kernel, dim3((unsigned)((MH + 255) / 256)), dim3(256), 0, s, w.dxh + I, LDX,
                               w.dh_pass, extra, M, H, a->dh, social_denc, a->whT, C);
            TNP_HIP(hipGetLastError());
        }
        float *t = dc_cur; dc_cur = dc_nxt; dc_nxt = t;
    }
    if (dc_cur != a->dc) TNP_HIP(hipMemcpyAsync(a->dc, dc_cur, MH * 4, hipMemcpyDeviceToDevice, s));
    if (stf && dpc_cur != a->st_dpc)
        TNP_HIP(hipMemcpyAsync(a->st_dpc, dpc_cur, (size_t)M * md->dims[0] * 4, hipMemcpyDeviceToDevice, s));
    return 0;
}

extern "C" TNP_API size_t tnp_abi_sizeof(int which) {
    switch (which) {
        case 0: return sizeof(tnp_lstm_model);
        case 1: return sizeof(tnp_lstm_extras);
        case 2: return sizeof(tnp_step_saves);
        case 3: return sizeof(tnp_train_saves);
        case 4: return sizeof(tnp_bwd_sweep);
        case 5: return sizeof(tnp_wgrad_problem);
        case 6: return sizeof(tnp_adam_tensor);
        case 7: return sizeof(tnp_transpose_problem);
        default: return 0;
    }
}
