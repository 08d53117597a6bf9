// First embedding layer of an occupancy / directional grid with the GRID BUILT IN THE KERNEL (reference
// lstm/gridbased_pooling.py:112-140, 227-305 + the first Linear of :308-335), gfx950.
//
// The two-launch form writes every ego's dense grid row to HBM (config 3: 16384 tracks x 288 floats = 18.9 MB per recurrent
// step) for the GEMM to read it back, and pays one more dependent launch per step (5.6 us at 2048 tracks, 12.5 us at 16384).
// Here a workgroup owns 16 egos x 128 output columns:
//   * prologue: the winner key of every (ego, cell) by LDS integer max over the ego's scene -- the reference's exact fp32 cell
//     arithmetic and "last writer in ascending j wins, out-of-range / absent / padded neighbours clobber cell 0", the
//     expressions of grid_build_kernel (pool_grid.hip) -- then the 16 grid rows as an LDS tile [16][K + 4] (K = C n^2);
//   * main loop: v_mfma_f32_16x16x4_f32 as in gemm_skinny.hip -- one wave per 16-column tile, its weight rows straight from
//     global memory into registers (float4 chunks in MFMA operand layout), the grid rows from the LDS tile, two accumulator
//     chains, no barrier;
//   * epilogue: bias + ReLU, the 16 x 16 tile into the pooled columns of the LSTM input.
// The two column-block workgroups of an ego tile (N = 256) both build the tile (16 egos x <= n_max pairs: cheaper than the
// launch and the grid's round trip through HBM).  The backward pass recomputes the dense grid itself (lstm_bwd.hip), so the
// training forward runs this kernel as well.
#include "tnp_internal.h"

namespace tnp {

typedef float gl_f32x4 __attribute__((ext_vector_type(4)));

struct GridLinearArgs {
    const float *obs1, *obs2;                                  // [M][2]
    const int32_t *row_base, *row_end, *row_padded;            // [M]: first row / one past the last row / padded slots of the row's scene
    int M, type, n, C;
    float cell, half_x, half_y, constant;
    const float *W; int ldw; const float *bias;                // [N][K] PyTorch layout, K = C n n
    int N, relu;
    float *out; int ldo;
    int tiles_n;
};

__device__ __forceinline__ float gl_nan_to_num(float v) {
    if (v != v) return 0.0f;
    if (__builtin_isinf(v)) return v > 0.0f ? 3.402823466e+38f : -3.402823466e+38f;
    return v;
}

constexpr int GL_EGOS = 16, GL_WAVES = 8, GL_MAXC = 8;

__global__ void __launch_bounds__(64 * GL_WAVES) grid_linear_kernel(const GridLinearArgs a) {
    extern __shared__ __attribute__((aligned(16))) float glsm[];
    const int G = a.n, ncell = G * G, K = a.C * ncell, KS = K + 4;
    float *At = glsm;                                           // [16][K + 4]
    int *keys = reinterpret_cast<int *>(At + GL_EGOS * KS);     // [16][ncell]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tn = blockIdx.x % a.tiles_n, tm = blockIdx.x / a.tiles_n;
    const int r0 = tm * GL_EGOS;
    const int l16 = lane & 15, kq = lane >> 4;

    // this wave's weight rows, requested as early as possible: chunk c of 16 k, element j of the float4 = operand of MFMA j
    const int col = min(tn * 16 * GL_WAVES + wave * 16 + l16, a.N - 1);
    const float *wrow = a.W + (size_t)col * a.ldw + 4 * kq;
    const int nch = K >> 4;
    float bias = 0.0f;
    if (a.bias) bias = a.bias[col];

    for (int i = tid; i < GL_EGOS * ncell; i += 64 * GL_WAVES) keys[i] = -1;
    __syncthreads();
    const float fG = (float)G;
    for (int e = wave; e < GL_EGOS; e += GL_WAVES) {            // votes: lanes over the neighbours of the ego's scene
        const int row = r0 + e;
        if (row >= a.M) continue;
        const int lo = a.row_base[row], ns = a.row_end[row] - lo, pad = a.row_padded[row], ki = row - lo;
        float2 pi = reinterpret_cast<const float2 *>(a.obs2)[row];
        if (pi.x != pi.x || pi.y != pi.y) { pi.x = -500.0f; pi.y = -500.0f; }                     // :247-249 sentinel
        for (int j = lane; j < ns; j += 64) {
            if (j == ki) continue;
            float2 pj = reinterpret_cast<const float2 *>(a.obs2)[lo + j];
            if (pj.x != pj.x || pj.y != pj.y) { pj.x = -500.0f; pj.y = -500.0f; }
            const float ox = __fadd_rn(__fdiv_rn(__fsub_rn(pj.x, pi.x), a.cell), a.half_x);       // :276
            const float oy = __fadd_rn(__fdiv_rn(__fsub_rn(pj.y, pi.y), a.cell), a.half_y);
            const bool inr = !(ox < 0.0f) && !(ox >= fG) && !(oy < 0.0f) && !(oy >= fG);
            const int cellid = inr ? ((int)ox * G + (int)oy) : 0;
            atomicMax(&keys[e * ncell + cellid], 2 * j + (inr ? 1 : 0));
        }
        if (ns < pad && lane == 0) atomicMax(&keys[e * ncell], 2 * (pad - 1));                    // padded slots: absent, highest j, cell 0
    }
    __syncthreads();
    for (int i = tid; i < GL_EGOS * K; i += 64 * GL_WAVES) {    // the grid rows (feature = channel * n^2 + cell)
        const int e = i / K, f = i - e * K;
        const int c = f / ncell, cellid = f - c * ncell;
        const int row = r0 + e;
        float v = 0.0f;
        if (row < a.M) {
            const int w = keys[e * ncell + cellid];
            v = a.constant;
            if (w >= 0 && (w & 1)) {
                if (a.type == TNP_POOL_OCCUPANCY) v = 1.0f;                                        // :266-267
                else {                                                                             // directional, :127-140
                    const int lo = a.row_base[row], j = lo + (w >> 1);
                    const float2 j2 = reinterpret_cast<const float2 *>(a.obs2)[j], j1 = reinterpret_cast<const float2 *>(a.obs1)[j];
                    const float2 i2 = reinterpret_cast<const float2 *>(a.obs2)[row], i1 = reinterpret_cast<const float2 *>(a.obs1)[row];
                    const float vj = c == 0 ? __fsub_rn(j2.x, j1.x) : __fsub_rn(j2.y, j1.y);
                    const float vi = c == 0 ? __fsub_rn(i2.x, i1.x) : __fsub_rn(i2.y, i1.y);
                    v = gl_nan_to_num(__fsub_rn(vj, vi));
                }
            }
        }
        At[e * KS + f] = v;
    }
    __syncthreads();

    gl_f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    const float *arow = At + l16 * KS + 4 * kq;
    for (int c0 = 0; c0 < nch; c0 += GL_MAXC) {
        gl_f32x4 wv[GL_MAXC];
#pragma unroll
        for (int i = 0; i < GL_MAXC; ++i) wv[i] = *reinterpret_cast<const gl_f32x4 *>(wrow + min(c0 + i, nch - 1) * 16);
#pragma unroll
        for (int i = 0; i < GL_MAXC; ++i) {
            if (c0 + i < nch) {
                const gl_f32x4 av = *reinterpret_cast<const gl_f32x4 *>(arow + (c0 + i) * 16);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, wv[i].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, wv[i].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, wv[i].z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, wv[i].w, acc1, 0, 0, 0);
            }
        }
    }
    const gl_f32x4 acc = acc0 + acc1;
    const int ocol = tn * 16 * GL_WAVES + wave * 16 + l16;
    if (ocol >= a.N) return;
#pragma unroll
    for (int r = 0; r < 4; ++r) {                               // accumulator register r of lane (l16, kq) = D[row 4 kq + r][column l16]
        const int row = r0 + 4 * kq + r;
        if (row >= a.M) continue;
        float v = acc[r] + bias;
        if (a.relu) v = v > 0.0f ? v : 0.0f;
        a.out[(size_t)row * a.ldo + ocol] = v;
    }
}

static size_t grid_linear_smem(int n, int C) {
    const size_t ncell = (size_t)n * n, K = (size_t)C * ncell;
    return (size_t)GL_EGOS * (K + 4) * 4 + (size_t)GL_EGOS * ncell * 4;
}

// occupancy / directional grids whose row is a whole number of 16-float chunks and fits the LDS tile
bool grid_linear_supported(int type, int n, int C, const float *W, int ldw, int n_max) {
    if (type != TNP_POOL_OCCUPANCY && type != TNP_POOL_DIRECTIONAL) return false;
    const int K = C * n * n;
    if (K % 16 != 0 || ldw % 4 != 0 || (reinterpret_cast<uintptr_t>(W) & 15) != 0 || n_max > 32767) return false;
    return grid_linear_smem(n, C) <= (size_t)120 * 1024;
}

int launch_grid_linear(const float *obs1, const float *obs2, const int32_t *row_base, const int32_t *row_end, const int32_t *row_padded,
                       int M, int type, int n, int C, float cell, float half_x, float half_y, float constant, const float *W, int ldw,
                       const float *bias, int N, int relu, float *out, int ldo, hipStream_t s) {
    if (M <= 0) return 0;
    GridLinearArgs a;
    a.obs1 = obs1; a.obs2 = obs2; a.row_base = row_base; a.row_end = row_end; a.row_padded = row_padded;
    a.M = M; a.type = type; a.n = n; a.C = C; a.cell = cell; a.half_x = half_x; a.half_y = half_y; a.constant = constant;
    a.W = W; a.ldw = ldw; a.bias = bias; a.N = N; a.relu = relu; a.out = out; a.ldo = ldo;
    a.tiles_n = (N + 16 * GL_WAVES - 1) / (16 * GL_WAVES);
    const size_t smem = grid_linear_smem(n, C);
    int dev = 0;
    TNP_HIP(hipGetDevice(&dev));
    static size_t attr[64] = {0};
    if (dev >= 64 || smem > attr[dev]) {
        TNP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(grid_linear_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        if (dev < 64) attr[dev] = smem;
    }
    const int blocks = ((M + GL_EGOS - 1) / GL_EGOS) * a.tiles_n;
    hipLaunchKernelGGL(grid_linear_kernel, dim3(blocks), dim3(64 * GL_WAVES), smem, s, a);
    TNP_HIP(hipGetLastError());
    return 0;
}

}  // namespace tnp
