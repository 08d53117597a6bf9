// Interaction-grid build for GridBasedPooling (reference lstm/gridbased_pooling.py:112-170, 227-305), gfx950.
//
// One workgroup (4 waves) handles EGOS egos of one scene; the scene's positions / velocities / per-track
// values are staged once in LDS (the O(N^2) relative-displacement work never touches HBM).  One wave owns one
// ego at a time: its lanes walk the neighbours j, compute the cell with the reference's exact fp32 arithmetic
//      oij = (pos_j - pos_i) / float32(cell_side) + n/2      (IEEE subtract, IEEE divide, IEEE add)
// and resolve collisions with an LDS integer max on  key = 2*j + in_range : the reference's index_put keeps
// the LAST writer in ascending j, and every out-of-range (or absent / padded) neighbour is redirected to
// cell 0 carrying the background constant, so "largest j wins, and if it was out of range the cell holds the
// constant" reproduces index_put exactly, including the cell-0 clobber.
// Output: the dense grid row [C*n*n] (feature = c*n*n + cell) written with coalesced 256-byte wave stores,
// and/or the compact int16 winner table [n*n] that a gather-GEMM can consume instead of the dense grid.
#include "tnp_internal.h"
#include "grid_build_body.h"

namespace tnp {

__global__ void __launch_bounds__(256) grid_build_kernel(const GridArgs a) {
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    grid_build_body(a, blockIdx.x, blockIdx.y, gsm, [&](int start, int ns, float2 *pos, float2 *vel) {
        for (int j = threadIdx.x; j < ns; j += 256) {
            const float2 p2 = reinterpret_cast<const float2 *>(a.obs2)[start + j];
            const float2 p1 = a.type == TNP_POOL_DIRECTIONAL ? reinterpret_cast<const float2 *>(a.obs1)[start + j] : p2;
            grid_stage_track(a.type, p1, p2, j, pos, vel);
        }
    });
}

// limits, LDS bytes and the vec4 flag of a grid launch (shared with the merged prepare + grid launch, lstm_seq.hip)
int grid_launch_plan(const GridArgs &a, GridArgs *planned, size_t *smem_bytes) {
    if (a.n_max > TNP_GRID_MAX_TRACKS)
        TNP_FAIL(-1, "grid pooling: %d tracks in one scene exceed the LDS-staged limit of %d", a.n_max,
                 TNP_GRID_MAX_TRACKS);
    if (a.n > 64) TNP_FAIL(-1, "grid pooling: n=%d cells per side not supported (max 64)", a.n);
    if (a.n_max > 32767) TNP_FAIL(-1, "winner table is int16");
    const int ncell = a.n * a.n;
    size_t smem = (size_t)a.n_max * 16 + (a.type == TNP_POOL_SOCIAL ? (((size_t)a.n_max * a.C + 3) & ~(size_t)3) * 4 : 0) +
                  (size_t)4 * ncell * 4;
    *smem_bytes = (smem + 15) & ~(size_t)15;
    *planned = a;
    planned->vec4 = (ncell % 4 == 0) && (a.ldg % 4 == 0) && a.grid && ((reinterpret_cast<uintptr_t>(a.grid) & 15) == 0);
    return 0;
}

int launch_grid(const GridArgs &a, hipStream_t s) {
    if (a.B <= 0 || a.n_max <= 0) return 0;
    GridArgs b;
    size_t smem = 0;
    const int rc = grid_launch_plan(a, &b, &smem);
    if (rc) return rc;
    static size_t attr = 0;
    if (smem > attr) {
        TNP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(grid_build_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr = smem;
    }
    dim3 grid((a.n_max + TNP_GRID_EGOS - 1) / TNP_GRID_EGOS, a.B);
    hipLaunchKernelGGL(grid_build_kernel, grid, dim3(256), smem, s, b);
    TNP_HIP(hipGetLastError());
    return 0;
}

// GridBasedPooling(pool_size, blur_size) behind the scatter (reference lstm/gridbased_pooling.py:297-304): the fine grid
// [M][C][G][G], G = n pool_size, is blurred by avg_pool2d(blur, stride 1, padding blur / 2, count_include_pad = True) -- a
// (G + 1)^2 map for an even blur -- and reduced by lp_pool2d(p = 1, pool_size) = avg_pool2d(pool_size) * pool_size^2 to
// [M][C][n][n].  ATen's order of operations: window sums row-major in float32, one IEEE divide by the FULL window size (padded
// elements count), the multiply after the second average.  One thread per output element; inference only.
__global__ void __launch_bounds__(256) grid_finish_kernel(const float *fine, int ldf, long total, int C, int n, int ps, int blur,
                                                          float *out, int ldo) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= total) return;
    const int j = (int)(q % n), i = (int)((q / n) % n), ch = (int)((q / ((long)n * n)) % C);
    const long row = q / ((long)n * n * C);
    const int G = n * ps, pad = blur / 2;
    const float *src = fine + (size_t)row * ldf + (size_t)ch * G * G;
    float acc = 0.0f;
    for (int y = i * ps; y < i * ps + ps; ++y)
        for (int x = j * ps; x < j * ps + ps; ++x) {
            float v;
            if (blur == 1) v = src[y * G + x];
            else {
                float b = 0.0f;
                for (int yy = max(y - pad, 0); yy < min(y - pad + blur, G); ++yy)
                    for (int xx = max(x - pad, 0); xx < min(x - pad + blur, G); ++xx) b = __fadd_rn(b, src[yy * G + xx]);
                v = __fdiv_rn(b, (float)(blur * blur));
            }
            acc = __fadd_rn(acc, v);
        }
    float r = acc;
    if (ps > 1) r = __fmul_rn(__fdiv_rn(acc, (float)(ps * ps)), (float)(ps * ps));
    out[(size_t)row * ldo + (size_t)ch * n * n + i * n + j] = r;
}

int launch_grid_finish(const float *fine, int ldf, int M, int C, int n, int pool_size, int blur_size, float *out, int ldo,
                       hipStream_t s) {
    const long total = (long)M * C * n * n;
    if (total <= 0) return 0;
    hipLaunchKernelGGL(grid_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, fine, ldf, total, C, n, pool_size,
                       blur_size, out, ldo);
    TNP_HIP(hipGetLastError());
    return 0;
}

// Cell of every ordered pair (ego, neighbour slot) for the scatter's backward pass: the reference's autograd gives
// EVERY in-range neighbour the gradient of its cell, overwritten duplicates included.
// out[row][j] = cell id of neighbour j of the row's scene as seen from the ego, -1 if j is the ego itself, absent,
// out of range or beyond the scene.  One thread per (row, j).
__global__ void pair_cells_kernel(const float *obs2, const int32_t *row_base, const int32_t *row_count, int M, int n_max,
                                  int G, float cell, float half_x, float half_y, int32_t *out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= M * n_max) return;
    const int row = q / n_max, j = q - row * n_max;
    const int base = row_base[row], ns = row_count[row];
    int res = -1;
    if (j < ns && base + j != row) {
        float2 pi = reinterpret_cast<const float2 *>(obs2)[row], pj = reinterpret_cast<const float2 *>(obs2)[base + j];
        if (pi.x != pi.x || pi.y != pi.y) { pi.x = -500.0f; pi.y = -500.0f; }
        if (pj.x != pj.x || pj.y != pj.y) { pj.x = -500.0f; pj.y = -500.0f; }
        const float ox = (pj.x - pi.x) / cell + half_x, oy = (pj.y - pi.y) / cell + half_y;
        const float fG = (float)G;
        if (!(ox < 0.0f) && !(ox >= fG) && !(oy < 0.0f) && !(oy >= fG)) res = (int)ox * G + (int)oy;
    }
    out[q] = res;
}

// What autograd lets through the scatter (reference gridbased_pooling.py:290-304), given the raw pair cells above.  Two
// rules on top of "every in-range neighbour receives the gradient of its cell":
//   * the grid leaves occupancy() through lp_pool2d(x, 1, 1) = sign(x) * relu(|x|) (torch 2.x), whose derivative at
//     x == 0 is ZERO: a cell whose final value is exactly 0 passes no gradient.  That is the case of cell (0, 0) whenever its
//     last writer is an out-of-range / absent / padded neighbour (they all write `constant` there) and
//     constant == 0: the genuine neighbours standing in cell (0, 0) then get NOTHING (found by the full-size training pin of
//     round 4: 0.2 % of the pairs of a 32-agent crowd, 13 % of the hidden_dim_encoding gradient after cancellation);
//   * the value that decides is the WINNER's (last writer in ascending j): `winner[row][j]` = slot of the winner of pair
//     (row, j)'s cell, for consumers that need its value (the directional grid masks per channel where the winner's
//     relative velocity is exactly 0, e.g. a winner without a finite velocity).
// cells[row][j] = raw cell, or -1 when the pair gets no gradient; winner = -2: the cell holds a non-zero `constant` (gradient
// passes, there is no winner value to test).  One thread per (row, j); raw and cells are different buffers (a thread reads
// the raw cells of the other slots of its row).
__global__ void pair_cells_mask_kernel(const int32_t *__restrict__ raw, const int32_t *__restrict__ row_base,
                                       const int32_t *__restrict__ row_count, const int32_t *__restrict__ row_padded,
                                       int pad_default, int M, int n_max, int clobber_blocks, int32_t *__restrict__ cells,
                                       int32_t *__restrict__ winner) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= M * n_max) return;
    const int row = q / n_max, j = q - row * n_max;
    const int c = raw[q];
    int res = c, win = -1;
    if (c > 0 && !winner) { cells[q] = c; return; }                  // only cell 0 can be clobbered; no winner wanted: done
    if (c >= 0) {
        const int base = row_base[row], ns = row_count[row], self = row - base;
        const int pad = row_padded ? row_padded[row] : pad_default;
        win = j;
        bool clobbered = (c == 0) && pad > ns;                       // a padded slot has the highest index and is absent
        if (!clobbered) {
            for (int k = ns - 1; k > j; --k) {                       // the LAST writer of the cell decides: scan from the top
                if (k == self) continue;
                const int ck = raw[(size_t)row * n_max + k];
                if (ck == c) { win = k; break; }
                if (c == 0 && ck < 0) { clobbered = true; break; }   // out-of-range / absent: writes `constant` to cell 0
            }
        }
        if (clobbered) {
            if (clobber_blocks) { res = -1; win = -1; }              // constant == 0: the cell's value is 0, no gradient
            else win = -2;                                           // the cell holds `constant` != 0: gradient passes, no winner value
        }
    }
    cells[q] = res;
    if (winner) winner[q] = win;
}

__global__ void mark_primaries_kernel(const int32_t *scene_start, int B, uint8_t *flag) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < B && scene_start[s + 1] > scene_start[s]) flag[scene_start[s]] = 1;
}

}  // namespace tnp

extern "C" TNP_API int tnp_mark_primaries(const int32_t *scene_start, int B, int M, uint8_t *primary_flag, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    TNP_HIP(hipMemsetAsync(primary_flag, 0, (size_t)M, s));
    if (B > 0) {
        hipLaunchKernelGGL(tnp::mark_primaries_kernel, dim3((B + 255) / 256), dim3(256), 0, s, scene_start, B,
                           primary_flag);
        TNP_HIP(hipGetLastError());
    }
    return 0;
}

extern "C" TNP_API int tnp_pool_pair_cells(const float *obs2, const int32_t *row_base, const int32_t *row_count, int M,
                                           int n_max, int n, float cell, float half_x, float half_y, int32_t *out,
                                           void *stream) {
    if (M <= 0 || n_max <= 0) return 0;
    const int tot = M * n_max;
    hipLaunchKernelGGL(tnp::pair_cells_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, obs2, row_base,
                       row_count, M, n_max, n, cell, half_x, half_y, out);
    TNP_HIP(hipGetLastError());
    return 0;
}

extern "C" TNP_API int tnp_pool_pair_cells_autograd(const float *obs2, const int32_t *row_base, const int32_t *row_count,
                                                    const int32_t *row_padded, int pad_default, int M, int n_max, int n,
                                                    float cell, float half_x, float half_y, float constant,
                                                    int32_t *raw_scratch, int32_t *cells, int32_t *winner, void *stream) {
    if (M <= 0 || n_max <= 0) return 0;
    if (!raw_scratch || !cells || raw_scratch == cells) TNP_FAIL(-1, "tnp_pool_pair_cells_autograd: raw_scratch and cells must be two buffers");
    const int tot = M * n_max;
    hipLaunchKernelGGL(tnp::pair_cells_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, obs2, row_base,
                       row_count, M, n_max, n, cell, half_x, half_y, raw_scratch);
    TNP_HIP(hipGetLastError());
    hipLaunchKernelGGL(tnp::pair_cells_mask_kernel, dim3((tot + 255) / 256), dim3(256), 0, (hipStream_t)stream, raw_scratch,
                       row_base, row_count, row_padded, pad_default, M, n_max, constant == 0.0f ? 1 : 0, cells, winner);
    TNP_HIP(hipGetLastError());
    return 0;
}

extern "C" TNP_API int tnp_pool_grid_forward(int type, const float *obs1, const float *obs2, const float *values, int ldv,
                                     const int32_t *scene_start, int B, int n_max, const int32_t *scene_slots, int n, int C,
                                     float cell, float half_x, float half_y, float constant, float *grid, int ldg,
                                     int16_t *winners, void *stream) {
    if (type < TNP_POOL_OCCUPANCY || type > TNP_POOL_SOCIAL) TNP_FAIL(-1, "unknown pooling type %d", type);
    if (type == TNP_POOL_SOCIAL && values == nullptr) TNP_FAIL(-1, "social pooling needs per-track values");
    tnp::GridArgs a;
    a.obs1 = obs1; a.obs2 = obs2; a.values = values; a.ldv = ldv; a.scene_start = scene_start;
    a.B = B; a.n_max = n_max; a.scene_slots = scene_slots; a.type = type; a.n = n; a.C = C;
    a.cell = cell; a.half_x = half_x; a.half_y = half_y; a.constant = constant;
    a.grid = grid; a.ldg = ldg; a.winners = winners;
    return tnp::launch_grid(a, (hipStream_t)stream);
}
