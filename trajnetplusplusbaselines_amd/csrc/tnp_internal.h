// Internal declarations shared by the HIP translation units of libtrajnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/trajnet_hip.h"
#include "../../include/trajnet_hip_profile.h"

namespace tnp {

// ---- error handling -------------------------------------------------------------------------
void set_error(const char *fmt, ...);
#define TNP_FAIL(code, ...) do { ::tnp::set_error(__VA_ARGS__); return (code); } while (0)
#define TNP_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    ::tnp::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); return -2; } } while (0)

// ---- tile-selection knobs (tnp_tuning_set; initial values from the environment, read once) ------------------------------
struct Tuning {
    int sparse_te, sparse_ncs;   // pinned tile of the sparse first layer (0 = automatic)          TNP_SPARSE_TILE="te,ncs"
    long sparse_min_wg;          // workgroups a tile must reach before a smaller one is tried (0 = one per CU)   TNP_SPARSE_MIN_WG
    int skinny_max_rows;         // rows up to which gemm_skinny.hip takes the dense layers                        TNP_SKINNY_MAX_M
    int skinny_gates_max_rows;   // ... and the LSTM gates                                                          TNP_SKINNY_GATES_MAX_M
    int sparse_wgrad_plan;       // sparse first layer's weight gradient: 16 * waves + batches per trip (0 = by size)  TNP_SPARSE_WGRAD_PLAN
    int wgrad_min_rows;          // dense weight gradients: fewest rows of K per split ...                          TNP_WGRAD_MIN_ROWS
    int wgrad_target_wgs;        // ... and the workgroup count a contraction is split towards                       TNP_WGRAD_TARGET
    int fuse_prepare_grid;       // occupancy / directional grids: track_prepare + grid in one launch (0 / 1 by size / 2) TNP_FUSE_PREPARE_GRID
};
Tuning &tuning();

// ---- GEMM on the matrix cores ---------------------------------------------------------------
enum { EPI_BIAS = 0, EPI_LSTM = 1 };

struct GemmArgs {
    // A = [A1 | A2] along K, row-major, K-contiguous
    const float *A1; int lda1; int K1;
    const float *A2; int lda2; int K2;
    // W = [B1 | B2] along K, PyTorch layout [N, K]
    const float *B1; int ldb1;
    const float *B2; int ldb2;
    const float *bias1; const float *bias2;
    int M, N;
    int vec_ok;           // all K / ld multiples of 4 and pointers 16-byte aligned
    int prio;             // pipelined kernel: static s_setprio for the second half of the waves
    int tiles_m, tiles_n;
    // EPI_BIAS
    float *C; int ldc; int relu;
    // EPI_LSTM: N == 4*H, tile columns = 4 gates x 32 hidden units
    int H; const float *h_in; float *h_out; const float *c_in; float *c_out; const uint8_t *mask;
    float *gates_out;     // optional [M, 4H]: post-activation i, f, g, o of the present rows (training saves)
    // EPI_BIAS, optional: C[row, col] = mask_act[row, col] > 0 ? value : 0 -- the ReLU backward of the layer below fused into
    // the data-gradient GEMM (what a separate relu_mask launch did)
    const float *mask_act; int ld_mask;
    // EPI_BIAS, optional: up to three masked COPIES of column ranges of the result (the value itself still goes to C):
    // seg[i].out[row, col - col0] = seg[i].act[row, col - col0] > 0 ? value : 0 for col0 <= col < col0 + n -- the ReLU
    // backwards of the input / goal embeddings and of the interaction vector, which read the data-gradient GEMM's output
    // (lstm_bwd.hip: what a relu_mask3 launch did after it)
    struct EpiSeg { int col0, n; const float *act; int ld_act; float *out; int ld_out; };
    EpiSeg seg[3]; int nseg;
};

// generic dense layer (variant selects the tile configuration)
int launch_linear(const GemmArgs &g, int variant, hipStream_t s);
// fused LSTM gates GEMM + cell update
int launch_lstm_gates(const GemmArgs &g, int variant, hipStream_t s);
// small batches (gemm_skinny.hip): 16-track tiles, operands straight into registers; variants 40 .. 45 / 30 .. 34
bool skinny_ok(const GemmArgs &g);
int launch_skinny_linear(const GemmArgs &g, int variant, hipStream_t s);
int launch_skinny_gates(const GemmArgs &g, int variant, hipStream_t s);
// ---- grid pooling ---------------------------------------------------------------------------
struct GridArgs {
    const float *obs1, *obs2;
    const float *values; int ldv;
    const int32_t *scene_start;
    int B, n_max, type, n, C;
    const int32_t *scene_slots;   // [B] padded slot count per scene, NULL = n_max for every scene
    float cell, half_x, half_y, constant;
    float *grid; int ldg;
    int16_t *winners;
    int vec4;  // set by launch_grid
};
int launch_grid(const GridArgs &a, hipStream_t s);
int grid_launch_plan(const GridArgs &a, GridArgs *planned, size_t *smem_bytes);   // checks, LDS bytes, vec4 flag
// pool_size / blur_size reduction of a fine grid [M][C][(n ps)^2] to [M][C][n^2] (lstm/gridbased_pooling.py:297-304)
int launch_grid_finish(const float *fine, int ldf, int M, int C, int n, int pool_size, int blur_size, float *out, int ldo,
                       hipStream_t s);

// ---- sparse pooling embedding (first MLP layer on the winner table) ---------------------------
bool sparse_supported(int C, int N1, int ncell);
size_t sparse_partial_bytes(int M, int N1, int ncell);
int launch_row_base(const int32_t *scene_start, int B, int32_t *row_base, hipStream_t s, int32_t *row_end = nullptr,
                    int32_t *row_padded = nullptr, const int32_t *scene_slots = nullptr, int n_max = 0);
// optional: build the winner tile inside the cell-split kernel from the positions (no grid kernel, no winner table)
struct SparseGridFuse {
    const float *obs2; const int32_t *row_end; const int32_t *row_padded; int G; float cell, half_x, half_y; int16_t *winners_out;
};
bool sparse_fuses_grid(int ncell, int n_max);
bool sparse_uses_regacc(int M, int ldv, int C, int ncell);   // launch_pool_embed_sparse will run pool_embed_regacc_kernel
int launch_pool_embed_sparse(const int16_t *winners, const float *enc, int ldv, const int32_t *row_base,
                             const float *Wp, const float *bias, int M, int ncell, int C, int N1, int relu,
                             float *out, int ldo, float *partial, hipStream_t s, const SparseGridFuse *fg = nullptr,
                             const float *Wq = nullptr);   // Wq: quad-major copy of the weights (tnp_lstm_model.Wp0_quad_major)

// ---- non-grid interaction modules (pool_nongrid.hip) -----------------------------------------
int launch_pool_nn(const float *obs1, const float *obs2, const int32_t *scene_start, int B, int n_sel, int in_dim,
                   const float *W, const float *bias, int d, float *out, int ldo, hipStream_t s, float *attrs = nullptr,
                   int n_max_hint = 0);   // largest scene if the caller knows it (0: unknown -> the lane-per-ego kernel)
int launch_pool_hiddenmlp(const float *obs1, const float *obs2, const float *henc, int ldh, int henc_relu,
                          const int32_t *scene_start, int B, int ms, int mv, int mh, const float *Ws, const float *bs,
                          const float *Wv, const float *bv, float *pooled, int ldp, hipStream_t s);

int launch_pool_attn_self(const float *obs1, const float *obs2, const float *henc, int ldh, int henc_relu, int M, int ms,
                          int mv, int mh, const float *bs, const float *bv, float fill, float *e_self, int lde,
                          hipStream_t s);
int launch_pool_attn_pair(const float *obs1, const float *obs2, const float *henc, int ldh, int henc_relu,
                          const int32_t *scene_start, int B, int n_max, const int32_t *scene_slots, int ms, int mv, int mh,
                          const float *Ws, const float *bs, const float *Wv, const float *bv, float fill, const float *u,
                          int ldu, float *ebar, int lde, hipStream_t s);

int launch_pool_traj(const float *obs1, const float *obs2, int M, const float *W, const float *bias, int P, float *out,
                     int ldo, double *scratch4, hipStream_t s, float *inputs = nullptr);

// ---- per-primary loss value (lstm/loss.py:23-91 / :107-135), shared by loss.hip and the sequence driver's fused form ----
__device__ __forceinline__ float gaussian_2d_dev(float mu1, float mu2, float s1, float s2, float rho, float x1, float x2) {
    // lstm/loss.py:23-50, same operation order
    const float norm1 = x1 - mu1, norm2 = x2 - mu2;
    const float s1s2 = s1 * s2;
    const float q1 = norm1 / s1, q2 = norm2 / s2;
    const float z = q1 * q1 + q2 * q2 - 2.0f * rho * norm1 * norm2 / s1s2;
    const float omr = 1.0f - rho * rho;
    const float num = expf(-z / (2.0f * omr));
    const float den = 6.283185307179586f * s1s2 * sqrtf(omr);
    return num / den;
}
// mode 0 = PredictionLoss (NLL with flat background), 1 = L2 (sum of the two squared errors)
__device__ __forceinline__ float primary_loss_value(int mode, float n0, float n1, float n2, float n3, float n4, float tx, float ty,
                                                    float bg) {
    if (mode == 0) {
        const float g_bg = gaussian_2d_dev(n0, n1, 3.0f, 3.0f, 0.0f, tx, ty);   // :73-76
        const float g = gaussian_2d_dev(n0, n1, n2, n3, n4, tx, ty);
        return -logf(0.01f + bg * g_bg + (0.99f - bg) * g);                     // :78-82
    }
    const float d0 = n0 - tx, d1 = n1 - ty;
    return d0 * d0 + d1 * d1;
}

// ---- profiling hook -------------------------------------------------------------------------
enum { PROF_GEMM1 = 0, PROF_ALL_GEMM = 1 };
void prof_before(int cls, hipStream_t s);
void prof_after(int cls, hipStream_t s);
// Events carried by the DISPATCH itself (hipExtLaunchKernelGGL start / stop events = the begin / end timestamps of the kernel's
// AQL packet, what rocprofv3's kernel trace reports) instead of two hipEventRecord calls around it, whose own markers add ~3 us
// to a bracketed launch.  prof_dispatch_events reserves the next event pair when profiling of `cls` is on; a launcher that
// supports it takes the pair with take_dispatch_events() and clears it.
bool prof_dispatch_events(int cls);
// a reserved pair that no launcher took (the launch went down a kernel path without dispatch events): give the slot back, so that
// tnp_profile_read never meets unrecorded events and the next launch does not inherit a stale pair.  Returns true if it did.
bool prof_dispatch_release();
bool take_dispatch_events(hipEvent_t *start, hipEvent_t *stop);

}  // namespace tnp
