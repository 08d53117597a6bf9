// Sequence driver of trajnetbaselines.lstm.LSTM on MI355X (reference lstm/lstm.py:91-264), gfx950 only.
//
// The reference keeps per-track hidden state as Python lists and re-stacks them every step; here the state is
// dense [M,H] fp32 in HBM with a per-step presence mask, and one recurrent step is four launches:
//
//   track_prepare   per track: finish step s-1 (Hidden2Normal of h, predicted position), select obs1/obs2 for
//                   step s (observed / teacher-forced / own prediction, primary rows patched with the model's
//                   prediction, lstm.py:240-250), presence mask, InputEmbedding (+goal embedding) into X[:,0:E],
//                   and the social hidden-state encoding Linear(H->C) of the PREVIOUS-step hidden state.
//   grid_build      pool_grid.hip
//   linear x L      grid embedding MLP on the matrix cores (gemm_f32_mfma.hip), last layer writes X[:,E:]
//   lstm_gates      [X | h] @ [W_ih | W_hh]^T + LSTMCell pointwise + masked state update (h ping-pong)
//
// All launches go to the caller's stream; nothing synchronises or allocates (graph capturable).
#include "tnp_internal.h"
#include "grid_build_body.h"
#define TNP_MAX_DEVICES_SEQ 64

#include <math.h>
#include <string.h>
#include <stdlib.h>
#include <string>

namespace tnp {

Tuning &tuning() {
    static Tuning t = [] {
        Tuning v = {0, 0, 0, 160, 512, 0, 128, 512, 1};
        const char *e = getenv("TNP_SPARSE_TILE");
        if (e && sscanf(e, "%d,%d", &v.sparse_te, &v.sparse_ncs) != 2) v.sparse_te = v.sparse_ncs = 0;
        if ((e = getenv("TNP_SPARSE_MIN_WG")) != nullptr) v.sparse_min_wg = atol(e);
        if ((e = getenv("TNP_SKINNY_MAX_M")) != nullptr) v.skinny_max_rows = v.skinny_gates_max_rows = atoi(e);
        if ((e = getenv("TNP_SKINNY_GATES_MAX_M")) != nullptr) v.skinny_gates_max_rows = atoi(e);
        if ((e = getenv("TNP_SPARSE_WGRAD_PLAN")) != nullptr) v.sparse_wgrad_plan = atoi(e);
        if ((e = getenv("TNP_WGRAD_MIN_ROWS")) != nullptr && atoi(e) > 0) v.wgrad_min_rows = atoi(e);
        if ((e = getenv("TNP_WGRAD_TARGET")) != nullptr && atoi(e) > 0) v.wgrad_target_wgs = atoi(e);
        if ((e = getenv("TNP_FUSE_PREPARE_GRID")) != nullptr) v.fuse_prepare_grid = atoi(e);
        return v;
    }();
    return t;
}

static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- profiling hook (bench.py roofline leg) -------------------------------------------------
static int g_prof_cls = -1;
static const int PROF_MAX = 32768;
static hipEvent_t g_prof_ev[2 * PROF_MAX];
static int g_prof_n = 0, g_prof_created = 0;
void prof_before(int cls, hipStream_t s) {
    if (g_prof_cls < 0 || (g_prof_cls != cls && g_prof_cls != PROF_ALL_GEMM) || g_prof_n >= PROF_MAX) return;
    while (g_prof_created < 2 * (g_prof_n + 1)) (void)hipEventCreate(&g_prof_ev[g_prof_created++]);
    (void)hipEventRecord(g_prof_ev[2 * g_prof_n], s);
}
void prof_after(int cls, hipStream_t s) {
    if (g_prof_cls < 0 || (g_prof_cls != cls && g_prof_cls != PROF_ALL_GEMM) || g_prof_n >= PROF_MAX) return;
    (void)hipEventRecord(g_prof_ev[2 * g_prof_n + 1], s);
    ++g_prof_n;
}
static hipEvent_t g_disp_ev[2] = {nullptr, nullptr};
static int g_prof_disp_n = 0, g_prof_disp_last = 0;
bool prof_dispatch_events(int cls) {
    if (g_prof_cls < 0 || (g_prof_cls != cls && g_prof_cls != PROF_ALL_GEMM) || g_prof_n >= PROF_MAX) return false;
    while (g_prof_created < 2 * (g_prof_n + 1)) (void)hipEventCreate(&g_prof_ev[g_prof_created++]);
    g_disp_ev[0] = g_prof_ev[2 * g_prof_n]; g_disp_ev[1] = g_prof_ev[2 * g_prof_n + 1];
    ++g_prof_n; ++g_prof_disp_n;
    return true;
}
bool prof_dispatch_release() {
    if (!g_disp_ev[0]) return false;
    g_disp_ev[0] = g_disp_ev[1] = nullptr;
    --g_prof_n; --g_prof_disp_n;
    return true;
}
bool take_dispatch_events(hipEvent_t *start, hipEvent_t *stop) {
    if (!g_disp_ev[0]) return false;
    *start = g_disp_ev[0]; *stop = g_disp_ev[1];
    g_disp_ev[0] = g_disp_ev[1] = nullptr;
    return true;
}

// ---- fills and copies as KERNELS ------------------------------------------------------------------
// The sequence driver is captured into hipGraphs (lstm/lstm.py: _forward_graphed).  With ROCm 7.2 a hipMemsetAsync captured
// as a memset node clears its whole range on the first replay only -- later replays leave more than half of the bytes as they
// were (tools/diag/graph_memset_probe.py: every size from 64 B to 1 MB) -- so the driver does not use memset / memcpy nodes.
__global__ void __launch_bounds__(256) fill_u32_kernel(uint32_t *p, uint32_t v, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = v;
}
__global__ void __launch_bounds__(256) fill_u8_kernel(uint8_t *p, uint8_t v, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = v;
}
__global__ void __launch_bounds__(256) copy_u32_kernel(uint32_t *dst, const uint32_t *src, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
static unsigned fill_blocks(size_t n) { const size_t b = (n + 255) / 256; return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b)); }
// bytes: a multiple of 4 on a 4-byte aligned address, or any count for a byte value
static int fill_bytes(void *p, int byte, size_t bytes, hipStream_t s) {
    if (bytes == 0) return 0;
    if (bytes % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & 3) == 0) {
        const uint32_t v = 0x01010101u * (uint32_t)(byte & 0xff);
        hipLaunchKernelGGL(fill_u32_kernel, dim3(fill_blocks(bytes / 4)), dim3(256), 0, s, reinterpret_cast<uint32_t *>(p), v, bytes / 4);
    } else {
        hipLaunchKernelGGL(fill_u8_kernel, dim3(fill_blocks(bytes)), dim3(256), 0, s, reinterpret_cast<uint8_t *>(p), (uint8_t)byte, bytes);
    }
    TNP_HIP(hipGetLastError());
    return 0;
}
static int copy_floats(float *dst, const float *src, size_t n, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(copy_u32_kernel, dim3(fill_blocks(n)), dim3(256), 0, s, reinterpret_cast<uint32_t *>(dst),
                       reinterpret_cast<const uint32_t *>(src), n);
    TNP_HIP(hipGetLastError());
    return 0;
}

// ---- per-track prepare kernel -----------------------------------------------------------------
struct PrepArgs {
    int M, H, E, goal_flag, goal_dim, C, I;  // I = row stride of X
    // finish previous step
    int have_prev;
    const float *h;              // [M,H] state after the previous step
    const uint8_t *mask_prev;    // [M]
    const float *obs2_prev;      // [M,2]
    float *normal_out;           // [M,5]  rel_pred[s-1]
    float *pos_out;              // [M,2]  pred[...]
    const float *Wn, *bn;
    // set up next step
    int have_next;
    const float *ext1, *ext2;    // [M,2] external frames or NULL
    const float *pos1;           // positions[-2] ([M,2]) or NULL
    const float *pos2;           // positions[-1] ([M,2]) when this launch does not compute it itself (have_prev == 0)
    int use_pos2;                // obs2 (all rows if ext2 == NULL, primary rows if patch2) <- the position just computed
    int patch1, patch2;          // primary rows of an external frame are replaced by the prediction
    const uint8_t *primary;      // [M]
    const float *goals;          // [M,2]
    float *obs1_buf, *obs2_buf;  // [M,2]
    uint8_t *mask;               // [M]
    float *X;                    // [M,I]
    const float *We, *be, *Wg, *bg;
    const float *Wh, *bh;        // social encoding [C,H]
    float *enc;                  // [M,C]
    // fused loss of the step being finished (primaries only)
    const float *loss_tgt;       // [M,2] or NULL
    float *loss_out;             // [M]
    int loss_mode; float loss_bg;
    const uint8_t *prim_prev;    // [M] primary flags (read when loss_tgt is set)
};

#define PREP_TRACKS 8

__device__ __forceinline__ float sigmoid_dev(float x) { return 1.0f / (1.0f + expf(-x)); }

// 256 threads = 8 tracks x 32 lanes.  Phase 1: the 5 + C dot products of length H per track (h row and the
// weight rows staged in LDS, weight stride H+1 -> conflict free).  Phase 2: per-track scalar bookkeeping.
// Phase 3: the E-2 (+ goal) embedding outputs, 32 lanes per track.
__device__ __forceinline__ void track_prepare_body(const PrepArgs &a, const int block, float *psm) {
    const int H = a.H;
    const int nout = (a.have_prev ? 5 : 0) + (a.have_next ? a.C : 0);  // rows of the stacked weight
    float *hs = psm;                        // [8][H]
    float *ws = hs + PREP_TRACKS * H;       // [nout][H+4]: rows 16-byte aligned, lanes (= outputs) 4 banks apart
    float *outs = ws + nout * (H + 4);      // [8][nout]
    float *ob = outs + PREP_TRACKS * (nout > 0 ? nout : 1);  // [8][8]: obs1.xy obs2.xy mask goal.xy
    const int tid = threadIdx.x;
    const int t_local = tid >> 5, l32 = tid & 31;
    const int m0 = block * PREP_TRACKS;
    const int m = m0 + t_local;
    const bool valid = m < a.M;

    // phase-2 operands of lane 0 are fetched now so their latency hides behind phase 1
    float pf_o2x = NAN, pf_o2y = NAN, pf_e1x = NAN, pf_e1y = NAN, pf_e2x = NAN, pf_e2y = NAN, pf_p1x = NAN, pf_p1y = NAN;
    int pf_maskprev = 0, pf_prim = 0, pf_lossprim = 0;
    float pf_tx = 0.0f, pf_ty = 0.0f;
    if (l32 == 0 && valid) {
        if (a.have_prev) {
            pf_o2x = a.obs2_prev[2 * m]; pf_o2y = a.obs2_prev[2 * m + 1]; pf_maskprev = a.mask_prev[m];
            if (a.loss_tgt) { pf_lossprim = a.prim_prev[m]; pf_tx = a.loss_tgt[2 * m]; pf_ty = a.loss_tgt[2 * m + 1]; }
        }
        if (a.have_next) {
            pf_prim = a.primary[m];
            if (a.ext1) { pf_e1x = a.ext1[2 * m]; pf_e1y = a.ext1[2 * m + 1]; }
            if (a.ext2) { pf_e2x = a.ext2[2 * m]; pf_e2y = a.ext2[2 * m + 1]; }
            if (a.pos1) { pf_p1x = a.pos1[2 * m]; pf_p1y = a.pos1[2 * m + 1]; }
        }
    }
    float pf_b0 = 0.0f;                                        // bias of this lane's first output row (phase 1)
    if (l32 < nout) pf_b0 = (a.have_prev && l32 < 5) ? a.bn[l32] : a.bh[l32 - (a.have_prev ? 5 : 0)];
    // InputEmbedding / goal embedding weights of this lane's outputs (phase 3) are fetched now as well
    constexpr int EPF = 2;                                     // outputs per lane held in registers (E, goal_dim <= 64)
    float pf_we[EPF][3], pf_wg[EPF][3];
    const bool epf = a.have_next && a.E <= 32 * EPF && (!a.goal_flag || a.goal_dim <= 32 * EPF);
    if (epf) {
#pragma unroll
        for (int i = 0; i < EPF; ++i) {
            const int o = l32 + 32 * i;
            pf_we[i][0] = pf_we[i][1] = pf_we[i][2] = 0.0f;
            pf_wg[i][0] = pf_wg[i][1] = pf_wg[i][2] = 0.0f;
            if (o < a.E - 2) { pf_we[i][0] = a.We[2 * o]; pf_we[i][1] = a.We[2 * o + 1]; pf_we[i][2] = a.be[o]; }
            if (a.goal_flag && o < a.goal_dim - 2) { pf_wg[i][0] = a.Wg[2 * o]; pf_wg[i][1] = a.Wg[2 * o + 1]; pf_wg[i][2] = a.bg[o]; }
        }
    }
    if (nout > 0) {
        // all global loads of the staging first, then the LDS stores: a load -> store loop with a runtime trip count exposes
        // one global round trip per iteration (this was 10 of the kernel's 20 k cycles)
        const int H4 = H >> 2;                                 // H % 4 == 0
        constexpr int HB = 2, WB = 6;                          // float4 per thread per batch
        for (int q0 = 0; q0 < PREP_TRACKS * H4; q0 += 256 * HB) {
            float4 v[HB];
#pragma unroll
            for (int i = 0; i < HB; ++i) {
                const int q = q0 + tid + 256 * i;
                const int t = q / H4, k4 = q - t * H4;
                v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < PREP_TRACKS * H4 && m0 + t < a.M) v[i] = reinterpret_cast<const float4 *>(a.h + (size_t)(m0 + t) * H)[k4];
            }
#pragma unroll
            for (int i = 0; i < HB; ++i) {
                const int q = q0 + tid + 256 * i;
                if (q < PREP_TRACKS * H4) reinterpret_cast<float4 *>(hs)[q] = v[i];
            }
        }
        for (int q0 = 0; q0 < nout * H4; q0 += 256 * WB) {
            float4 v[WB];
#pragma unroll
            for (int i = 0; i < WB; ++i) {
                const int q = q0 + tid + 256 * i;
                const int o = q / H4, k4 = q - o * H4;
                v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < nout * H4) {
                    const float *row = (a.have_prev && o < 5) ? a.Wn + (size_t)o * H : a.Wh + (size_t)(o - (a.have_prev ? 5 : 0)) * H;
                    v[i] = reinterpret_cast<const float4 *>(row)[k4];
                }
            }
#pragma unroll
            for (int i = 0; i < WB; ++i) {
                const int q = q0 + tid + 256 * i;
                const int o = q / H4, k4 = q - o * H4;
                if (q < nout * H4) {
                    *reinterpret_cast<float4 *>(ws + o * (H + 4) + 4 * k4) = v[i];
                }
            }
        }
    }
    __syncthreads();
    for (int o = l32; o < nout; o += 32) {
        const float *hr = hs + t_local * H;
        const float *wr = ws + o * (H + 4);
        float acc;
        if (o == l32) acc = pf_b0;
        else if (a.have_prev && o < 5) acc = a.bn[o];
        else acc = a.bh[o - (a.have_prev ? 5 : 0)];
        // four interleaved partial sums (H % 4 == 0): breaks the 128-long dependent FMA chain
        float s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll 4
        for (int k = 0; k < H; k += 4) {
            const float4 hv = *reinterpret_cast<const float4 *>(hr + k), wv = *reinterpret_cast<const float4 *>(wr + k);
            acc = fmaf(hv.x, wv.x, acc);
            s1 = fmaf(hv.y, wv.y, s1);
            s2 = fmaf(hv.z, wv.z, s2);
            s3 = fmaf(hv.w, wv.w, s3);
        }
        outs[t_local * nout + o] = (acc + s1) + (s2 + s3);
    }
    __syncthreads();

    // ---- phase 2: one lane per track ----
    if (l32 == 0 && valid) {
        float px = NAN, py = NAN;  // position predicted by the previous step
        if (!a.have_prev && a.pos2) { px = a.pos2[2 * m]; py = a.pos2[2 * m + 1]; }
        if (a.have_prev) {
            const float *o = outs + t_local * nout;
            float n0 = NAN, n1 = NAN, n2 = NAN, n3 = NAN, n4 = NAN;
            if (pf_maskprev) {  // Hidden2Normal, lstm/modules.py:56-64
                n0 = o[0]; n1 = o[1];
                n2 = 0.01f + 0.2f * sigmoid_dev(o[2]);
                n3 = 0.01f + 0.2f * sigmoid_dev(o[3]);
                n4 = 0.7f * sigmoid_dev(o[4]);
            }
            float *no = a.normal_out + (size_t)m * 5;
            no[0] = n0; no[1] = n1; no[2] = n2; no[3] = n3; no[4] = n4;
            if (pf_lossprim) a.loss_out[m] = primary_loss_value(a.loss_mode, n0, n1, n2, n3, n4, pf_tx, pf_ty, a.loss_bg);
            px = pf_o2x + n0;      // positions.append(obs2 + normal[:, :2]), lstm.py:232,255
            py = pf_o2y + n1;
            a.pos_out[2 * m] = px; a.pos_out[2 * m + 1] = py;
        }
        if (a.have_next) {
            const bool prim = pf_prim != 0;
            float o1x, o1y, o2x, o2y;
            if (a.ext1 && !(a.patch1 && prim)) { o1x = pf_e1x; o1y = pf_e1y; }
            else { o1x = pf_p1x; o1y = pf_p1y; }
            if (a.ext2 && !(a.patch2 && prim)) { o2x = pf_e2x; o2y = pf_e2y; }
            else { o2x = px; o2y = py; }
            const bool present = (o1x == o1x) && (o2x == o2x);  // lstm.py:118
            a.obs1_buf[2 * m] = o1x; a.obs1_buf[2 * m + 1] = o1y;
            a.obs2_buf[2 * m] = o2x; a.obs2_buf[2 * m + 1] = o2y;
            a.mask[m] = present ? 1 : 0;
            float *b = ob + t_local * 8;
            b[0] = o1x; b[1] = o1y; b[2] = o2x; b[3] = o2y;
            if (a.goal_flag) {  // lstm.py:132-139
                const float dx = o2x - a.goals[2 * m], dy = o2y - a.goals[2 * m + 1];
                const float nf = sqrtf(dx * dx + dy * dy);
                float gx = dx / nf, gy = dy / nf;
                if (nf == 0.0f) { gx = 0.0f; gy = 0.0f; }
                b[4] = gx; b[5] = gy;
            }
            if (a.enc) {
                const float *o = outs + t_local * nout + (a.have_prev ? 5 : 0);
                for (int c = 0; c < a.C; ++c) a.enc[(size_t)m * a.C + c] = o[c];
            }
        }
    }
    __syncthreads();

    // ---- phase 3: InputEmbedding (lstm/modules.py:24-30): relu(W (4 v) + b) ++ 0,0 ----
    if (a.have_next && valid) {
        const float *b = ob + t_local * 8;
        const float vx = (b[2] - b[0]) * 4.0f, vy = (b[3] - b[1]) * 4.0f;  // lstm.py:127
        float *xr = a.X + (size_t)m * a.I;
        if (epf) {
#pragma unroll
            for (int i = 0; i < EPF; ++i) {
                const int o = l32 + 32 * i;
                if (o < a.E) {
                    float v = 0.0f;
                    if (o < a.E - 2) {
                        v = fmaf(vy, pf_we[i][1], fmaf(vx, pf_we[i][0], pf_we[i][2]));
                        v = v > 0.0f ? v : 0.0f;
                    }
                    xr[o] = v;
                }
                if (a.goal_flag && o < a.goal_dim) {
                    const float gx = b[4] * 4.0f, gy = b[5] * 4.0f;
                    float v = 0.0f;
                    if (o < a.goal_dim - 2) {
                        v = fmaf(gy, pf_wg[i][1], fmaf(gx, pf_wg[i][0], pf_wg[i][2]));
                        v = v > 0.0f ? v : 0.0f;
                    }
                    xr[a.E + o] = v;
                }
            }
            return;
        }
        for (int o = l32; o < a.E; o += 32) {
            float v = 0.0f;
            if (o < a.E - 2) {
                v = fmaf(vy, a.We[2 * o + 1], fmaf(vx, a.We[2 * o], a.be[o]));
                v = v > 0.0f ? v : 0.0f;  // absent track (NaN velocity) -> 0; the row is masked anyway
            }
            xr[o] = v;
        }
        if (a.goal_flag) {
            const float gx = b[4] * 4.0f, gy = b[5] * 4.0f;
            for (int o = l32; o < a.goal_dim; o += 32) {
                float v = 0.0f;
                if (o < a.goal_dim - 2) {
                    v = fmaf(gy, a.Wg[2 * o + 1], fmaf(gx, a.Wg[2 * o], a.bg[o]));
                    v = v > 0.0f ? v : 0.0f;
                }
                xr[a.E + o] = v;
            }
        }
    }
}

__global__ void __launch_bounds__(256) track_prepare_kernel(const PrepArgs a) {
    extern __shared__ __attribute__((aligned(16))) float psm[];
    track_prepare_body(a, blockIdx.x, psm);
}

// ---- track_prepare + grid_build in ONE launch (occupancy / directional grids) ------------------------------------------------
// The grid of step s needs the positions track_prepare writes for step s: two dependent ~5.6 us launches per recurrent step,
// a sixth of the step at a strong-scaling shard (2048 tracks) and an eighth of an S-GAN generator step.  Here the grid's
// workgroups do not wait for those buffers: a workgroup (four egos of one scene) forms the positions of ITS scene from what
// track_prepare itself reads -- the previous state (Hidden2Normal's two mean rows, same four partial sums in the same order:
// the same bits), the previous obs2, the external frames and the primaries' patches -- so both kinds of workgroups run side
// by side in one launch.  ~2 x 128 FMAs per track of the scene and workgroup, against a launch boundary.
__global__ void __launch_bounds__(256) prepare_grid_kernel(const PrepArgs pa, const GridArgs ga, int prep_blocks, int ego_blocks,
                                                           int wn_offset /* floats: behind the grid's own LDS */) {
    extern __shared__ __attribute__((aligned(16))) float pgsm[];
    if ((int)blockIdx.x < prep_blocks) { track_prepare_body(pa, blockIdx.x, pgsm); return; }
    const int gid = (int)blockIdx.x - prep_blocks;
    const int sc = gid / ego_blocks, bx = gid - sc * ego_blocks;
    grid_build_body(ga, bx, sc, pgsm, [&](int start, int ns, float2 *pos, float2 *vel) {
        const int tid = threadIdx.x, q = tid & 3, H = pa.H;
        float *wn = pgsm + wn_offset;                                        // rows 0 / 1 of Hidden2Normal, once per workgroup
        if (pa.have_prev) {
            for (int k = tid; k < 2 * H; k += 256) wn[k] = pa.Wn[k];
            __syncthreads();
        }
        for (int j0 = 0; j0 < ns; j0 += 64) {                                // four lanes per track
            const int j = j0 + (tid >> 2);
            const bool on = j < ns;
            const int m = start + (on ? j : 0);
            // every operand that does not depend on the dot products is requested first (their round trip runs beside the h rows')
            float2 prev2 = make_float2(NAN, NAN), e1 = make_float2(NAN, NAN), e2 = make_float2(NAN, NAN);
            bool was = false;
            if (pa.have_prev) { prev2 = reinterpret_cast<const float2 *>(pa.obs2_prev)[m]; was = pa.mask_prev[m] != 0; }
            else if (pa.pos2) prev2 = reinterpret_cast<const float2 *>(pa.pos2)[m];
            const bool prim = pa.primary[m] != 0;
            const bool take1 = pa.ext1 && !(pa.patch1 && prim), take2 = pa.ext2 && !(pa.patch2 && prim);
            if (take1) e1 = reinterpret_cast<const float2 *>(pa.ext1)[m];
            else if (pa.pos1) e1 = reinterpret_cast<const float2 *>(pa.pos1)[m];
            if (take2) e2 = reinterpret_cast<const float2 *>(pa.ext2)[m];
            float px = prev2.x, py = prev2.y;                                // position predicted by the previous step
            if (pa.have_prev) {
                // rows 0 / 1 of Hidden2Normal as track_prepare sums them: lane q = partial sum q (k = q mod 4, ascending, the
                // first one starting from the bias), combined as (p0 + p1) + (p2 + p3)
                float p0 = q == 0 ? pa.bn[0] : 0.0f, p1 = q == 0 ? pa.bn[1] : 0.0f;
                const float *hr = pa.h + (size_t)m * H + q, *w0 = wn + q, *w1 = wn + H + q;
                // (the h elements of the first 64 tracks requested ahead of the weight staging -- 32 more registers -- made the launch
                // slower: 8.6 -> 9.3 us)
#pragma unroll 16
                for (int k = 0; k < H; k += 4) {
                    const float hv = hr[k];
                    p0 = fmaf(hv, w0[k], p0);
                    p1 = fmaf(hv, w1[k], p1);
                }
                p0 += __shfl_xor(p0, 1); p1 += __shfl_xor(p1, 1);
                p0 += __shfl_xor(p0, 2); p1 += __shfl_xor(p1, 2);
                px = prev2.x + (was ? p0 : NAN);
                py = prev2.y + (was ? p1 : NAN);
            }
            const float2 o1 = e1;
            const float2 o2 = take2 ? e2 : make_float2(px, py);
            if (on && q == 0) grid_stage_track(ga.type, o1, o2, j, pos, vel);
        }
    });
}

static size_t prepare_smem_bytes(const PrepArgs &a) {
    const int nout = (a.have_prev ? 5 : 0) + (a.have_next ? a.C : 0);
    return ((size_t)PREP_TRACKS * a.H + (size_t)nout * (a.H + 4) + (size_t)PREP_TRACKS * (nout > 0 ? nout : 1) + PREP_TRACKS * 8) *
           sizeof(float);
}

static int launch_prepare_grid(const PrepArgs &a, const GridArgs &g, hipStream_t s) {
    size_t psm = prepare_smem_bytes(a), gsm = 0;
    if (psm > 60000) TNP_FAIL(-1, "track_prepare: hidden_dim %d too large for the LDS staging", a.H);
    GridArgs gb;
    const int rc = grid_launch_plan(g, &gb, &gsm);
    if (rc) return rc;
    const int wn_offset = (int)(gsm / sizeof(float));
    gsm += (size_t)2 * a.H * sizeof(float);
    const size_t smem = psm > gsm ? psm : gsm;
    // (egos per grid workgroup: every workgroup of a scene forms the scene's positions again, so more egos per workgroup would
    // mean fewer repetitions -- and a longer ego loop, which is what counts: 4 / 8 / 16 egos -> 0.591 / 0.621 / 0.698 ms per
    // forward at 32 x 64 directional, 1.488 / 1.525 / 1.675 ms per S-GAN forward)
    const int prep_blocks = (a.M + PREP_TRACKS - 1) / PREP_TRACKS, ego_blocks = (g.n_max + TNP_GRID_EGOS - 1) / TNP_GRID_EGOS;
    static size_t attr[TNP_MAX_DEVICES_SEQ] = {};
    int dev = 0;
    TNP_HIP(hipGetDevice(&dev));
    if (dev >= 0 && dev < TNP_MAX_DEVICES_SEQ && smem > attr[dev] && smem > 48 * 1024) {
        TNP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(prepare_grid_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)smem));
        attr[dev] = smem;
    }
    hipLaunchKernelGGL(prepare_grid_kernel, dim3(prep_blocks + ego_blocks * g.B), dim3(256), smem, s, a, gb, prep_blocks, ego_blocks,
                       wn_offset);
    TNP_HIP(hipGetLastError());
    return 0;
}

static int launch_prepare(const PrepArgs &a, hipStream_t s) {
    const int nout = (a.have_prev ? 5 : 0) + (a.have_next ? a.C : 0);
    size_t smem = ((size_t)PREP_TRACKS * a.H + (size_t)nout * (a.H + 4) + (size_t)PREP_TRACKS * (nout > 0 ? nout : 1) +
                   PREP_TRACKS * 8) * sizeof(float);
    if (smem > 60000) TNP_FAIL(-1, "track_prepare: hidden_dim %d too large for the LDS staging", a.H);
    const int blocks = (a.M + PREP_TRACKS - 1) / PREP_TRACKS;
    hipLaunchKernelGGL(track_prepare_kernel, dim3(blocks), dim3(256), smem, s, a);
    TNP_HIP(hipGetLastError());
    return 0;
}

// ---- workspace layout ---------------------------------------------------------------------------
struct Workspace {
    float *h[2];
    float *c;
    float *obs1, *obs2;
    float *X;
    float *enc;
    float *grid;
    float *y[2];
    uint8_t *mask;
    int16_t *winners;
    int32_t *row_base;
    float *partial;
    int sparse;   // first embedding layer runs on the winner table (pool_embed_sparse.hip)
    int I, Fin, ldg;
    // stateful interaction encoders (NearestNeighborLSTM / TrajectronPooling): pool_lstm state, all-ones mask
    float *ph[2], *pc;
    uint8_t *ones;
    double *scratch;
    int pcur;
    int to_hidden;   // pooled vector goes to hplus (= h + pooled, the LSTMCell's hidden operand) instead of X
    float *hplus;
    float *pdst;     // where the interaction module writes its [M, P] result, and its leading dimension
    int pld;
    float *gates_save;   // training: post-activation gates of the step
    float *nn_attrs_save;
    float *pc_out, *pgates_save, *traj_in_save;   // stateful interaction encoders under tnp_lstm_forward_train
    float *pvec_save;                             // pool_to_input=False: the interaction vector before it is added to h
    int32_t *row_end;                             // sparse path: one past the last row of every row's scene
    int32_t *row_padded;                          // sparse path: slots the reference pads the row's scene to
    int fuse_grid, save_winners;                  // winner tile built inside the sparse kernel; table wanted by the caller
    int grid_prebuilt;                            // this step's grid came with the merged prepare + grid launch (run_step_body skips its own)
    float *obs2_alt; uint8_t *mask_alt;           // second copies for that launch: its grid workgroups read the PREVIOUS obs2 / mask of
                                                  // a whole scene while its prepare workgroups write the new ones
    size_t bytes;
    bool fine; float *grid_fine; int ldg_fine;   // pool_size / blur_size != 1: the fine grid in front of grid_finish_kernel
};

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static int plan_workspace(const tnp_lstm_model *md, int M, void *base, Workspace &w) {
    const int H = md->H, E = md->E;
    const int GD = md->goal_flag ? md->goal_dim : 0;
    const bool pool = md->pool_type != TNP_POOL_NONE;
    const int P = pool ? md->P : 0;
    // variant bit 17: LSTM(pool_to_input=False) -- the interaction vector is added to the hidden state
    // (lstm/lstm.py:150-151) instead of being concatenated to the input embedding
    w.to_hidden = pool && ((md->variant >> 17) & 1);
    w.I = E + GD + (w.to_hidden ? 0 : P);
    const bool grid_pool = pool && md->pool_type <= TNP_POOL_SOCIAL;
    w.Fin = grid_pool ? md->C * md->n * md->n : 0;
    w.ldg = (w.Fin + 3) & ~3;
    size_t off = 0;
    char *b = reinterpret_cast<char *>(base);
    auto take = [&](size_t nbytes) { char *p = b ? b + off : nullptr; off += align256(nbytes); return p; };
    w.h[0] = (float *)take((size_t)M * H * 4);
    w.h[1] = (float *)take((size_t)M * H * 4);
    w.c = (float *)take((size_t)M * H * 4);
    // (right behind h[0], h[1], c: the one fill that zeroes the state at the start of a forward pass zeroes these as well)
    w.obs1 = (float *)take((size_t)M * 2 * 4);
    w.obs2 = (float *)take((size_t)M * 2 * 4);
    w.X = (float *)take((size_t)M * w.I * 4);
    w.enc = (float *)take((size_t)M * (md->C > 0 ? md->C : 1) * 4);
    w.grid = (float *)take((size_t)M * (w.ldg > 0 ? w.ldg : 4) * 4);
    // pool_size / blur_size (lstm/gridbased_pooling.py:297-304): the fine grid and, for an even blur_size, the (G + 1)^2 blurred map
    w.fine = grid_pool && (md->pool_size > 1 || md->blur_size > 1);
    w.grid_fine = nullptr;
    if (w.fine) {
        const int G = md->n * (md->pool_size > 1 ? md->pool_size : 1);
        w.ldg_fine = (md->C * G * G + 3) & ~3;
        w.grid_fine = (float *)take((size_t)M * w.ldg_fine * 4);
    }
    int maxmid = 4;
    for (int l = 1; grid_pool && l < md->n_layers; ++l) if (md->dims[l] > maxmid) maxmid = md->dims[l];
    if (md->pool_type == TNP_POOL_HIDDENMLP) maxmid = md->dims[0] + md->dims[1] + md->dims[2];   // pooled [M, mlp_dim]
    if (md->pool_type == TNP_POOL_ATTNMLP) maxmid = md->dims[0] + md->dims[1] + md->dims[2] + 4;  // u [M, mlp_dim + 4]
    if (md->pool_type == TNP_POOL_NNLSTM || md->pool_type == TNP_POOL_TRAJ) maxmid = md->P;      // features [M, P]
    w.y[0] = (float *)take((size_t)M * maxmid * 4);
    w.y[1] = (float *)take((size_t)M * maxmid * 4);
    w.mask = (uint8_t *)take((size_t)M);
    w.mask_alt = (uint8_t *)take((size_t)M);
    w.obs2_alt = (float *)take((size_t)M * 2 * 4);
    w.sparse = grid_pool && !w.fine && md->pool_type == TNP_POOL_SOCIAL && md->Wp0_cell_major != nullptr && md->constant == 0.0f &&
               ((md->variant >> 16) & 1) == 0 && sparse_supported(md->C, md->dims[1], md->n * md->n) && (w.I % 4 == 0);
    w.winners = nullptr; w.row_base = nullptr; w.partial = nullptr; w.row_end = nullptr; w.row_padded = nullptr; w.fuse_grid = 0; w.grid_prebuilt = 0;
    w.save_winners = 0;
    if (w.sparse) {
        w.winners = (int16_t *)take((size_t)M * md->n * md->n * sizeof(int16_t));
        w.row_base = (int32_t *)take((size_t)M * sizeof(int32_t));
        w.row_end = (int32_t *)take((size_t)M * sizeof(int32_t));
        w.row_padded = (int32_t *)take((size_t)M * sizeof(int32_t));
        // winner tile built inside the sparse kernel (no grid kernel, no table) whenever the tile fits in LDS
        w.fuse_grid = sparse_fuses_grid(md->n * md->n, 32767) && (size_t)M * md->C * sizeof(float) < ((size_t)1 << 32);
        const size_t pb = sparse_partial_bytes(M, md->dims[1], md->n * md->n);
        if (pb) w.partial = (float *)take(pb);
    }
    w.ph[0] = w.ph[1] = w.pc = nullptr; w.ones = nullptr; w.scratch = nullptr; w.pcur = 0;
    w.hplus = w.to_hidden ? (float *)take((size_t)M * md->H * 4) : nullptr;
    w.pdst = w.to_hidden ? w.hplus : (pool ? w.X + (w.I - P) : nullptr);
    w.pld = w.to_hidden ? md->H : w.I;
    w.gates_save = nullptr;
    w.nn_attrs_save = nullptr;
    w.pc_out = nullptr; w.pgates_save = nullptr; w.traj_in_save = nullptr; w.pvec_save = nullptr;
    if (md->pool_type == TNP_POOL_NNLSTM || md->pool_type == TNP_POOL_TRAJ) {
        const int Hp = md->dims[0];
        w.ph[0] = (float *)take((size_t)M * Hp * 4);
        w.ph[1] = (float *)take((size_t)M * Hp * 4);
        w.pc = (float *)take((size_t)M * Hp * 4);
        w.ones = (uint8_t *)take((size_t)M);
        w.scratch = (double *)take(4 * sizeof(double));
    }
    w.bytes = off;
    return 0;
}

static int validate_model(const tnp_lstm_model *md) {
    if (!md) TNP_FAIL(-1, "null model");
    if (md->pool_type != TNP_POOL_NONE && ((md->variant >> 17) & 1) && md->P != md->H)
        TNP_FAIL(-1, "pool_to_input=False adds the interaction vector to the hidden state: out_dim %d must equal hidden_dim %d", md->P, md->H);
    if (md->H <= 0 || md->H % 32 != 0) TNP_FAIL(-1, "hidden_dim must be a positive multiple of 32 (got %d)", md->H);
    if (md->E < 4) TNP_FAIL(-1, "embedding_dim too small (%d)", md->E);
    if (md->pool_type == TNP_POOL_NN) {
        if (md->n < 1 || md->n > 8 || md->P <= 0 || md->P % md->n != 0) TNP_FAIL(-1, "NearestNeighborMLP: n=%d must divide out_dim=%d, n <= 8", md->n, md->P);
        if (md->C != 2 && md->C != 4) TNP_FAIL(-1, "NearestNeighborMLP: input_dim %d", md->C);
        return 0;
    }
    if (md->pool_type == TNP_POOL_NNLSTM || md->pool_type == TNP_POOL_TRAJ) {
        const int Hp = md->dims[0];
        if (Hp <= 0 || Hp % 32 != 0) TNP_FAIL(-1, "interaction-encoder LSTM: hidden_dim %d must be a positive multiple of 32", Hp);
        if (md->P <= 0 || md->P % 4 != 0) TNP_FAIL(-1, "interaction encoder: out_dim %d must be a multiple of 4", md->P);
        if (md->pool_type == TNP_POOL_NNLSTM && (md->n < 1 || md->n > 8 || md->P % md->n != 0 || md->C != 4))
            TNP_FAIL(-1, "NearestNeighborLSTM: n=%d must divide out_dim=%d (n <= 8)", md->n, md->P);
        for (int k = 0; k < 3; ++k)
            if (!md->Wx[k] || !md->bx[k]) TNP_FAIL(-1, "interaction encoder: pool_lstm / hidden2pool weights (Wx/bx) missing");
        return 0;
    }
    if (md->pool_type == TNP_POOL_HIDDENMLP || md->pool_type == TNP_POOL_ATTNMLP) {
        if (md->pool_type == TNP_POOL_ATTNMLP && (!md->Wx[0] || !md->Wx[1] || !md->Wx[2] || !md->bx[0] || !md->bx[2]))
            TNP_FAIL(-1, "AttentionMLPPooling: folded attention matrices (Wx/bx) missing");
        if (md->dims[0] <= 0 || md->dims[1] < 0 || md->dims[2] < 0 || md->C != md->dims[2] || md->P <= 0)
            TNP_FAIL(-1, "HiddenStateMLPPooling: bad dims %d/%d/%d", md->dims[0], md->dims[1], md->dims[2]);
        if (md->C > 64) TNP_FAIL(-1, "HiddenStateMLPPooling: mlp_dim_hidden %d > 64 not supported", md->C);
        return 0;
    }
    if (md->pool_type < TNP_POOL_NONE || md->pool_type > TNP_POOL_SOCIAL) TNP_FAIL(-1, "unknown pool_type %d", md->pool_type);
    if (md->pool_type != TNP_POOL_NONE) {
        if (md->n_layers < 1 || md->n_layers > 3) TNP_FAIL(-1, "embedding MLP depth %d not in 1..3", md->n_layers);
        if (md->dims[0] != md->C * md->n * md->n) TNP_FAIL(-1, "dims[0]=%d != C*n*n=%d", md->dims[0], md->C * md->n * md->n);
        if (md->dims[md->n_layers] != md->P) TNP_FAIL(-1, "dims[last] != P");
        if (md->C > 64) TNP_FAIL(-1, "pooling_dim %d > 64 not supported", md->C);
    }
    return 0;
}

// y[q] += x[q] on float4 groups
__global__ void add_rows_kernel(float *__restrict__ y, const float *__restrict__ x, long n4) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n4) return;
    float4 a = reinterpret_cast<float4 *>(y)[q];
    const float4 b = reinterpret_cast<const float4 *>(x)[q];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    reinterpret_cast<float4 *>(y)[q] = a;
}

// the LSTM gates GEMM of one step: [X | h] x [W_ih | W_hh]^T + LSTMCell pointwise + masked state update
static void fill_gates_args(GemmArgs &g, const tnp_lstm_model *md, int decoder, const Workspace &w, const float *h_in, float *h_out,
                            const float *c_in, float *c_out, int M) {
    const int H = md->H;
    memset(&g, 0, sizeof(g));
    g.A1 = w.X; g.lda1 = w.I; g.K1 = w.I;
    g.A2 = w.to_hidden ? w.hplus : h_in; g.lda2 = H; g.K2 = H;
    g.B1 = decoder ? md->dec_Wih : md->enc_Wih; g.ldb1 = w.I;
    g.B2 = decoder ? md->dec_Whh : md->enc_Whh; g.ldb2 = H;
    g.bias1 = decoder ? md->dec_bih : md->enc_bih;
    g.bias2 = decoder ? md->dec_bhh : md->enc_bhh;
    g.M = M; g.N = 4 * H; g.H = H;
    g.h_in = h_in; g.h_out = h_out; g.c_in = c_in; g.c_out = c_out; g.mask = w.mask;
    g.gates_out = w.gates_save;
}

// pool + gates of one step; obs/mask/X[:,0:E+GD]/enc already prepared
static GridArgs step_grid_args(const tnp_lstm_model *md, const Workspace &w, const int32_t *scene_start, int B, int n_max,
                               const int32_t *scene_slots) {
    GridArgs ga;
    memset(&ga, 0, sizeof(ga));
    ga.obs1 = w.obs1; ga.obs2 = w.obs2; ga.values = w.enc; ga.ldv = md->C; ga.scene_start = scene_start;
    ga.B = B; ga.n_max = n_max; ga.scene_slots = scene_slots; ga.type = md->pool_type; ga.n = md->n; ga.C = md->C;
    ga.cell = md->cell; ga.half_x = md->half_x; ga.half_y = md->half_y; ga.constant = md->constant;
    ga.grid = w.sparse ? nullptr : w.grid; ga.ldg = w.ldg; ga.winners = w.sparse ? w.winners : nullptr;
    if (w.fine) { ga.n = md->n * (md->pool_size > 1 ? md->pool_size : 1); ga.grid = w.grid_fine; ga.ldg = w.ldg_fine; }
    return ga;
}

// track_prepare of a step and -- occupancy / directional grids -- the step's grid in the same launch (prepare_grid_kernel)
static int prepare_step(const tnp_lstm_model *md, Workspace &w, const PrepArgs &p, const int32_t *scene_start, int B, int n_max,
                        const int32_t *scene_slots, hipStream_t s) {
    const bool grid_pool = md->pool_type == TNP_POOL_OCCUPANCY || md->pool_type == TNP_POOL_DIRECTIONAL;
    // ... while the repeated position work is small: every grid workgroup (four egos) reads the h rows of its whole scene, M *
    // n_max / 4 rows per launch.  64-agent scenes: 32 / 64 / 128 / 256 scenes -> 0.584 / 0.883 / 1.367 / 2.503 ms per forward merged
    // against 0.617 / 0.879 / 1.278 / 2.318 ms with two launches; 128 x 32 (S-GAN): 1.486 against 1.556.
    const int fuse = tuning().fuse_prepare_grid;                     // 0 = never, 1 = by size, 2 = always
    if (p.have_next && grid_pool && !w.sparse && !w.fine && B > 0 && n_max > 0 &&
        (fuse == 2 || (fuse == 1 && (long)p.M * n_max <= 196608))) {
        // track_prepare updates obs2 / mask in place (a thread reads its track's previous value, then writes the new one); the
        // grid workgroups of the same launch read the previous values of whole scenes: the new ones go to the second copies
        PrepArgs q = p;
        if (q.have_prev && q.obs2_prev == q.obs2_buf) { q.obs2_buf = w.obs2_alt; w.obs2_alt = w.obs2; w.obs2 = q.obs2_buf; }
        if (q.have_prev && q.mask_prev == q.mask) { q.mask = w.mask_alt; w.mask_alt = w.mask; w.mask = q.mask; }
        const int rc = launch_prepare_grid(q, step_grid_args(md, w, scene_start, B, n_max, scene_slots), s);
        if (rc == 0) w.grid_prebuilt = 1;
        return rc;
    }
    return launch_prepare(p, s);
}

static int run_step_body(const tnp_lstm_model *md, int decoder, Workspace &w, const float *h_in, float *h_out,
                         const float *c_in, float *c_out, const int32_t *scene_start, int B, int M, int n_max,
                         const int32_t *scene_slots, hipStream_t s) {
    const int H = md->H;
    if (md->pool_type == TNP_POOL_NN) {          // NearestNeighborMLP straight into the pooled columns of X
        int rc = launch_pool_nn(w.obs1, w.obs2, scene_start, B, md->n, md->C, md->Wp[0], md->bp[0], md->P / md->n,
                                w.pdst, w.pld, s, w.nn_attrs_save, n_max);
        if (rc) return rc;
    } else if (md->pool_type == TNP_POOL_HIDDENMLP) {   // pair embeddings + max-pool, then the projection GEMM
        const int ms = md->dims[0], mv = md->dims[1], mh = md->dims[2];
        int rc = launch_pool_hiddenmlp(w.obs1, w.obs2, mh > 0 ? w.enc : nullptr, mh, 1, scene_start, B, ms, mv, mh,
                                       md->Wp[0], md->bp[0], md->Wp[1], md->bp[1], w.y[0], ms + mh + mv, s);
        if (rc) return rc;
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.A1 = w.y[0]; g.lda1 = ms + mh + mv; g.K1 = ms + mh + mv;
        g.B1 = md->Wp[2]; g.ldb1 = ms + mh + mv;
        g.bias1 = md->bp[2];
        g.M = M; g.N = md->P; g.relu = 0;
        g.C = w.pdst; g.ldc = w.pld;
        rc = launch_linear(g, 0, s);
        if (rc) return rc;
    } else if (md->pool_type == TNP_POOL_NNLSTM || md->pool_type == TNP_POOL_TRAJ) {
        // features -> interaction-encoder LSTMCell over ALL tracks (no presence mask, :450 / :531) -> hidden2pool
        const int Hp = md->dims[0];
        int rc;
        if (md->pool_type == TNP_POOL_NNLSTM)
            rc = launch_pool_nn(w.obs1, w.obs2, scene_start, B, md->n, 4, md->Wp[0], md->bp[0], md->P / md->n, w.y[0], md->P, s,
                                w.nn_attrs_save, n_max);
        else
            rc = launch_pool_traj(w.obs1, w.obs2, M, md->Wp[0], md->bp[0], md->P, w.y[0], md->P, w.scratch, s, w.traj_in_save);
        if (rc) return rc;
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.A1 = w.y[0]; g.lda1 = md->P; g.K1 = md->P;
        g.A2 = w.ph[w.pcur]; g.lda2 = Hp; g.K2 = Hp;
        g.B1 = md->Wx[0]; g.ldb1 = md->P; g.B2 = md->Wx[1]; g.ldb2 = Hp;
        g.bias1 = md->bx[0]; g.bias2 = md->bx[1];
        g.M = M; g.N = 4 * Hp; g.H = Hp;
        g.h_in = w.ph[w.pcur]; g.h_out = w.ph[w.pcur ^ 1]; g.c_in = w.pc; g.c_out = w.pc_out ? w.pc_out : w.pc; g.mask = w.ones;
        g.gates_out = w.pgates_save;
        rc = launch_lstm_gates(g, 0, s);
        if (rc) return rc;
        w.pcur ^= 1;
        memset(&g, 0, sizeof(g));
        g.A1 = w.ph[w.pcur]; g.lda1 = Hp; g.K1 = Hp; g.B1 = md->Wx[2]; g.ldb1 = Hp; g.bias1 = md->bx[2];
        g.M = M; g.N = md->P; g.C = w.pdst; g.ldc = w.pld;
        rc = launch_linear(g, 0, s);
        if (rc) return rc;
    } else if (md->pool_type == TNP_POOL_ATTNMLP) {     // AttentionMLPPooling with the linear maps folded (Wx / bx)
        const int ms = md->dims[0], mv = md->dims[1], mh = md->dims[2], D = ms + mh + mv;
        const float *henc = mh > 0 ? w.enc : nullptr;
        int rc = launch_pool_attn_self(w.obs1, w.obs2, henc, mh, 1, M, ms, mv, mh, md->bp[0], md->bp[1], md->constant,
                                       w.y[0], D, s);
        if (rc) return rc;
        GemmArgs g;
        memset(&g, 0, sizeof(g));                      // q = Wq e_self + bq
        g.A1 = w.y[0]; g.lda1 = D; g.K1 = D; g.B1 = md->Wx[0]; g.ldb1 = D; g.bias1 = md->bx[0];
        g.M = M; g.N = D; g.C = w.y[1]; g.ldc = D;
        rc = launch_linear(g, 0, s);
        if (rc) return rc;
        memset(&g, 0, sizeof(g));                      // u = [Wk^T q ; bk . q ; 0 0 0]
        g.A1 = w.y[1]; g.lda1 = D; g.K1 = D; g.B1 = md->Wx[1]; g.ldb1 = D;
        g.M = M; g.N = D + 4; g.C = w.y[0]; g.ldc = D + 4;
        rc = launch_linear(g, 0, s);
        if (rc) return rc;
        rc = launch_pool_attn_pair(w.obs1, w.obs2, henc, mh, 1, scene_start, B, n_max, scene_slots, ms, mv, mh, md->Wp[0], md->bp[0],
                                   md->Wp[1], md->bp[1], md->constant, w.y[0], D + 4, w.y[1], D, s);
        if (rc) return rc;
        memset(&g, 0, sizeof(g));                      // pooled = Wfin ebar + bfin
        g.A1 = w.y[1]; g.lda1 = D; g.K1 = D; g.B1 = md->Wx[2]; g.ldb1 = D; g.bias1 = md->bx[2];
        g.M = M; g.N = md->P; g.C = w.pdst; g.ldc = w.pld;
        rc = launch_linear(g, 0, s);
        if (rc) return rc;
    } else if (md->pool_type != TNP_POOL_NONE) {
        const GridArgs ga = step_grid_args(md, w, scene_start, B, n_max, scene_slots);
        const bool fused_grid = w.sparse && w.fuse_grid && n_max <= 32767;
        int rc = (fused_grid || w.grid_prebuilt) ? 0 : launch_grid(ga, s);
        w.grid_prebuilt = 0;
        if (rc) return rc;
        if (w.fine) {
            rc = launch_grid_finish(w.grid_fine, w.ldg_fine, M, md->C, md->n, md->pool_size > 1 ? md->pool_size : 1,
                                    md->blur_size > 1 ? md->blur_size : 1, w.grid, w.ldg, s);
            if (rc) return rc;
        }
        const float *src = w.grid;
        int lds = w.ldg;
        int l0 = 0;
        if (w.sparse) {  // first layer straight from the winner table
            const bool last = (md->n_layers == 1);
            float *dst = last ? w.pdst : w.y[0];
            const int ldo = last ? w.pld : md->dims[1];
            // the dominant kernel of the step is timed by the events of its own dispatch when the register-accumulator
            // kernel runs (bench.py's roofline leg), by a bracket of two recorded events otherwise
            const bool disp = sparse_uses_regacc(M, md->C, md->C, md->n * md->n) && prof_dispatch_events(PROF_GEMM1);
            if (!disp) prof_before(PROF_GEMM1, s);
            SparseGridFuse fg;
            fg.obs2 = w.obs2; fg.row_end = w.row_end; fg.row_padded = w.row_padded; fg.G = md->n;
            fg.cell = md->cell; fg.half_x = md->half_x; fg.half_y = md->half_y;
            fg.winners_out = w.save_winners ? w.winners : nullptr;
            rc = launch_pool_embed_sparse(w.winners, w.enc, md->C, w.row_base, md->Wp0_cell_major, md->bp[0], M,
                                          md->n * md->n, md->C, md->dims[1], 1, dst, ldo, w.partial, s,
                                          fused_grid ? &fg : nullptr, md->Wp0_quad_major);
            if (!disp) prof_after(PROF_GEMM1, s);
            else prof_dispatch_release();        // (no-op when the register-accumulator launch took the pair)
            if (rc) return rc;
            src = dst; lds = ldo; l0 = 1;
        }
        for (int l = l0; l < md->n_layers; ++l) {
            const bool last = (l == md->n_layers - 1);
            GemmArgs g;
            memset(&g, 0, sizeof(g));
            g.A1 = src; g.lda1 = lds; g.K1 = md->dims[l];
            g.B1 = md->Wp[l]; g.ldb1 = md->dims[l];
            g.bias1 = md->bp[l];
            g.M = M; g.N = md->dims[l + 1];
            g.relu = 1;
            if (last) { g.C = w.pdst; g.ldc = w.pld; }
            else { g.C = w.y[l & 1]; g.ldc = md->dims[l + 1]; }
            const int cls = (l == 0) ? PROF_GEMM1 : PROF_ALL_GEMM;
            prof_before(cls, s);
            rc = launch_linear(g, (l == 0) ? (md->variant & 0xff) : 0, s);
            prof_after(cls, s);
            if (rc) return rc;
            src = g.C; lds = g.ldc;
        }
    }
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    if (w.to_hidden) {   // hplus = h_in + pooled (rows of absent tracks are never used: their state is copied through)
        const long tot4 = (long)M * H / 4;
        if (w.pvec_save) { int rcc = copy_floats(w.pvec_save, w.hplus, (size_t)M * H, s); if (rcc) return rcc; }
        hipLaunchKernelGGL(add_rows_kernel, dim3((unsigned)((tot4 + 255) / 256)), dim3(256), 0, s, w.hplus, h_in, tot4);
        TNP_HIP(hipGetLastError());
    }
    fill_gates_args(g, md, decoder, w, h_in, h_out, c_in, c_out, M);
    prof_before(PROF_ALL_GEMM, s);
    int rc = launch_lstm_gates(g, (md->variant >> 8) & 0xff, s);
    prof_after(PROF_ALL_GEMM, s);
    return rc;
}

static void fill_prep_common(PrepArgs &p, const tnp_lstm_model *md, const Workspace &w, int M) {
    memset(&p, 0, sizeof(p));
    p.M = M; p.H = md->H; p.E = md->E; p.goal_flag = md->goal_flag; p.goal_dim = md->goal_dim;
    const bool wants_enc = md->pool_type == TNP_POOL_SOCIAL ||
                           ((md->pool_type == TNP_POOL_HIDDENMLP || md->pool_type == TNP_POOL_ATTNMLP) && md->C > 0);
    p.C = wants_enc ? md->C : 0;
    p.I = w.I;
    p.Wn = md->Wn; p.bn = md->bn; p.We = md->We; p.be = md->be; p.Wg = md->Wg; p.bg = md->bg;
    p.Wh = md->Wh; p.bh = md->bh;
    p.obs1_buf = w.obs1; p.obs2_buf = w.obs2; p.mask = w.mask; p.X = w.X;
    p.enc = wants_enc ? w.enc : nullptr;
}

// y[q] = x[q] * g[q] on float4 groups (H % 4 == 0)
__global__ void scale_rows_kernel(float *__restrict__ y, const float *__restrict__ x, const float *__restrict__ g, long n4) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n4) return;
    const float4 a = reinterpret_cast<const float4 *>(x)[q], b = reinterpret_cast<const float4 *>(g)[q];
    reinterpret_cast<float4 *>(y)[q] = make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}

// h[:, H-nd:H] = z (one noise vector shared by all tracks, sgan/sgan.py:213-216)
// group > 0: tracks [g*group, (g+1)*group) carry noise vector g (several generator samples batched as replicated scenes)
__global__ void noise_broadcast_kernel(float *h, int M, int H, int nd, const float *z, int group) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= M * nd) return;
    const int m = q / nd, k = q - m * nd;
    h[(size_t)m * H + (H - nd) + k] = z[(group > 0 ? (m / group) * nd : 0) + k];
}

}  // namespace tnp

using namespace tnp;

extern "C" TNP_API int tnp_abi_version(void) { return TNP_ABI_VERSION; }
extern "C" TNP_API const char *tnp_last_error(void) { return tnp::g_err; }

extern "C" TNP_API size_t tnp_lstm_workspace_bytes(const tnp_lstm_model *model, int M, int B) {
    (void)B;
    if (validate_model(model)) return 0;
    Workspace w;
    plan_workspace(model, M > 0 ? M : 1, nullptr, w);
    return w.bytes;
}

extern "C" TNP_API int tnp_lstm_sparse_first_layer(const tnp_lstm_model *model, int M) {
    if (validate_model(model)) return -1;
    Workspace w;
    plan_workspace(model, M > 0 ? M : 1, nullptr, w);
    return w.sparse ? 1 : 0;
}

static int lstm_forward_impl(const tnp_lstm_model *md, const float *observed, int T_obs, int M,
                             const float *goals, const int32_t *scene_start, const uint8_t *primary_flag,
                             int B, int n_max, const int32_t *scene_slots, const float *truth, int T_dec, float *rel_pred,
                             float *pred, void *workspace, size_t workspace_bytes, const tnp_lstm_extras *ex,
                             const tnp_train_saves *sv, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    int rc = validate_model(md);
    if (rc) return rc;
    const bool stateful = md->pool_type == TNP_POOL_NNLSTM || md->pool_type == TNP_POOL_TRAJ;
    if (sv && stateful && (!sv->ph_all || !sv->pc_all || !sv->pgates_all || !sv->act_all[0]))
        TNP_FAIL(-1, "tnp_lstm_forward_train: stateful interaction encoders need ph_all, pc_all, pgates_all, act_all[0]");
    if (sv && md->pool_type != TNP_POOL_NONE && ((md->variant >> 17) & 1) && !sv->pvec_all)
        TNP_FAIL(-1, "tnp_lstm_forward_train: pool_to_input=False needs pvec_all");
    if (sv && (md->pool_size > 1 || md->blur_size > 1))
        TNP_FAIL(-1, "tnp_lstm_forward_train: pool_size / blur_size != 1 run in inference only (no backward through the blur / pooling reduction)");
    if (sv && (!sv->h_all || !sv->c_all || !sv->X_all || !sv->gates_all || !sv->obs1_all || !sv->obs2_all))
        TNP_FAIL(-1, "tnp_lstm_forward_train: h_all, c_all, X_all, gates_all, obs1_all, obs2_all are required");
    if (T_obs < 2) TNP_FAIL(-1, "need at least 2 observed frames (got %d)", T_obs);
    if (T_dec < 0) TNP_FAIL(-1, "negative decoder length");
    if (M <= 0 || B <= 0) return 0;
    if (md->goal_flag && !goals) TNP_FAIL(-1, "goal_flag set but goals == NULL");
    if (ex && ex->loss_targets && (!ex->loss_values || ex->loss_steps < 1 || ex->loss_steps > (T_obs - 1) + T_dec ||
                                   (ex->loss_mode != 0 && ex->loss_mode != 1)))
        TNP_FAIL(-1, "tnp_lstm_forward_ex: bad fused-loss request (loss_steps %d, mode %d)", ex->loss_steps, ex->loss_mode);
    const bool noisy = ex && ex->noise_dim > 0;
    if (noisy && (ex->noise_dim >= md->H || !ex->W_ctx || !ex->b_ctx || !ex->noise || ex->noise_group_tracks < 0))
        TNP_FAIL(-1, "tnp_lstm_forward_ex: bad noise interface (noise_dim %d)", ex->noise_dim);
    const bool scaled = ex && ex->h_scale != nullptr;      // VAE: h <- h * vae_decoder(z) between encoder and decoder
    if (noisy && scaled) TNP_FAIL(-1, "tnp_lstm_forward_ex: the noise interface and h_scale are exclusive");
    Workspace w;
    plan_workspace(md, M, workspace, w);
    if (workspace == nullptr || workspace_bytes < w.bytes)
        TNP_FAIL(-1, "workspace too small: need %zu bytes, got %zu", w.bytes, workspace_bytes);
    const size_t F = (size_t)M * 2;
    const int H = md->H;
    if (w.sparse) { rc = launch_row_base(scene_start, B, w.row_base, s, w.row_end, w.row_padded, scene_slots, n_max); if (rc) return rc; }
    const size_t MH = (size_t)M * H;
    // training: the states of all steps stay in the caller's [steps + 1, M, H] buffers instead of the ping-pong pair
    float *hcur = sv ? sv->h_all : w.h[0];
    if (!sv && w.c > w.h[0]) {      // lstm.py:207-210; h[0], h[1], c lie one after the other: one fill
        rc = fill_bytes(hcur, 0, (size_t)(reinterpret_cast<char *>(w.c) + MH * 4 - reinterpret_cast<char *>(w.h[0])), s); if (rc) return rc;
    } else {
        rc = fill_bytes(hcur, 0, MH * 4, s); if (rc) return rc;
        rc = fill_bytes(sv ? sv->c_all : w.c, 0, MH * 4, s); if (rc) return rc;
    }
    if (w.ph[0]) {  // pool.reset() (lstm/lstm.py:213-216): zero interaction-encoder state, all tracks "present"
        rc = fill_bytes((sv && stateful) ? sv->ph_all : w.ph[0], 0, (size_t)M * md->dims[0] * 4, s); if (rc) return rc;
        rc = fill_bytes((sv && stateful) ? sv->pc_all : w.pc, 0, (size_t)M * md->dims[0] * 4, s); if (rc) return rc;
        rc = fill_bytes(w.ones, 1, (size_t)M, s); if (rc) return rc;
    }
    int npos = 0, nnorm = 0, cur = 0;
    if (T_obs == 2) {  // lstm.py:222-223
        rc = copy_floats(pred, observed + F, F, s); if (rc) return rc;
        npos = 1;
    }
    const int n_steps = (T_obs - 1) + T_dec;
    for (int st = 0; st <= n_steps; ++st) {
        const float *obs2_prev = w.obs2;
        if (sv && st < n_steps) {   // this step's intermediates go straight into the caller's per-step slices
            const size_t r = (size_t)st * M;
            w.X = sv->X_all + r * w.I;
            w.pdst = w.to_hidden ? w.hplus : (md->pool_type != TNP_POOL_NONE ? w.X + (w.I - md->P) : nullptr);
            if (sv->act_all[0] && md->pool_type != TNP_POOL_ATTNMLP)   // grid MLP: first hidden layer; HiddenStateMLPPooling: the max-pooled vector
                w.y[0] = sv->act_all[0] + r * (md->pool_type == TNP_POOL_HIDDENMLP ? md->dims[0] + md->dims[1] + md->dims[2]
                                               : (stateful ? md->P : md->dims[1]));          // stateful: the pool_lstm's input features
            if (stateful) {   // interaction-encoder state before / after every step, its gates, the embedding's inputs
                const size_t MHp = (size_t)M * md->dims[0];
                w.ph[0] = sv->ph_all + (size_t)st * MHp; w.ph[1] = sv->ph_all + (size_t)(st + 1) * MHp; w.pcur = 0;
                w.pc = sv->pc_all + (size_t)st * MHp; w.pc_out = sv->pc_all + (size_t)(st + 1) * MHp;
                w.pgates_save = sv->pgates_all + r * 4 * md->dims[0];
                w.traj_in_save = sv->traj_in_all ? sv->traj_in_all + r * 8 : nullptr;
            }
            if (sv->act_all[1]) w.y[1] = sv->act_all[1] + r * md->dims[2];
            if (sv->enc_all) w.enc = sv->enc_all + r * md->C;
            w.gates_save = sv->gates_all + r * 4 * H;
            w.nn_attrs_save = sv->nn_attrs_all ? sv->nn_attrs_all + r * md->n * (md->pool_type == TNP_POOL_NNLSTM ? 4 : md->C) : nullptr;
            if (sv->winners_all && w.sparse) { w.winners = sv->winners_all + r * md->n * md->n; w.save_winners = 1; }
            w.pvec_save = (w.to_hidden && sv->pvec_all) ? sv->pvec_all + r * H : nullptr;
            w.obs1 = sv->obs1_all + r * 2;
            w.obs2 = sv->obs2_all + r * 2;
        }
        PrepArgs p;
        fill_prep_common(p, md, w, M);
        p.h = hcur;
        p.primary = primary_flag;
        p.goals = goals;
        p.have_prev = st > 0;
        if (p.have_prev) {
            p.mask_prev = w.mask;       // read before this launch overwrites it: each thread owns one track
            p.obs2_prev = obs2_prev;
            p.normal_out = rel_pred + (size_t)nnorm * M * 5;
            if (ex && ex->loss_targets && nnorm >= n_steps - ex->loss_steps) {   // loss of this output, fused
                const int t = nnorm - (n_steps - ex->loss_steps);
                p.loss_tgt = ex->loss_targets + (size_t)t * M * 2; p.loss_out = ex->loss_values + (size_t)t * M;
                p.loss_mode = ex->loss_mode; p.loss_bg = ex->loss_background_rate; p.prim_prev = primary_flag;
            }
            p.pos_out = pred + (size_t)npos * F;
            ++nnorm; ++npos;            // the entry being written is positions[-1] for the next step
        }
        p.have_next = st < n_steps;
        int decoder = 0;
        if (p.have_next) {
            if (st < T_obs - 1) {       // encoder, lstm.py:226-232
                p.ext1 = observed + (size_t)st * F;
                p.ext2 = observed + (size_t)(st + 1) * F;
            } else {                    // decoder step k, lstm.py:240-250
                const int k = st - (T_obs - 1);
                decoder = 1;
                // positions[-2] at the time of this step = entry npos-2 (the one just written is npos-1)
                p.pos1 = pred + (size_t)(npos - 2) * F;
                if (k == 0) { p.ext1 = observed + (size_t)(T_obs - 1) * F; p.patch1 = 1; }
                else if (truth) { p.ext1 = truth + (size_t)(k - 1) * F; p.patch1 = 1; }
                else { p.ext1 = nullptr; }
                if (truth) { p.ext2 = truth + (size_t)k * F; p.patch2 = 1; }
                else { p.ext2 = nullptr; }
                p.use_pos2 = 1;
            }
        }
        if ((noisy || scaled) && p.have_next && st == T_obs - 1) {
            // S-GAN generator (sgan/sgan.py:200-221, 366): the last encoder step is finished on the clean hidden state,
            // then h <- [relu(W_ctx h + b_ctx) | z] for every track, then the first decoder step is prepared
            PrepArgs pa = p;
            pa.have_next = 0;
            rc = launch_prepare(pa, s);
            if (rc) return rc;
            const float *clean = hcur;
            float *noisy = w.h[cur ^ 1];
            if (sv) {   // h_all[st] becomes the decoder's input state; the clean encoder state is kept beside it
                if (!sv->h_clean) TNP_FAIL(-1, "tnp_lstm_forward_train: noise interface without h_clean");
                { int rcc = copy_floats(sv->h_clean, hcur, MH, s); if (rcc) return rcc; }
                clean = sv->h_clean;
                noisy = hcur;
            }
            if (scaled) {   // VAE add_noise (vae/vae.py:101-105): hidden <- hidden * vae_decoder(z), cell state unchanged
                const long tot4 = (long)MH / 4;
                hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((tot4 + 255) / 256)), dim3(256), 0, s, noisy, clean, ex->h_scale, tot4);
                TNP_HIP(hipGetLastError());
            } else {
            GemmArgs g;
            memset(&g, 0, sizeof(g));
            g.A1 = clean; g.lda1 = H; g.K1 = H;
            g.B1 = ex->W_ctx; g.ldb1 = H; g.bias1 = ex->b_ctx;
            g.M = M; g.N = H - ex->noise_dim; g.C = noisy; g.ldc = H; g.relu = 1;
            rc = launch_linear(g, 0, s);
            if (rc) return rc;
            const int tot = M * ex->noise_dim;
            hipLaunchKernelGGL(noise_broadcast_kernel, dim3((tot + 255) / 256), dim3(256), 0, s, noisy, M, H,
                               ex->noise_dim, ex->noise, ex->noise_group_tracks);
            TNP_HIP(hipGetLastError());
            }
            if (!sv) cur ^= 1;
            hcur = noisy;
            p.h = hcur;
            p.have_prev = 0;
            p.pos2 = pred + (size_t)(npos - 1) * F;
        }
        rc = prepare_step(md, w, p, scene_start, B, n_max, scene_slots, s);
        if (rc) return rc;
        if (p.have_next) {
            float *hnext = sv ? sv->h_all + (size_t)(st + 1) * MH : w.h[cur ^ 1];
            const float *c_in = sv ? sv->c_all + (size_t)st * MH : w.c;
            float *c_out = sv ? sv->c_all + (size_t)(st + 1) * MH : w.c;
            rc = run_step_body(md, decoder, w, hcur, hnext, c_in, c_out, scene_start, B, M, n_max, scene_slots, s);
            if (rc) return rc;
            cur ^= 1;
            hcur = hnext;
        }
    }
    if (ex && ex->h_final) { rc = copy_floats(ex->h_final, hcur, MH, s); if (rc) return rc; }
    return 0;
}

extern "C" TNP_API int tnp_lstm_forward_train(const tnp_lstm_model *md, const float *observed, int T_obs, int M,
                                      const float *goals, const int32_t *scene_start, const uint8_t *primary_flag,
                                      int B, int n_max, const int32_t *scene_slots, const float *truth, int T_dec,
                                      float *rel_pred, float *pred, void *workspace, size_t workspace_bytes,
                                      const tnp_lstm_extras *extras, const tnp_train_saves *saves, void *stream) {
    if (!saves) TNP_FAIL(-1, "tnp_lstm_forward_train: saves == NULL");
    return lstm_forward_impl(md, observed, T_obs, M, goals, scene_start, primary_flag, B, n_max, scene_slots, truth, T_dec, rel_pred,
                             pred, workspace, workspace_bytes, extras, saves, stream);
}

extern "C" TNP_API int tnp_lstm_forward(const tnp_lstm_model *md, const float *observed, int T_obs, int M,
                                const float *goals, const int32_t *scene_start, const uint8_t *primary_flag,
                                int B, int n_max, const int32_t *scene_slots, const float *truth, int T_dec, float *rel_pred,
                                float *pred, void *workspace, size_t workspace_bytes, void *stream) {
    return lstm_forward_impl(md, observed, T_obs, M, goals, scene_start, primary_flag, B, n_max, scene_slots, truth, T_dec, rel_pred,
                             pred, workspace, workspace_bytes, nullptr, nullptr, stream);
}

extern "C" TNP_API int tnp_lstm_forward_ex(const tnp_lstm_model *md, const float *observed, int T_obs, int M,
                                   const float *goals, const int32_t *scene_start, const uint8_t *primary_flag,
                                   int B, int n_max, const int32_t *scene_slots, const float *truth, int T_dec, float *rel_pred,
                                   float *pred, void *workspace, size_t workspace_bytes, const tnp_lstm_extras *extras,
                                   void *stream) {
    return lstm_forward_impl(md, observed, T_obs, M, goals, scene_start, primary_flag, B, n_max, scene_slots, truth, T_dec, rel_pred,
                             pred, workspace, workspace_bytes, extras, nullptr, stream);
}

static int lstm_step_impl(const tnp_lstm_model *md, int decoder, const float *h_in, const float *c_in,
                             const float *obs1, const float *obs2, const float *goals, const int32_t *scene_start,
                             int B, int M, int n_max, const int32_t *scene_slots, float *h_out, float *c_out, float *normal,
                             void *workspace, size_t workspace_bytes, const tnp_step_saves *sv, void *stream) {
    hipStream_t s = (hipStream_t)stream;
    int rc = validate_model(md);
    if (rc) return rc;
    if (M <= 0 || B <= 0) return 0;
    if (h_in == h_out) TNP_FAIL(-1, "tnp_lstm_step: h_in and h_out must not alias");
    if (md->pool_type == TNP_POOL_NNLSTM || md->pool_type == TNP_POOL_TRAJ)
        TNP_FAIL(-1, "tnp_lstm_step: stateful interaction encoders (pool_lstm) only run inside tnp_lstm_forward");
    if (sv && (md->pool_size > 1 || md->blur_size > 1))
        TNP_FAIL(-1, "tnp_lstm_step_train: pool_size / blur_size != 1 run in inference only");
    Workspace w;
    plan_workspace(md, M, workspace, w);
    if (workspace == nullptr || workspace_bytes < w.bytes)
        TNP_FAIL(-1, "workspace too small: need %zu bytes, got %zu", w.bytes, workspace_bytes);
    if (sv) {   // training: the step writes its intermediates straight into the caller's buffers
        if (sv->X) { w.X = sv->X; w.pdst = w.to_hidden ? w.hplus : (md->pool_type != TNP_POOL_NONE ? w.X + (w.I - md->P) : nullptr); }
        if (sv->act[0]) w.y[0] = sv->act[0];
        if (sv->act[1]) w.y[1] = sv->act[1];
        if (sv->enc) w.enc = sv->enc;
        w.gates_save = sv->gates;
        w.nn_attrs_save = sv->nn_attrs;
        if (sv->winners && w.sparse) { w.winners = sv->winners; w.save_winners = 1; }
    }
    if (w.sparse) { rc = launch_row_base(scene_start, B, w.row_base, s, w.row_end, w.row_padded, scene_slots, n_max); if (rc) return rc; }
    PrepArgs p;
    fill_prep_common(p, md, w, M);
    p.h = h_in; p.goals = goals;
    p.have_prev = 0; p.have_next = 1;
    p.ext1 = obs1; p.ext2 = obs2;
    // primary flags are only consulted when patching; none here
    p.primary = w.mask;  // any valid [M] byte buffer (unused: patch1 = patch2 = 0)
    rc = prepare_step(md, w, p, scene_start, B, n_max, scene_slots, s);
    if (rc) return rc;
    rc = run_step_body(md, decoder, w, h_in, h_out, c_in, c_out, scene_start, B, M, n_max, scene_slots, s);
    if (rc) return rc;
    // Hidden2Normal of the new state; positions are the caller's business in step mode
    PrepArgs q;
    fill_prep_common(q, md, w, M);
    q.h = h_out; q.have_prev = 1; q.have_next = 0;
    q.mask_prev = w.mask; q.obs2_prev = w.obs2; q.normal_out = normal; q.pos_out = w.obs1;  // scratch
    return launch_prepare(q, s);
}

extern "C" TNP_API int tnp_lstm_step(const tnp_lstm_model *md, int decoder, const float *h_in, const float *c_in,
                             const float *obs1, const float *obs2, const float *goals, const int32_t *scene_start,
                             int B, int M, int n_max, const int32_t *scene_slots, float *h_out, float *c_out, float *normal,
                             void *workspace, size_t workspace_bytes, void *stream) {
    return lstm_step_impl(md, decoder, h_in, c_in, obs1, obs2, goals, scene_start, B, M, n_max, scene_slots, h_out, c_out, normal,
                          workspace, workspace_bytes, nullptr, stream);
}

extern "C" TNP_API int tnp_lstm_step_train(const tnp_lstm_model *md, int decoder, const float *h_in, const float *c_in,
                                   const float *obs1, const float *obs2, const float *goals,
                                   const int32_t *scene_start, int B, int M, int n_max, const int32_t *scene_slots,
                                   float *h_out, float *c_out, float *normal, const tnp_step_saves *saves, void *workspace,
                                   size_t workspace_bytes, void *stream) {
    return lstm_step_impl(md, decoder, h_in, c_in, obs1, obs2, goals, scene_start, B, M, n_max, scene_slots, h_out, c_out, normal,
                          workspace, workspace_bytes, saves, stream);
}

extern "C" TNP_API int tnp_linear_forward(const float *A, int lda, const float *W, int ldw, const float *bias, float *C,
                                  int ldc, int M, int N, int K, int relu, int variant, void *stream) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A1 = A; g.lda1 = lda; g.K1 = K; g.B1 = W; g.ldb1 = ldw; g.bias1 = bias;
    g.M = M; g.N = N; g.C = C; g.ldc = ldc; g.relu = relu;
    if (M <= 0 || N <= 0) return 0;
    prof_before(PROF_GEMM1, (hipStream_t)stream);
    int rc = launch_linear(g, variant, (hipStream_t)stream);
    prof_after(PROF_GEMM1, (hipStream_t)stream);
    return rc;
}

extern "C" TNP_API int tnp_tuning_set(const char *key, long value) {
    tnp::Tuning &t = tnp::tuning();
    const std::string k = key ? key : "";
    if (k == "sparse_tile") { t.sparse_te = (int)(value >> 8); t.sparse_ncs = (int)(value & 0xff); }
    else if (k == "sparse_min_wg") t.sparse_min_wg = value;
    else if (k == "skinny_max_rows") t.skinny_max_rows = (int)value;
    else if (k == "skinny_gates_max_rows") t.skinny_gates_max_rows = (int)value;
    else if (k == "fuse_prepare_grid") t.fuse_prepare_grid = (int)value;
    else if (k == "sparse_wgrad_plan") t.sparse_wgrad_plan = (int)value;
    else if (k == "wgrad_min_rows") { if (value <= 0) TNP_FAIL(-1, "tnp_tuning_set: wgrad_min_rows must be positive"); t.wgrad_min_rows = (int)value; }
    else if (k == "wgrad_target_wgs") { if (value <= 0) TNP_FAIL(-1, "tnp_tuning_set: wgrad_target_wgs must be positive"); t.wgrad_target_wgs = (int)value; }
    else TNP_FAIL(-1, "tnp_tuning_set: unknown key '%s' (sparse_tile, sparse_min_wg, skinny_max_rows, skinny_gates_max_rows, "
                  "sparse_wgrad_plan, wgrad_min_rows, wgrad_target_wgs, fuse_prepare_grid)", k.c_str());
    return 0;
}

extern "C" TNP_API int tnp_profile_begin(int which) {
    if (which != PROF_GEMM1 && which != PROF_ALL_GEMM) TNP_FAIL(-1, "tnp_profile_begin: which must be 0 or 1");
    tnp::g_prof_cls = which;
    tnp::g_prof_n = 0; tnp::g_prof_disp_n = 0;
    return 0;
}
extern "C" TNP_API int tnp_profile_dispatch_timed(void) { return tnp::g_prof_disp_last; }
extern "C" TNP_API int tnp_profile_read(double *total_ms, int *launches) {
    double tot = 0.0;
    for (int i = 0; i < tnp::g_prof_n; ++i) {
        TNP_HIP(hipEventSynchronize(tnp::g_prof_ev[2 * i + 1]));
        float ms = 0.f;
        TNP_HIP(hipEventElapsedTime(&ms, tnp::g_prof_ev[2 * i], tnp::g_prof_ev[2 * i + 1]));
        tot += ms;
    }
    if (total_ms) *total_ms = tot;
    if (launches) *launches = tnp::g_prof_n;
    tnp::g_prof_disp_last = tnp::g_prof_disp_n;
    tnp::g_prof_n = 0; tnp::g_prof_disp_n = 0;
    return 0;
}
extern "C" TNP_API int tnp_profile_end(void) { tnp::g_prof_cls = -1; tnp::g_prof_n = 0; return 0; }

