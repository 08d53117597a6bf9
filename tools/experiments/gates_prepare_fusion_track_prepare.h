// The per-track "prepare" work between two recurrent steps (finish step s: Hidden2Normal, predicted position, fused loss;
// set up step s + 1: obs1 / obs2 selection, presence mask, InputEmbedding, social encoding) as a device function, so that it
// can run as a kernel of its own (lstm_seq.hip: track_prepare_kernel) or in the tail of the small-batch gates kernel
// (gemm_skinny.hip: the workgroup that arrives last for a 16-track tile runs it for those tracks).
#pragma once
#include "tnp_internal.h"

namespace tnp {

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


__device__ __forceinline__ float sigmoid_dev(float x) { return 1.0f / (1.0f + expf(-x)); }

// 256 threads = 8 tracks x 32 lanes.  Phase 1: the 5 + C dot products of length H per track (h row and the
// weight rows staged in LDS, weight stride H+1 -> conflict free).  Phase 2: per-track scalar bookkeeping.
// Phase 3: the E-2 (+ goal) embedding outputs, 32 lanes per track.
// HC: the rows of a.h were written INSIDE this launch by other workgroups (write-through stores, agent scope): read them with
// agent-scope (sc1) loads, which miss the non-coherent caches instead of returning a line from two steps ago
template <int NT, bool HC = false>
__device__ __forceinline__ void track_prepare_body(const PrepArgs &a, const int m0, float *psm) {
    constexpr int PREP_TRACKS = NT / 32;
    const int H = a.H;
    const int nout = (a.have_prev ? 5 : 0) + (a.have_next ? a.C : 0);  // rows of the stacked weight
    float *hs = psm;                        // [8][H]
    float *ws = hs + PREP_TRACKS * H;       // [nout][H+4]: rows 16-byte aligned, lanes (= outputs) 4 banks apart
    float *outs = ws + nout * (H + 4);      // [8][nout]
    float *ob = outs + PREP_TRACKS * (nout > 0 ? nout : 1);  // [8][8]: obs1.xy obs2.xy mask goal.xy
    const int tid = threadIdx.x;
    const int t_local = tid >> 5, l32 = tid & 31;
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
        for (int q0 = 0; q0 < PREP_TRACKS * H4; q0 += NT * HB) {
            float4 v[HB];
#pragma unroll
            for (int i = 0; i < HB; ++i) {
                const int q = q0 + tid + NT * i;
                const int t = q / H4, k4 = q - t * H4;
                v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < PREP_TRACKS * H4 && m0 + t < a.M) {
                    const float4 *src = reinterpret_cast<const float4 *>(a.h + (size_t)(m0 + t) * H) + k4;
                    if constexpr (HC) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[i]) : "v"(src) : "memory");
                    else v[i] = *src;
                }
            }
            if constexpr (HC) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < HB; ++i) {
                const int q = q0 + tid + NT * i;
                if (q < PREP_TRACKS * H4) reinterpret_cast<float4 *>(hs)[q] = v[i];
            }
        }
        for (int q0 = 0; q0 < nout * H4; q0 += NT * WB) {
            float4 v[WB];
#pragma unroll
            for (int i = 0; i < WB; ++i) {
                const int q = q0 + tid + NT * i;
                const int o = q / H4, k4 = q - o * H4;
                v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (q < nout * H4) {
                    const float *row = (a.have_prev && o < 5) ? a.Wn + (size_t)o * H : a.Wh + (size_t)(o - (a.have_prev ? 5 : 0)) * H;
                    v[i] = reinterpret_cast<const float4 *>(row)[k4];
                }
            }
#pragma unroll
            for (int i = 0; i < WB; ++i) {
                const int q = q0 + tid + NT * i;
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


// LDS floats the body needs for NT threads (NT / 32 tracks)
static inline size_t track_prepare_smem_floats(const PrepArgs &a, int tracks) {
    const int nout = (a.have_prev ? 5 : 0) + (a.have_next ? a.C : 0);
    return (size_t)tracks * a.H + (size_t)nout * (a.H + 4) + (size_t)tracks * (nout > 0 ? nout : 1) + (size_t)tracks * 8;
}

}  // namespace tnp
