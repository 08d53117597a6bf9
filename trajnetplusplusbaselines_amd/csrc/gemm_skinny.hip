// fp32 GEMMs of the recurrent step for SMALL batches (one scene per call: the reference's evaluator, lstm/trajnet_evaluator.py:
// 15-19,61-62; batch_size 8: its trainer, lstm/trainer.py:96-133), gfx950.
//
//   C[M,N] = epilogue( [A1|A2][M,K] @ [B1|B2][N,K]^T + bias1 (+ bias2) )          (GemmArgs, tnp_internal.h)
//
// With a few hundred tracks the 32 x 32 x 2 tiles of gemm_f32_mfma.hip give a handful of workgroups on an empty chip; what a
// launch then costs is ONE workgroup's latency chain (global -> registers -> LDS ring -> barrier -> MFMA -> split-K swap ->
// epilogue: 9.4 us for the second embedding layer and 14.2 us for the gates at 36 tracks, profiles/round5_small_*), while the
// whole layer is 1 MB of weights.  Here:
//   * v_mfma_f32_16x16x4_f32 tiles (16 tracks): three times the workgroups of a 32-row tile at 36 tracks, every product still
//     an exact fp32 fma;
//   * NO LDS staging: lane (i = l & 15, q = l >> 4) loads float4 chunks row i, k = 16 c + 4 q .. + 3 of both operands straight
//     into registers -- element j of the chunk is the operand of the j-th MFMA of that chunk (the assignment of k to MFMA slots
//     is free as long as both operands use the same one), so all loads of a wave are in flight at once and the only barrier is
//     the one of the split-K reduction;
//   * K is split over the KS waves of the workgroup AND the chunks of a wave alternate between two accumulators (a dependent
//     16x16x4 chain issues every 40 cycles, two chains every 32); partial tiles are added in fixed order through LDS:
//     deterministic, batch-invariant (a row's sum does not depend on M);
//   * EPI_LSTM: the MFMA's A operand is the WEIGHT tile -- 16 rows = 4 hidden units x 4 gates -- and B the 16 tracks, so a lane's
//     four accumulator registers are i, f, g, o of ONE (track, unit): torch.nn.LSTMCell's pointwise part runs in registers.
// Blocks b, b + 8, ... share an XCD and a weight column tile (tn = b % tiles_n): each XCD's L2 keeps its slice of the weights.
#include "tnp_internal.h"
#include "lstm_cell.h"

namespace tnp {

typedef float sk_f32x4 __attribute__((ext_vector_type(4)));

// CT: 16-wide column tiles (EPI_BIAS) / 4-unit groups (EPI_LSTM) per workgroup; KS: K split; MAXC: chunks of 16 k in flight
template <int EPI, int CT, int KS, int MAXC>
__global__ void __launch_bounds__(64 * CT * KS) gemm_skinny_kernel(const GemmArgs g) {
    __shared__ float red[(KS > 1 ? KS - 1 : 1) * CT * 4 * 64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = wave % CT, ks = wave / CT;
    const int tn = blockIdx.x % g.tiles_n, tm = blockIdx.x / g.tiles_n;
    const int l16 = lane & 15, kq = lane >> 4;
    const int K1 = g.K1, K = g.K1 + g.K2;
    const int nch = K >> 4;
    const int cpw = (nch + KS - 1) / KS;
    const int cb = ks * cpw, ce = min(cb + cpw, nch);

    // P = the MFMA's A operand rows (16 x K), Q = its B operand rows (16 x K); both K-contiguous, each possibly two sources
    const float *p1, *p2, *q1, *q2;
    {
        const int track = min(tm * 16 + l16, g.M - 1);
        const float *a1 = g.A1 + (size_t)track * g.lda1, *a2 = g.K2 > 0 ? g.A2 + (size_t)track * g.lda2 : a1;
        int wr;
        if (EPI == EPI_LSTM) wr = (l16 & 3) * g.H + min((tn * CT + ct) * 4 + (l16 >> 2), g.H - 1);
        else wr = min((tn * CT + ct) * 16 + l16, g.N - 1);
        const float *b1 = g.B1 + (size_t)wr * g.ldb1, *b2 = g.K2 > 0 ? g.B2 + (size_t)wr * g.ldb2 : b1;
        if (EPI == EPI_LSTM) { p1 = b1; p2 = b2; q1 = a1; q2 = a2; }
        else { p1 = a1; p2 = a2; q1 = b1; q2 = b2; }
        p1 += 4 * kq; q1 += 4 * kq; p2 += 4 * kq; q2 += 4 * kq;
    }
    // epilogue operands of the waves that run it (k group 0), requested BEFORE the main loop: behind it they would be one more
    // exposed global round trip at the tail of a kernel that is nothing but its latency chain
    float e_bias[4] = {0.0f, 0.0f, 0.0f, 0.0f}, e_c = 0.0f;
    int e_present = 0;
    if (ks == 0) {
        if (EPI == EPI_LSTM) {
            const int unit = min((tn * CT + ct) * 4 + kq, g.H - 1), track = min(tm * 16 + l16, g.M - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) e_bias[r] = g.bias1[r * g.H + unit] + g.bias2[r * g.H + unit];
            e_c = g.c_in[(size_t)track * g.H + unit];
            e_present = g.mask[track];
        } else {
            const int col = min((tn * CT + ct) * 16 + l16, g.N - 1);
            e_bias[0] = g.bias1 ? g.bias1[col] : 0.0f;
            if (g.bias2) e_bias[0] += g.bias2[col];
        }
    }
    sk_f32x4 acc0 = {0.0f, 0.0f, 0.0f, 0.0f}, acc1 = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int c0 = cb; c0 < ce; c0 += MAXC) {
        sk_f32x4 pv[MAXC], qv[MAXC];
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int k = min(c0 + i, ce - 1) * 16;                             // wave-uniform; a chunk never straddles the sources
            const bool first = k < K1;
            pv[i] = *reinterpret_cast<const sk_f32x4 *>(first ? p1 + k : p2 + (k - K1));
            qv[i] = *reinterpret_cast<const sk_f32x4 *>(first ? q1 + k : q2 + (k - K1));
        }
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            if (c0 + i < ce) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[i].x, qv[i].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[i].y, qv[i].y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[i].z, qv[i].z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(pv[i].w, qv[i].w, acc1, 0, 0, 0);
            }
        }
    }
    sk_f32x4 acc = acc0 + acc1;
    if (KS > 1) {                                                               // partial tiles: k group 0 + 1 + ... in order
        if (ks > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(((ks - 1) * CT + ct) * 4 + r) * 64 + lane] = acc[r];
        }
        __syncthreads();
        if (ks > 0) return;
#pragma unroll
        for (int k = 1; k < KS; ++k)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += red[(((k - 1) * CT + ct) * 4 + r) * 64 + lane];
    }
    // accumulator register r of lane (l16, kq) = D[row 4 kq + r][column l16]
    if (EPI == EPI_LSTM) {
        const int unit = (tn * CT + ct) * 4 + kq, track = tm * 16 + l16;
        if (unit >= g.H) return;
        lstm_cell_store_pf(g, track, unit, acc[0] + e_bias[0], acc[1] + e_bias[1], acc[2] + e_bias[2], acc[3] + e_bias[3], e_c, e_present);
    } else {
        const int col = (tn * CT + ct) * 16 + l16;
        if (col >= g.N) return;
        const float b = e_bias[0];
        // masked copies (GemmArgs::seg, the ReLU backwards fused into the data-gradient GEMM): a column belongs to at most one
        const float *sact = nullptr; float *sout = nullptr;
        int sc = 0, lda_s = 0, ldo_s = 0;
        for (int si = 0; si < g.nseg; ++si) {
            const GemmArgs::EpiSeg &sg = g.seg[si];
            const int c = col - sg.col0;
            if (c >= 0 && c < sg.n) { sact = sg.act; sout = sg.out; sc = c; lda_s = sg.ld_act; ldo_s = sg.ld_out; }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = tm * 16 + 4 * kq + r;
            if (row >= g.M) continue;
            const float raw = acc[r] + b;
            float v = raw;
            if (g.relu) v = v > 0.0f ? v : 0.0f;
            if (g.mask_act) v = g.mask_act[(size_t)row * g.ld_mask + col] > 0.0f ? v : 0.0f;
            g.C[(size_t)row * g.ldc + col] = v;
            if (sout) sout[(size_t)row * ldo_s + sc] = sact[(size_t)row * lda_s + sc] > 0.0f ? raw : 0.0f;
        }
    }
}

// K chunks of 16 that never straddle the two sources, float4 loads
bool skinny_ok(const GemmArgs &g) {
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (g.M <= 0 || g.K1 <= 0 || g.K1 % 16 != 0 || g.K2 % 16 != 0) return false;
    if (g.lda1 % 4 != 0 || g.ldb1 % 4 != 0 || !al(g.A1) || !al(g.B1)) return false;
    if (g.K2 > 0 && (g.lda2 % 4 != 0 || g.ldb2 % 4 != 0 || !al(g.A2) || !al(g.B2))) return false;
    return true;
}

template <int EPI, int CT, int KS>
static int launch_skinny_t(GemmArgs g, hipStream_t s) {
    g.tiles_m = (g.M + 15) / 16;
    g.tiles_n = EPI == EPI_LSTM ? (g.H + 4 * CT - 1) / (4 * CT) : (g.N + 16 * CT - 1) / (16 * CT);
    hipLaunchKernelGGL((gemm_skinny_kernel<EPI, CT, KS, 8>), dim3(g.tiles_m * g.tiles_n), dim3(64 * CT * KS), 0, s, g);
    TNP_HIP(hipGetLastError());
    return 0;
}

// variant: 40 .. 45 (dense layer), 30 .. 33 (gates); see launch_linear / launch_lstm_gates for the automatic choice
int launch_skinny_linear(const GemmArgs &g, int variant, hipStream_t s) {
    if (!skinny_ok(g)) TNP_FAIL(-1, "skinny GEMM: K (%d + %d) must come in chunks of 16 floats, 16-byte aligned rows", g.K1, g.K2);
    switch (variant) {
        case 40: return launch_skinny_t<EPI_BIAS, 1, 8>(g, s);    // 16 x 16, K over eight waves
        case 41: return launch_skinny_t<EPI_BIAS, 2, 4>(g, s);    // 16 x 32, K over four
        case 42: return launch_skinny_t<EPI_BIAS, 4, 4>(g, s);    // 16 x 64, K over four (sixteen waves)
        case 43: return launch_skinny_t<EPI_BIAS, 2, 8>(g, s);    // 16 x 32, K over eight (sixteen waves)
        case 44: return launch_skinny_t<EPI_BIAS, 4, 2>(g, s);    // 16 x 64, K over two
        case 45: return launch_skinny_t<EPI_BIAS, 1, 4>(g, s);    // 16 x 16, K over four
        default: TNP_FAIL(-1, "skinny GEMM: unknown variant %d (40 .. 45)", variant);
    }
}

int launch_skinny_gates(const GemmArgs &g, int variant, hipStream_t s) {
    if (!skinny_ok(g)) TNP_FAIL(-1, "skinny gates GEMM: K (%d + %d) must come in chunks of 16 floats, 16-byte aligned rows", g.K1, g.K2);
    switch (variant) {
        case 30: return launch_skinny_t<EPI_LSTM, 2, 4>(g, s);    // 16 tracks x 8 units, K over four
        case 31: return launch_skinny_t<EPI_LSTM, 4, 4>(g, s);    // 16 tracks x 16 units, K over four (sixteen waves)
        case 32: return launch_skinny_t<EPI_LSTM, 2, 8>(g, s);    // 16 tracks x 8 units, K over eight (sixteen waves)
        case 33: return launch_skinny_t<EPI_LSTM, 1, 8>(g, s);    // 16 tracks x 4 units, K over eight
        case 34: return launch_skinny_t<EPI_LSTM, 1, 4>(g, s);    // 16 tracks x 4 units, K over four
        default: TNP_FAIL(-1, "skinny gates GEMM: unknown variant %d (30 .. 34)", variant);
    }
}

}  // namespace tnp
