// fp32 GEMM on the CDNA4 matrix cores for the TrajNet++ hot path (gfx950 only).
//
//   C[M,N] = epilogue( [A1|A2][M,K] @ [B1|B2][N,K]^T + bias1 (+ bias2) )
//
// * v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate): every product is an exact fp32 fma, which the
//   1e-4 ADE/FDE parity bar against the reference's fp32 CPU path needs.  64 cycles per instruction per
//   SIMD, so the kernel is matrix-pipe bound and everything else (LDS, address math) hides under it.
// * Both operands are K-contiguous ("NT" GEMM): A rows are tracks, B rows are PyTorch [out,in] weight rows,
//   so the reference's state_dict tensors are consumed in place.  A and B may each be the concatenation of
//   two matrices along K: the LSTM gates GEMM reads [x | h] and [W_ih | W_hh] without any packing copy.
// * 64-wide wavefronts: one wave owns 1 x AN blocks of 32x32 (16 accumulator VGPRs each).  A workgroup is
//   WM x WN x WK waves; WK > 1 splits K across waves of the SAME workgroup (reduced through LDS), which is how
//   the small GEMMs of the step (M = 2048 tracks) still put an MFMA stream on all 1024 SIMDs of the chip.
// * Operand tiles are staged global -> registers -> LDS (row stride BK+4 floats: conflict-free for the
//   16-lane groups of ds_read_b128), double buffered, one barrier per K step.  Each lane reads 4 consecutive
//   k per ds_read_b128; the four MFMAs of a k8 step pair k with k+4 (same pairing for A and B).
// * Workgroup -> tile mapping is XCD aware: blocks b, b+8, ... share an XCD (and its 4 MiB L2), so each XCD
//   gets a contiguous chunk of a panel-ordered tile list (4 tile-rows x many tile-columns share A rows / W rows).
// * EPI_LSTM fuses torch.nn.LSTMCell's pointwise part (reference lstm/lstm.py:154): the 128 tile columns are
//   the i,f,g,o gates of 32 hidden units, so one lane holds all four gates of its (track, unit) pairs.
#include "tnp_internal.h"

namespace tnp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// Branch-free operand fetch: the address is always legal (invalid lanes are pointed at `safe`), the raw value
// is kept in registers and zeroed only when it is written to LDS (after the MFMAs of the current K step), so
// hipcc emits straight-line global_load_dwordx4 and waits for them behind the matrix work, not in front of it.
template <bool VEC>
__device__ __forceinline__ f32x4 load4_dual(const float *__restrict__ r1, int K1, const float *__restrict__ r2,
                                            int K2, int k, bool row_ok, const float *__restrict__ safe,
                                            unsigned &okbits, int bit) {
    f32x4 v;
    if (VEC) {  // K1, K2 multiples of 4: a chunk never straddles a source or the end of K
        const bool ok = row_ok && (k < K1 + K2);
        const float *p = (k < K1) ? (r1 + k) : (r2 + (k - K1));
        p = ok ? p : safe;
        v = *reinterpret_cast<const f32x4 *>(p);
        okbits |= ok ? (0xFu << bit) : 0u;
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kk = k + q;
            const bool ok = row_ok && (kk < K1 + K2);
            const float *p = (kk < K1) ? (r1 + kk) : (r2 + (kk - K1));
            p = ok ? p : safe;
            v[q] = *p;
            okbits |= ok ? (1u << (bit + q)) : 0u;
        }
    }
    return v;
}

// blocks b, b+8, b+16, ... run on the same XCD: hand each XCD a contiguous chunk of the panel-ordered tiles
__device__ __forceinline__ void tile_coords(int bid, int tiles_m, int tiles_n, int &tm, int &tn) {
    const int T = tiles_m * tiles_n;
    int lin = bid;
    if (T >= 16) {
        const int q = T >> 3, r = T & 7;
        const int xcd = bid & 7, slot = bid >> 3;
        lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    constexpr int PM = 4;
    const int per_panel = PM * tiles_n;
    const int p = lin / per_panel;
    const int within = lin - p * per_panel;
    const int rows = min(PM, tiles_m - p * PM);
    tm = p * PM + within % rows;
    tn = within / rows;
}

__device__ __forceinline__ float sigmoidf_acc(float x) { return 1.0f / (1.0f + expf(-x)); }

template <int WM, int WN, int WK, int AN, int BK, int PF, int EPI, bool VEC>
__global__ void __launch_bounds__(64 * WM * WN * WK) gemm_nt_kernel(const GemmArgs g) {
    constexpr int BM = 32 * WM;
    constexpr int BN = 32 * WN * AN;
    constexpr int NT = 64 * WM * WN * WK;
    constexpr int WMN = WM * WN;
    constexpr int LDS_STRIDE = BK + 4;
    constexpr int ROWS = BM + BN;
    constexpr int GROUP_FLOATS = ROWS * LDS_STRIDE;
    constexpr int C4 = BK / 4;
    constexpr int CHUNKS = WK * ROWS * C4;
    static_assert(CHUNKS % NT == 0, "loader must tile evenly");
    constexpr int CH = CHUNKS / NT;
    static_assert(EPI != EPI_LSTM || (WN == 1 && AN == 4), "LSTM epilogue: 4 gate blocks per wave");

    extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][WK][GROUP_FLOATS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int kg = wave / WMN;
    const int wq = wave - kg * WMN;
    const int wm = wq / WN;
    const int wn = wq - wm * WN;

    int tm, tn;
    tile_coords(blockIdx.x, g.tiles_m, g.tiles_n, tm, tn);
    const int m0 = tm * BM;
    const int n0 = tn * BN;  // EPI_LSTM: tn indexes a block of 32 hidden units

    const int K = g.K1 + g.K2;
    const int Kg = (((K + WK - 1) / WK) + BK - 1) / BK * BK;  // K span of one k-group
    const int KT = Kg / BK;

    // kernel arguments into scalars once (selecting between struct fields per lane would otherwise turn
    // into per-lane loads of the argument block)
    const float *const gA1 = g.A1, *const gA2 = g.A2, *const gB1 = g.B1, *const gB2 = g.B2;
    const int lda1 = g.lda1, lda2 = g.lda2, ldb1 = g.ldb1, ldb2 = g.ldb2;
    const int K1 = g.K1, K2 = g.K2, gM = g.M, gN = g.N, gH = g.H;

    static_assert(CH * 4 <= 64, "ok bits");
    static_assert(PF >= 1 && PF <= 4, "prefetch depth");
    // PF register stage sets: the tile of K step t lives in set t % PF from the moment its loads are issued
    // (PF steps before it is needed) until it is written to LDS, so global-load latency has PF-1 full
    // compute phases (+ the current one) to hide under.
    f32x4 stage[PF][CH];
    unsigned long long okbits[PF];

    auto load_stage = [&](int kt, f32x4 (&st)[CH], unsigned long long &okb) {
        unsigned lo = 0u, hi = 0u;
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int idx = tid + c * NT;
            const int grp = idx / (ROWS * C4);
            const int rem = idx - grp * (ROWS * C4);
            const int row = rem / C4;
            const int c4 = rem - row * C4;
            const int k = grp * Kg + kt * BK + c4 * 4;
            const bool is_a = row < BM;
            // source row: a track of A, or a weight row of W
            int wr;
            bool ok;
            if (EPI == EPI_LSTM) {
                const int cc = row - BM;
                const int unit = tn * 32 + (cc & 31);
                wr = (cc >> 5) * gH + unit;
                ok = unit < gH;
            } else {
                wr = n0 + (row - BM);
                ok = wr < gN;
            }
            const int m = m0 + row;
            const bool row_ok = (is_a ? (m < gM) : ok) && (kt < KT);
            const size_t r = is_a ? (size_t)m : (size_t)wr;
            const float *r1 = (is_a ? gA1 : gB1) + r * (size_t)(is_a ? lda1 : ldb1);
            const float *r2 = (is_a ? gA2 : gB2) + r * (size_t)(is_a ? lda2 : ldb2);
            if (c < 8) st[c] = load4_dual<VEC>(r1, K1, r2, K2, k, row_ok, gA1, lo, (c & 7) * 4);
            else st[c] = load4_dual<VEC>(r1, K1, r2, K2, k, row_ok, gA1, hi, (c & 7) * 4);
        }
        okb = ((unsigned long long)hi << 32) | lo;
    };
    auto store_stage = [&](int buf, const f32x4 (&st)[CH], unsigned long long okb) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int idx = tid + c * NT;
            const int grp = idx / (ROWS * C4);
            const int rem = idx - grp * (ROWS * C4);
            const int row = rem / C4;
            const int c4 = rem - row * C4;
            float *dst = smem + (size_t)(buf * WK + grp) * GROUP_FLOATS + row * LDS_STRIDE + c4 * 4;
            f32x4 v = st[c];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = ((okb >> (c * 4 + q)) & 1ull) ? v[q] : 0.0f;
            *reinterpret_cast<f32x4 *>(dst) = v;
        }
    };

    f32x16 acc[AN];
#pragma unroll
    for (int an = 0; an < AN; ++an)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[an][r] = 0.0f;

    // prologue: tile 0 -> LDS buffer 0, tiles 1..PF in flight (tile t in set t % PF)
    load_stage(0, stage[0], okbits[0]);
#pragma unroll
    for (int p = 1; p < PF; ++p) load_stage(p, stage[p], okbits[p]);
    store_stage(0, stage[0], okbits[0]);
    load_stage(PF, stage[0], okbits[0]);
    __syncthreads();

    const int a_off = (wm * 32 + (lane & 31)) * LDS_STRIDE + (lane >> 5) * 4;
    const int b_off = BM * LDS_STRIDE + (wn * AN * 32 + (lane & 31)) * LDS_STRIDE + (lane >> 5) * 4;

    auto compute = [&](int buf) {
        const float *base = smem + (size_t)(buf * WK + kg) * GROUP_FLOATS;
#pragma unroll
        for (int k8 = 0; k8 < BK / 8; ++k8) {
            const f32x4 a4 = *reinterpret_cast<const f32x4 *>(base + a_off + k8 * 8);
            f32x4 b4[AN];
#pragma unroll
            for (int an = 0; an < AN; ++an)
                b4[an] = *reinterpret_cast<const f32x4 *>(base + b_off + an * 32 * LDS_STRIDE + k8 * 8);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int an = 0; an < AN; ++an)
                    acc[an] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[q], b4[an][q], acc[an], 0, 0, 0);
        }
    };

    // main loop, unrolled by PF so that the register-set index is a compile-time constant.
    // Iteration kt: tile kt+1 (issued PF steps ago) goes registers -> LDS buffer (kt+1)&1 (free since the
    // barrier of iteration kt-1), its set is refilled with tile kt+1+PF, then the MFMAs of tile kt run.
    for (int kt0 = 0; kt0 < KT; kt0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int kt = kt0 + p;
            if (kt < KT) {
                const int set = (p + 1) % PF;  // == (kt + 1) % PF because kt0 % PF == 0
                if (kt + 1 < KT) {
                    store_stage((kt + 1) & 1, stage[set], okbits[set]);
                    load_stage(kt + 1 + PF, stage[set], okbits[set]);
                }
                compute(kt & 1);
                __syncthreads();
            }
        }
    }

    // ---- reduce the k-groups through LDS (deterministic order) ----
    if (WK > 1) {
        float *red = smem;
        if (kg > 0) {
#pragma unroll
            for (int an = 0; an < AN; ++an)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    red[(((kg - 1) * WMN + wq) * AN * 16 + an * 16 + r) * 64 + lane] = acc[an][r];
        }
        __syncthreads();
        if (kg == 0) {
            for (int gk = 1; gk < WK; ++gk)
#pragma unroll
                for (int an = 0; an < AN; ++an)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[an][r] += red[(((gk - 1) * WMN + wq) * AN * 16 + an * 16 + r) * 64 + lane];
        }
    }
    if (kg != 0) return;

    // ---- epilogue: accumulator (lane, reg r) <-> row (r&3) + 8*(r>>2) + 4*(lane>>5), column lane&31 ----
    const int rbase = m0 + wm * 32 + 4 * (lane >> 5);
    if (EPI == EPI_BIAS) {
#pragma unroll
        for (int an = 0; an < AN; ++an) {
            const int col = n0 + (wn * AN + an) * 32 + (lane & 31);
            if (col >= g.N) continue;
            float b = g.bias1 ? g.bias1[col] : 0.0f;
            if (g.bias2) b += g.bias2[col];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < g.M) {
                    float v = acc[an][r] + b;
                    if (g.relu) v = v > 0.0f ? v : 0.0f;
                    g.C[(size_t)row * g.ldc + col] = v;
                }
            }
        }
    } else {
        const int H = g.H;
        const int unit = tn * 32 + (lane & 31);
        if (unit < H) {
            float bi = g.bias1[unit] + g.bias2[unit];
            float bf = g.bias1[H + unit] + g.bias2[H + unit];
            float bg = g.bias1[2 * H + unit] + g.bias2[2 * H + unit];
            float bo = g.bias1[3 * H + unit] + g.bias2[3 * H + unit];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < g.M) {
                    const size_t o = (size_t)row * H + unit;
                    if (g.mask[row]) {
                        const float ig = sigmoidf_acc(acc[0][r] + bi);
                        const float fg = sigmoidf_acc(acc[1][r] + bf);
                        const float gg = tanhf(acc[2][r] + bg);
                        const float og = sigmoidf_acc(acc[3][r] + bo);
                        const float cn = fg * g.c_in[o] + ig * gg;
                        g.c_out[o] = cn;
                        g.h_out[o] = og * tanhf(cn);
                    } else {  // absent track: state frozen (reference lstm/lstm.py:118-124,158-166)
                        g.c_out[o] = g.c_in[o];
                        g.h_out[o] = g.h_in[o];
                    }
                }
            }
        }
    }
}

template <int WM, int WN, int WK, int AN, int BK, int PF, int EPI, bool VEC>
static int launch_cfg_v(GemmArgs g, hipStream_t s) {
    constexpr int BM = 32 * WM, BN = 32 * WN * AN, NT = 64 * WM * WN * WK;
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = (EPI == EPI_LSTM) ? (g.H + 31) / 32 : (g.N + BN - 1) / BN;
    const size_t smem = (size_t)2 * WK * (BM + BN) * (BK + 4) * sizeof(float);
    auto kern = gemm_nt_kernel<WM, WN, WK, AN, BK, PF, EPI, VEC>;
    static bool attr_set = false;  // one flag per template instantiation
    if (!attr_set) {
        TNP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set = true;
    }
    const int blocks = g.tiles_m * g.tiles_n;
    if (blocks <= 0) return 0;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(NT), smem, s, g);
    TNP_HIP(hipGetLastError());
    return 0;
}

template <int WM, int WN, int WK, int AN, int BK, int PF, int EPI>
static int launch_cfg(const GemmArgs &g, hipStream_t s) {
    if (g.vec_ok) return launch_cfg_v<WM, WN, WK, AN, BK, PF, EPI, true>(g, s);
    // rare fallback (K or a leading dimension not a multiple of 4): scalar loader, shallow prefetch
    return launch_cfg_v<WM, WN, WK, AN, BK, 1, EPI, false>(g, s);
}

static int check_vec(GemmArgs &g) {
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    bool ok = (g.K1 % 4 == 0) && (g.lda1 % 4 == 0) && (g.ldb1 % 4 == 0) && al(g.A1) && al(g.B1);
    if (g.K2 > 0) ok = ok && (g.K2 % 4 == 0) && (g.lda2 % 4 == 0) && (g.ldb2 % 4 == 0) && al(g.A2) && al(g.B2);
    g.vec_ok = ok ? 1 : 0;
    return 0;
}

int launch_linear(const GemmArgs &g_in, int variant, hipStream_t s) {
    GemmArgs g = g_in;
    check_vec(g);
    if (variant == 0) {
        const long big_tiles = (long)((g.M + 127) / 128) * ((g.N + 63) / 64);
        variant = (big_tiles >= 192) ? 1 : 3;
    }
    switch (variant) {
        //                         WM WN WK AN BK PF
        case 1: return launch_cfg<4, 2, 1, 1, 32, 2, EPI_BIAS>(g, s);   // 128x64, 8 waves
        case 2: return launch_cfg<2, 4, 1, 1, 32, 2, EPI_BIAS>(g, s);   // 64x128, 8 waves
        case 3: return launch_cfg<1, 2, 2, 1, 32, 3, EPI_BIAS>(g, s);   // 32x64, split-K 2
        case 4: return launch_cfg<4, 2, 1, 1, 32, 1, EPI_BIAS>(g, s);   // 128x64, prefetch 1 (round-1 baseline)
        case 5: return launch_cfg<4, 2, 1, 1, 32, 3, EPI_BIAS>(g, s);   // 128x64, prefetch 3
        case 6: return launch_cfg<2, 4, 1, 1, 32, 3, EPI_BIAS>(g, s);   // 64x128, prefetch 3
        case 7: return launch_cfg<4, 2, 1, 1, 64, 2, EPI_BIAS>(g, s);   // 128x64, BK 64
        case 8: return launch_cfg<2, 4, 1, 1, 64, 2, EPI_BIAS>(g, s);   // 64x128, BK 64
        case 9: return launch_cfg<1, 2, 2, 1, 32, 1, EPI_BIAS>(g, s);   // 32x64 split-K 2, prefetch 1
        case 10: return launch_cfg<1, 2, 2, 1, 32, 2, EPI_BIAS>(g, s);  // 32x64 split-K 2, prefetch 2
        case 11: return launch_cfg<1, 2, 2, 1, 64, 2, EPI_BIAS>(g, s);  // 32x64 split-K 2, BK 64
        case 12: return launch_cfg<1, 1, 4, 1, 32, 3, EPI_BIAS>(g, s);  // 32x32 split-K 4
        case 13: return launch_cfg<2, 2, 2, 1, 32, 3, EPI_BIAS>(g, s);  // 64x64 split-K 2, 8 waves
        case 14: return launch_cfg<4, 1, 1, 2, 32, 2, EPI_BIAS>(g, s);  // 128x64, 4 waves x 2 blocks
        default: TNP_FAIL(-1, "tnp_linear_forward: unknown variant %d", variant);
    }
}

int launch_lstm_gates(const GemmArgs &g_in, int variant, hipStream_t s) {
    GemmArgs g = g_in;
    check_vec(g);
    if (g.H % 32 != 0) TNP_FAIL(-1, "LSTM hidden_dim must be a multiple of 32 (got %d)", g.H);
    if (variant == 0) variant = (g.M >= 4096) ? 1 : 2;
    switch (variant) {
        case 1: return launch_cfg<2, 1, 2, 4, 32, 2, EPI_LSTM>(g, s);  // 64 tracks x 32 units, split-K 2
        case 2: return launch_cfg<1, 1, 4, 4, 16, 3, EPI_LSTM>(g, s);  // 32 tracks x 32 units, split-K 4
        case 3: return launch_cfg<1, 1, 4, 4, 16, 1, EPI_LSTM>(g, s);  // same, prefetch 1 (round-1 baseline)
        case 4: return launch_cfg<1, 1, 4, 4, 16, 2, EPI_LSTM>(g, s);  // same, prefetch 2
        case 5: return launch_cfg<2, 1, 2, 4, 32, 1, EPI_LSTM>(g, s);  // 64 tracks, prefetch 1
        case 6: return launch_cfg<2, 1, 2, 4, 16, 3, EPI_LSTM>(g, s);  // 64 tracks, BK 16, prefetch 3
        default: TNP_FAIL(-1, "lstm gates: unknown variant %d", variant);
    }
}

}  // namespace tnp
