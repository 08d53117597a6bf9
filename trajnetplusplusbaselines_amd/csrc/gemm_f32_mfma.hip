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
#include <cstdlib>
#include "tnp_internal.h"
#include "lstm_cell.h"

#include <stdlib.h>
#include <utility>

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

// Split-K reduction fused with the LSTM epilogue, spread over ALL WK k-group waves: wave kg owns accumulator
// registers [kg*16/WK, (kg+1)*16/WK) of the 32x32 block (x 4 gates), receives the other waves' partials for those
// through LDS, and runs the transcendental epilogue for its rows only.  Partials are added in k-group order, so the
// result is bit-identical to the single-wave reduction it replaces.  LDS: (WK-1)*WMN*4*16*64 floats.
struct LstmPrefetch {           // epilogue operands of one wave, fetched before the main loop (gemm_nt_pipe)
    float bias[4];              // bias_ih + bias_hh of the lane's unit, per gate
    float c[16];                // c_in of the wave's (row, unit) entries, r = kg * RN + rr
    int present[16];
};

template <int WK, int WMN>
__device__ __forceinline__ void lstm_reduce_epilogue(const GemmArgs &g, f32x16 (&acc)[4], float *red, int kg, int wq,
                                                     int lane, int rbase, int tn, const LstmPrefetch *pf = nullptr) {
    constexpr int RN = 16 / WK;
#pragma unroll
    for (int o = 0; o < WK; ++o) {
        if (o == kg) continue;
        const int slot = kg - (kg > o ? 1 : 0);
#pragma unroll
        for (int an = 0; an < 4; ++an)
#pragma unroll
            for (int rr = 0; rr < RN; ++rr)
                red[((((o * (WK - 1) + slot) * WMN + wq) * 4 + an) * RN + rr) * 64 + lane] = acc[an][o * RN + rr];
    }
    __syncthreads();
    const int H = g.H;
    const int unit = tn * 32 + (lane & 31);
    if (unit >= H) return;
    float bias[4];
#pragma unroll
    for (int an = 0; an < 4; ++an) bias[an] = pf ? pf->bias[an] : g.bias1[an * H + unit] + g.bias2[an * H + unit];
#pragma unroll
    for (int o = 0; o < WK; ++o) {
        if (o != kg) continue;
#pragma unroll
        for (int rr = 0; rr < RN; ++rr) {
            float pre[4];
#pragma unroll
            for (int an = 0; an < 4; ++an) {
                float s = 0.0f;
#pragma unroll
                for (int k = 0; k < WK; ++k) {
                    const float v = (k == o) ? acc[an][o * RN + rr]
                                             : red[((((o * (WK - 1) + (k - (k > o ? 1 : 0))) * WMN + wq) * 4 + an) * RN + rr) * 64 + lane];
                    s = (k == 0) ? v : s + v;
                }
                pre[an] = s + bias[an];
            }
            const int r = o * RN + rr;
            if (pf) lstm_cell_store_pf(g, rbase + (r & 3) + 8 * (r >> 2), unit, pre[0], pre[1], pre[2], pre[3], pf->c[rr], pf->present[rr]);
            else lstm_cell_store(g, rbase + (r & 3) + 8 * (r >> 2), unit, pre[0], pre[1], pre[2], pre[3]);
        }
    }
}

template <int AN, int EPI>
__device__ __forceinline__ void epilogue(const GemmArgs &g, f32x16 (&acc)[AN], int rbase, int n0, int wn, int tn,
                                         int lane, const float *pbias = nullptr) {
    // ---- epilogue: accumulator (lane, reg r) <-> row (r&3) + 8*(r>>2) + 4*(lane>>5), column lane&31 ----
    if (EPI == EPI_BIAS) {
#pragma unroll
        for (int an = 0; an < AN; ++an) {
            const int col = n0 + (wn * AN + an) * 32 + (lane & 31);
            if (col >= g.N) continue;
            float b;
            if (pbias) b = pbias[an];
            else { b = g.bias1 ? g.bias1[col] : 0.0f; if (g.bias2) b += g.bias2[col]; }
            // masked copies (GemmArgs::seg): a column belongs to at most one segment; pick it per lane and request its
            // sixteen activation values BEFORE the stores of the value itself, so that their round trip runs under them
            const float *sact = nullptr; float *sout = nullptr;
            int sc = 0, lda_s = 0, ldo_s = 0;
            for (int si = 0; si < g.nseg; ++si) {
                const GemmArgs::EpiSeg &sg = g.seg[si];
                const int c = col - sg.col0;
                if (c >= 0 && c < sg.n) { sact = sg.act; sout = sg.out; sc = c; lda_s = sg.ld_act; ldo_s = sg.ld_out; }
            }
            float ak[16];
            if (g.nseg > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(rbase + (r & 3) + 8 * (r >> 2), g.M - 1);
                    ak[r] = sout ? sact[(size_t)row * lda_s + sc] : 0.0f;
                }
            }
            float mk[16];
            if (g.mask_act) {                                   // all sixteen mask operands in flight before the first use
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = min(rbase + (r & 3) + 8 * (r >> 2), g.M - 1);
                    mk[r] = g.mask_act[(size_t)row * g.ld_mask + col];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rbase + (r & 3) + 8 * (r >> 2);
                if (row < g.M) {
                    float v = acc[an][r] + b;
                    if (g.relu) v = v > 0.0f ? v : 0.0f;
                    if (g.mask_act) v = mk[r] > 0.0f ? v : 0.0f;
                    g.C[(size_t)row * g.ldc + col] = v;
                }
            }
            if (g.nseg > 0 && sout) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rbase + (r & 3) + 8 * (r >> 2);
                    if (row < g.M) sout[(size_t)row * ldo_s + sc] = ak[r] > 0.0f ? acc[an][r] + b : 0.0f;
                }
            }
        }
    } else if constexpr (AN == 4) {   // (the gate-split tiles, AN == 1, swap their gate blocks through LDS and never come here)
        const int H = g.H;
        const int unit = tn * 32 + (lane & 31);
        if (unit < H) {
            float bi = g.bias1[unit] + g.bias2[unit];
            float bf = g.bias1[H + unit] + g.bias2[H + unit];
            float bg = g.bias1[2 * H + unit] + g.bias2[2 * H + unit];
            float bo = g.bias1[3 * H + unit] + g.bias2[3 * H + unit];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                lstm_cell_store(g, rbase + (r & 3) + 8 * (r >> 2), unit, acc[0][r] + bi, acc[1][r] + bf,
                                acc[2][r] + bg, acc[3][r] + bo);
            }
        }
    }
}

template <int WM, int WN, int WK, int AN, int BK, int EPI, bool VEC>
__global__ void __launch_bounds__(64 * WM * WN * WK) gemm_nt_general(const GemmArgs g) {
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
    f32x4 stage[CH];
    unsigned long long okbits = 0ull;

    auto load_stage = [&](int kt) {
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
            const bool row_ok = is_a ? (m < gM) : ok;
            const size_t r = is_a ? (size_t)m : (size_t)wr;
            const float *r1 = (is_a ? gA1 : gB1) + r * (size_t)(is_a ? lda1 : ldb1);
            const float *r2 = (is_a ? gA2 : gB2) + r * (size_t)(is_a ? lda2 : ldb2);
            if (c < 8) stage[c] = load4_dual<VEC>(r1, K1, r2, K2, k, row_ok, gA1, lo, (c & 7) * 4);
            else stage[c] = load4_dual<VEC>(r1, K1, r2, K2, k, row_ok, gA1, hi, (c & 7) * 4);
        }
        okbits = ((unsigned long long)hi << 32) | lo;
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const int idx = tid + c * NT;
            const int grp = idx / (ROWS * C4);
            const int rem = idx - grp * (ROWS * C4);
            const int row = rem / C4;
            const int c4 = rem - row * C4;
            float *dst = smem + (size_t)(buf * WK + grp) * GROUP_FLOATS + row * LDS_STRIDE + c4 * 4;
            f32x4 v = stage[c];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = ((okbits >> (c * 4 + q)) & 1ull) ? v[q] : 0.0f;
            *reinterpret_cast<f32x4 *>(dst) = v;
        }
    };

    f32x16 acc[AN];
#pragma unroll
    for (int an = 0; an < AN; ++an)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[an][r] = 0.0f;

    load_stage(0);
    store_stage(0);
    __syncthreads();

    const int a_off = (wm * 32 + (lane & 31)) * LDS_STRIDE + (lane >> 5) * 4;
    const int b_off = BM * LDS_STRIDE + (wn * AN * 32 + (lane & 31)) * LDS_STRIDE + (lane >> 5) * 4;

    // Fragment reads are software-pipelined one k8 step ahead: the ds_read_b128 of step k8+1 are issued before the
    // four MFMAs of step k8, so LDS latency hides under the matrix pipe instead of idling both waves of a SIMD.
    auto read_frags = [&](const float *base, int k8, f32x4 &a4, f32x4 (&b4)[AN]) {
        a4 = *reinterpret_cast<const f32x4 *>(base + a_off + k8 * 8);
#pragma unroll
        for (int an = 0; an < AN; ++an)
            b4[an] = *reinterpret_cast<const f32x4 *>(base + b_off + an * 32 * LDS_STRIDE + k8 * 8);
    };
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) load_stage(kt + 1);
        const float *base = smem + (size_t)((kt & 1) * WK + kg) * GROUP_FLOATS;
        f32x4 a4[2];
        f32x4 b4[2][AN];
        read_frags(base, 0, a4[0], b4[0]);
#pragma unroll
        for (int k8 = 0; k8 < BK / 8; ++k8) {
            if (k8 + 1 < BK / 8) read_frags(base, k8 + 1, a4[(k8 + 1) & 1], b4[(k8 + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);  // keep the next step's ds_reads ahead of this step's MFMAs
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int an = 0; an < AN; ++an)
                    acc[an] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[k8 & 1][q], b4[k8 & 1][an][q], acc[an], 0, 0, 0);
        }
        if (kt + 1 < KT) store_stage((kt + 1) & 1);
        __syncthreads();
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
    epilogue<AN, EPI>(g, acc, m0 + wm * 32 + 4 * (lane >> 5), n0, wn, tn, lane);
}

// ---------------------------------------------------------------------------------------------------------
// Fast path: K1, K2 multiples of BK, K multiple of WK*BK, 16-byte aligned rows.  Per K step a lane issues CH
// global_load_dwordx4 through running pointers (one 64-bit add each, no masks: out-of-range rows are clamped to
// the last valid row and their results are never stored), so the non-MFMA part of a K step is a few dozen
// instructions instead of a few hundred -- the barrier re-aligns all waves every K step, which serialises
// whatever VALU work precedes the MFMA chain with the matrix pipe.
// ---------------------------------------------------------------------------------------------------------
template <int WM, int WN, int WK, int AN, int BK, int EPI, bool DUAL>
__global__ void __launch_bounds__(64 * WM * WN * WK) gemm_nt_fast(const GemmArgs g) {
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
    const int n0 = tn * BN;

    const int K1 = g.K1;
    const int Kg = (g.K1 + g.K2) / WK;  // multiple of BK
    const int KT = Kg / BK;
    const float *cur[CH];   // running source pointer (source 1)
    const float *alt[CH];   // DUAL: source-2 pointer such that alt + k addresses column k - K1 of source 2
    int kbeg[CH];           // DUAL: first k of this chunk
    int ldsoff[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int idx = tid + c * NT;
        const int grp = idx / (ROWS * C4);
        const int rem = idx - grp * (ROWS * C4);
        const int row = rem / C4;
        const int c4 = rem - row * C4;
        const bool is_a = row < BM;
        int wr;
        if (EPI == EPI_LSTM) {
            const int cc = row - BM;
            wr = (cc >> 5) * g.H + tn * 32 + (cc & 31);
        } else {
            wr = min(n0 + (row - BM), g.N - 1);
        }
        const size_t r = is_a ? (size_t)min(m0 + row, g.M - 1) : (size_t)wr;
        const int k0 = grp * Kg + c4 * 4;
        cur[c] = (is_a ? g.A1 : g.B1) + r * (size_t)(is_a ? g.lda1 : g.ldb1) + k0;
        if (DUAL) {
            alt[c] = (is_a ? g.A2 : g.B2) + r * (size_t)(is_a ? g.lda2 : g.ldb2) + (k0 - K1);
            kbeg[c] = k0;
        }
        ldsoff[c] = grp * GROUP_FLOATS + row * LDS_STRIDE + c4 * 4;
    }

    f32x4 stage[CH];
    auto load_stage = [&](int kt) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const float *p = cur[c] + kt * BK;
            if (DUAL) p = (kbeg[c] + kt * BK < K1) ? p : (alt[c] + kt * BK);
            stage[c] = *reinterpret_cast<const f32x4 *>(p);
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int c = 0; c < CH; ++c)
            *reinterpret_cast<f32x4 *>(smem + (size_t)buf * WK * GROUP_FLOATS + ldsoff[c]) = stage[c];
    };

    f32x16 acc[AN];
#pragma unroll
    for (int an = 0; an < AN; ++an)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[an][r] = 0.0f;

    load_stage(0);
    store_stage(0);
    __syncthreads();

    const int a_off = (wm * 32 + (lane & 31)) * LDS_STRIDE + (lane >> 5) * 4;
    const int b_off = BM * LDS_STRIDE + (wn * AN * 32 + (lane & 31)) * LDS_STRIDE + (lane >> 5) * 4;

    // Fragment reads are software-pipelined one k8 step ahead: the ds_read_b128 of step k8+1 are issued before the
    // four MFMAs of step k8, so LDS latency hides under the matrix pipe instead of idling both waves of a SIMD.
    auto read_frags = [&](const float *base, int k8, f32x4 &a4, f32x4 (&b4)[AN]) {
        a4 = *reinterpret_cast<const f32x4 *>(base + a_off + k8 * 8);
#pragma unroll
        for (int an = 0; an < AN; ++an)
            b4[an] = *reinterpret_cast<const f32x4 *>(base + b_off + an * 32 * LDS_STRIDE + k8 * 8);
    };
    for (int kt = 0; kt < KT; ++kt) {
        if (kt + 1 < KT) load_stage(kt + 1);
        const float *base = smem + (size_t)((kt & 1) * WK + kg) * GROUP_FLOATS;
        f32x4 a4[2];
        f32x4 b4[2][AN];
        read_frags(base, 0, a4[0], b4[0]);
#pragma unroll
        for (int k8 = 0; k8 < BK / 8; ++k8) {
            if (k8 + 1 < BK / 8) read_frags(base, k8 + 1, a4[(k8 + 1) & 1], b4[(k8 + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);  // keep the next step's ds_reads ahead of this step's MFMAs
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int an = 0; an < AN; ++an)
                    acc[an] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[k8 & 1][q], b4[k8 & 1][an][q], acc[an], 0, 0, 0);
        }
        if (kt + 1 < KT) store_stage((kt + 1) & 1);
        __syncthreads();
    }

    if constexpr (WK > 1 && EPI == EPI_LSTM) {
        lstm_reduce_epilogue<WK, WMN>(g, acc, smem, kg, wq, lane, m0 + wm * 32 + 4 * (lane >> 5), tn);
        return;
    } else if (WK > 1) {
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
    epilogue<AN, EPI>(g, acc, m0 + wm * 32 + 4 * (lane >> 5), n0, wn, tn, lane);
}

// ---------------------------------------------------------------------------------------------------------
// Pipelined fast path ("mid-stream barrier").  PMC on the kernel above shows the matrix pipe busy only ~67 %:
// the end-of-step barrier aligns all waves, so their non-MFMA work (vmcnt wait, ds_write, barrier, first ds_read)
// coincides and the pipe drains once per K step.  Here the LDS ring is three tiles deep and the single barrier of
// a K step sits in the MIDDLE of the step's MFMA stream:
//     first half : MFMAs of tile t   | registers -> LDS of tile t+1 (its loads were issued half a step earlier)
//     barrier    : tile t+1 visible; everybody is past tile t-1, whose buffer tile t+2 will reuse
//     second half: MFMAs of tile t   | global loads of tile t+2 -> registers | first fragments of tile t+1
// so there is no barrier at the step boundary, fragments are always one k8 step ahead of the MFMAs that use
// them, and a wave that waits at the barrier still has its last MFMA in the pipe while its SIMD partner issues.
// SCHED = 1 (what launch_pipe instantiates): the global loads of tile t+2 move into the first half as well, and
// sched_group_barrier interleaves one MFMA with at most one LDS write / LDS read / global load, so the memory
// instructions issue in the shadow of the matrix pipe instead of in a clump in front of it.  The epilogue's operands
// (bias; previous cell state and presence mask of the LSTM epilogue) are fetched before the main loop: all workgroups
// run in lockstep, so a load issued after it exposes its whole latency.  What is left (gemm_probe, second embedding layer
// 14.6 us): matrix pipe 7.6 us, launch / kernarg / first tile ~2.7 us, the rest in waits on the tile loads and the
// barrier; a direct-to-LDS variant (global_load_lds_dwordx4, swizzled unpadded ring) is exactly as fast, so not kept.
// ---------------------------------------------------------------------------------------------------------
// Template switches beyond the tile shape are for tools/experiments/gemm_probe.hip: DBG shader-clock stamps per wave (buffer
// passed as g.gates_out, g.C for the LSTM epilogue), ACC2 two accumulator chains per tile, GABL timing ablations.
template <int WM, int WN, int WK, int AN, int BK, int EPI, bool DUAL, int DBG = 0, bool ACC2 = false, int SCHED = 0, int GABL = 0>
__device__ __forceinline__ void gemm_nt_pipe_body(const GemmArgs &g, const int bid) {
#define GP_T(k) do { if constexpr (DBG) { if ((threadIdx.x & 63) == 0) reinterpret_cast<long long *>(EPI == EPI_LSTM ? (float *)g.C : g.gates_out)[(blockIdx.x * (WM * WN * WK) + (threadIdx.x >> 6)) * 8 + (k)] = (long long)__builtin_readcyclecounter(); } } while (0)
    GP_T(0);
    constexpr int BM = 32 * WM;
    constexpr int BN = 32 * WN * AN;
    constexpr int NT = 64 * WM * WN * WK;
    constexpr int WMN = WM * WN;
    constexpr int LDS_STRIDE = BK + 4;
    constexpr int ROWS = BM + BN;
    constexpr int GROUP_FLOATS = ROWS * LDS_STRIDE;
    constexpr int BUF_FLOATS = WK * GROUP_FLOATS;
    constexpr int C4 = BK / 4;
    constexpr int CHUNKS = WK * ROWS * C4;
    static_assert(CHUNKS % NT == 0, "loader must tile evenly");
    constexpr int CH = CHUNKS / NT;
    constexpr int NK8 = BK / 8;
    static_assert(NK8 >= 2 && NK8 % 2 == 0, "BK must be a multiple of 16");
    constexpr int WRITE_AT = NK8 / 2 - 1;
    // LSTM epilogue: either a wave owns all four gate blocks of its 32 units (WN == 1, AN == 4; K split over the waves), or
    // "gate split": four waves own one gate block each over the whole K (WN == 4, AN == 1, WK == 1) and swap tiles through LDS
    // Round 5: the gate split also comes with the K range over two wave quartets (WK == 2: eight waves, two per SIMD -- a
    // wave's LDS / global waits are covered by its SIMD partner's MFMAs); the quartets' partial tiles meet in the LDS swap
    constexpr bool GSPLIT = EPI == EPI_LSTM && WM == 1 && WN == 4 && AN == 1 && (WK == 1 || WK == 2);
    static_assert(EPI != EPI_LSTM || (WN == 1 && AN == 4) || GSPLIT, "LSTM epilogue: 4 gate blocks per wave or gate split");

    extern __shared__ __attribute__((aligned(16))) float smem[];  // [3][WK][GROUP_FLOATS]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int kg = wave / WMN;
    const int wq = wave - kg * WMN;
    const int wm = wq / WN;
    const int wn = wq - wm * WN;

    int tm, tn;
    tile_coords(bid, g.tiles_m, g.tiles_n, tm, tn);
    const int m0 = tm * BM;
    const int n0 = tn * BN;

    const int K1 = g.K1;
    const int Kg = (g.K1 + g.K2) / WK;  // multiple of BK
    const int KT = Kg / BK;

    const float *cur[CH];
    const float *alt[CH];
    int kbeg[CH];
    int ldsoff[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int idx = tid + c * NT;
        const int grp = idx / (ROWS * C4);
        const int rem = idx - grp * (ROWS * C4);
        const int row = rem / C4;
        const int c4 = rem - row * C4;
        const bool is_a = row < BM;
        int wr;
        if (EPI == EPI_LSTM) {
            const int cc = row - BM;
            wr = (cc >> 5) * g.H + tn * 32 + (cc & 31);
        } else {
            wr = min(n0 + (row - BM), g.N - 1);
        }
        const size_t r = is_a ? (size_t)min(m0 + row, g.M - 1) : (size_t)wr;
        const int k0 = grp * Kg + c4 * 4;
        cur[c] = (is_a ? g.A1 : g.B1) + r * (size_t)(is_a ? g.lda1 : g.ldb1) + k0;
        if (DUAL) {
            alt[c] = (is_a ? g.A2 : g.B2) + r * (size_t)(is_a ? g.lda2 : g.ldb2) + (k0 - K1);
            kbeg[c] = k0;
        }
        ldsoff[c] = grp * GROUP_FLOATS + row * LDS_STRIDE + c4 * 4;
    }

    f32x4 stage[CH];
    auto load_stage = [&](int kt) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const float *p = cur[c] + kt * BK;
            if (DUAL) p = (kbeg[c] + kt * BK < K1) ? p : (alt[c] + kt * BK);
            stage[c] = *reinterpret_cast<const f32x4 *>(p);
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int c = 0; c < CH; ++c)
            *reinterpret_cast<f32x4 *>(smem + (size_t)buf * BUF_FLOATS + ldsoff[c]) = stage[c];
    };

    f32x16 acc[AN];
    f32x16 accb[ACC2 ? AN : 1];     // ACC2 (probe): odd k-pairs accumulate separately -- two independent MFMA chains per tile
#pragma unroll
    for (int an = 0; an < AN; ++an)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[an][r] = 0.0f; if (ACC2) accb[an][r] = 0.0f; }

    const int a_off = kg * GROUP_FLOATS + (wm * 32 + (lane & 31)) * LDS_STRIDE + (lane >> 5) * 4;
    const int b_off = kg * GROUP_FLOATS + BM * LDS_STRIDE + (wn * AN * 32 + (lane & 31)) * LDS_STRIDE + (lane >> 5) * 4;
    auto read_frags = [&](int buf, int k8, f32x4 &a4, f32x4 (&b4)[AN]) {
        const float *base = smem + (size_t)buf * BUF_FLOATS;
        a4 = *reinterpret_cast<const f32x4 *>(base + a_off + k8 * 8);
#pragma unroll
        for (int an = 0; an < AN; ++an)
            b4[an] = *reinterpret_cast<const f32x4 *>(base + b_off + an * 32 * LDS_STRIDE + k8 * 8);
    };

    // optional static priority for the second-dispatched half of the waves (they share SIMDs with the first
    // half): breaks the phase lock in which both waves of a SIMD do their non-MFMA work at the same time
    if (g.prio && __builtin_amdgcn_readfirstlane(wave) >= (WM * WN * WK) / 2) __builtin_amdgcn_s_setprio(1);

    // prologue: tile 0 in LDS buffer 0, tile 1 in registers, first fragments of tile 0 in flight
    load_stage(0);
    // epilogue operands (bias, previous cell state, presence mask) are fetched now: every workgroup runs in lockstep, so a
    // load issued after the main loop would expose its full latency on the whole chip
    LstmPrefetch lpf;
    float ebias[AN];
    if constexpr ((EPI == EPI_LSTM && WK > 1) || GSPLIT) {
        constexpr int RN = GSPLIT ? 4 / WK : 16 / WK;
        const int unit = tn * 32 + (lane & 31), uc = unit < g.H ? unit : g.H - 1;
#pragma unroll
        for (int an = 0; an < 4; ++an) lpf.bias[an] = g.bias1[an * g.H + uc] + g.bias2[an * g.H + uc];
#pragma unroll
        for (int rr = 0; rr < RN; ++rr) {
            const int r = (GSPLIT ? kg * 4 + wn : kg) * RN + rr;
            const int row = min(m0 + wm * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2), g.M - 1);
            lpf.c[rr] = g.c_in[(size_t)row * g.H + uc];
            lpf.present[rr] = g.mask[row];
        }
    } else if constexpr (EPI == EPI_BIAS) {
#pragma unroll
        for (int an = 0; an < AN; ++an) {
            const int col = min(n0 + (wn * AN + an) * 32 + (lane & 31), g.N - 1);
            ebias[an] = (g.bias1 ? g.bias1[col] : 0.0f) + (g.bias2 ? g.bias2[col] : 0.0f);
        }
    }
    GP_T(1);
    store_stage(0);
    if (KT > 1) load_stage(1);
    __syncthreads();
    GP_T(2);
    f32x4 a4[2];
    f32x4 b4[2][AN];
    read_frags(0, 0, a4[0], b4[0]);

    int buf_cur = 0;
    for (int kt = 0; kt < KT; ++kt) {
        const int buf_nxt = (buf_cur == 2) ? 0 : buf_cur + 1;
        const bool has_next = kt + 1 < KT;
#pragma unroll
        for (int k8 = 0; k8 < NK8; ++k8) {
            // fragments one k8 step ahead (the first ones of the next tile once its buffer is published)
            if constexpr (!(GABL & 8)) {
            if (k8 + 1 < NK8) read_frags(buf_cur, k8 + 1, a4[(k8 + 1) & 1], b4[(k8 + 1) & 1]);
            else if (has_next) read_frags(buf_nxt, 0, a4[(k8 + 1) & 1], b4[(k8 + 1) & 1]);
            }
            if constexpr (!(GABL & 2)) if (k8 == WRITE_AT && has_next) store_stage(buf_nxt);
            if constexpr (SCHED) {
                // the global loads of tile t+2 are issued in the same k8 step as the LDS writes of tile t+1 (the stage
                // registers are free as soon as those writes have issued) and everything is interleaved with the MFMAs
                if constexpr (!(GABL & 1)) if (k8 == WRITE_AT && kt + 2 < KT) load_stage(kt + 2);
            } else {
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int an = 0; an < AN; ++an) {
                    if (ACC2 && (q & 1)) accb[an] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[k8 & 1][q], b4[k8 & 1][an][q], accb[an], 0, 0, 0);
                    else acc[an] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[k8 & 1][q], b4[k8 & 1][an][q], acc[an], 0, 0, 0);
                }
            if constexpr (SCHED) {
                // one MFMA, then at most one LDS write, one LDS read and one global load, repeated
#pragma unroll
                for (int i = 0; i < 4 * AN; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, (CH + 4 * AN - 1) / (4 * AN), 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, (1 + AN + 4 * AN - 1) / (4 * AN), 0);
                    __builtin_amdgcn_sched_group_barrier(0x020, (CH + 4 * AN - 1) / (4 * AN), 0);
                }
                if (k8 == WRITE_AT && !(GABL & 4)) {
                    __builtin_amdgcn_sched_barrier(0);
                    __syncthreads();
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else if (k8 == WRITE_AT) {
                __builtin_amdgcn_sched_barrier(0);
                __syncthreads();
                if (kt + 2 < KT) load_stage(kt + 2);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        buf_cur = buf_nxt;
    }
    if constexpr (ACC2) {
#pragma unroll
        for (int an = 0; an < AN; ++an)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[an][r] += accb[an][r];
    }
    GP_T(3);

    if constexpr (GSPLIT) {
        // wave (kg, wn) holds the pre-activations of gate `wn` for the tile's 32 tracks x 32 units over K group kg; the tiles
        // are swapped through LDS and wave w = 4 kg + wn runs the cell update of accumulator registers RN w .. RN w + RN - 1
        // (8 tracks x 32 units with one K group, 4 x 32 with two; the K groups' partials are added group 0 + group 1)
        constexpr int RN = 4 / WK;
        __syncthreads();  // everybody is done with the tile ring before it is reused
        float *red = smem;
#pragma unroll
        for (int r = 0; r < 16; ++r) red[((kg * 4 + wn) * 16 + r) * 64 + lane] = acc[0][r];
        __syncthreads();
        const int unit = tn * 32 + (lane & 31);
        if (unit < g.H) {
            const int rbase = m0 + 4 * (lane >> 5);
#pragma unroll
            for (int rr = 0; rr < RN; ++rr) {
                const int r = (kg * 4 + wn) * RN + rr;
                float pre[4];
#pragma unroll
                for (int gt = 0; gt < 4; ++gt) {
                    float v = red[(gt * 16 + r) * 64 + lane];
                    if constexpr (WK == 2) v += red[((4 + gt) * 16 + r) * 64 + lane];
                    pre[gt] = v + lpf.bias[gt];
                }
                lstm_cell_store_pf(g, rbase + (r & 3) + 8 * (r >> 2), unit, pre[0], pre[1], pre[2], pre[3], lpf.c[rr], lpf.present[rr]);
            }
        }
        GP_T(5);
        return;
    } else if constexpr (WK > 1 && EPI == EPI_LSTM) {
        __syncthreads();  // everybody is done with the tile ring before it is reused for the reduction
        lstm_reduce_epilogue<WK, WMN>(g, acc, smem, kg, wq, lane, m0 + wm * 32 + 4 * (lane >> 5), tn, &lpf);
        GP_T(5);
        return;
    } else if (WK > 1) {
        __syncthreads();  // everybody is done with the tile ring before it is reused for the reduction
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
    GP_T(4);
    if (kg != 0) return;
    epilogue<AN, EPI>(g, acc, m0 + wm * 32 + 4 * (lane >> 5), n0, wn, tn, lane, EPI == EPI_BIAS ? ebias : nullptr);
    GP_T(5);
#undef GP_T
}

template <int WM, int WN, int WK, int AN, int BK, int EPI, bool DUAL, int DBG = 0, bool ACC2 = false, int SCHED = 0, int GABL = 0>
__global__ void __launch_bounds__(64 * WM * WN * WK) gemm_nt_pipe(const GemmArgs g) {
    gemm_nt_pipe_body<WM, WN, WK, AN, BK, EPI, DUAL, DBG, ACC2, SCHED, GABL>(g, blockIdx.x);
}

template <typename KernT>
static int launch_kernel(KernT kern, GemmArgs &g, int BM, int BN, int NT, size_t smem, bool lstm, hipStream_t s,
                         bool &attr_set) {
    g.tiles_m = (g.M + BM - 1) / BM;
    g.tiles_n = lstm ? (g.H + 31) / 32 : (g.N + BN - 1) / BN;
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

// general (masked) kernel: any M, N, K
template <int WM, int WN, int WK, int AN, int BK, int EPI>
static int launch_general(GemmArgs g, hipStream_t s) {
    constexpr int BM = 32 * WM, BN = 32 * WN * AN, NT = 64 * WM * WN * WK;
    const size_t smem = (size_t)2 * WK * (BM + BN) * (BK + 4) * sizeof(float);
    static bool attr_v = false, attr_s = false;
    if (g.vec_ok) return launch_kernel(gemm_nt_general<WM, WN, WK, AN, BK, EPI, true>, g, BM, BN, NT, smem,
                                       EPI == EPI_LSTM, s, attr_v);
    return launch_kernel(gemm_nt_general<WM, WN, WK, AN, BK, EPI, false>, g, BM, BN, NT, smem, EPI == EPI_LSTM, s,
                         attr_s);
}

static bool fast_ok(const GemmArgs &g, int WK, int BK) {
    if (!g.vec_ok) return false;
    if (g.K1 <= 0 || g.K1 % BK != 0 || g.K2 % BK != 0) return false;
    return ((g.K1 + g.K2) % (WK * BK)) == 0;
}

// fast kernel when the shape allows it, otherwise the general kernel of a fallback tile configuration
template <int WM, int WN, int WK, int AN, int BK, int EPI>
static int launch_fast(GemmArgs g, hipStream_t s) {
    constexpr int BM = 32 * WM, BN = 32 * WN * AN, NT = 64 * WM * WN * WK;
    const size_t smem = (size_t)2 * WK * (BM + BN) * (BK + 4) * sizeof(float);
    static bool attr_1 = false, attr_2 = false;
    if (g.K2 > 0) return launch_kernel(gemm_nt_fast<WM, WN, WK, AN, BK, EPI, true>, g, BM, BN, NT, smem,
                                       EPI == EPI_LSTM, s, attr_2);
    return launch_kernel(gemm_nt_fast<WM, WN, WK, AN, BK, EPI, false>, g, BM, BN, NT, smem, EPI == EPI_LSTM, s,
                         attr_1);
}

static int check_vec(GemmArgs &g) {
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    bool ok = (g.K1 % 4 == 0) && (g.lda1 % 4 == 0) && (g.ldb1 % 4 == 0) && al(g.A1) && al(g.B1);
    if (g.K2 > 0) ok = ok && (g.K2 % 4 == 0) && (g.lda2 % 4 == 0) && (g.ldb2 % 4 == 0) && al(g.A2) && al(g.B2);
    g.vec_ok = ok ? 1 : 0;
    return 0;
}

template <int WM, int WN, int WK, int AN, int BK, int EPI>
static int launch_pipe(GemmArgs g, hipStream_t s) {
    constexpr int BM = 32 * WM, BN = 32 * WN * AN, NT = 64 * WM * WN * WK;
    constexpr size_t smem = (size_t)3 * WK * (BM + BN) * (BK + 4) * sizeof(float);
    static_assert(smem <= 163840, "LDS ring does not fit");
    static bool attr_1 = false, attr_2 = false;
    // interleaved schedule (SCHED = 1) + static priority for the second half of the waves: 16.1 -> 14.6 us on the second
    // embedding layer, 18.2 -> 17.8 us on the gates (tools/experiments/gemm_probe.hip); results are bit-identical
    g.prio = 1;
    if (g.K2 > 0) return launch_kernel(gemm_nt_pipe<WM, WN, WK, AN, BK, EPI, true, 0, false, 1>, g, BM, BN, NT, smem,
                                       EPI == EPI_LSTM, s, attr_2);
    return launch_kernel(gemm_nt_pipe<WM, WN, WK, AN, BK, EPI, false, 0, false, 1>, g, BM, BN, NT, smem,
                         EPI == EPI_LSTM, s, attr_1);
}

#define TNP_TRY_FAST(WM, WN, WK, AN, BK, EPI) \
    if (fast_ok(g, WK, BK)) return launch_fast<WM, WN, WK, AN, BK, EPI>(g, s)
#define TNP_TRY_PIPE(WM, WN, WK, AN, BK, EPI) \
    if (fast_ok(g, WK, BK)) return launch_pipe<WM, WN, WK, AN, BK, EPI>(g, s)

static int compute_units() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cu > 0) n = cu;
        else n = 256;
    }
    return n;
}

// Rows up to which the 16-track kernels of gemm_skinny.hip take the step's dense layers (Tuning::skinny_max_rows, 160) and
// which of them.  Device time per launch (rocprofv3, tools/diag/gemm_device_times.py -> profiles/round6_gemm_device_times.txt):
// second embedding layer [M,1024]x[256,1024]^T at 36 tracks 6.8 us (v40) against 9.1 (v29), its data gradient 5.1 (v45)
// against 5.7, the gates' data gradient 5.1 against 6.8; from ~200 tracks on the 32 x 32 tiles with K over eight waves are
// ahead again (310 tracks: 8.8 against 10.3).  One column tile per workgroup (variants 40 / 45) is the best form throughout.
static int skinny_max_rows() { return tuning().skinny_max_rows; }
static int skinny_linear_variant(const GemmArgs &g) { return g.K1 + g.K2 >= 1024 ? 40 : 45; }

int launch_linear(const GemmArgs &g_in, int variant, hipStream_t s) {
    GemmArgs g = g_in;
    check_vec(g);
    const long big_tiles = (long)((g.M + 127) / 128) * ((g.N + 63) / 64);
    // tile selection measured on MI355X (profiles/archive/round1_b_variant_sweep.jsonl): 64x64 tiles when they fill the chip, else
    // the pipelined 32x64 split-K 2 kernel; `variant` pins one of the two (12 / 24), anything else is an error
    if (variant >= 40 && variant <= 45) return launch_skinny_linear(g, variant, s);
    if (variant == 0 && g.M <= skinny_max_rows() && skinny_ok(g)) {
        // SMALL batches (one scene, a batch_size-8 training batch): 16-track tiles with the operands straight in registers
        // (gemm_skinny.hip; tools/diag/small_step_probe.py, profiles/round6_small_step_probe.txt)
        return launch_skinny_linear(g, skinny_linear_variant(g), s);
    }
    if (variant == 0) {
        variant = (big_tiles >= 192) ? 12 : 24;
        // 64 x 64 tiles, at most four rounds of workgroups: the PIPELINED form (three-stage LDS ring, interleaved schedule) of the
        // same tile -- the layer-2 data gradient of training, [2048, 256] x [1024, 256]^T: 15.8 -> 13.2 us; at 16 k rows the plain
        // double-buffered kernel is ahead again (83.5 vs 88.2 us), tools/diag/small_gemm_probe.py
        if (variant == 12 && g.K2 <= 0 && fast_ok(g, 1, 32) &&
            (long)((g.M + 63) / 64) * ((g.N + 63) / 64) <= 4L * compute_units()) variant = 33;
        // 32x64 tiles that need a second round of workgroups while 32x128 tiles fit in one (the backward data-gradient GEMM,
        // N = 448: 448 against 256 workgroups on 256 CUs): 22.8 -> 18.0 us (round 3 sweep run_r3v, git history)
        const long b64 = (long)((g.M + 31) / 32) * ((g.N + 63) / 64), b128 = (long)((g.M + 31) / 32) * ((g.N + 127) / 128);
        if (variant == 24 && b64 > compute_units() && b128 <= compute_units() && fast_ok(g, 1, 32)) variant = 25;
        // SMALL batches with a long contraction (one scene, a batch_size-8 training batch: the second embedding layer,
        // K = 1024): a handful of workgroups on an empty chip wait for their own K loop, so 32 x 32 tiles with the K range
        // split over eight (four) waves of the workgroup finish sooner -- 15.0 -> 9.5 us for M <= 1024, N = 256, K = 1024
        // (tools/diag/small_gemm_probe.py, profiles/round5_small_gemm_probe.txt); no gain for K = 256, a loss once the
        // 32 x 32 tiles need a second round of workgroups
        const long b32 = (long)((g.M + 31) / 32) * ((g.N + 31) / 32);
        if (variant == 24 && g.K2 <= 0 && g.K1 >= 512 && b32 <= compute_units()) {
            if (fast_ok(g, 8, 16)) variant = 29;
            else if (fast_ok(g, 4, 16)) variant = 27;
        }
        // Round 5: long contractions on the 32 x 64 tile with the K range over FOUR wave pairs (eight waves = two per SIMD:
        // a wave's LDS / global waits are covered by its partner's MFMAs) instead of two -- the second embedding layer at
        // config 2 15.4 -> 14.9 us, the gates' data gradient (K = 512) 25.6 -> 24.3 us (profiles/round5_small_gemm_probe.txt);
        // 1.031 -> 1.046 M scene-steps/s on the headline (profiles/round5_eight_wave_gemms.txt)
        if (variant == 24 && g.K2 <= 0 && g.K1 >= 512 && fast_ok(g, 4, 16)) variant = 30;
        // ... and the 32 x 128 tile likewise with the K range over two wave quartets (the gates' data gradient in training,
        // [2048, 512] x [448, 512]^T: 14.8 -> 13.9 us)
        if (variant == 25 && g.K2 <= 0 && g.K1 >= 512 && fast_ok(g, 2, 32)) variant = 31;
    }
    switch (variant) {
        case 12: TNP_TRY_FAST(2, 2, 1, 1, 32, EPI_BIAS); break;  // 64x64, 4 waves
        case 24:
            TNP_TRY_PIPE(1, 2, 2, 1, 32, EPI_BIAS);              // pipelined: 32x64, split-K 2
            // K a multiple of 96 but not of 64 (the directional grid: 2 x 12 x 12 = 288 inputs): split-K 3 keeps the
            // pipelined kernel instead of falling to the masked general one
            TNP_TRY_PIPE(1, 2, 3, 1, 32, EPI_BIAS);
            break;
        case 25: TNP_TRY_PIPE(1, 4, 1, 1, 32, EPI_BIAS); break;  // pipelined: 32x128, four waves over the whole K
        case 26: TNP_TRY_PIPE(1, 2, 2, 2, 32, EPI_BIAS); break;  // pipelined: 32x128, split-K 2, two blocks per wave
        case 27: TNP_TRY_PIPE(1, 1, 4, 1, 16, EPI_BIAS); break;  // 32x32, split-K 4, 61 KB (two workgroups per CU): small batches when K % 128 != 0
        case 29: TNP_TRY_PIPE(1, 1, 8, 1, 16, EPI_BIAS); break;  // 32x32, split-K 8 (eight waves, 120 KB): small batches, long K
        case 33: TNP_TRY_PIPE(2, 2, 1, 1, 32, EPI_BIAS); break;  // 64x64 pipelined, four waves: up to four rounds of workgroups
        case 31: TNP_TRY_PIPE(1, 4, 2, 1, 32, EPI_BIAS); break;  // 32x128, K range over two wave quartets (eight waves)
        case 30: TNP_TRY_PIPE(1, 2, 4, 1, 16, EPI_BIAS); break;  // 32x64, split-K 4 (eight waves): long K at one workgroup per CU
        default: TNP_FAIL(-1, "tnp_linear_forward: unknown variant %d (0 = automatic, 12, 24 .. 27, 29 .. 31, 33, 40 .. 45)", variant);
    }
    // shape not eligible for the fast path: masked general kernel
    if (big_tiles >= 192) return launch_general<4, 2, 1, 1, 32, EPI_BIAS>(g, s);
    return launch_general<1, 2, 2, 1, 32, EPI_BIAS>(g, s);
}

// [x | h] x [W_ih | W_hh]^T is contracted h FIRST ([h | x] x [W_hh | W_ih]^T: the same sum in another order) in every
// variant: the pooled columns of x -- the last thing the step produces -- enter the sum last.
static void gates_h_first(GemmArgs &g) {
    if (g.K2 <= 0) return;
    std::swap(g.A1, g.A2); std::swap(g.lda1, g.lda2); std::swap(g.K1, g.K2);
    std::swap(g.B1, g.B2); std::swap(g.ldb1, g.ldb2); std::swap(g.bias1, g.bias2);
}

int launch_lstm_gates(const GemmArgs &g_in, int variant, hipStream_t s) {
    GemmArgs g = g_in;
    gates_h_first(g);
    check_vec(g);
    if (g.H % 32 != 0) TNP_FAIL(-1, "LSTM hidden_dim must be a multiple of 32 (got %d)", g.H);
    // measured on MI355X (profiles/archive/round1_b_variant_sweep.jsonl, tools/experiments/gemm_probe.hip): big batches the 128-track
    // kernel, else "gate split" (four waves own one gate block each over the whole K and swap tiles through LDS: no split-K
    // reduction, half the staged chunks per thread; 17.9 -> 16.1 us at config 2) when K1, K2 are multiples of 32
    // round 5 (tools/diag/gates_variant_sweep.py, profiles/round5_gates_variant_sweep.txt): one round of workgroups (M <= 2048 at
    // H = 128) -> the gate split with the K range over two wave quartets (22: eight waves, two per SIMD; 1.046 -> 1.068 M
    // scene-steps/s on the headline); several rounds -> the four-wave gate split (21: two workgroups co-resident per CU; at
    // 4096 tracks 2.13 ms per forward against 2.47 with the 128-track tiles); from 8192 tracks the 128-track kernel (5; round 6: its pipelined form 6)
    if (variant >= 30 && variant <= 34) return launch_skinny_gates(g, variant, s);
    // up to 512 tracks the 16-track gates kernel (16 tracks x 4 units per workgroup, K over four waves: variant 34) -- one
    // 36-agent scene 8.0 us against 14.2, a batch_size-8 training batch (310 tracks) 11.2 against 14.7 (profiles/round6_small_*)
    if (variant == 0 && g.M <= tuning().skinny_gates_max_rows && g.H % 4 == 0 && skinny_ok(g)) return launch_skinny_gates(g, 34, s);
    if (variant == 0) {
        const long wgs = (long)((g.M + 31) / 32) * ((g.H + 31) / 32);
        if (g.M >= 8192) variant = 6;   // 128-track tiles, pipelined: config 3 (16384 tracks) 2.47 -> 2.31 ms per forward, 8192 tracks 1.37 -> 1.27
                                        // (tools/diag/gates_big_probe.py; bit-identical to variant 5: same tile, same sum order)
        else if (wgs <= compute_units() && fast_ok(g, 2, 32)) variant = 22;
        else variant = fast_ok(g, 1, 32) ? 21 : 20;
    }
    switch (variant) {
        case 5: TNP_TRY_FAST(4, 1, 2, 4, 16, EPI_LSTM); break;   // 128 tracks x 32 units, 8 waves
        case 6: TNP_TRY_PIPE(4, 1, 2, 4, 16, EPI_LSTM); TNP_TRY_FAST(4, 1, 2, 4, 16, EPI_LSTM); break;   // the same tile on the three-stage ring (round 6)
        case 20: TNP_TRY_PIPE(1, 1, 4, 4, 16, EPI_LSTM); break;  // pipelined: 32 tracks x 32 units, split-K 4
        case 21: TNP_TRY_PIPE(1, 4, 1, 1, 32, EPI_LSTM); TNP_TRY_PIPE(1, 1, 4, 4, 16, EPI_LSTM); break;   // gate split
        case 22: TNP_TRY_PIPE(1, 4, 2, 1, 32, EPI_LSTM); TNP_TRY_PIPE(1, 4, 1, 1, 32, EPI_LSTM); TNP_TRY_PIPE(1, 1, 4, 4, 16, EPI_LSTM); break;   // gate split x K split 2 (eight waves)
        default: TNP_FAIL(-1, "lstm gates: unknown variant %d (0 = automatic, 5, 6, 20, 21, 22, 30 .. 34)", variant);
    }
    if (g.M >= 4096) return launch_general<2, 1, 2, 4, 32, EPI_LSTM>(g, s);
    return launch_general<1, 1, 4, 4, 16, EPI_LSTM>(g, s);
}

}  // namespace tnp

