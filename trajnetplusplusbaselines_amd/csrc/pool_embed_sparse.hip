// Sparse first layer of the grid-embedding MLP (reference lstm/gridbased_pooling.py:107-109 applied to the social
// grid of :145-170), gfx950.
//
// The social grid of an ego holds at most N-1 occupied cells out of n*n, each carrying the C-vector of ONE
// neighbour, so   Linear(C*n*n -> N1)(grid)[i, o] = b[o] + sum_{occupied cells c} sum_ch W[o, ch*n*n + c] * enc[j(i,c), ch].
// At BASELINE config 2 (n = 16, C = 16, 31 neighbours) that is 8.3x fewer multiply-adds than the dense GEMM,
// but the non-zeros are unstructured inside any 32-wide block, so the matrix cores cannot skip them.  The fp32
// VECTOR rate of CDNA4 equals its fp32 MFMA rate (157.3 TFLOP/s), so the sparse sum runs on the VALU:
//
//   * one workgroup = 128 egos x 256 outputs x a range of cells; lane <-> output column, 4 waves = 256 columns;
//   * the [128 x 256] fp32 accumulator tile lives in LDS (128 KiB of the CU's 160 KiB) because the ego of a
//     (cell, ego) hit is data dependent; a hit is  acc[ego][lane] += sum_ch w[ch] * enc[j][ch]  with the cell's
//     C weights w[ch] = W'[c][ch][o] held in VGPRs (W' = cell-major copy of the weight, o contiguous -> 256-byte
//     coalesced wave loads) and reused by every ego of the tile that has cell c occupied (~15 per cell);
//   * enc[j] is wave-uniform (scalar loads / broadcast), hits of one cell touch distinct egos, so they are
//     processed 4 at a time to overlap the LDS read-modify-write latencies;
//   * the int16 winner table of the tile's cell range is staged transposed in LDS and scanned with ballots;
//   * workgroup -> (output block, cell range) is XCD aware: the weight slice an XCD streams (2 MiB at config 2)
//     stays in its 4 MiB L2, so the 16.8 MiB weight is read from HBM / Infinity Cache once per launch.
// Cell ranges (split S) give enough workgroups for 256 CUs; their partial tiles are summed in fixed order by
// a small reduce kernel that also applies bias + ReLU (deterministic, no atomics).
#include "tnp_internal.h"
#include <stdlib.h>

namespace tnp {

#define SP_TE 128   // egos per workgroup
#define SP_OB 256   // outputs per workgroup (4 waves x 64 lanes)
#define SP_U 4      // hits in flight
#define SP_EQ 4     // ego groups per workgroup (waves per SIMD)

struct SparseArgs {
    const int16_t *winners;  // [M][ncell]
    const float *enc;        // [M][ldv]
    int ldv;
    const int32_t *row_base; // [M] first row of the row's scene
    const float *Wp;         // [ncell][C][N1]
    const float *bias;       // [N1]
    int M, ncell, C, N1;
    int S, cps;              // cell split, cells per split
    int ego_tiles, out_blocks;
    float *out;              // S == 1: [M][ldo] final (bias + relu applied); S > 1: partial [S][M][N1]
    int ldo;
    int relu;
};

// SP_EQ ego groups x 4 column sets = SP_EQ*4 waves per workgroup.  Wave (q, cs) owns the accumulator entries
// (egos of group q) x (columns of set cs): every (ego, column) address has exactly one writer wave, so the LDS
// float adds are race free and their order is program order (deterministic), while SP_EQ waves per SIMD hide the
// scalar-load and LDS latencies of each other.
template <int C, int EQ, bool ATOMIC>
__global__ void __launch_bounds__(256 * EQ) pool_embed_sparse_kernel(const SparseArgs a) {
    extern __shared__ __attribute__((aligned(16))) float ssm[];
    float *acc = ssm;                                                  // [SP_TE][SP_OB]
    int16_t *wl = reinterpret_cast<int16_t *>(ssm + SP_TE * SP_OB);    // [cps][SP_TE + 2]
    constexpr int WLS = SP_TE + 2;
    constexpr int NTH = 256 * EQ;
    constexpr int EPG = SP_TE / EQ;                                    // egos per group (<= 64)
    static_assert(EPG * EQ == SP_TE, "ego groups");
    constexpr int NH = (EPG + 63) / 64;                                // 64-lane chunks of the ego group

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cs = wave & 3, eq = wave >> 2;
    // XCD-aware decomposition: blocks b, b+8, ... share an XCD; give each XCD a fixed set of (output block, split)
    const int ncombo = a.out_blocks * a.S;
    int combo, et;
    if ((ncombo & 7) == 0) {
        const int cpx = ncombo >> 3, xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
        combo = xcd * cpx + (l % cpx);
        et = l / cpx;
    } else {
        combo = blockIdx.x % ncombo;
        et = blockIdx.x / ncombo;
    }
    const int ob = combo % a.out_blocks, sp = combo / a.out_blocks;
    const int row0 = et * SP_TE;
    const int c0 = sp * a.cps;
    const int ncl = min(a.cps, a.ncell - c0);                          // cells of this workgroup
    const int o = ob * SP_OB + cs * 64 + lane;
    const bool o_ok = o < a.N1;
    const int oc = o_ok ? o : (a.N1 - 1);

    for (int q = tid; q < SP_TE * SP_OB; q += NTH) acc[q] = 0.0f;
    for (int q = tid; q < SP_TE * a.cps; q += NTH) {
        const int e = q / a.cps, cc = q - e * a.cps;
        const int row = row0 + e;
        int16_t v = -1;
        if (row < a.M && cc < ncl) v = a.winners[(size_t)row * a.ncell + c0 + cc];
        wl[cc * WLS + e] = v;
    }
    int rbv[NH];
#pragma unroll
    for (int hh = 0; hh < NH; ++hh)
        rbv[hh] = a.row_base[min(row0 + min(eq * EPG + hh * 64 + lane, SP_TE - 1), a.M - 1)];
    __syncthreads();

    float *accl = acc + cs * 64 + lane;                                // this lane's column of the tile
    const float *wcol = a.Wp + oc;

    auto load_w = [&](float (&w)[C], int cc) {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) w[ch] = wcol[((size_t)(c0 + cc) * C + ch) * a.N1];
    };
    auto process = [&](const float (&w)[C], int cc) {
#pragma unroll
      for (int hh = 0; hh < NH; ++hh) {
        const int lego = hh * 64 + lane;                               // ego index inside the group
        const int wv = (lego < EPG) ? (int)wl[cc * WLS + eq * EPG + lego] : -1;
        const int rb = rbv[hh];
        unsigned long long mask = __ballot(wv >= 0);
        while (mask) {
            int eg[SP_U];
            const float *ep[SP_U];
            bool ok[SP_U];
#pragma unroll
            for (int u = 0; u < SP_U; ++u) {
                ok[u] = mask != 0ull;
                const int b = ok[u] ? (__ffsll((long long)mask) - 1) : 0;
                if (ok[u]) mask &= mask - 1ull;
                const int wj = __builtin_amdgcn_readlane(wv, b);
                const int base = __builtin_amdgcn_readlane(rb, b);
                const int j = ok[u] ? (base + wj) : 0;
                eg[u] = eq * EPG + hh * 64 + b;
                ep[u] = a.enc + (size_t)j * a.ldv;
            }
            if (ATOMIC) {
#pragma unroll
                for (int u = 0; u < SP_U; ++u) {
                    float c = 0.0f;
#pragma unroll
                    for (int ch = 0; ch < C; ++ch) c = fmaf(w[ch], ep[u][ch], c);
                    if (ok[u]) __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float *)(accl + eg[u] * SP_OB), c, 0, 0, false);
                }
            } else {
                float av[SP_U];
#pragma unroll
                for (int u = 0; u < SP_U; ++u) av[u] = accl[eg[u] * SP_OB];
#pragma unroll
                for (int u = 0; u < SP_U; ++u) {
#pragma unroll
                    for (int ch = 0; ch < C; ++ch) av[u] = fmaf(w[ch], ep[u][ch], av[u]);
                }
#pragma unroll
                for (int u = 0; u < SP_U; ++u)
                    if (ok[u]) accl[eg[u] * SP_OB] = av[u];
            }
        }
      }
    };
    // the C weights of the next cell are in flight while the hits of the current cell are processed
    float wA[C], wB[C];
    if (ncl > 0) load_w(wA, 0);
    for (int cc = 0; cc < ncl; cc += 2) {
        if (cc + 1 < ncl) load_w(wB, cc + 1);
        process(wA, cc);
        if (cc + 1 < ncl) {
            if (cc + 2 < ncl) load_w(wA, cc + 2);
            process(wB, cc + 1);
        }
    }
    __syncthreads();   // all adds of the tile have landed before it is read back

    // epilogue: wave (q, cs) writes its egos' rows of its columns
    if (o_ok) {
        if (a.S == 1) {
            const float b = a.bias ? a.bias[o] : 0.0f;
            for (int e = eq * EPG; e < (eq + 1) * EPG; ++e) {
                const int row = row0 + e;
                if (row >= a.M) break;
                float v = accl[e * SP_OB] + b;
                if (a.relu) v = v > 0.0f ? v : 0.0f;
                a.out[(size_t)row * a.ldo + o] = v;
            }
        } else {
            float *pp = a.out + (size_t)sp * a.M * a.N1;
            for (int e = eq * EPG; e < (eq + 1) * EPG; ++e) {
                const int row = row0 + e;
                if (row >= a.M) break;
                pp[(size_t)row * a.N1 + o] = accl[e * SP_OB];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Staged variant: per pass over a range of cells, the workgroup first builds a compact hit list in LDS --
// (ego, the neighbour's C values) for every occupied (ego, cell) of the tile, grouped by cell -- with all
// lanes gathering in parallel (lane <-> (ego, cell) entry of the winner table).  The accumulation loop then
// touches only LDS (broadcast reads of the staged values, read-modify-write of the accumulator tile) plus the
// double-buffered weight loads, and needs almost no scalar bookkeeping per hit.
// ---------------------------------------------------------------------------------------------------------
template <int C, int EQ>
__global__ void __launch_bounds__(256 * EQ) pool_embed_sparse_staged_kernel(const SparseArgs a, int hcap) {
    extern __shared__ __attribute__((aligned(16))) float ssm[];
    constexpr int NTH = 256 * EQ;
    constexpr int EPG = SP_TE / EQ;
    float *acc = ssm;                                        // [SP_TE][SP_OB]
    float *henc = acc + SP_TE * SP_OB;                       // [hcap][C]
    int *hego = reinterpret_cast<int *>(henc + (size_t)hcap * C);   // [hcap]
    int *cstart = hego + hcap;                               // [cps + 1]  start of each cell's hits in the pass
    int *cursor = cstart + 132;                              // [cps]
    int *cnt = cursor + 132;                                 // [cps]  hits per cell (whole split)
    int *ctl = cnt + 132;                                    // [4]    pass bounds

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cs = wave & 3, eq = wave >> 2;
    const int ncombo = a.out_blocks * a.S;
    int combo, et;
    if ((ncombo & 7) == 0) {
        const int cpx = ncombo >> 3, xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
        combo = xcd * cpx + (l % cpx);
        et = l / cpx;
    } else {
        combo = blockIdx.x % ncombo;
        et = blockIdx.x / ncombo;
    }
    const int ob = combo % a.out_blocks, sp = combo / a.out_blocks;
    const int row0 = et * SP_TE;
    const int c0 = sp * a.cps;
    const int ncl = min(a.cps, a.ncell - c0);
    const int o = ob * SP_OB + cs * 64 + lane;
    const bool o_ok = o < a.N1;
    const int oc = o_ok ? o : (a.N1 - 1);

    for (int q = tid; q < SP_TE * SP_OB; q += NTH) acc[q] = 0.0f;
    for (int q = tid; q < 132; q += NTH) cnt[q] = 0;
    __syncthreads();
    // hits per cell of this tile / split
    for (int q = tid; q < SP_TE * ncl; q += NTH) {
        const int e = q / ncl, cc = q - e * ncl;
        const int row = row0 + e;
        if (row < a.M && a.winners[(size_t)row * a.ncell + c0 + cc] >= 0) atomicAdd(&cnt[cc], 1);
    }
    __syncthreads();

    float *accl = acc + cs * 64 + lane;
    const float *wcol = a.Wp + oc;
    auto load_w = [&](float (&w)[C], int cc) {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) w[ch] = wcol[((size_t)(c0 + cc) * C + ch) * a.N1];
    };
    auto process = [&](const float (&w)[C], int cc) {
        const int k0 = cstart[cc], k1 = cstart[cc + 1];
        for (int k = k0; k < k1; ++k) {
            const int ego = __builtin_amdgcn_readfirstlane(hego[k]);
            if (ego / EPG != eq) continue;                    // single writer per (ego, column)
            const float *ev = henc + (size_t)k * C;
            float av = accl[ego * SP_OB];
#pragma unroll
            for (int ch = 0; ch < C; ch += 4) {
                const float4 e4 = *reinterpret_cast<const float4 *>(ev + ch);
                av = fmaf(w[ch], e4.x, av); av = fmaf(w[ch + 1], e4.y, av);
                av = fmaf(w[ch + 2], e4.z, av); av = fmaf(w[ch + 3], e4.w, av);
            }
            accl[ego * SP_OB] = av;
        }
    };

    int pass_lo = 0;
    while (pass_lo < ncl) {
        // pass bounds: as many cells as fit the staging buffer (at least one)
        if (tid == 0) {
            int tot = 0, c = pass_lo;
            while (c < ncl && (c == pass_lo || tot + cnt[c] <= hcap)) { cstart[c - pass_lo] = tot; tot += cnt[c]; ++c; }
            cstart[c - pass_lo] = tot;
            ctl[0] = c;
        }
        for (int q = tid; q < 132; q += NTH) cursor[q] = 0;
        __syncthreads();
        const int pass_hi = ctl[0];
        const int npc = pass_hi - pass_lo;
        // fill: lane <-> (ego, cell) entry; gather the neighbour's values next to the ego id
        for (int q = tid; q < SP_TE * npc; q += NTH) {
            const int e = q / npc, cc = q - e * npc;
            const int row = row0 + e;
            if (row >= a.M) continue;
            const int wv = a.winners[(size_t)row * a.ncell + c0 + pass_lo + cc];
            if (wv < 0) continue;
            const int slot = cstart[cc] + atomicAdd(&cursor[cc], 1);
            if (slot >= hcap) continue;                       // a single over-full cell: cannot happen (<= SP_TE hits)
            hego[slot] = e;
            const float *src = a.enc + (size_t)(a.row_base[row] + wv) * a.ldv;
#pragma unroll
            for (int ch = 0; ch < C; ch += 4)
                *reinterpret_cast<float4 *>(henc + (size_t)slot * C + ch) = *reinterpret_cast<const float4 *>(src + ch);
        }
        __syncthreads();
        float wA[C], wB[C];
        load_w(wA, pass_lo);
        for (int cc = 0; cc < npc; cc += 2) {
            if (cc + 1 < npc) load_w(wB, pass_lo + cc + 1);
            process(wA, cc);
            if (cc + 1 < npc) {
                if (cc + 2 < npc) load_w(wA, pass_lo + cc + 2);
                process(wB, cc + 1);
            }
        }
        __syncthreads();
        pass_lo = pass_hi;
    }

    if (o_ok) {
        if (a.S == 1) {
            const float b = a.bias ? a.bias[o] : 0.0f;
            for (int e = eq * EPG; e < (eq + 1) * EPG; ++e) {
                const int row = row0 + e;
                if (row >= a.M) break;
                float v = accl[e * SP_OB] + b;
                if (a.relu) v = v > 0.0f ? v : 0.0f;
                a.out[(size_t)row * a.ldo + o] = v;
            }
        } else {
            float *pp = a.out + (size_t)sp * a.M * a.N1;
            for (int e = eq * EPG; e < (eq + 1) * EPG; ++e) {
                const int row = row0 + e;
                if (row >= a.M) break;
                pp[(size_t)row * a.N1 + o] = accl[e * SP_OB];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// List variant: a compact per-(cell, ego group) hit list  (neighbour row j << 8 | ego)  is built in LDS first
// (lane <-> winner-table entry, integer LDS atomics for the slots), so the accumulation loop carries almost no
// scalar bookkeeping: per hit one broadcast LDS read, one s_load_dwordx16 of the neighbour's values, C FMAs with
// SGPR operands and the LDS read-modify-write of the accumulator entry.  EQ ego groups x 4 column sets of waves
// keep 4 waves per SIMD in flight; a wave only walks the hits of its own ego group (single writer per entry).
// ---------------------------------------------------------------------------------------------------------
template <int C, int EQ>
__global__ void __launch_bounds__(256 * EQ) pool_embed_sparse_list_kernel(const SparseArgs a, int hcap) {
    extern __shared__ __attribute__((aligned(16))) float ssm[];
    constexpr int NTH = 256 * EQ;
    constexpr int EPG = SP_TE / EQ;
    constexpr int TAB = 128 * EQ + 4;                        // >= cps * EQ + 1
    float *acc = ssm;                                        // [SP_TE][SP_OB]
    int *hits = reinterpret_cast<int *>(acc + SP_TE * SP_OB);   // [hcap]
    int *cstart = hits + hcap;                               // [TAB] start of (cell, group) lists
    int *cursor = cstart + TAB;                              // [TAB]
    int *cnt = cursor + TAB;                                 // [TAB]
    int *ctl = cnt + TAB;                                    // [4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cs = wave & 3, eq = wave >> 2;
    const int ncombo = a.out_blocks * a.S;
    int combo, et;
    if ((ncombo & 7) == 0) {
        const int cpx = ncombo >> 3, xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
        combo = xcd * cpx + (l % cpx);
        et = l / cpx;
    } else {
        combo = blockIdx.x % ncombo;
        et = blockIdx.x / ncombo;
    }
    const int ob = combo % a.out_blocks, sp = combo / a.out_blocks;
    const int row0 = et * SP_TE;
    const int c0 = sp * a.cps;
    const int ncl = min(a.cps, a.ncell - c0);
    const int o = ob * SP_OB + cs * 64 + lane;
    const bool o_ok = o < a.N1;
    const int oc = o_ok ? o : (a.N1 - 1);

    for (int q = tid; q < SP_TE * SP_OB; q += NTH) acc[q] = 0.0f;
    for (int q = tid; q < TAB; q += NTH) cnt[q] = 0;
    __syncthreads();
    for (int q = tid; q < SP_TE * ncl; q += NTH) {
        const int e = q / ncl, cc = q - e * ncl;
        const int row = row0 + e;
        if (row < a.M && a.winners[(size_t)row * a.ncell + c0 + cc] >= 0) atomicAdd(&cnt[cc * EQ + e / EPG], 1);
    }
    __syncthreads();

    float *accl = acc + cs * 64 + lane;
    const float *wcol = a.Wp + oc;
    const float *encp = a.enc;
    const int ldv = a.ldv;
    auto load_w = [&](float (&w)[C], int cc) {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) w[ch] = wcol[((size_t)(c0 + cc) * C + ch) * a.N1];
    };
    auto process = [&](const float (&w)[C], int cc) {
        int k = cstart[cc * EQ + eq];
        const int k1 = cstart[cc * EQ + eq + 1];
        for (; k + 1 < k1; k += 2) {                          // two hits (distinct egos) in flight
            const int h0 = __builtin_amdgcn_readfirstlane(hits[k]), h1 = __builtin_amdgcn_readfirstlane(hits[k + 1]);
            const float *e0 = encp + (size_t)(h0 >> 8) * ldv, *e1 = encp + (size_t)(h1 >> 8) * ldv;
            const int g0 = (h0 & 255) * SP_OB, g1 = (h1 & 255) * SP_OB;
            float a0 = accl[g0], a1 = accl[g1];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) { a0 = fmaf(w[ch], e0[ch], a0); a1 = fmaf(w[ch], e1[ch], a1); }
            accl[g0] = a0; accl[g1] = a1;
        }
        if (k < k1) {
            const int h0 = __builtin_amdgcn_readfirstlane(hits[k]);
            const float *e0 = encp + (size_t)(h0 >> 8) * ldv;
            const int g0 = (h0 & 255) * SP_OB;
            float a0 = accl[g0];
#pragma unroll
            for (int ch = 0; ch < C; ++ch) a0 = fmaf(w[ch], e0[ch], a0);
            accl[g0] = a0;
        }
    };

    int pass_lo = 0;
    while (pass_lo < ncl) {
        if (tid == 0) {   // pass bounds: as many cells as fit the hit list (at least one)
            int tot = 0, c = pass_lo;
            while (c < ncl) {
                int cell_tot = 0;
                for (int gq = 0; gq < EQ; ++gq) cell_tot += cnt[c * EQ + gq];
                if (c > pass_lo && tot + cell_tot > hcap) break;
                for (int gq = 0; gq < EQ; ++gq) { cstart[(c - pass_lo) * EQ + gq] = tot; tot += cnt[c * EQ + gq]; }
                ++c;
            }
            cstart[(c - pass_lo) * EQ] = tot;
            ctl[0] = c;
        }
        for (int q = tid; q < TAB; q += NTH) cursor[q] = 0;
        __syncthreads();
        const int pass_hi = ctl[0];
        const int npc = pass_hi - pass_lo;
        for (int q = tid; q < SP_TE * npc; q += NTH) {
            const int e = q / npc, cc = q - e * npc;
            const int row = row0 + e;
            if (row >= a.M) continue;
            const int wv = a.winners[(size_t)row * a.ncell + c0 + pass_lo + cc];
            if (wv < 0) continue;
            const int li = cc * EQ + e / EPG;
            const int slot = cstart[li] + atomicAdd(&cursor[li], 1);
            if (slot < hcap) hits[slot] = ((a.row_base[row] + wv) << 8) | e;
        }
        __syncthreads();
        float wA[C], wB[C];
        load_w(wA, pass_lo);
        for (int cc = 0; cc < npc; cc += 2) {
            if (cc + 1 < npc) load_w(wB, pass_lo + cc + 1);
            process(wA, cc);
            if (cc + 1 < npc) {
                if (cc + 2 < npc) load_w(wA, pass_lo + cc + 2);
                process(wB, cc + 1);
            }
        }
        __syncthreads();
        pass_lo = pass_hi;
    }

    if (o_ok) {
        if (a.S == 1) {
            const float b = a.bias ? a.bias[o] : 0.0f;
            for (int e = eq * EPG; e < (eq + 1) * EPG; ++e) {
                const int row = row0 + e;
                if (row >= a.M) break;
                float v = accl[e * SP_OB] + b;
                if (a.relu) v = v > 0.0f ? v : 0.0f;
                a.out[(size_t)row * a.ldo + o] = v;
            }
        } else {
            float *pp = a.out + (size_t)sp * a.M * a.N1;
            for (int e = eq * EPG; e < (eq + 1) * EPG; ++e) {
                const int row = row0 + e;
                if (row >= a.M) break;
                pp[(size_t)row * a.N1 + o] = accl[e * SP_OB];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// Register-accumulator variant.  Wave (q, cs) keeps the accumulators of its 32 egos x 64 columns in 32 VGPRs
// (lane <-> column): the ego of a hit is wave-uniform, so "acc[ego] += c" is a scalar jump into a 32-way switch
// of single v_add instructions -- no LDS traffic for the accumulators at all.  LDS only holds the compact
// per-(cell, ego group) hit lists (neighbour row << 8 | ego).  Per hit: one s_load_dwordx16 of the neighbour's
// values (SGPR FMA operands), C FMAs, one add.
// ---------------------------------------------------------------------------------------------------------
#define SP_ACC_CASE(i) case i: acc[i] += c; break;
#define SP_ACC_SWITCH(el, c) switch (el) { \
    SP_ACC_CASE(0) SP_ACC_CASE(1) SP_ACC_CASE(2) SP_ACC_CASE(3) SP_ACC_CASE(4) SP_ACC_CASE(5) SP_ACC_CASE(6) SP_ACC_CASE(7) \
    SP_ACC_CASE(8) SP_ACC_CASE(9) SP_ACC_CASE(10) SP_ACC_CASE(11) SP_ACC_CASE(12) SP_ACC_CASE(13) SP_ACC_CASE(14) SP_ACC_CASE(15) \
    SP_ACC_CASE(16) SP_ACC_CASE(17) SP_ACC_CASE(18) SP_ACC_CASE(19) SP_ACC_CASE(20) SP_ACC_CASE(21) SP_ACC_CASE(22) SP_ACC_CASE(23) \
    SP_ACC_CASE(24) SP_ACC_CASE(25) SP_ACC_CASE(26) SP_ACC_CASE(27) SP_ACC_CASE(28) SP_ACC_CASE(29) SP_ACC_CASE(30) SP_ACC_CASE(31) \
    default: break; }

template <int C>
__global__ void __launch_bounds__(1024) pool_embed_sparse_reg_kernel(const SparseArgs a, int hcap) {
    extern __shared__ __attribute__((aligned(16))) float ssm[];
    constexpr int EQ = 4, NTH = 1024, EPG = 32;
    constexpr int TAB = 128 * EQ + 4;
    int *hits = reinterpret_cast<int *>(ssm);                // [hcap]
    int *cstart = hits + hcap;                               // [TAB]
    int *cursor = cstart + TAB;                              // [TAB]
    int *cnt = cursor + TAB;                                 // [TAB]
    int *ctl = cnt + TAB;                                    // [4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cs = wave & 3, eq = wave >> 2;
    const int ncombo = a.out_blocks * a.S;
    int combo, et;
    if ((ncombo & 7) == 0) {
        const int cpx = ncombo >> 3, xcd = blockIdx.x & 7, l = blockIdx.x >> 3;
        combo = xcd * cpx + (l % cpx);
        et = l / cpx;
    } else {
        combo = blockIdx.x % ncombo;
        et = blockIdx.x / ncombo;
    }
    const int ob = combo % a.out_blocks, sp = combo / a.out_blocks;
    const int row0 = et * SP_TE;
    const int c0 = sp * a.cps;
    const int ncl = min(a.cps, a.ncell - c0);
    const int o = ob * SP_OB + cs * 64 + lane;
    const bool o_ok = o < a.N1;
    const int oc = o_ok ? o : (a.N1 - 1);

    for (int q = tid; q < TAB; q += NTH) cnt[q] = 0;
    __syncthreads();
    for (int q = tid; q < SP_TE * ncl; q += NTH) {
        const int e = q / ncl, cc = q - e * ncl;
        const int row = row0 + e;
        if (row < a.M && a.winners[(size_t)row * a.ncell + c0 + cc] >= 0) atomicAdd(&cnt[cc * EQ + e / EPG], 1);
    }
    __syncthreads();

    float acc[EPG];
#pragma unroll
    for (int i = 0; i < EPG; ++i) acc[i] = 0.0f;
    const float *wcol = a.Wp + oc;
    const float *encp = a.enc;
    const int ldv = a.ldv;
    auto load_w = [&](float (&w)[C], int cc) {
#pragma unroll
        for (int ch = 0; ch < C; ++ch) w[ch] = wcol[((size_t)(c0 + cc) * C + ch) * a.N1];
    };
    auto process = [&](const float (&w)[C], int cc) {
        int k = cstart[cc * EQ + eq];
        const int k1 = cstart[cc * EQ + eq + 1];
        for (; k < k1; ++k) {
            const int h0 = __builtin_amdgcn_readfirstlane(hits[k]);
            const float *e0 = encp + (size_t)(h0 >> 8) * ldv;
            float c = 0.0f;
#pragma unroll
            for (int ch = 0; ch < C; ++ch) c = fmaf(w[ch], e0[ch], c);
            const int el = (h0 & 255) - eq * EPG;
            SP_ACC_SWITCH(el, c)
        }
    };

    int pass_lo = 0;
    while (pass_lo < ncl) {
        if (tid == 0) {
            int tot = 0, c = pass_lo;
            while (c < ncl) {
                int cell_tot = 0;
                for (int gq = 0; gq < EQ; ++gq) cell_tot += cnt[c * EQ + gq];
                if (c > pass_lo && tot + cell_tot > hcap) break;
                for (int gq = 0; gq < EQ; ++gq) { cstart[(c - pass_lo) * EQ + gq] = tot; tot += cnt[c * EQ + gq]; }
                ++c;
            }
            cstart[(c - pass_lo) * EQ] = tot;
            ctl[0] = c;
        }
        for (int q = tid; q < TAB; q += NTH) cursor[q] = 0;
        __syncthreads();
        const int pass_hi = ctl[0];
        const int npc = pass_hi - pass_lo;
        for (int q = tid; q < SP_TE * npc; q += NTH) {
            const int e = q / npc, cc = q - e * npc;
            const int row = row0 + e;
            if (row >= a.M) continue;
            const int wv = a.winners[(size_t)row * a.ncell + c0 + pass_lo + cc];
            if (wv < 0) continue;
            const int li = cc * EQ + e / EPG;
            const int slot = cstart[li] + atomicAdd(&cursor[li], 1);
            if (slot < hcap) hits[slot] = ((a.row_base[row] + wv) << 8) | e;
        }
        __syncthreads();
        float wA[C], wB[C];
        load_w(wA, pass_lo);
        for (int cc = 0; cc < npc; cc += 2) {
            if (cc + 1 < npc) load_w(wB, pass_lo + cc + 1);
            process(wA, cc);
            if (cc + 1 < npc) {
                if (cc + 2 < npc) load_w(wA, pass_lo + cc + 2);
                process(wB, cc + 1);
            }
        }
        __syncthreads();
        pass_lo = pass_hi;
    }

    if (o_ok) {
        const float b = (a.S == 1 && a.bias) ? a.bias[o] : 0.0f;
        float *pp = (a.S == 1) ? a.out : a.out + (size_t)sp * a.M * a.N1;
        const int ld = (a.S == 1) ? a.ldo : a.N1;
#pragma unroll
        for (int i = 0; i < EPG; ++i) {
            const int row = row0 + eq * EPG + i;
            if (row < a.M) {
                float v = acc[i] + b;
                if (a.S == 1 && a.relu) v = v > 0.0f ? v : 0.0f;
                pp[(size_t)row * ld + o] = v;
            }
        }
    }
}

// Y[m][o] = act(bias[o] + sum_s partial[s][m][o]), fixed summation order
__global__ void __launch_bounds__(256) sparse_reduce_kernel(const float *partial, int S, int M, int N1, const float *bias,
                                                            int relu, float *out, int ldo) {
    const size_t total4 = (size_t)M * N1 / 4;
    const size_t plane = (size_t)M * N1;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < total4; q += (size_t)gridDim.x * blockDim.x) {
        const size_t idx = q * 4;
        const int m = (int)(idx / N1), oo = (int)(idx - (size_t)m * N1);
        float4 s = *reinterpret_cast<const float4 *>(partial + idx);
        for (int k = 1; k < S; ++k) {
            const float4 p = *reinterpret_cast<const float4 *>(partial + k * plane + idx);
            s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
        }
        if (bias) { s.x += bias[oo]; s.y += bias[oo + 1]; s.z += bias[oo + 2]; s.w += bias[oo + 3]; }
        if (relu) { s.x = s.x > 0.f ? s.x : 0.f; s.y = s.y > 0.f ? s.y : 0.f; s.z = s.z > 0.f ? s.z : 0.f; s.w = s.w > 0.f ? s.w : 0.f; }
        *reinterpret_cast<float4 *>(out + (size_t)m * ldo + oo) = s;
    }
}

__global__ void row_base_kernel(const int32_t *scene_start, int B, int32_t *row_base) {
    const int s = blockIdx.x;
    if (s >= B) return;
    const int lo = scene_start[s], hi = scene_start[s + 1];
    for (int r = lo + threadIdx.x; r < hi; r += blockDim.x) row_base[r] = lo;
}

int launch_row_base(const int32_t *scene_start, int B, int32_t *row_base, hipStream_t s) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(row_base_kernel, dim3(B), dim3(64), 0, s, scene_start, B, row_base);
    TNP_HIP(hipGetLastError());
    return 0;
}

bool sparse_supported(int C, int N1, int ncell) {
    return (C == 4 || C == 8 || C == 16 || C == 32) && N1 >= 4 && (N1 % 4 == 0) && ncell >= 1;
}

void sparse_plan(int M, int N1, int ncell, int &S, int &cps, int &ego_tiles, int &out_blocks) {
    ego_tiles = (M + SP_TE - 1) / SP_TE;
    out_blocks = (N1 + SP_OB - 1) / SP_OB;
    S = 1;
    while (S < 8 && ((long)ego_tiles * out_blocks * S < 200 || (ncell + S - 1) / S > 112) && S < ncell) S *= 2;
    cps = (ncell + S - 1) / S;
}

size_t sparse_partial_bytes(int M, int N1, int ncell) {
    int S, cps, et, obk;
    sparse_plan(M, N1, ncell, S, cps, et, obk);
    return S > 1 ? (size_t)S * M * N1 * sizeof(float) : 0;
}

int launch_pool_embed_sparse(const int16_t *winners, const float *enc, int ldv, const int32_t *row_base,
                             const float *Wp, const float *bias, int M, int ncell, int C, int N1, int relu,
                             float *out, int ldo, float *partial, hipStream_t s) {
    if (M <= 0) return 0;
    if (!sparse_supported(C, N1, ncell)) TNP_FAIL(-1, "sparse pooling embedding: unsupported C=%d N1=%d", C, N1);
    SparseArgs a;
    a.winners = winners; a.enc = enc; a.ldv = ldv; a.row_base = row_base; a.Wp = Wp; a.bias = bias;
    a.M = M; a.ncell = ncell; a.C = C; a.N1 = N1; a.relu = relu; a.ldo = ldo;
    sparse_plan(M, N1, ncell, a.S, a.cps, a.ego_tiles, a.out_blocks);
    if (a.cps > 120) TNP_FAIL(-1, "sparse pooling embedding: %d cells per split exceed the LDS winner tile", a.cps);
    if (a.S > 1 && !partial) TNP_FAIL(-1, "sparse pooling embedding: partial workspace missing");
    a.out = (a.S > 1) ? partial : out;
    if (a.S > 1) a.ldo = N1;
    const size_t smem = (size_t)SP_TE * SP_OB * 4 + (((size_t)a.cps * (SP_TE + 2) * 2 + 15) & ~(size_t)15);
    const int blocks = a.ego_tiles * a.out_blocks * a.S;
    static int sp_variant = -1;
    if (sp_variant < 0) { const char *e = getenv("TNP_SPARSE_VARIANT"); sp_variant = e ? atoi(e) : 4; }  // 4 = 16 waves, read-modify-write accumulators (measured best)
#define SP_LAUNCH2(CC, EQ, AT) { static bool set = false; if (!set) { TNP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>( \
        pool_embed_sparse_kernel<CC, EQ, AT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
        hipLaunchKernelGGL((pool_embed_sparse_kernel<CC, EQ, AT>), dim3(blocks), dim3(256 * EQ), smem, s, a); }
    // staged kernel: LDS = accumulator tile + hit list (hcap hits x (C values + ego id)) + per-cell tables
    const int hcap = (int)((163840 - (size_t)SP_TE * SP_OB * 4 - 4 * 132 * 4 - 64) / ((size_t)C * 4 + 4)) & ~3;
    const size_t smem_st = (size_t)SP_TE * SP_OB * 4 + (size_t)hcap * (C * 4 + 4) + 4 * 132 * 4;
#define SP_LAUNCH3(CC, EQ) { static bool set = false; if (!set) { TNP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>( \
        pool_embed_sparse_staged_kernel<CC, EQ>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
        hipLaunchKernelGGL((pool_embed_sparse_staged_kernel<CC, EQ>), dim3(blocks), dim3(256 * EQ), smem_st, s, a, hcap); }
    // list kernel: LDS = accumulator tile + hit list + 3 tables of 128*EQ+4 ints + ctl
#define SP_LAUNCH5(CC, EQ) { const int tab = 128 * EQ + 4; \
        const int hc = (int)((163840 - (size_t)SP_TE * SP_OB * 4 - (size_t)3 * tab * 4 - 64) / 4) & ~3; \
        const size_t sm = (size_t)SP_TE * SP_OB * 4 + (size_t)hc * 4 + (size_t)3 * tab * 4 + 64; \
        static bool set = false; if (!set) { TNP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>( \
        pool_embed_sparse_list_kernel<CC, EQ>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
        hipLaunchKernelGGL((pool_embed_sparse_list_kernel<CC, EQ>), dim3(blocks), dim3(256 * EQ), sm, s, a, hc); }
#define SP_LAUNCH6(CC) { const int tab = 128 * 4 + 4; const int hc = 16384; \
        const size_t sm = (size_t)hc * 4 + (size_t)3 * tab * 4 + 64; \
        static bool set = false; if (!set) { TNP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>( \
        pool_embed_sparse_reg_kernel<CC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
        hipLaunchKernelGGL((pool_embed_sparse_reg_kernel<CC>), dim3(blocks), dim3(1024), sm, s, a, hc); }
#define SP_LAUNCH(CC) { switch (sp_variant) { \
        case 30: SP_LAUNCH6(CC) break; \
        case 20: SP_LAUNCH5(CC, 1) break; case 21: SP_LAUNCH5(CC, 2) break; case 22: SP_LAUNCH5(CC, 4) break; \
        case 10: SP_LAUNCH3(CC, 1) break; case 11: SP_LAUNCH3(CC, 2) break; case 12: SP_LAUNCH3(CC, 4) break; \
        case 1: SP_LAUNCH2(CC, 1, true) break; case 2: SP_LAUNCH2(CC, 2, false) break; case 3: SP_LAUNCH2(CC, 2, true) break; \
        case 4: SP_LAUNCH2(CC, 4, false) break; case 5: SP_LAUNCH2(CC, 4, true) break; default: SP_LAUNCH2(CC, 1, false) break; } }
    if (C == 4) SP_LAUNCH(4) else if (C == 8) SP_LAUNCH(8) else if (C == 16) SP_LAUNCH(16) else SP_LAUNCH(32)
    TNP_HIP(hipGetLastError());
    if (a.S > 1) {
        const size_t total4 = (size_t)M * N1 / 4;
        const int rblocks = (int)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
        hipLaunchKernelGGL(sparse_reduce_kernel, dim3(rblocks), dim3(256), 0, s, partial, a.S, M, N1, bias, relu, out, ldo);
        TNP_HIP(hipGetLastError());
    }
    return 0;
}

}  // namespace tnp

extern "C" TNP_API size_t tnp_pool_embed_sparse_workspace_bytes(int M, int N1, int ncell) {
    return tnp::sparse_partial_bytes(M, N1, ncell);
}

extern "C" TNP_API int tnp_row_base(const int32_t *scene_start, int B, int32_t *row_base, void *stream) {
    return tnp::launch_row_base(scene_start, B, row_base, (hipStream_t)stream);
}

extern "C" TNP_API int tnp_pool_embed_sparse_forward(const int16_t *winners, const float *values, int ldv,
                                                     const int32_t *row_base, const float *W_cell_major,
                                                     const float *bias, int M, int ncell, int C, int N1, int relu,
                                                     float *out, int ldo, void *workspace, size_t workspace_bytes,
                                                     void *stream) {
    const size_t need = tnp::sparse_partial_bytes(M, N1, ncell);
    if (need > 0 && (workspace == nullptr || workspace_bytes < need))
        TNP_FAIL(-1, "tnp_pool_embed_sparse_forward: workspace too small (need %zu bytes)", need);
    if (ldo % 4 != 0 || (reinterpret_cast<uintptr_t>(out) & 15)) TNP_FAIL(-1, "output must be 16-byte aligned, ldo %% 4 == 0");
    return tnp::launch_pool_embed_sparse(winners, values, ldv, row_base, W_cell_major, bias, M, ncell, C, N1, relu, out,
                                         ldo, reinterpret_cast<float *>(workspace), (hipStream_t)stream);
}
