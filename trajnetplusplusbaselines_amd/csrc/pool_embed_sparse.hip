// Sparse first layer of the grid-embedding MLP (reference lstm/gridbased_pooling.py:107-109 applied to the social
// grid of :145-170), gfx950.
//
// The social grid of an ego holds at most N-1 occupied cells out of n*n, each carrying the C-vector of ONE
// neighbour, so   Linear(C*n*n -> N1)(grid)[i, o] = b[o] + sum_{occupied cells c} sum_ch W[o, ch*n*n + c] * enc[j(i,c), ch].
// At BASELINE config 2 (n = 16, C = 16, 31 neighbours) that is 8.3x fewer multiply-adds than the dense GEMM,
// but the non-zeros are unstructured inside any 32-wide block, so the matrix cores cannot skip them.  The fp32
// VECTOR rate of CDNA4 equals its fp32 MFMA rate (157.3 TFLOP/s), so the sparse sum runs on the VALU.
//
// Three kernels, chosen by launch_pool_embed_sparse:
//   * pool_embed_regacc_kernel (default: C <= 16, grids up to 512 cells) -- 64 egos x 128 columns per workgroup, 16 waves =
//     8 cell groups x 2 column sets, a wave's accumulators in VGPRs indexed by the wave-uniform ego, weights streamed from
//     a quad-major copy; described in front of the kernel;
//   * pool_embed_cellsplit_kernel (C = 32, or encodings beyond 32-bit row offsets; up to 440 cells) -- round 1's default,
//     one workgroup = 32 egos x 256 output columns, 16 waves:
//       - wave (q, cs): cells c with c % 4 == q, columns cs*64 + lane.  The C weights of the current cell W'[c][ch][o]
//         (cell-major copy of the weight, o contiguous -> 256-byte coalesced wave loads) sit in VGPRs, prefetched one
//         occupied cell ahead, and are reused by every ego of the tile that has cell c occupied;
//       - the int16 winner table of the tile is staged transposed in LDS; a cell's hits are found with one ballot (lane
//         <-> ego), and a hit costs: s_ff1 + bit clear, one v_readlane (32-bit byte offset of the neighbour's row), one
//         SMEM load of the C-float row (inline asm, two hits in flight before the single s_waitcnt), one LDS
//         read-modify-write of acc[q][ego][column] and C/2 v_pk_fma_f32;
//       - every wave group q owns a private 32 x 256 accumulator copy (4 x 32 KiB of LDS): one writer wave per entry,
//         adds in program order (deterministic), the four copies summed in the epilogue with bias + ReLU fused;
//   * pool_embed_sparse_kernel + sparse_reduce_kernel for larger grids: 128 egos x 256 columns x a RANGE of cells per
//     workgroup, partial tiles summed in a second launch.
// All three share the hit discovery (winner per (ego, cell), "last writer in ascending j wins", cell-0 clobber) and the
// packed-FMA hit; blocks b, b+8, ... share an XCD and a column block, whose weight slice stays in that XCD's L2.
#include "tnp_internal.h"
#include <hip/hip_ext.h>
#include <stdlib.h>

// Timing ablations and shader-clock stamps exist only in builds of tools/experiments/*.hip, which define
// TNP_EXPERIMENT_HOOKS before including this file; the library compiles none of it.
#ifdef TNP_EXPERIMENT_HOOKS
#define TNP_ABL_TPARAM , int ABL = 0
#define TNP_ABL_ARG , 0
#define TNP_ABL(bits) ((ABL & (bits)) != 0)
#else
#define TNP_ABL_TPARAM
#define TNP_ABL_ARG
#define TNP_ABL(bits) false
#endif

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
    // fused grid build (cell-split kernel, FG): the winner tile is computed from the positions instead of read
    const float *obs2;           // [M][2] current positions (NaN = absent)
    const int32_t *row_end;      // [M] one past the last row of the row's scene
    const int32_t *row_padded;   // [M] slots the reference pads the row's scene to (cell-0 clobber rule)
    int G;                       // cells per side
    float cell, half_x, half_y;
    int16_t *winners_out;        // optional [M][ncell]: the winner table for the training backward
};

// SP_EQ ego groups x 4 column sets = SP_EQ*4 waves per workgroup.  Wave (q, cs) owns the accumulator entries
// (egos of group q) x (columns of set cs): every (ego, column) address has exactly one writer wave, so the LDS
// float adds are race free and their order is program order (deterministic), while SP_EQ waves per SIMD hide the
// scalar-load and LDS latencies of each other.
template <int C, int EQ>
__global__ void __launch_bounds__(256 * EQ) pool_embed_sparse_kernel(const SparseArgs a) {
    constexpr int PF = 2;
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

    // uniform (SGPR) row base + 32-bit lane offset -> saddr-form global loads, no 64-bit per-lane address chains
    const unsigned ocu = (unsigned)oc;
    auto load_w = [&](float (&w)[C], int cc) {
        const float *wb = a.Wp + (size_t)(c0 + cc) * C * a.N1;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) w[ch] = (wb + (size_t)ch * a.N1)[ocu];
    };
    auto process = [&](float (&w)[C], int cc) {
        // pin the wait for THIS cell's weights here (straight-line code, so the compiler emits vmcnt(#younger
        // loads) and the prefetched sets stay in flight); inside the hit loop it would fall back to vmcnt(0)
#pragma unroll
        for (int ch = 0; ch < C; ++ch) asm("" : "+v"(w[ch]));
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
            {
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
    // the C weights of the next PF-1 cells are in flight while the hits of the current cell are processed
    float w[PF][C];
#pragma unroll
    for (int p = 0; p < PF - 1; ++p)
        if (p < ncl) load_w(w[p], p);
    for (int cc = 0; cc < ncl; cc += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const int cur = cc + p;
            if (cur < ncl) {
                if (cur + PF - 1 < ncl) load_w(w[(p + PF - 1) % PF], cur + PF - 1);
                process(w[p], cur);
            }
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

__global__ void row_base_kernel(const int32_t *scene_start, int B, int32_t *row_base, int32_t *row_end, int32_t *row_padded,
                                const int32_t *scene_slots, int n_max) {
    const int s = blockIdx.x;
    if (s >= B) return;
    const int lo = scene_start[s], hi = scene_start[s + 1];
    const int pad = scene_slots ? scene_slots[s] : n_max;
    for (int r = lo + threadIdx.x; r < hi; r += blockDim.x) {
        row_base[r] = lo;
        if (row_end) row_end[r] = hi;
        if (row_padded) row_padded[r] = pad;
    }
}

int launch_row_base(const int32_t *scene_start, int B, int32_t *row_base, hipStream_t s, int32_t *row_end, int32_t *row_padded,
                    const int32_t *scene_slots, int n_max) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(row_base_kernel, dim3(B), dim3(64), 0, s, scene_start, B, row_base, row_end, row_padded, scene_slots,
                       n_max);
    TNP_HIP(hipGetLastError());
    return 0;
}


constexpr int TL_TE = 32;    // egos per tile
constexpr int TL_OB = 256;   // output columns per workgroup
constexpr int TL_NQ = 4;     // cell groups of one workgroup (cell c belongs to group c % 4)
constexpr int TL_MAXCELL_LDS = 440;   // the int16 winner tile [ncell][34] must fit beside the 128 KiB accumulators

// ---------------------------------------------------------------------------------------------------------
// Cell-split kernel: the hit discovery of pool_embed_sparse_kernel (winner tile in LDS, ballot, scalar loads of
// the neighbour rows straight from `enc`, which stays hot in the scalar cache) on a tile of 32 egos x 256
// columns, with the CELLS split over the 4 wave groups of one workgroup (cell c -> group c % 4, each with its own
// accumulator copy).  The copies are summed in the epilogue with bias + ReLU fused: no partial sums in HBM, no
// reduce kernel, and the fixed per-workgroup work (zeroing, winner staging, epilogue) shrinks 4x.
// ---------------------------------------------------------------------------------------------------------

// Scalar (SMEM) load of one C-float neighbour row, issued as inline asm so that the U loads of a hit group are all
// in flight before the single s_waitcnt (the compiler otherwise sinks each load next to its use, exposing one
// scalar-load latency per hit).
// lowest set bit of a wave-uniform mask, cleared in place: s_ff1_i32_b64 + s_bitset0_b64 (the C expression
// mask &= mask - 1 costs three SALU instructions)
__device__ __forceinline__ int pop_bit(unsigned long long &mask) {
    const int b = __builtin_ctzll(mask);
    asm("s_bitset0_b64 %0, %1" : "+s"(mask) : "s"(b));
    return b;
}
__device__ __forceinline__ int pop_bit32(unsigned &mask) {
    const int b = __builtin_ctz(mask);
    asm("s_bitset0_b32 %0, %1" : "+s"(mask) : "s"(b));
    return b;
}

template <int C> struct SRow;
template <> struct SRow<4> { typedef float type __attribute__((ext_vector_type(4))); };
template <> struct SRow<8> { typedef float type __attribute__((ext_vector_type(8))); };
template <> struct SRow<16> { typedef float type __attribute__((ext_vector_type(16))); };
template <int C>
__device__ __forceinline__ void sload_row(typename SRow<C>::type &v, const float *base, unsigned byte_off) {
    if constexpr (C == 4) asm volatile("s_load_dwordx4 %0, %1, %2" : "=s"(v) : "s"(base), "s"(byte_off));
    else if constexpr (C == 8) asm volatile("s_load_dwordx8 %0, %1, %2" : "=s"(v) : "s"(base), "s"(byte_off));
    else asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(v) : "s"(base), "s"(byte_off));
}

// SASM: neighbour rows addressed with a 32-bit byte offset and loaded by inline-asm scalar loads (C <= 16); FG: the winner
// tile is built from the positions inside the kernel instead of read from the table.
// ABL (tools/experiments/sparse_ablate.hip only; the library instantiates 0): timing ablations -- 1 no weight loads,
// 2 no hits, 4 no neighbour-row loads, 8 no accumulator read-modify-write, 16 no cell loop at all.
template <int C, bool SASM, bool FG TNP_ABL_TPARAM, int TE = TL_TE, int OB = TL_OB, int NQ = TL_NQ, typename WT = int16_t>
__global__ void __launch_bounds__(64 * NQ * (OB / 64)) pool_embed_cellsplit_kernel(const SparseArgs a) {
    constexpr int NCS = OB / 64, U = 2;
    constexpr int WLS = TE + (sizeof(WT) == 1 ? 4 : 2), NTH = 64 * NQ * NCS;
    extern __shared__ __attribute__((aligned(16))) float csm[];
    float *acc = csm;                                                        // [NQ][TE][OB]
    WT *wl = reinterpret_cast<WT *>(csm + NQ * TE * OB);    // [ncell][WLS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: the cell loop and its addresses stay scalar
    const int cs = wave % NCS, q = wave / NCS;
    const int ob = blockIdx.x % a.out_blocks, tile = blockIdx.x / a.out_blocks;   // blocks b, b+8, .. share an XCD
    const int row0 = tile * TE;
    const int o = ob * OB + cs * 64 + lane;
    const unsigned ocu = (unsigned)(o < a.N1 ? o : a.N1 - 1);

    float *accl = acc + (size_t)q * TE * OB + cs * 64 + lane;
    constexpr int NW = NTH / 64;
    int *socc = reinterpret_cast<int *>(wl + (((size_t)a.ncell * WLS * sizeof(WT) + 15) & ~(size_t)15) / sizeof(WT));   // [ncell] cell has a hit in the tile
    int *wkey = reinterpret_cast<int *>(acc);                                     // FG: [TE][ncell] keys (not yet the accumulators)
    int *sg = wkey + TE * a.ncell;                                                // FG: lo / ns / ki / pad [TE] each, then x / y
    float *sp = reinterpret_cast<float *>(sg + 4 * TE);
    for (int c = tid; c < a.ncell; c += NTH) socc[c] = 0;
    if constexpr (FG) {
        // Winner tile straight from the positions (what grid_build_kernel computes, pool_grid.hip: the reference's exact
        // fp32 cell arithmetic, LDS integer max on key = 2*j + in_range = "last writer in ascending j wins", cell-0
        // clobber by out-of-range / absent / padded neighbours).  One wave per ego, lanes over the neighbours of its
        // scene; the int32 keys live in the (not yet zeroed) accumulator space.  The column-block workgroups of a tile
        // repeat this (TE x <= n_max pairs): cheaper than a kernel launch and the winner table's HBM round trip.
        // The egos' scene geometry is staged first (one thread per ego), so that the per-ego loop has one global load
        // per neighbour chunk and those of EU egos are in flight together.
        {
            const int4 m1 = {-1, -1, -1, -1};
            for (int idx = tid; idx < TE * a.ncell / 4; idx += NTH) reinterpret_cast<int4 *>(wkey)[idx] = m1;
        }
        if (tid < TE) {
            const int row = row0 + tid;
            int lo = 0, ns = 0, pad = 0;
            float2 pi = {-500.0f, -500.0f};
            if (row < a.M) {
                lo = a.row_base[row]; ns = a.row_end[row] - lo; pad = a.row_padded[row];
                pi = reinterpret_cast<const float2 *>(a.obs2)[row];
                if (pi.x != pi.x || pi.y != pi.y) { pi.x = -500.0f; pi.y = -500.0f; }
            }
            sg[tid] = lo; sg[TE + tid] = ns; sg[2 * TE + tid] = row - lo; sg[3 * TE + tid] = pad;
            sp[tid] = pi.x; sp[TE + tid] = pi.y;
        }
        __syncthreads();
        const float fG = (float)a.G;
        constexpr int EU = (TE / NW) >= 8 ? 8 : ((TE / NW) >= 4 ? 4 : ((TE / NW) >= 2 ? 2 : 1));
        auto pair = [&](int e, int j, float2 pj) {
            if (j == sg[2 * TE + e]) return;
            if (pj.x != pj.x || pj.y != pj.y) { pj.x = -500.0f; pj.y = -500.0f; }
            const float ox = __fadd_rn(__fdiv_rn(__fsub_rn(pj.x, sp[e]), a.cell), a.half_x);
            const float oy = __fadd_rn(__fdiv_rn(__fsub_rn(pj.y, sp[TE + e]), a.cell), a.half_y);
            const bool inr = !(ox < 0.0f) && !(ox >= fG) && !(oy < 0.0f) && !(oy >= fG);
            const int cellid = inr ? ((int)ox * a.G + (int)oy) : 0;
            atomicMax(&wkey[e * a.ncell + cellid], 2 * j + (inr ? 1 : 0));
        };
        for (int e0 = wave * EU; e0 < TE; e0 += NW * EU) {
            float2 pj[EU];
#pragma unroll
            for (int u = 0; u < EU; ++u)
                if (lane < sg[TE + e0 + u]) pj[u] = reinterpret_cast<const float2 *>(a.obs2)[sg[e0 + u] + lane];
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                const int e = e0 + u, lo = sg[e], ns = sg[TE + e], pad = sg[3 * TE + e];
                if (lane < ns) pair(e, lane, pj[u]);
                for (int j = lane + 64; j < ns; j += 64) pair(e, j, reinterpret_cast<const float2 *>(a.obs2)[lo + j]);
                if (ns < pad && lane == 0) atomicMax(&wkey[e * a.ncell], 2 * (pad - 1));
            }
        }
        __syncthreads();
    }
    if constexpr (!FG) __syncthreads();                                          // socc zeroed
    // winner tile, transposed ([cell][ego]: a cell's hits are one ballot), and the cells that have a hit at all
    for (int e = wave; e < TE; e += NW) {
        const int row = row0 + e;
        for (int c0 = 0; c0 < a.ncell; c0 += 256) {                       // four independent 64-cell chunks in flight
            int kk[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * 64 + lane;
                kk[u] = -1;
                if (c < a.ncell && row < a.M) {
                    if constexpr (FG) { const int k = wkey[e * a.ncell + c]; kk[u] = (k >= 0 && (k & 1)) ? (k >> 1) : -1; }
                    else kk[u] = a.winners[(size_t)row * a.ncell + c];
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = c0 + u * 64 + lane;
                if (c >= a.ncell) continue;
                if constexpr (FG) if (a.winners_out && ob == 0 && row < a.M) a.winners_out[(size_t)row * a.ncell + c] = (int16_t)kk[u];
                wl[c * WLS + e] = (WT)kk[u];
                if (kk[u] >= 0) socc[c] = 1;
            }
        }
    }
    int rb;
    if constexpr (FG) rb = sg[lane & (TE - 1)];
    else rb = a.row_base[min(row0 + (lane & (TE - 1)), a.M - 1)];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < TE; ++e) accl[e * OB] = 0.0f;      // own column of own copy: no barrier needed before the cell loop
    asm volatile("" : "+v"(rb));    // landed here: a compiler-placed vmcnt(0) inside the cell loop would drain the weight prefetch

    // Weight loads of the lean path are inline asm (saddr form: scalar base of the (cell, channel) row + this lane's
    // 32-bit byte offset) with MANUAL vmcnt waits: the compiler's own placement serialised the two weight sets (64-bit
    // VALU address chains in registers of the set still in flight => vmcnt(0) before every issue).  Every load_w issues
    // exactly C loads, so "this set has landed, the younger set may still be in flight" is s_waitcnt vmcnt(C).
    constexpr bool ASMW = SASM && C <= 16;
    const unsigned vo = ocu * 4u;
    auto load_w = [&](float (&w)[C], int c) {
        const float *wb = a.Wp + (size_t)c * C * a.N1;
        if constexpr (TNP_ABL(1)) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch) w[ch] = 1.0f + ch;
        } else if constexpr (ASMW) {
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
                asm volatile("global_load_dword %0, %1, %2" : "=v"(w[ch]) : "v"(vo), "s"(wb + (size_t)ch * a.N1) : "memory");
        } else {
#pragma unroll
            for (int ch = 0; ch < C; ++ch) w[ch] = (wb + (size_t)ch * a.N1)[ocu];
        }
    };
    auto wait_w = [&](float (&w)[C]) {
        if constexpr (TNP_ABL(1)) {
        } else if constexpr (ASMW) {
            static_assert(C == 4 || C == 8 || C == 16, "vmcnt immediate");
            if constexpr (C == 16)
                asm volatile("s_waitcnt vmcnt(16)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]),
                             "+v"(w[7]), "+v"(w[8]), "+v"(w[9]), "+v"(w[10]), "+v"(w[11]), "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15]));
            else if constexpr (C == 8)
                asm volatile("s_waitcnt vmcnt(8)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
            else
                asm volatile("s_waitcnt vmcnt(4)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
        } else {
#pragma unroll
            for (int ch = 0; ch < C; ++ch) asm("" : "+v"(w[ch]));
        }
    };
    auto process = [&](float (&w)[C], int c) {
        const int wv = lane < TE ? (int)wl[c * WLS + lane] : -1;
        unsigned long long mask = __ballot(wv >= 0);
        // wait for THIS cell's weights only: the loads of the next cell (issued unconditionally by the caller, so that
        // their number is known) stay in flight behind them -- vmcnt counts in order
        wait_w(w);
        if constexpr (TNP_ABL(2)) { if (mask == 1234567ull) accl[0] = w[0] + w[C - 1]; return; }
        if constexpr (SASM && C <= 16) {
            // lean path: the byte offset of every lane's neighbour row is one VALU op per cell; a hit then costs
            // ff1 + bit clear + one readlane + one SMEM load (SGPR offset) + LDS read-modify-write + C FMAs
            const unsigned off = __umul24((unsigned)(rb + wv), (unsigned)(a.ldv * 4));   // rows, row bytes < 2^24: full-rate multiply
            auto one = [&](int b, typename SRow<C>::type &ev, float &av) {
                if constexpr (TNP_ABL(4)) asm volatile("" : "=s"(ev) : "s"(__builtin_amdgcn_readlane((int)off, b)));
                else sload_row<C>(ev, a.enc, (unsigned)__builtin_amdgcn_readlane((int)off, b));
                if constexpr (TNP_ABL(8)) av = 0.0f; else av = accl[b * OB];
            };
            // even / odd channels accumulate in the two halves of one packed register: C/2 v_pk_fma_f32 with the
            // weight pair (w[2k], w[2k+1]) and the SGPR pair (e[2k], e[2k+1]) instead of C v_fmac_f32
            typedef float f2 __attribute__((ext_vector_type(2)));
            auto fin = [&](int b, const typename SRow<C>::type &ev, float av) {
                f2 p = {av, 0.0f};
#pragma unroll
                for (int k = 0; k < C / 2; ++k) {
                    const f2 wk = {w[2 * k], w[2 * k + 1]};
                    const f2 ek = {ev[2 * k], ev[2 * k + 1]};
                    p = __builtin_elementwise_fma(wk, ek, p);
                }
                if constexpr (TNP_ABL(8)) { if (p.x + p.y == 1.2345e33f) accl[b * OB] = p.x; } else accl[b * OB] = p.x + p.y;
            };
            {
                while (mask & (mask - 1ull)) {                      // at least two hits left: both in flight before the wait
                    const int b0 = pop_bit(mask);
                    const int b1 = pop_bit(mask);
                    typename SRow<C>::type e0, e1;
                    float a0, a1;
                    one(b0, e0, a0);
                    one(b1, e1, a1);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(e0), "+s"(e1));
                    fin(b0, e0, a0);
                    fin(b1, e1, a1);
                }
            }
            while (mask) {
                const int b0 = pop_bit(mask);
                typename SRow<C>::type e0;
                float a0;
                one(b0, e0, a0);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(e0));
                fin(b0, e0, a0);
            }
            return;
        }
        while (mask) {
            int eg[U];
            const float *ep[U];
            bool ok[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                ok[u] = mask != 0ull;
                const int b = ok[u] ? (__ffsll((long long)mask) - 1) : 0;
                if (ok[u]) mask &= mask - 1ull;
                const int wj = __builtin_amdgcn_readlane(wv, b);
                const int base = __builtin_amdgcn_readlane(rb, b);
                const int j = ok[u] ? (base + wj) : 0;
                eg[u] = b;
                ep[u] = a.enc + (size_t)j * a.ldv;
            }
            float av[U];
#pragma unroll
            for (int u = 0; u < U; ++u) av[u] = accl[eg[u] * OB];
#pragma unroll
            for (int u = 0; u < U; ++u) {
#pragma unroll
                for (int ch = 0; ch < C; ++ch) av[u] = fmaf(w[ch], ep[u][ch], av[u]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (ok[u]) accl[eg[u] * OB] = av[u];
        }
    };
    // Occupancy of this wave group's cells (cell q + NQ*k <-> bit k): only cells with at least one hit in the tile are
    // visited, and only their weights are loaded.
    const int nk = (a.ncell - q + NQ - 1) / NQ;                      // <= 128 (ncell <= 440 + NQ)
    unsigned long long occ[2];
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const int k = hh * 64 + lane;
        occ[hh] = __ballot(k < nk && socc[q + NQ * (k < nk ? k : 0)] != 0);
    }
    auto pop = [&]() -> int {                                         // next occupied cell of the group, or -1
        if (occ[0]) { const int k = __ffsll((long long)occ[0]) - 1; occ[0] &= occ[0] - 1ull; return q + NQ * k; }
        if (occ[1]) { const int k = __ffsll((long long)occ[1]) - 1; occ[1] &= occ[1] - 1ull; return q + NQ * (64 + k); }
        return -1;
    };
    // Two weight sets: the next occupied cell's C weights are in flight while the current cell's hits are processed.
    // Every load_w is unconditional (past the last occupied cell it re-reads the last one): only then does the compiler
    // know how many loads are younger than the set it waits for and emit vmcnt(C) instead of vmcnt(0).
    float wA[C], wB[C];
    int ca = pop();
    if constexpr (TNP_ABL(16)) ca = -1;
    if (ca >= 0) {
        load_w(wA, ca);
        while (true) {
            const int cb = pop();
            load_w(wB, cb >= 0 ? cb : ca);
            process(wA, ca);
            if (cb < 0) break;
            ca = pop();
            load_w(wA, ca >= 0 ? ca : cb);
            process(wB, cb);
            if (ca < 0) break;
        }
        if constexpr (ASMW && !TNP_ABL(1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    for (int g = tid; g < TE * OB / 4; g += NTH) {
        const int e = g / (OB / 4), c4 = (g - e * (OB / 4)) * 4;
        const int row = row0 + e, oo = ob * OB + c4;
        if (row >= a.M || oo >= a.N1) continue;
        float4 v = *reinterpret_cast<const float4 *>(acc + (size_t)e * OB + c4);
#pragma unroll
        for (int qq = 1; qq < NQ; ++qq) {
            const float4 t = *reinterpret_cast<const float4 *>(acc + ((size_t)qq * TE + e) * OB + c4);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        if (a.bias) {
            const float4 b = *reinterpret_cast<const float4 *>(a.bias + oo);
            v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
        }
        if (a.relu) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
        *reinterpret_cast<float4 *>(a.out + (size_t)row * a.ldo + oo) = v;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Register-accumulator kernel (default for C <= 16 and grids up to 417 cells): 64 egos x 128 columns per workgroup,
// 16 waves = 8 cell groups x 2 column sets.  Same hit discovery as the cell-split kernel (winner tile in LDS, one ballot
// per cell with lane <-> ego, scalar loads of the neighbour rows), but
//   * a wave's accumulators -- 64 egos x its lane's column -- live in VGPRs (two 32-element vectors): the ego of a hit
//     is wave-uniform, so acc[ego] is a register access through s_set_gpr_idx; no LDS read-modify-write per hit and no
//     accumulator copies in LDS, which is what capped the cell-split kernel's tile at 32 egos;
//   * a weight register set W'[c][.][o] serves the hits of 64 egos instead of 32: the L2 -> CU weight stream (1 GB per
//     launch at BASELINE config 2, ~45 us at the ~25 TB/s the L1 fill path sustains) is halved;
//   * weight loads are inline asm with manual vmcnt (see load_w) so that the next cell's set really is in flight;
//   * the winner keys are built transposed ([cell][ego]) and read by the cell loop as they are: no conversion pass;
//   * the 8 cell groups' partial sums are combined once at the end through LDS, 32 egos per round, in fixed order
//     (group 0 + 1 + ... + 7, then bias and activation): deterministic, independent of the ego's position in the tile.
// Measured (tools/experiments/sparse_ablate.hip, config-2 crowd, 11.7 hits per ego): 32.5 us with quad-major weights (39 us
// with the cell-major copy) against 50 us for the cell-split kernel.  ABL: timing ablations for that harness (1 no weight
// loads, 2 no hits, 16 no cell loop, 32 no votes, 128 no epilogue, 256 shader-clock stamps); the library instantiates 0.
// ---------------------------------------------------------------------------------------------------------
constexpr int RA_TE = 64, RA_OB = 128, RA_NQ = 8, RA_NCS = RA_OB / 64, RA_RED = 32;
typedef float ra_f32x32 __attribute__((ext_vector_type(32)));

// (row stride of the transposed key table = TE + 1: odd, conflict-free both ways)
static size_t ra_smem_bytes(int ncell, int te = RA_TE, int ncs = RA_NCS) {
    const size_t keys = (((size_t)ncell * (te + 1) * 4 + 15) & ~(size_t)15) + (size_t)ncell * 4 + 6 * te * 4;
    const size_t red = (size_t)RA_NQ * (te < RA_RED ? te : RA_RED) * (64 * ncs) * 4;
    return keys > red ? keys : red;
}

// QW: weights in the quad-major layout W''[c][o / 64][ch / 4][o % 64][ch % 4] (one scalar base per cell and column block,
// one 16-byte load per lane and channel quad); else the cell-major W'[c][ch][o].
//
// SMALL batches (round 6): TE (egos per tile: 4 .. 64) and NCS (column sets: OB = 64 NCS) are template parameters.  At 2048
// tracks the 64 x 128 tile gives exactly one round of 256 workgroups; one scene (36 tracks) on that tile is 8 workgroups,
// each streaming 2 MB of weights through ONE CU's L1 (64 B / clk: 13 us) with 248 CUs idle.  A smaller tile has fewer hits
// per wave AND fewer occupied cells, i.e. less weight traffic per workgroup, and needs no partial sums (the cell split across
// workgroups of the fallback kernel would).  NQ = 8 and the cell -> group map stay, so an ego's sum has the same order on
// every tile: the result does not depend on the tile size, bit for bit.
template <int C, bool FG, bool QW TNP_ABL_TPARAM, int TE = RA_TE, int NCS = RA_NCS>
__global__ void __launch_bounds__(64 * RA_NQ * NCS) pool_embed_regacc_kernel(const SparseArgs a) {
    constexpr int OB = 64 * NCS, NQ = RA_NQ, NTH = 64 * NQ * NCS, NW = NTH / 64;
    constexpr int RED = TE < RA_RED ? TE : RA_RED;                                    // egos per epilogue round
    static_assert(TE == 4 || TE == 8 || TE == 16 || TE == 32 || TE == 64, "ego tile");
    extern __shared__ __attribute__((aligned(16))) float rsm[];
    // keys [ncell][KS]: key of (cell, ego) = 2 * neighbour + in_range, -1 = empty -- written by LDS integer max ("last writer
    // in ascending j wins"), read by the cell loop as one row per cell (lane <-> ego)
    constexpr int KS = TE + 1;
    int *keyT = reinterpret_cast<int *>(rsm);
    int *socc = keyT + (((size_t)a.ncell * KS + 3) & ~(size_t)3);                   // [ncell] cell may have a hit in the tile
    int *sg = socc + a.ncell;                                                       // lo / ns / ki / pad [TE] each, then x / y
    float *sp = reinterpret_cast<float *>(sg + 4 * TE);
    float *red = rsm;                                                               // epilogue: [NQ][RED][OB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cs = wave % NCS, q = wave / NCS;
    const int ob = blockIdx.x % a.out_blocks, tile = blockIdx.x / a.out_blocks;     // blocks b, b+8, .. share an XCD
    const int row0 = tile * TE;
    const int o = ob * OB + cs * 64 + lane;
    const unsigned ocu = (unsigned)(o < a.N1 ? o : a.N1 - 1);
#ifdef TNP_EXPERIMENT_HOOKS
    // harness only (ABL & 256): shader-clock stamps per wave and phase into the buffer passed in place of `winners`
    long long *dbg = nullptr;
    if constexpr (TNP_ABL(256)) dbg = reinterpret_cast<long long *>(const_cast<int16_t *>(a.winners));
#define RA_T(k) do { if constexpr (TNP_ABL(256)) { if (lane == 0) dbg[(blockIdx.x * 16 + wave) * 8 + (k)] = (long long)__builtin_readcyclecounter(); } } while (0)
#else
#define RA_T(k) do { } while (0)
#endif
    RA_T(0);
#ifdef TNP_EXPERIMENT_HOOKS
    // harness only (ABL & 1024): what folding track_prepare into this prologue would add -- the tile's 64 state rows (H = 128)
    // and the 21 x 128 head / encoding weights through LDS, 64 x 21 dot products, the encodings written back to global
    // memory for this workgroup's own scalar row loads and waited for.  a.winners_out carries {h, W21, enc_out} here.
    if constexpr (TNP_ABL(1024)) {
        const float *const *pp = reinterpret_cast<const float *const *>(a.winners_out);
        const float *hsrc = pp[0], *wsrc = pp[1];
        float *eout = const_cast<float *>(pp[2]);
        float *hs = rsm, *ws21 = rsm + TE * 132;
        for (int i = tid; i < TE * 32; i += NTH) {                                  // 64 rows x 128 floats, float4 each
            const int r = i >> 5, k4 = i & 31;
            const float4 v = reinterpret_cast<const float4 *>(hsrc + (size_t)min(row0 + r, a.M - 1) * 128)[k4];
            *reinterpret_cast<float4 *>(hs + r * 132 + 4 * k4) = v;
        }
        for (int i = tid; i < 21 * 32; i += NTH) {
            const int o = i >> 5, k4 = i & 31;
            *reinterpret_cast<float4 *>(ws21 + o * 132 + 4 * k4) = reinterpret_cast<const float4 *>(wsrc + o * 128)[k4];
        }
        __syncthreads();
        for (int i = tid; i < TE * 21; i += NTH) {
            const int r = i / 21, o = i - r * 21;
            float acc0 = 0.0f, acc1 = 0.0f, acc2 = 0.0f, acc3 = 0.0f;
#pragma unroll 8
            for (int k = 0; k < 128; k += 4) {
                const float4 hv = *reinterpret_cast<const float4 *>(hs + r * 132 + k), wv = *reinterpret_cast<const float4 *>(ws21 + o * 132 + k);
                acc0 = fmaf(hv.x, wv.x, acc0); acc1 = fmaf(hv.y, wv.y, acc1); acc2 = fmaf(hv.z, wv.z, acc2); acc3 = fmaf(hv.w, wv.w, acc3);
            }
            if (row0 + r < a.M) eout[(size_t)(row0 + r) * 21 + o] = (acc0 + acc1) + (acc2 + acc3);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                             // the stores have reached L2 before the scalar loads
        __syncthreads();
    }
#endif

    for (int c = tid; c < a.ncell; c += NTH) socc[c] = 0;
    {
        const int4 m1 = {-1, -1, -1, -1};
        const int nk4 = (a.ncell * KS + 3) / 4;
        for (int idx = tid; idx < nk4; idx += NTH) reinterpret_cast<int4 *>(keyT)[idx] = m1;
    }
    if constexpr (FG) {
        // Winner keys straight from the positions (what grid_build_kernel computes, pool_grid.hip: the reference's exact fp32
        // cell arithmetic, integer max on key = 2*j + in_range, cell-0 clobber by out-of-range / absent / padded
        // neighbours).  The egos' scene geometry is staged first (one thread per ego); a wave then has the LDS reads and
        // the neighbour positions of its TE / 16 egos in flight together: two global round trips for the whole tile.
        if (tid < TE) {
            const int row = row0 + tid;
            int lo = 0, ns = 0, pad = 0;
            float2 pi = {-500.0f, -500.0f};
            if (row < a.M) {
                lo = a.row_base[row]; ns = a.row_end[row] - lo; pad = a.row_padded[row];
                pi = reinterpret_cast<const float2 *>(a.obs2)[row];
                if (pi.x != pi.x || pi.y != pi.y) { pi.x = -500.0f; pi.y = -500.0f; }
            }
            sg[tid] = lo; sg[TE + tid] = ns; sg[2 * TE + tid] = row - lo; sg[3 * TE + tid] = pad;
            sp[tid] = pi.x; sp[TE + tid] = pi.y;
        }
        __syncthreads();
        RA_T(1);
        const float fG = (float)a.G;
        constexpr int EU = TE >= NW ? TE / NW : 1;                  // egos per wave (tiles smaller than the wave count: waves >= TE sit out)
        static_assert(TE < NW || TE % NW == 0, "egos per wave");
        // branch-free votes: every lane computes its (ego, neighbour) pair, invalid ones are masked at the atomic only.  A
        // cell other than 0 that receives an in-range vote ends up with an in-range winner; cell 0 may still be clobbered
        // (then the cell loop finds no hit there: harmless).
        auto vote = [&](int e, int j, bool valid, float2 pj, float px, float py) {
            if (pj.x != pj.x || pj.y != pj.y) { pj.x = -500.0f; pj.y = -500.0f; }
            const float ox = __fadd_rn(__fdiv_rn(__fsub_rn(pj.x, px), a.cell), a.half_x);
            const float oy = __fadd_rn(__fdiv_rn(__fsub_rn(pj.y, py), a.cell), a.half_y);
            const bool inr = !(ox < 0.0f) && !(ox >= fG) && !(oy < 0.0f) && !(oy >= fG);
            const int cellid = inr ? ((int)ox * a.G + (int)oy) : 0;
            if (valid) {
                atomicMax(&keyT[cellid * KS + e], 2 * j + (inr ? 1 : 0));
                if (inr) socc[cellid] = 1;
            }
        };
        if (!TNP_ABL(32) && (TE >= NW || wave < TE)) {
            const int e0 = wave * EU;
            int lo[EU], ns[EU], ki[EU], pad[EU];
            float px[EU], py[EU];
            float2 pj[EU];
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                const int e = e0 + u;
                lo[u] = sg[e]; ns[u] = sg[TE + e]; ki[u] = sg[2 * TE + e]; pad[u] = sg[3 * TE + e];
                px[u] = sp[e]; py[u] = sp[TE + e];
            }
            bool small = true;                                       // wave-uniform: all of the wave's scenes have <= 32 tracks
#pragma unroll
            for (int u = 0; u < EU; ++u) small = small && ns[u] <= 32;
            if (EU % 2 == 0 && __builtin_amdgcn_readfirstlane((int)small)) {
                // two egos per pass: lanes 0-31 vote for ego e0 + 2 p, lanes 32-63 for ego e0 + 2 p + 1 (half the vector work
                // of the one-ego-per-pass form below for scenes that fill half a wave; the votes themselves are the same)
                const int hi = lane >> 5, j = lane & 31;
                float2 pq[EU >= 2 ? EU / 2 : 1];
#pragma unroll
                for (int p = 0; p < EU / 2; ++p) {
                    const int lo_ = hi ? lo[2 * p + 1] : lo[2 * p], ns_ = hi ? ns[2 * p + 1] : ns[2 * p];
                    pq[p] = reinterpret_cast<const float2 *>(a.obs2)[lo_ + (j < ns_ ? j : 0)];
                }
#pragma unroll
                for (int p = 0; p < EU / 2; ++p) {
                    const int ns_ = hi ? ns[2 * p + 1] : ns[2 * p], ki_ = hi ? ki[2 * p + 1] : ki[2 * p];
                    vote(e0 + 2 * p + hi, j, j < ns_ && j != ki_, pq[p], hi ? px[2 * p + 1] : px[2 * p], hi ? py[2 * p + 1] : py[2 * p]);
                }
#pragma unroll
                for (int u = 0; u < EU; ++u)
                    if (ns[u] < pad[u] && lane == 0) atomicMax(&keyT[e0 + u], 2 * (pad[u] - 1));
            } else {
#pragma unroll
            for (int u = 0; u < EU; ++u)
                pj[u] = reinterpret_cast<const float2 *>(a.obs2)[lo[u] + (lane < ns[u] ? lane : 0)];   // ns == 0: row 0, unused
#pragma unroll
            for (int u = 0; u < EU; ++u) vote(e0 + u, lane, lane < ns[u] && lane != ki[u], pj[u], px[u], py[u]);
#pragma unroll
            for (int u = 0; u < EU; ++u) {
                for (int j = lane + 64; j < ns[u]; j += 64)                          // scenes of more than 64 tracks
                    vote(e0 + u, j, j != ki[u], reinterpret_cast<const float2 *>(a.obs2)[lo[u] + j], px[u], py[u]);
                if (ns[u] < pad[u] && lane == 0) atomicMax(&keyT[e0 + u], 2 * (pad[u] - 1));
            }
            }
        }
        __syncthreads();
        RA_T(2);
        if (!TNP_ABL(1024) && a.winners_out && ob == 0) {                           // training: the winner table for the backward
            for (int e = wave; e < TE; e += NW) {
                const int row = row0 + e;
                if (row >= a.M) continue;
                for (int c = lane; c < a.ncell; c += 64) {
                    const int kq = keyT[c * KS + e];
                    a.winners_out[(size_t)row * a.ncell + c] = (kq >= 0 && (kq & 1)) ? (int16_t)(kq >> 1) : (int16_t)-1;
                }
            }
        }
    } else {
        __syncthreads();
        for (int e = wave; e < TE; e += NW) {                                       // keys from the given winner table
            const int row = row0 + e;
            if (row >= a.M) continue;
            for (int c = lane; c < a.ncell; c += 64) {
                const int w = a.winners[(size_t)row * a.ncell + c];
                if (w >= 0) { keyT[c * KS + e] = 2 * w + 1; socc[c] = 1; }
            }
        }
        __syncthreads();
    }
    int rb;
    if constexpr (FG) rb = sg[TE == 64 ? lane : (lane & (TE - 1))];
    else rb = a.row_base[min(row0 + (TE == 64 ? lane : (lane & (TE - 1))), a.M - 1)];
    asm volatile("" : "+v"(rb));
    RA_T(3);

    const int col2 = 2 * (tid & (OB / 2 - 1));                                      // this thread's column pair in the epilogue
    float2 bias2 = {0.0f, 0.0f};                                                    // fetched here: after the cell loop its latency would be exposed
    if (a.bias && ob * OB + col2 + 1 < a.N1) bias2 = *reinterpret_cast<const float2 *>(a.bias + ob * OB + col2);
    asm volatile("" : "+v"(bias2));                                                 // landed before the loop (see rb above)
    // egos 0..31 / 32..63 of the tile, this lane's column (tiles of <= 32 egos: accA alone, TE entries)
    constexpr int AE = TE < 32 ? TE : 32;
    typedef float ra_acc __attribute__((ext_vector_type(AE)));
    ra_acc accA, accB;
#pragma unroll
    for (int i = 0; i < AE; ++i) { accA[i] = 0.0f; accB[i] = 0.0f; }

    const unsigned vo = ocu * 4u;
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr bool QUAD = QW;
    const char *wq_base = reinterpret_cast<const char *>(a.Wp) +
                          (size_t)min(ob * NCS + cs, (a.N1 >> 6) - 1) * (C * 64 * 4);   // a column set past N1 (N1 % 128 == 64) re-reads the last block
    const unsigned wq_stride = (unsigned)(a.N1 >> 6) * (C * 64 * 4);
    struct WS { float f[QUAD ? 1 : C]; f4 q[QUAD ? C / 4 : 1]; };
    auto wch = [](const WS &w, int ch) -> float { if constexpr (QUAD) return w.q[ch >> 2][ch & 3]; else return w.f[ch]; };
    auto load_w = [&](WS &ws, int c) {
        float (&w)[QUAD ? 1 : C] = ws.f;
        if constexpr (TNP_ABL(1)) {
#pragma unroll
            for (int ch = 0; ch < (QUAD ? 1 : C); ++ch) w[ch] = 1.0f + ch;
        } else if constexpr (QUAD) {
            // one scalar base per cell (the column block's C x 64 weights are contiguous), channel quad = immediate offset.  The
            // scalar pipe is what bounds this loop: the base is wq_base (column block, set up once) + c * wq_stride, a 32-bit
            // product (the launcher checks ncell * N1 * C * 4 < 2^31) -- three scalar instructions per cell
            const char *wb = wq_base + __builtin_amdgcn_readfirstlane((unsigned)c * wq_stride);   // wave-uniform by construction; say so
            const unsigned vl = lane * 16u;
#pragma unroll
            for (int kq = 0; kq < C / 4; ++kq)
                asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(ws.q[kq]) : "v"(vl), "s"(wb), "i"(kq * 1024) : "memory");
        } else {
            const float *wb = a.Wp + (size_t)c * C * a.N1;
#pragma unroll
            for (int ch = 0; ch < C; ++ch)
                asm volatile("global_load_dword %0, %1, %2" : "=v"(w[ch]) : "v"(vo), "s"(wb + (size_t)ch * a.N1) : "memory");
        }
    };
    auto wait_w = [&](WS &ws) {
        float (&w)[QUAD ? 1 : C] = ws.f;
        static_assert(C == 4 || C == 8 || C == 16, "vmcnt immediate");
        if constexpr (TNP_ABL(1)) {
        } else if constexpr (QUAD && C == 16) {
            asm volatile("s_waitcnt vmcnt(4)" : "+v"(ws.q[0]), "+v"(ws.q[QUAD ? 1 : 0]), "+v"(ws.q[QUAD ? 2 : 0]), "+v"(ws.q[QUAD ? 3 : 0]));
        } else if constexpr (QUAD && C == 8) {
            asm volatile("s_waitcnt vmcnt(2)" : "+v"(ws.q[0]), "+v"(ws.q[QUAD ? 1 : 0]));
        } else if constexpr (QUAD) {
            asm volatile("s_waitcnt vmcnt(1)" : "+v"(ws.q[0]));
        }
        else if constexpr (C == 16)
            asm volatile("s_waitcnt vmcnt(16)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]),
                         "+v"(w[7]), "+v"(w[8]), "+v"(w[9]), "+v"(w[10]), "+v"(w[11]), "+v"(w[12]), "+v"(w[13]), "+v"(w[14]), "+v"(w[15]));
        else if constexpr (C == 8)
            asm volatile("s_waitcnt vmcnt(8)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]), "+v"(w[4]), "+v"(w[5]), "+v"(w[6]), "+v"(w[7]));
        else
            asm volatile("s_waitcnt vmcnt(4)" : "+v"(w[0]), "+v"(w[1]), "+v"(w[2]), "+v"(w[3]));
    };
    typedef float f2 __attribute__((ext_vector_type(2)));
    // one hit: acc[ego] += sum_ch w[ch] * e[ch] (even / odd channels in the two halves of a packed register: C/2
    // v_pk_fma_f32).  The ego is wave-uniform, so acc[ego] is a relative register access which the compiler lowers to
    // s_set_gpr_idx_on / v_mov / s_set_gpr_idx_off around the read and around the write.  Keep this exact shape (read,
    // chain, write): with `acc[b] += t` the compiler moves the vectors to scratch memory (launch_pool_embed_sparse refuses
    // a build whose kernel has a private segment).  A hand-written single  s_set_gpr_idx_on b, SRC1|DST ; v_add_f32 acc0,
    // t, acc0 ; s_set_gpr_idx_off  on pinned registers is 2 % faster but corrupts ~1 launch in 150 on gfx950 (wrong sums in
    // a few egos of one wave, or a wild scalar load), whatever wait states surround it -- tools/experiments/README.md.
    auto fin = [&](const WS &w, const typename SRow<C>::type &ev, float av) -> float {
        f2 p = {av, 0.0f};
#pragma unroll
        for (int k = 0; k < C / 2; ++k) {
            const f2 wk = {wch(w, 2 * k), wch(w, 2 * k + 1)};
            const f2 ek = {ev[2 * k], ev[2 * k + 1]};
            p = __builtin_elementwise_fma(wk, ek, p);
        }
        return p.x + p.y;
    };
    // the hits of one half of the tile (32 egos, bits of `m`) against the accumulator vector of that half
    auto half = [&](const WS &w, ra_acc &acc, unsigned m, unsigned off, int lane0) {
        // (pops as s_ff1 + s_bitset0 and a hit counter for the loop test: `m &= m - 1` twice plus `m & (m - 1)` are seven
        // scalar instructions per pair of hits, this is six less one)
        int left = __popc(m);
        for (; left >= 2; left -= 2) {                                              // two hits: both rows in flight before the wait
            const int b0 = pop_bit32(m);
            const int b1 = pop_bit32(m);
            typename SRow<C>::type e0, e1;
            sload_row<C>(e0, a.enc, (unsigned)__builtin_amdgcn_readlane((int)off, lane0 + b0));
            sload_row<C>(e1, a.enc, (unsigned)__builtin_amdgcn_readlane((int)off, lane0 + b1));
            const float a0 = acc[b0], a1 = acc[b1];
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(e0), "+s"(e1));
            acc[b0] = fin(w, e0, a0);
            acc[b1] = fin(w, e1, a1);
        }
        if (left) {
            const int b0 = __builtin_ctz(m);
            typename SRow<C>::type e0;
            sload_row<C>(e0, a.enc, (unsigned)__builtin_amdgcn_readlane((int)off, lane0 + b0));
            const float a0 = acc[b0];
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(e0));
            acc[b0] = fin(w, e0, a0);
        }
    };
    auto process = [&](WS &w, int kq) {                                              // kq = keyT[cell][lane]: lane <-> ego of the tile
        if constexpr (TE < 64) kq = lane < TE ? kq : -1;                             // (lanes past the tile read the next cell's row)
        const int wv = (kq >= 0 && (kq & 1)) ? (kq >> 1) : -1;
        const unsigned long long mask = __ballot(wv >= 0);
        wait_w(w);
        if constexpr (TNP_ABL(2)) { if (mask == 1234567ull) accA[0] += wch(w, 0) + wch(w, C - 1); return; }
        const unsigned off = __umul24((unsigned)(rb + wv), (unsigned)(a.ldv * 4));
        half(w, accA, (unsigned)mask, off, 0);
        if constexpr (TE == 64) half(w, accB, (unsigned)(mask >> 32), off, 32);
    };
    // cells of this wave's group with a hit in the tile; group q takes cell NQ k + ((q - k) mod NQ) of every block k of NQ cells
    const int nk = (a.ncell + NQ - 1) / NQ;                                          // <= 64: ncell <= 512
    auto cell_of = [&](int k) { return NQ * k + ((q - k) & (NQ - 1)); };
    unsigned long long occ;
    {
        const int c = cell_of(lane);
        occ = __ballot(lane < nk && c < a.ncell && socc[c < a.ncell ? c : 0] != 0);
    }
    // Two weight sets: the next occupied cell's weights are in flight while the current cell's hits are processed; every
    // load_w is unconditional (past the last occupied cell it re-reads the last one) so that vmcnt(C) is exact.  The visit
    // count is known up front, so the loop carries one counter and pops without tests (scalar instructions per visited cell
    // are what this loop pays for: 28 -> 17).
    WS wA, wB;
    if constexpr (TNP_ABL(16)) occ = 0ull;
    int left = __popcll(occ);                                                        // cells still to visit
    if (left > 0) {
        int ca = cell_of(pop_bit(occ)), cb;
        load_w(wA, ca);
        int ka = keyT[ca * KS + lane], kb;                                           // a cell's key row is read one cell ahead as well
        while (true) {
            cb = ca;
            if (left > 1) cb = cell_of(pop_bit(occ));
            load_w(wB, cb);
            kb = keyT[cb * KS + lane];
            process(wA, ka);
            if (--left == 0) break;
            ca = cb;
            if (left > 1) ca = cell_of(pop_bit(occ));
            load_w(wA, ca);
            ka = keyT[ca * KS + lane];
            process(wB, kb);
            if (--left == 0) break;
        }
        if constexpr (!TNP_ABL(1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }

    RA_T(4);
    // ---- the 8 cell groups' partial sums, 32 egos per round: every wave leaves its partials in LDS, then wave w sums the
    //      copies of egos w and w + 16 of the round in fixed order (group 0 + 1 + ... + 7), bias + activation, coalesced rows
    if constexpr (TNP_ABL(128)) { if (accA[3] + accB[5] == 1.234e30f) a.out[tid] = 0.0f; return; }
    // (thread t takes the column pairs t, t + NTH, ... of the round's RED x OB / 2: ego (t + NTH h) / (OB / 2), the same pair every time)
    constexpr int ITEMS = RED * (OB / 2);
#pragma unroll
    for (int r = 0; r < TE / RED; ++r) {
        __syncthreads();                                                            // prologue data / previous round no longer read
#pragma unroll
        for (int e = 0; e < RED; ++e)
            red[(q * RED + e) * OB + cs * 64 + lane] = r == 0 ? accA[e] : accB[e];
        __syncthreads();
#pragma unroll
        for (int h = 0; h < (ITEMS + NTH - 1) / NTH; ++h) {
            const int e = (tid + NTH * h) / (OB / 2);
            if (ITEMS % NTH != 0 && e >= RED) break;
            float2 v = *reinterpret_cast<const float2 *>(red + (size_t)e * OB + col2);
#pragma unroll
            for (int qq = 1; qq < NQ; ++qq) {
                const float2 t = *reinterpret_cast<const float2 *>(red + ((size_t)qq * RED + e) * OB + col2);
                v.x += t.x; v.y += t.y;
            }
            v.x += bias2.x; v.y += bias2.y;
            if (a.relu) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); }
            const int row = row0 + RED * r + e;
            if (row < a.M && ob * OB + col2 + 1 < a.N1) *reinterpret_cast<float2 *>(a.out + (size_t)row * a.ldo + ob * OB + col2) = v;
        }
    }
    RA_T(5);
#undef RA_T
}

// workgroups a smaller tile must reach before the next smaller one is tried: about half a round (a workgroup of the smaller
// tile streams fewer weights, but all of them together stream more)
static long sparse_tile_min_workgroups() {
    if (tuning().sparse_min_wg > 0) return tuning().sparse_min_wg;
    static long cus = 0;
    if (cus == 0) {
        int dev = 0, cu = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cu <= 0) cu = 256;
        cus = cu;
    }
    return cus;
}

bool regacc_supported(int C, int ncell) { return (C == 4 || C == 8 || C == 16) && ncell <= 64 * RA_NQ && ra_smem_bytes(ncell) <= (size_t)160 * 1024; }

bool sparse_supported(int C, int N1, int ncell) {
    return (C == 4 || C == 8 || C == 16 || C == 32) && N1 >= 4 && (N1 % 4 == 0) && ncell >= 1;
}

// fallback plan (pool_embed_sparse_kernel): cell ranges across workgroups
void sparse_plan(int M, int N1, int ncell, int &S, int &cps, int &ego_tiles, int &out_blocks) {
    ego_tiles = (M + SP_TE - 1) / SP_TE;
    out_blocks = (N1 + SP_OB - 1) / SP_OB;
    S = 1;
    while (S < 8 && ((long)ego_tiles * out_blocks * S < 200 || (ncell + S - 1) / S > 112) && S < ncell) S *= 2;
    cps = (ncell + S - 1) / S;
}

// the cell-split kernel needs no workspace; the fallback needs its partial sums
size_t sparse_partial_bytes(int M, int N1, int ncell) {
    if (ncell <= TL_MAXCELL_LDS) return 0;
    int S, cps, et, obk;
    sparse_plan(M, N1, ncell, S, cps, et, obk);
    return S > 1 ? (size_t)S * M * N1 * sizeof(float) : 0;
}

bool sparse_fuses_grid(int ncell, int n_max) { return ncell <= TL_MAXCELL_LDS && n_max <= 32767; }
bool sparse_uses_regacc(int M, int ldv, int C, int ncell) {
    return M > 0 && (size_t)M * ldv * sizeof(float) < ((size_t)1 << 32) && regacc_supported(C, ncell);
}

int launch_pool_embed_sparse(const int16_t *winners, const float *enc, int ldv, const int32_t *row_base,
                             const float *Wp, const float *bias, int M, int ncell, int C, int N1, int relu,
                             float *out, int ldo, float *partial, hipStream_t s, const SparseGridFuse *fg, const float *Wq) {
    if (M <= 0) return 0;
    if (!sparse_supported(C, N1, ncell)) TNP_FAIL(-1, "sparse pooling embedding: unsupported C=%d N1=%d", C, N1);
    SparseArgs a;
    a.winners = winners; a.enc = enc; a.ldv = ldv; a.row_base = row_base; a.Wp = Wp; a.bias = bias;
    a.M = M; a.ncell = ncell; a.C = C; a.N1 = N1; a.relu = relu; a.ldo = ldo;
    a.obs2 = nullptr; a.row_end = nullptr; a.row_padded = nullptr; a.G = 0; a.cell = 1.0f; a.half_x = a.half_y = 0.0f; a.winners_out = nullptr;
    const bool lean_rows = (size_t)M * ldv * sizeof(float) < ((size_t)1 << 32);   // 32-bit byte offsets of the neighbour rows
    if (lean_rows && regacc_supported(C, ncell)) {                                // register accumulators, 64-ego tiles
        a.out = out;
        a.S = 1; a.cps = ncell; a.ego_tiles = (M + RA_TE - 1) / RA_TE; a.out_blocks = (N1 + RA_OB - 1) / RA_OB;
        if (fg) {
            a.obs2 = fg->obs2; a.row_end = fg->row_end; a.row_padded = fg->row_padded; a.G = fg->G;
            a.cell = fg->cell; a.half_x = fg->half_x; a.half_y = fg->half_y; a.winners_out = fg->winners_out;
        }
        const bool quad = Wq != nullptr && fg != nullptr && N1 % 64 == 0 && C % 4 == 0 &&
                          (size_t)ncell * N1 * C * 4 < ((size_t)1 << 31);   // 32-bit byte offsets of the cells' weight blocks
        if (quad) a.Wp = Wq;
        // Tile: 64 egos x 128 columns when that fills the chip (2048 tracks x 1024 columns = 256 workgroups), else the largest
        // tile that still gives about one workgroup per CU (quad-major weights only) -- see the note in front of the kernel.
        // Tuning::sparse_te / sparse_ncs (TNP_SPARSE_TILE="te,ncs", tnp_tuning_set) pin one: tests, tools/diag/small_step_probe.py.
        int te = RA_TE, ncs = RA_NCS;
        if (quad) {
            static const int tiles[][2] = {{64, 2}, {32, 2}, {32, 1}, {16, 1}, {8, 1}, {4, 1}};
            const int pin_te = tuning().sparse_te, pin_ncs = tuning().sparse_ncs;
            const long want = sparse_tile_min_workgroups();
            for (const auto &t : tiles) {
                te = t[0]; ncs = t[1];
                if (pin_te > 0 ? (te == pin_te && ncs == pin_ncs) : ((long)((M + te - 1) / te) * ((N1 + 64 * ncs - 1) / (64 * ncs)) >= want)) break;
            }
            if (pin_te > 0 && !(te == pin_te && ncs == pin_ncs)) TNP_FAIL(-1, "TNP_SPARSE_TILE=%d,%d is not a tile of the sparse first layer", pin_te, pin_ncs);
        }
        a.ego_tiles = (M + te - 1) / te; a.out_blocks = (N1 + 64 * ncs - 1) / (64 * ncs);
        const size_t rsmem = ra_smem_bytes(ncell, te, ncs) + 256;   // (+ 256: lanes past a small tile read behind the last key row)
        const int rblocks = a.ego_tiles * a.out_blocks;
        hipEvent_t pe0 = nullptr, pe1 = nullptr;
        const bool pev = take_dispatch_events(&pe0, &pe1);   // profiling: events of the dispatch itself (tnp_internal.h)
#define RA_LAUNCH(CC, FGB, QB, TEE, NCC) { static bool set = false; auto kern = pool_embed_regacc_kernel<CC, FGB, QB TNP_ABL_ARG, TEE, NCC>; \
        if (!set) { hipFuncAttributes fa; \
        TNP_HIP(hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(kern))); \
        if (fa.localSizeBytes != 0) TNP_FAIL(-3, "pool_embed_regacc_kernel: accumulators left the register file (%zu bytes of scratch)", (size_t)fa.localSizeBytes); \
        TNP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
        if (pev) hipExtLaunchKernelGGL(kern, dim3(rblocks), dim3(64 * RA_NQ * NCC), rsmem, s, pe0, pe1, 0, a); \
        else hipLaunchKernelGGL(kern, dim3(rblocks), dim3(64 * RA_NQ * NCC), rsmem, s, a); }
#define RA_SWITCH(CC) { if (quad) { \
            if (te == 64) RA_LAUNCH(CC, true, true, 64, 2) else if (te == 32 && ncs == 2) RA_LAUNCH(CC, true, true, 32, 2) \
            else if (te == 32) RA_LAUNCH(CC, true, true, 32, 1) else if (te == 16) RA_LAUNCH(CC, true, true, 16, 1) \
            else if (te == 8) RA_LAUNCH(CC, true, true, 8, 1) else RA_LAUNCH(CC, true, true, 4, 1) } \
        else if (fg) RA_LAUNCH(CC, true, false, 64, 2) else RA_LAUNCH(CC, false, false, 64, 2) }
        if (C == 4) RA_SWITCH(4) else if (C == 8) RA_SWITCH(8) else RA_SWITCH(16)
        TNP_HIP(hipGetLastError());
        return 0;
    }
    if (ncell <= TL_MAXCELL_LDS) {
        a.out = out;
        a.S = 1; a.cps = ncell; a.ego_tiles = (M + TL_TE - 1) / TL_TE; a.out_blocks = (N1 + TL_OB - 1) / TL_OB;
        const size_t csmem = (size_t)TL_NQ * TL_TE * TL_OB * 4 + (((size_t)ncell * (TL_TE + 2) * 2 + 15) & ~(size_t)15) + (size_t)ncell * 4;
        const int cblocks = a.ego_tiles * a.out_blocks;
        // the lean path addresses neighbour rows with a 32-bit byte offset
        const bool lean = (size_t)M * ldv * sizeof(float) < ((size_t)1 << 32);
        if (fg && !lean) TNP_FAIL(-1, "sparse pooling embedding: fused grid build needs 32-bit row offsets");
        if (fg) {   // winner tile computed in the kernel: no grid kernel, no winner table
            a.obs2 = fg->obs2; a.row_end = fg->row_end; a.row_padded = fg->row_padded; a.G = fg->G;
            a.cell = fg->cell; a.half_x = fg->half_x; a.half_y = fg->half_y; a.winners_out = fg->winners_out;
        }
#define CS_LAUNCH(CC, SA, FGB) { static bool set = false; if (!set) { TNP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>( \
        pool_embed_cellsplit_kernel<CC, SA, FGB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
        hipLaunchKernelGGL((pool_embed_cellsplit_kernel<CC, SA, FGB>), dim3(cblocks), dim3(1024), csmem, s, a); }
#define CS_SWITCH(CC) { if (fg) CS_LAUNCH(CC, true, true) else if (lean) CS_LAUNCH(CC, true, false) else CS_LAUNCH(CC, false, false) }
        if (C == 4) CS_SWITCH(4) else if (C == 8) CS_SWITCH(8) else if (C == 16) CS_SWITCH(16) else CS_SWITCH(32)
        TNP_HIP(hipGetLastError());
        return 0;
    }
    if (fg) TNP_FAIL(-1, "sparse pooling embedding: fused grid build needs a grid of at most %d cells", TL_MAXCELL_LDS);
    // ---- fallback for grids too large for the LDS winner tile: cell ranges across workgroups + reduce ----
    sparse_plan(M, N1, ncell, a.S, a.cps, a.ego_tiles, a.out_blocks);
    if (a.cps > 120) TNP_FAIL(-1, "sparse pooling embedding: %d cells per split exceed the LDS winner tile", a.cps);
    if (a.S > 1 && !partial) TNP_FAIL(-1, "sparse pooling embedding: partial workspace missing");
    a.out = (a.S > 1) ? partial : out;
    if (a.S > 1) a.ldo = N1;
    const size_t smem = (size_t)SP_TE * SP_OB * 4 + (((size_t)a.cps * (SP_TE + 2) * 2 + 15) & ~(size_t)15);
    const int blocks = a.ego_tiles * a.out_blocks * a.S;
#define SP_LAUNCH(CC) { static bool set = false; if (!set) { TNP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>( \
        pool_embed_sparse_kernel<CC, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
        hipLaunchKernelGGL((pool_embed_sparse_kernel<CC, 4>), dim3(blocks), dim3(1024), smem, s, a); }
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
