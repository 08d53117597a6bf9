// Weight-gradient GEMM of the training sweep (gfx950 only):   dW[Mo, No] = dY[K, Mo]^T @ X[K, No],  K = steps * tracks.
//
// The reference gets these from autograd (one small GEMM per step, accumulated); here every parameter has ONE
// contraction over the stacked steps (lstm/training.py), i.e. a small output (<= 512 x 1024) and a very long K
// (38 912 rows at config 2).  The forward GEMM kernel (gemm_f32_mfma.hip) wants K-contiguous operands and tiles M x N
// only, which left these launches on 80 of 256 CUs after two LDS-tiled transposes.  This kernel
// * reads both operands as they lie in HBM (row k of dY / X is contiguous): lane l of a wave loads
//   dY[k + l/32][i0 + l%32] -- exactly the v_mfma_f32_32x32x2_f32 operand layout, 128-byte coalesced, no LDS, no
//   barriers, no transposes;
// * splits K across workgroups (grid.y) so that ~2 workgroups per CU stream independent K ranges; every wave owns a
//   64 x 64 output block (four accumulators, one operand load per MFMA), a workgroup 128 x 128;
// * writes per-split partial tiles and reduces them in a fixed order (deterministic, no atomics); the bias gradient
//   (column sums of dY) rides along on the A operand.
// Exact fp32 products and fp32 accumulation, like the forward kernel.
#include "tnp_internal.h"

namespace tnp {

typedef float wg_f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_U = 8;   // k-pairs per unrolled iteration (16 rows of K)

// one workgroup: region `region` (128 x 128 outputs) of split `ks` of one contraction
__device__ __forceinline__ void wgrad_tn_block(const float *__restrict__ A, int lda, const float *__restrict__ B, int ldb, int Mo,
                                               int No, int K, int kchunk, float *__restrict__ part,
                                               float *__restrict__ bias_part, int region, int ks) {
    const int regions_n = (No + 127) >> 7;
    const int rm = region / regions_n, rn = region - rm * regions_n;
    const int lane = threadIdx.x & 63, li = lane & 31, kh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);         // wave-uniform tile coordinates and edge tests
    const int i0 = rm * 128 + (wave & 1) * 64, j0 = rn * 128 + (wave >> 1) * 64;
    if (i0 >= Mo || j0 >= No) return;           // no barriers in this kernel: idle waves just leave
    const bool m1 = i0 + 32 < Mo, n1 = j0 + 32 < No;
    const int kb = ks * kchunk, ke = min(K, kb + kchunk);
    const int ia0 = min(i0 + li, Mo - 1), ia1 = min(i0 + 32 + li, Mo - 1);
    const int jb0 = min(j0 + li, No - 1), jb1 = min(j0 + 32 + li, No - 1);
    wg_f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc00[r] = 0.f; acc01[r] = 0.f; acc10[r] = 0.f; acc11[r] = 0.f; }
    const bool want_bias = bias_part != nullptr && rn == 0 && (wave >> 1) == 0;
    float bs0 = 0.f, bs1 = 0.f;
    auto finish = [&]() {
        float *P = part + (size_t)ks * Mo * No;
        auto store = [&](const wg_f32x16 &acc, int ib, int jb) {
            const int col = jb + li;
            if (col >= No) return;
    #pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = ib + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (row < Mo) P[(size_t)row * No + col] = acc[r];
            }
        };
        store(acc00, i0, j0);
        if (n1) store(acc01, i0, j0 + 32);
        if (m1) store(acc10, i0 + 32, j0);
        if (m1 && n1) store(acc11, i0 + 32, j0 + 32);
        if (want_bias) {
            bs0 += __shfl_xor(bs0, 32);
            bs1 += __shfl_xor(bs1, 32);
            float *bp = bias_part + (size_t)ks * Mo;
            if (kh == 0) {
                if (i0 + li < Mo) bp[i0 + li] = bs0;
                if (m1 && i0 + 32 + li < Mo) bp[i0 + 32 + li] = bs1;
            }
        }
    };
    if (m1 && n1 && ((ke - kb) % (2 * WG_U)) == 0) {
        // Whole stages of a block with both halves present (every split but a ragged last one): two operand stages of 16
        // rows of K, the 32 loads of stage s + 1 issued before the 32 MFMAs of stage s.  The loads are inline asm with a
        // hand-placed s_waitcnt: written as plain C++ the compiler (a) hoists the bias adds of a stage above its MFMAs, which
        // drags the wait for ALL outstanding loads in front of them, and (b) parks the accumulators in VGPRs across the
        // back edge (64 v_accvgpr_write per trip) -- 574 us for the grouped launch of config 2 without any prefetch (every
        // 16 rows of K paid a memory round trip in front of 2048 cycles of matrix work), 450 us with the C++ prefetch.
        // Addresses: scalar base of row k (wave-uniform) + a per-lane byte offset that never changes.
        // Same order of the additions as the edge loop below: bit-identical results.
        const unsigned oa0 = (unsigned)(kh * lda + ia0) * 4u, oa1 = (unsigned)(kh * lda + ia1) * 4u;
        const unsigned ob0 = (unsigned)(kh * ldb + jb0) * 4u, ob1 = (unsigned)(kh * ldb + jb1) * 4u;
        const float *ab = A + (size_t)kb * lda, *bb = B + (size_t)kb * ldb;
        float a0[2][WG_U], a1[2][WG_U], b0[2][WG_U], b1[2][WG_U];
#define WG_LD(dst, off, base) asm volatile("global_load_dword %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base) : "memory")
        int rem = (ke - kb) / (2 * WG_U);                     // stages not yet requested (<= 0: look-ahead past the end)
        auto load = [&](int st) {                             // past the last stage: the last one again (valid memory, never used)
#pragma unroll
            for (int u = 0; u < WG_U; ++u) {
                const float *pa = ab + (size_t)(2 * u) * lda, *pb = bb + (size_t)(2 * u) * ldb;
                WG_LD(a0[st][u], oa0, pa); WG_LD(b0[st][u], ob0, pb);
                WG_LD(a1[st][u], oa1, pa); WG_LD(b1[st][u], ob1, pb);
            }
            --rem;                                            // ab / bb: the next stage to request, if there is one
            ab += rem > 0 ? (size_t)(2 * WG_U) * lda : 0; bb += rem > 0 ? (size_t)(2 * WG_U) * ldb : 0;
        };
#undef WG_LD
        static_assert(WG_U == 8, "the wait statements name eight registers per operand");
#define WG_R8(x) "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
        auto wait = [&](int st, bool more) {                 // stage st has landed (`more`: the next stage's 32 loads stay in flight)
            if (more) asm volatile("s_waitcnt vmcnt(32)" : WG_R8(a0[st]), WG_R8(b0[st]));
            else asm volatile("s_waitcnt vmcnt(0)" : WG_R8(a0[st]), WG_R8(b0[st]));
            asm volatile("" : WG_R8(a1[st]), WG_R8(b1[st]));
        };
#undef WG_R8
        auto mma = [&](int st) {
#pragma unroll
            for (int u = 0; u < WG_U; ++u) {
                acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[st][u], b0[st][u], acc00, 0, 0, 0);
                acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[st][u], b1[st][u], acc01, 0, 0, 0);
                acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[st][u], b0[st][u], acc10, 0, 0, 0);
                acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[st][u], b1[st][u], acc11, 0, 0, 0);
            }
            if (want_bias) {
#pragma unroll
                for (int u = 0; u < WG_U; ++u) { bs0 += a0[st][u]; bs1 += a1[st][u]; }
            }
        };
        // one back edge, no exit from the middle, no branch in the body (anything else and the accumulators get copied
        // between VGPRs and AGPRs around every stage, or a stage's registers are moved before their wait)
        const int stages = rem, pairs = stages >> 1;
        if (stages > 0) {
            load(0);
            for (int p = 0; p < pairs; ++p) {
                load(1);
                wait(0, true);
                mma(0);
                load(0);
                wait(1, true);
                mma(1);
            }
            wait(0, false);                                   // drains the look-ahead loads too
            if (stages & 1) mma(0);
        }
        finish();                       // (not shared with the edge path: a join would keep both paths' accumulators alive)
        return;
    }
    {
        for (int k = kb; k < ke; k += 2 * WG_U) {
            float a0[WG_U], a1[WG_U], b0[WG_U], b1[WG_U];
#pragma unroll
            for (int u = 0; u < WG_U; ++u) {
                const int kk = k + 2 * u + kh;
                const bool ok = kk < ke;
                const size_t row = (size_t)(ok ? kk : ke - 1);
                const float *ar = A + row * lda, *br = B + row * ldb;
                float va0 = ar[ia0], vb0 = br[jb0];
                float va1 = m1 ? ar[ia1] : 0.f, vb1 = n1 ? br[jb1] : 0.f;
                a0[u] = ok ? va0 : 0.f; a1[u] = ok ? va1 : 0.f;
                b0[u] = ok ? vb0 : 0.f; b1[u] = ok ? vb1 : 0.f;
            }
#pragma unroll
            for (int u = 0; u < WG_U; ++u) {
                acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b0[u], acc00, 0, 0, 0);
                if (n1) acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], b1[u], acc01, 0, 0, 0);
                if (m1) acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b0[u], acc10, 0, 0, 0);
                if (m1 && n1) acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], b1[u], acc11, 0, 0, 0);
                bs0 += a0[u]; bs1 += a1[u];
            }
        }
    }
    finish();
}

// A contraction WITHOUT a weight part (No == 0): column sums of dY only -- the bias gradient of a layer whose weight gradient is
// formed elsewhere (the sparse first embedding layer, lstm_bwd.hip).  It used to be an ATen `sum(0)` behind the sweep (a
// semaphore fill + a 21-38 us reduce launch on an otherwise idle chip); here its workgroups stream dY beside the matrix-pipe
// work of the grouped launch.  Region = 128 columns, thread t takes column t & 127 and every second row of the split
// (512-byte coalesced rows), eight loads in flight; the two halves are added through LDS, the splits by the reduce launch in
// split order: deterministic.
__device__ __forceinline__ void wgrad_bias_block(const float *__restrict__ A, int lda, int Mo, int K, int kchunk,
                                                 float *__restrict__ bias_part, int region, int ks) {
    __shared__ float half_sum[128];
    const int t = threadIdx.x, col = region * 128 + (t & 127), half = t >> 7;
    const int kb = ks * kchunk, ke = min(K, kb + kchunk);
    const int c = min(col, Mo - 1);
    float acc = 0.f;
    int k = kb + half;
    for (; k + 14 < ke; k += 16) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = A[(size_t)(k + 2 * u) * lda + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; k < ke; k += 2) acc += A[(size_t)k * lda + c];
    if (half) half_sum[t & 127] = acc;
    __syncthreads();
    if (!half && col < Mo) bias_part[(size_t)ks * Mo + col] = acc + half_sum[t];
}

__global__ void __launch_bounds__(256) wgrad_tn_kernel(const float *__restrict__ A, int lda, const float *__restrict__ B,
                                                       int ldb, int Mo, int No, int K, int kchunk,
                                                       float *__restrict__ part, float *__restrict__ bias_part) {
    wgrad_tn_block(A, lda, B, ldb, Mo, No, K, kchunk, part, bias_part, blockIdx.x, blockIdx.y);
}

// ---- grouped form: several contractions in ONE launch (+ one reduce launch) ------------------------------------------------
// An optimisation step has eight of these (second embedding layer, the two LSTM cells' W_ih / W_hh, output head, input
// embedding, social encoding); five are tiny (<= 64 x 128 outputs) and cost a ~27 us launch + a ~7 us reduce each for
// microseconds of matrix work.  Here every contraction keeps exactly the plan -- split count, chunk, summation order -- of
// its stand-alone launch (so the results are bit-identical to tnp_wgrad), and the workgroups of all of them fill the chip
// together.
#define WG_MAX_PROBLEMS 12
struct WgradGroup {
    const float *A[WG_MAX_PROBLEMS], *B[WG_MAX_PROBLEMS];
    float *part[WG_MAX_PROBLEMS], *bias_part[WG_MAX_PROBLEMS], *out[WG_MAX_PROBLEMS], *bout[WG_MAX_PROBLEMS];
    int lda[WG_MAX_PROBLEMS], ldb[WG_MAX_PROBLEMS], Mo[WG_MAX_PROBLEMS], No[WG_MAX_PROBLEMS], K[WG_MAX_PROBLEMS];
    int kchunk[WG_MAX_PROBLEMS], regions[WG_MAX_PROBLEMS], SK[WG_MAX_PROBLEMS], ldo[WG_MAX_PROBLEMS];
    int first_block[WG_MAX_PROBLEMS + 1];     // wgrad_tn_group_kernel: block ranges of problems tn_order[0], tn_order[1], ...
    int tn_order[WG_MAX_PROBLEMS];            // longest workgroups first (the launch ends with the short ones: no long tail)
    int first_rblock[WG_MAX_PROBLEMS + 1];    // wgrad_reduce_group_kernel
    int count;
};

__global__ void __launch_bounds__(256) wgrad_tn_group_kernel(const WgradGroup g) {
    int q = 0;
    while (q + 1 < g.count && (int)blockIdx.x >= g.first_block[q + 1]) ++q;
    const int local = (int)blockIdx.x - g.first_block[q], p = g.tn_order[q];
    const int region = local % g.regions[p], ks = local / g.regions[p];
    if (g.No[p] == 0) { wgrad_bias_block(g.A[p], g.lda[p], g.Mo[p], g.K[p], g.kchunk[p], g.bias_part[p], region, ks); return; }
    wgrad_tn_block(g.A[p], g.lda[p], g.B[p], g.ldb[p], g.Mo[p], g.No[p], g.K[p], g.kchunk[p], g.part[p], g.bias_part[p], region, ks);
}

// out[e] = sum over the splits of part[s][e], ascending s.  One launch covers the weight tile and (blocks past it) the bias
// column; eight partials are in flight before the first add (the adds stay in split order: bit-identical to a serial loop).
__device__ __forceinline__ void wgrad_reduce_block(const float *__restrict__ part, int SK, long n, int cols, float *__restrict__ out,
                                                   int ldo, const float *__restrict__ bpart, long nb, float *__restrict__ bout, long block) {
    const long nblk = (n + 255) / 256;
    long e = block * 256 + threadIdx.x;
    long cnt = n;
    const float *src = part;
    bool bias = false;
    if (block >= nblk) { e -= nblk * 256; cnt = nb; src = bpart; bias = true; }
    if (e >= cnt) return;
    float acc = src[e];
    constexpr int U = 8;
    int s = 1;
    for (; s + U <= SK; s += U) {
        float v[U];
#pragma unroll
        for (int i = 0; i < U; ++i) v[i] = src[(size_t)(s + i) * cnt + e];
#pragma unroll
        for (int i = 0; i < U; ++i) acc += v[i];
    }
    for (; s < SK; ++s) acc += src[(size_t)s * cnt + e];
    if (bias) bout[e] = acc;
    else out[(e / cols) * ldo + (e % cols)] = acc;
}

__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float *__restrict__ part, int SK, long n, int cols, float *__restrict__ out,
                                                           int ldo, const float *__restrict__ bpart, long nb, float *__restrict__ bout) {
    wgrad_reduce_block(part, SK, n, cols, out, ldo, bpart, nb, bout, (long)blockIdx.x);
}

__global__ void __launch_bounds__(256) wgrad_reduce_group_kernel(const WgradGroup g) {
    int p = 0;
    while (p + 1 < g.count && (int)blockIdx.x >= g.first_rblock[p + 1]) ++p;
    wgrad_reduce_block(g.part[p], g.SK[p], (long)g.Mo[p] * g.No[p], g.No[p], g.out[p], g.ldo[p], g.bias_part[p], (long)g.Mo[p], g.bout[p],
                       (long)((int)blockIdx.x - g.first_rblock[p]));
}

static void wgrad_plan(int Mo, int No, int K, int &regions, int &SK, int &kchunk) {
    regions = ((Mo + 127) / 128) * (No > 0 ? (No + 127) / 128 : 1);      // No == 0: column sums only (wgrad_bias_block)
    const int target = tuning().wgrad_target_wgs, min_rows = tuning().wgrad_min_rows;
    int want = (target + regions - 1) / regions;   // 512: ~2 workgroups per CU; in the grouped launch 256 - 512 measure the same
                                                   // (365 us), 768 / 1024 / 2048: 384 / 389 / 410 us (round 3 sweep run_r3w, git history)
    if (want > 128) want = 128;
    const int maxsk = (K + min_rows - 1) / min_rows;   // min_rows 128 (256 until round 6: batch_size 8, K ~ 5900: 100 + 11 -> 87 + 16 us)
    if (want > maxsk) want = maxsk;
    if (want < 1) want = 1;
    kchunk = (K + want - 1) / want;
    kchunk = (kchunk + 2 * WG_U - 1) / (2 * WG_U) * (2 * WG_U);
    SK = (K + kchunk - 1) / kchunk;
}

}  // namespace tnp

extern "C" TNP_API size_t tnp_wgrad_workspace_bytes(int Mo, int No, int K) {
    if (Mo <= 0 || No <= 0 || K <= 0) return 0;
    int regions, SK, kchunk;
    tnp::wgrad_plan(Mo, No, K, regions, SK, kchunk);
    return ((size_t)SK * Mo * No + (size_t)SK * Mo) * sizeof(float);
}

extern "C" TNP_API int tnp_wgrad(const float *dy, int ld_dy, const float *x, int ld_x, int K, int Mo, int No, float *dw, int ld_dw,
                                 float *dbias, void *workspace, size_t workspace_bytes, void *stream) {
    if (Mo <= 0 || No <= 0) return 0;
    if (K <= 0) TNP_FAIL(-1, "tnp_wgrad: K = %d", K);
    int regions, SK, kchunk;
    tnp::wgrad_plan(Mo, No, K, regions, SK, kchunk);
    const size_t need = ((size_t)SK * Mo * No + (size_t)SK * Mo) * sizeof(float);
    if (!workspace || workspace_bytes < need) TNP_FAIL(-1, "tnp_wgrad: workspace too small (need %zu bytes, got %zu)", need, workspace_bytes);
    float *part = (float *)workspace, *bpart = part + (size_t)SK * Mo * No;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(tnp::wgrad_tn_kernel, dim3(regions, SK), dim3(256), 0, s, dy, ld_dy, x, ld_x, Mo, No, K, kchunk, part,
                       dbias ? bpart : nullptr);
    TNP_HIP(hipGetLastError());
    const long n = (long)Mo * No;
    const unsigned rblocks = (unsigned)((n + 255) / 256) + (dbias ? (unsigned)((Mo + 255) / 256) : 0u);
    hipLaunchKernelGGL(tnp::wgrad_reduce_kernel, dim3(rblocks), dim3(256), 0, s, part, SK, n, No, dw, ld_dw,
                       dbias ? bpart : nullptr, (long)Mo, dbias);
    TNP_HIP(hipGetLastError());
    return 0;
}


static size_t wgrad_problem_bytes(int Mo, int No, int K) {
    int regions, SK, kchunk;
    tnp::wgrad_plan(Mo, No, K, regions, SK, kchunk);
    return (((size_t)SK * Mo * No + (size_t)SK * Mo) * sizeof(float) + 255) & ~(size_t)255;
}

extern "C" TNP_API size_t tnp_wgrad_grouped_workspace_bytes(const tnp_wgrad_problem *problems, int n) {
    size_t total = 0;
    for (int i = 0; i < n; ++i)
        if (problems[i].Mo > 0 && (problems[i].No > 0 || (problems[i].No == 0 && problems[i].dbias)) && problems[i].K > 0)
            total += wgrad_problem_bytes(problems[i].Mo, problems[i].No, problems[i].K);
    return total;
}

extern "C" TNP_API int tnp_wgrad_grouped(const tnp_wgrad_problem *problems, int n, void *workspace, size_t workspace_bytes,
                                         void *stream) {
    if (n <= 0) return 0;
    if (!problems) TNP_FAIL(-1, "tnp_wgrad_grouped: problems == NULL");
    const size_t need = tnp_wgrad_grouped_workspace_bytes(problems, n);
    if (!workspace || workspace_bytes < need) TNP_FAIL(-1, "tnp_wgrad_grouped: workspace too small (need %zu bytes, got %zu)", need, workspace_bytes);
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    int done = 0;
    while (done < n) {
        tnp::WgradGroup g;
        g.count = 0; g.first_block[0] = 0; g.first_rblock[0] = 0;
        for (; done < n && g.count < WG_MAX_PROBLEMS; ++done) {
            const tnp_wgrad_problem &q = problems[done];
            const bool bias_only = q.No == 0 && q.dbias != nullptr;
            if (q.Mo <= 0 || (q.No <= 0 && !bias_only)) continue;
            if (q.K <= 0 || !q.dy || (!bias_only && (!q.x || !q.dw)))
                TNP_FAIL(-1, "tnp_wgrad_grouped: problem %d: K = %d or a NULL pointer", done, q.K);
            const int c = g.count;
            int regions, SK, kchunk;
            tnp::wgrad_plan(q.Mo, q.No, q.K, regions, SK, kchunk);
            g.A[c] = q.dy; g.lda[c] = q.ld_dy; g.B[c] = q.x; g.ldb[c] = q.ld_x; g.Mo[c] = q.Mo; g.No[c] = q.No; g.K[c] = q.K;
            g.kchunk[c] = kchunk; g.regions[c] = regions; g.SK[c] = SK; g.out[c] = q.dw; g.ldo[c] = q.ld_dw; g.bout[c] = q.dbias;
            g.part[c] = (float *)ws;
            g.bias_part[c] = q.dbias ? g.part[c] + (size_t)SK * q.Mo * q.No : nullptr;
            ws += wgrad_problem_bytes(q.Mo, q.No, q.K);
            const long nel = (long)q.Mo * q.No;
            g.first_rblock[c + 1] = g.first_rblock[c] + (int)((nel + 255) / 256) + (q.dbias ? (q.Mo + 255) / 256 : 0);
            ++g.count;
        }
        if (g.count == 0) break;
        // block order of the contraction launch: problems by the estimated duration of one of their workgroups, longest
        // first (rows of K per split; x 3 for blocks that take the edge loop, which has no operand prefetch).  With the
        // queue order the five tiny contractions came last and their 50-us edge-loop workgroups ran on an empty chip
        // (1.2 of 2 waves per SIMD resident on average, profiles/archive/round3_q_pmc_train.md).  Plans, partial buffers and the
        // reduce launch are untouched: results do not depend on the order.
        {
            long cost[WG_MAX_PROBLEMS];
            for (int c = 0; c < g.count; ++c) {
                const bool edge = g.Mo[c] % 64 != 0 || g.No[c] % 64 != 0;
                cost[c] = g.No[c] == 0 ? (long)g.kchunk[c] / 8 : (long)g.kchunk[c] * (edge ? 3 : 1);
                g.tn_order[c] = c;
            }
            for (int a = 1; a < g.count; ++a)                       // insertion sort, stable: equal costs keep the queue order
                for (int b = a; b > 0 && cost[g.tn_order[b]] > cost[g.tn_order[b - 1]]; --b) {
                    const int t = g.tn_order[b]; g.tn_order[b] = g.tn_order[b - 1]; g.tn_order[b - 1] = t;
                }
            for (int c = 0; c < g.count; ++c) g.first_block[c + 1] = g.first_block[c] + g.regions[g.tn_order[c]] * g.SK[g.tn_order[c]];
        }
        hipLaunchKernelGGL(tnp::wgrad_tn_group_kernel, dim3(g.first_block[g.count]), dim3(256), 0, s, g);
        TNP_HIP(hipGetLastError());
        hipLaunchKernelGGL(tnp::wgrad_reduce_group_kernel, dim3(g.first_rblock[g.count]), dim3(256), 0, s, g);
        TNP_HIP(hipGetLastError());
    }
    return 0;
}
