// First layer of the social grid embedding, cell-major on the matrix cores (reference lstm/gridbased_pooling.py:107-109
// applied to the social grid of :145-170), gfx950.
//
//   y1[i, o] = act(b[o] + sum over the occupied cells c of ego i of  sum_ch W'[c][ch][o] * enc[j(i,c)][ch])
//
// pool_embed_sparse.hip walks this ego-major on the VALU: a weight register set W'[c][.][o] serves the two or three
// egos of a 32-ego tile that have cell c occupied, so that kernel is pinned to its L2 -> CU weight stream (1 GB per
// launch at BASELINE config 2).  Here the loop is CELL-major over a 128-ego tile: the hits (ego, neighbour) of the tile
// that fall into cell c form the N dimension of a v_mfma_f32_16x16x4_f32, the cell's [C x 16] weight block is the A
// operand, and K = C:
//
//   D[o (16 columns)][hit (16 hits)] = sum_ch  W'[c][ch][o]  *  enc[j(hit)][ch]
//
// One workgroup = 128 egos x 64 output columns, 4 waves, wave w owns columns 16w..16w+15 of the block: every
// accumulator entry acc[ego][column] (fp32 tile in LDS) has exactly one writer wave and its adds happen in program
// order (cells ascending, egos ascending) -> deterministic, bit-reproducible.  After the MFMAs lane l holds, for hit
// l & 15, the four consecutive columns 4 (l >> 4) .. +3: one 128-bit LDS read-modify-write per 16 hits x 16 columns x C
// channels.  Exact fp32 products and sums (the MFMA is an fmaf chain over ch), like the VALU kernel.  The main loop is
// software-pipelined over PAIRS of 16-hit groups (two independent accumulators keep the matrix pipe issuing every 32
// cycles): stage 0 reads the pair's cell ids and hit entries (LDS), stage 1 its weights (global, L2-resident) and
// neighbour encodings (LDS), stage 2 issues the 2 x C/4 MFMAs, stage 3 adds the previous pair's results into the tile.
//
// Prologue (per workgroup, repeated by the 16 column blocks of an ego tile -- 128 x 31 pairs, cheaper than a kernel and
// a table round trip): positions and encodings of the tile's scenes are staged in LDS; the winner keys are built
// exactly as grid_build_kernel does it (IEEE fp32 cell arithmetic, LDS integer max on key = 2 j + in_range = "last
// writer in ascending j wins", cell-0 clobber by out-of-range / absent / padded neighbours) 32 egos at a time and
// compressed into a transposed int8 winner tile [cell][ego]; one thread per cell then turns its 128 bytes into the
// cell's hit list, padded to 16-hit groups (16-bit entries: ego of the tile << 8 | encoding row of the tile's scenes).
// The 128-ego tile makes a cell's list ~8 hits long at config 2 (one MFMA group per cell) and cuts the weight stream
// to 256 MB per launch.
#include "tnp_internal.h"
#include <stdlib.h>
#include <type_traits>

namespace tnp {

static long long *g_cm_dbg = nullptr;
#define CM_T(k) do { if (a.dbg && tid == 0) a.dbg[blockIdx.x * 8 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)

constexpr int CM_TE = 128;          // egos per workgroup
constexpr int CM_OB = 64;           // output columns per workgroup (4 waves x 16)
constexpr int CM_ACCLD = 80;        // accumulator row stride (floats), = 16 mod 64: a row's 16 columns of a wave start at bank
                                    // 16 * ((row + wave) mod 4) -- four rows with distinct (row mod 4) never conflict
constexpr int CM_WLD = CM_TE + 16;   // winner table row stride (bytes): threads reading whole rows hit distinct banks
constexpr int CM_ACC_ROWS = CM_TE + 16;   // + one scratch row per hit slot: padded hit slots add their garbage there
constexpr int CM_ENC_ROWS = 256;    // rows of the tile's scenes: 128 egos + the rest of the first / last scene (<= 254)
constexpr int CM_MAX_SCENE = 64;    // largest scene the tile layout above admits
constexpr int CM_HITS_CAP = CM_TE * (CM_MAX_SCENE - 1);
constexpr int CM_ITEM_PAD = 16;     // items past the end that the pipelined loop may read (padded with "no hit")
constexpr int CM_NONE = 0xFF;       // winner byte: untouched cell
constexpr int CM_CLOB = 0xFE;       // winner byte: clobbered by an out-of-range / absent / padded neighbour

struct CellMajorArgs {
    const float *obs2;               // [M][2]
    const int32_t *row_base, *row_end, *row_padded;   // [M]
    const float *enc; int ldv;       // [M][ldv]
    const float *Wp;                 // [ncell][N1/16][4][16][C/4] (tnp_lstm_model.Wp0_mfma)
    const float *bias;               // [N1] or NULL
    int M, ncell, G, N1, out_blocks, relu, item_cap;
    float cell, half_x, half_y;
    float *out; int ldo;
    int16_t *winners_out;            // optional [M][ncell]
    int abl;
    long long *dbg;
};

typedef float f32x4 __attribute__((ext_vector_type(4)));

static int cm_item_cap(int ncell) { return 2 * ncell + CM_HITS_CAP / 16; }  // 16-hit groups: ceil(n/16) (+1 with residue slotting) per cell
struct CmLayout { size_t acc, win_lo, win_hi, enc, pos, entries, item_woff, meta, total; };
static CmLayout cm_layout(int ncell, int C) {
    CmLayout L;
    auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
    size_t off = 0;
    // the winner tiles are dead once the hit lists exist: the accumulator tile takes over their space
    L.acc = off; L.win_lo = off;
    L.win_hi = off + al((size_t)ncell * CM_WLD);
    const size_t wb = 2 * al((size_t)ncell * CM_WLD), ab = al((size_t)CM_ACC_ROWS * CM_ACCLD * 4);
    off += wb > ab ? wb : ab;
    L.enc = off; off += al((size_t)CM_ENC_ROWS * (C + 4) * 4);   // rows padded by 16 bytes: 16 hits' rows spread over the banks
    L.pos = off; off += al((size_t)CM_ENC_ROWS * 8);
    L.entries = off; off += al((size_t)(cm_item_cap(ncell) + CM_ITEM_PAD) * 32);
    L.item_woff = off; off += al((size_t)(cm_item_cap(ncell) + CM_ITEM_PAD) * 4);
    L.meta = off; off += al((size_t)CM_TE * 6 + 64);
    L.total = off;
    return L;
}

// bytes of a dword that differ from 0xFF, as 0x80 flags
__device__ __forceinline__ uint32_t cm_nonff(uint32_t d) {
    const uint32_t x = ~d;
    return (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
}
// winners of four egos: the upper-half table where it was written (hit or clobber), else the lower-half table
__device__ __forceinline__ uint32_t cm_merge(uint32_t lo, uint32_t hi) {
    const uint32_t bm = (cm_nonff(hi) >> 7) * 0xFFu;
    return (hi & bm) | (lo & ~bm);
}
// hit flags (0x80 per byte) of merged winners: neither untouched (0xFF) nor clobbered (0xFE)
__device__ __forceinline__ uint32_t cm_hits(uint32_t merged) { return cm_nonff(merged | 0x01010101u); }

template <int C, int ABL = 0>
__global__ void __launch_bounds__(256) pool_embed_cellmajor_kernel(const CellMajorArgs a, const CmLayout L) {
    constexpr int KS = C / 4;                                   // MFMA k-steps: K = 4 * KS = C
    extern __shared__ __attribute__((aligned(16))) unsigned char cm_smem[];
    const int ncell = a.ncell;
    float *acc = reinterpret_cast<float *>(cm_smem + L.acc);                        // [128 + 16][CM_ACCLD]   (main loop)
    uint8_t *win_lo = cm_smem + L.win_lo, *win_hi = cm_smem + L.win_hi;             // [ncell][128] each      (prologue)
    constexpr int ELD = C + 4;                                  // staged encoding row stride (floats)
    float *encl = reinterpret_cast<float *>(cm_smem + L.enc);                       // [CM_ENC_ROWS][ELD]
    float2 *posl = reinterpret_cast<float2 *>(cm_smem + L.pos);                     // [CM_ENC_ROWS] positions, sentinel applied
    uint16_t *entries = reinterpret_cast<uint16_t *>(cm_smem + L.entries);          // [items][16]: accumulator row << 8 | encoding row
    uint32_t *item_woff = reinterpret_cast<uint32_t *>(cm_smem + L.item_woff);      // [items] byte offset of the cell's weight block
    int16_t *rbl = reinterpret_cast<int16_t *>(cm_smem + L.meta);                   // [128] row_base - lo_t
    int16_t *nsl = rbl + CM_TE;                                                     // [128] tracks of the ego's scene
    int16_t *padl = nsl + CM_TE;                                                    // [128] slots the reference pads it to
    int *wsum = reinterpret_cast<int *>(padl + CM_TE);                              // [4]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ob = blockIdx.x % a.out_blocks, tile = blockIdx.x / a.out_blocks;     // blocks b, b+8, .. share an XCD
    const int row0 = tile * CM_TE;
    const int nrows = min(CM_TE, a.M - row0);
    // Rows of the tile's scenes: scenes hold at most CM_MAX_SCENE tracks, so they lie inside the fixed window
    // [row0 - 63, row0 + 128 + 63) -- no dependent load (first scene's start) in front of the staging loads.
    const int lo_t = max(row0 - (CM_MAX_SCENE - 1), 0);
    const int nenc = min(a.M - lo_t, CM_ENC_ROWS - 2);

    CM_T(0);
    // ---- prologue 1: positions / encodings of the tile's scenes, per-ego scene geometry, cleared winner tiles ----
    for (int idx = tid; idx < nenc * KS; idx += 256) {
        const int r = idx / KS, q = idx - r * KS;
        *reinterpret_cast<float4 *>(encl + r * ELD + 4 * q) = *reinterpret_cast<const float4 *>(a.enc + (size_t)(lo_t + r) * a.ldv + 4 * q);
    }
    for (int r = tid; r < nenc; r += 256) {
        float2 p = reinterpret_cast<const float2 *>(a.obs2)[lo_t + r];
        if (p.x != p.x || p.y != p.y) { p.x = -500.0f; p.y = -500.0f; }            // gridbased_pooling.py:247-249
        posl[r] = p;
    }
    if (tid < CM_TE) {
        const bool ok = tid < nrows;
        const int rb = ok ? a.row_base[row0 + tid] : lo_t;
        rbl[tid] = (int16_t)(rb - lo_t);
        nsl[tid] = (int16_t)(ok ? a.row_end[row0 + tid] - rb : 0);
        padl[tid] = (int16_t)(ok ? min(a.row_padded[row0 + tid], 32767) : 0);
    }
    {
        const uint4 ff = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        const int nq = (int)((L.win_hi - L.win_lo) / 16) + ncell * (CM_WLD / 16);  // both tables (and the gap between them)
        for (int idx = tid; idx < nq; idx += 256) reinterpret_cast<uint4 *>(win_lo)[idx] = ff;
    }
    __syncthreads();

    CM_T(1);
    // ---- prologue 2: winners.  Two threads per ego walk the lower / upper half of its neighbours in ascending j and
    // store the neighbour's encoding row into their table at [cell][ego]: the program order of a thread's stores is
    // index_put's "last writer in ascending j wins"; an out-of-range / absent neighbour stores "clobbered" into cell 0
    // (:281-282), and so does the padded slot after the loop.  Upper half beats lower half when the lists are built ----
    if constexpr (!(ABL & 2)) {
        const int eg = tid & (CM_TE - 1), half = tid >> 7;
        const int ns = nsl[eg], rb = rbl[eg], ki = row0 + eg - lo_t - rb;
        const float fG = (float)a.G;
        const float2 pi = posl[min(rb + ki, CM_ENC_ROWS - 1)];
        uint8_t *col = (half ? win_hi : win_lo) + eg;
        const int mid = (ns + 1) >> 1;
        const int j0 = half ? mid : 0, j1 = half ? ns : mid;
#pragma unroll 4
        for (int j = j0; j < j1; ++j) {
            const float2 pj = posl[rb + j];
            const float ox = __fadd_rn(__fdiv_rn(__fsub_rn(pj.x, pi.x), a.cell), a.half_x);   // :276
            const float oy = __fadd_rn(__fdiv_rn(__fsub_rn(pj.y, pi.y), a.cell), a.half_y);
            const bool inr = !(ox < 0.0f) && !(ox >= fG) && !(oy < 0.0f) && !(oy >= fG);      // :278-279
            const int cellid = inr ? ((int)ox * a.G + (int)oy) : 0;                            // :281-287
            if (j != ki) col[cellid * CM_WLD] = inr ? (uint8_t)(rb + j) : (uint8_t)CM_CLOB;    // the ego itself is no neighbour (:259-263)
        }
        if (half && ns > 0 && ns < padl[eg]) col[0] = (uint8_t)CM_CLOB;   // slots the reference pads this scene to (lstm.py:29)
    }
    __syncthreads();
    if (a.winners_out && ob == 0) {                                  // the winner table, for the training backward
        for (int idx = tid; idx < CM_TE * ncell; idx += 256) {
            const int e = idx / ncell, c = idx - e * ncell, row = row0 + e;
            const int hi = win_hi[c * CM_WLD + e], v = hi != CM_NONE ? hi : (int)win_lo[c * CM_WLD + e];
            if (row < a.M) a.winners_out[(size_t)row * ncell + c] = v < CM_CLOB ? (int16_t)(v - rbl[e]) : (int16_t)-1;
        }
    }

    CM_T(2);
    // ---- prologue 3: hit lists.  Thread <-> cell (tid, then tid + 256 for grids of more than 256 cells): the cell's 128
    // merged winners are read into registers once; count, scan of the 16-hit group counts over the cells, fill ----
    int nitems = 0;
    const uint32_t cell_bytes = (uint32_t)C * (uint32_t)a.N1 * 4u;   // one cell's block of W''
    for (int c0 = 0; c0 < ncell; c0 += 256) {
        const int c = c0 + tid;
        uint32_t mw[CM_TE / 4];
        int n = 0;
        if (c < ncell) {
            const uint4 *pl = reinterpret_cast<const uint4 *>(win_lo + c * CM_WLD), *ph = reinterpret_cast<const uint4 *>(win_hi + c * CM_WLD);
#pragma unroll
            for (int q = 0; q < CM_TE / 16; ++q) {
                const uint4 l = pl[q], h = ph[q];
                mw[4 * q] = cm_merge(l.x, h.x); mw[4 * q + 1] = cm_merge(l.y, h.y);
                mw[4 * q + 2] = cm_merge(l.z, h.z); mw[4 * q + 3] = cm_merge(l.w, h.w);
            }
#pragma unroll
            for (int q = 0; q < CM_TE / 4; ++q) n += __builtin_popcount(cm_hits(mw[q]));
        }
        // Slots of a 16-hit group: the MFMA leaves hits 4q .. 4q+3 in lane group q, and the four lane groups of one LDS
        // instruction then touch four accumulator rows -- conflict free when those rows differ mod 4.  So a hit of ego e
        // (accumulator row e) takes a slot of quad e mod 4 = its byte position in the merged dword; a cell needs as many
        // groups as its fullest residue class has hits / 4.
        uint32_t cnt4 = 0;                                                      // per-residue hit counts, one per byte
        if (c < ncell) {
#pragma unroll
            for (int q = 0; q < CM_TE / 4; ++q) cnt4 += cm_hits(mw[q]) >> 7;
        }
        const uint32_t mx = max(max(cnt4 & 0xff, (cnt4 >> 8) & 0xff), max((cnt4 >> 16) & 0xff, cnt4 >> 24));
        // at most one group more than the hits need: beyond that (a lopsided cell) the slots are filled in ego order and
        // that cell's adds may conflict -- the item count stays bounded by cm_item_cap()
        const bool by_residue = (int)((mx + 3) >> 2) <= ((n + 15) >> 4) + 1;
        const int groups = by_residue ? (int)((mx + 3) >> 2) : ((n + 15) >> 4);
        int incl = groups;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(incl, d); if (lane >= d) incl += t; }
        __syncthreads();                                                       // wsum of the previous pass has been read
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int item0 = nitems + incl - groups;
        for (int w = 0; w < wave; ++w) item0 += wsum[w];
        nitems += wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (groups > 0 && item0 + groups <= a.item_cap) {
            for (int g = 0; g < groups; ++g) item_woff[item0 + g] = (uint32_t)c * cell_bytes;
            uint16_t *dst = entries + item0 * 16;
            for (int r = 0; r < groups * 16; ++r) dst[r] = (uint16_t)(((CM_TE + 4 * (r & 3) + ((r >> 2) & 3)) << 8) | 0xff);   // scratch row of the slot, garbage encoding
            uint32_t run = 0;                                                  // hits placed so far, per residue
#pragma unroll
            for (int q = 0; q < CM_TE / 4; ++q) {
                const uint32_t d = mw[q];
                uint32_t m = cm_hits(d);
                while (m) {
                    const int byte = __builtin_ctz(m) >> 3;
                    m &= m - 1u;
                    const int k = by_residue ? (run >> (8 * byte)) & 0xff : run;
                    run += by_residue ? 1u << (8 * byte) : 1u;
                    dst[by_residue ? (k >> 2) * 16 + 4 * byte + (k & 3) : k] = (uint16_t)(((4 * q + byte) << 8) | ((d >> (8 * byte)) & 0xff));
                }
            }
        }
    }
    nitems = min(nitems, a.item_cap);
    if constexpr (ABL & 1) nitems = 0;
    entries[nitems * 16 + tid] = (uint16_t)(((CM_TE + 4 * (tid & 3) + ((tid >> 2) & 3)) << 8) | 0xff);   // what the pipeline reads past the end
    if (tid < CM_ITEM_PAD) item_woff[nitems + tid] = 0;
    __syncthreads();
    {   // the accumulator tile takes over the winner tiles' space
        const float4 z = {0.f, 0.f, 0.f, 0.f};
        for (int idx = tid; idx < CM_ACC_ROWS * CM_ACCLD / 4; idx += 256) reinterpret_cast<float4 *>(acc)[idx] = z;
    }
    __syncthreads();

    CM_T(3);
    // ---- main loop over pairs of items ----
    const int hl = lane & 15, kq = lane >> 4;
    // this lane's A operands of a cell: KS consecutive floats at W''[c][4 ob + wave][kq][hl][.]
    const char *wlane = reinterpret_cast<const char *>(a.Wp + ((size_t)(ob * (CM_OB / 16) + wave) * 64 + lane) * KS);
    const float *encq = encl + KS * kq;                                       // + enc row * ELD

    struct Head { uint32_t woff[2]; int ent[2]; uint2 rows[2]; };
    // Four register sets of everything, rotated by a fully unrolled 4-step body (no register moves, so no load has to be
    // waited for before its consumer): in step r of pair p, set r holds pair p's operands, set r+1 receives pair p+1's
    // encodings (LDS), set r+3 receives pair p+3's weights (global: three steps of latency cover) and set r+3 still holds
    // pair p-1's results, which are added to the accumulator tile now.  The non-MFMA work of a step is cut into chunks
    // that are issued BETWEEN the step's MFMAs (an MFMA occupies the matrix pipe for 32 cycles, during which the wave can
    // issue ~6 other instructions).
    // MFMA orientation: A = encodings (16 hit slots x K), B = weights (K x 16 columns), D[hit slot][column]: lane l holds
    // column l & 15 of the hit slots 4 (l >> 4) .. +3, i.e. one LDS instruction per register touches four accumulator rows
    // (one per lane group) x 16 consecutive columns -- conflict free thanks to the residue slotting of the hit lists.
    constexpr int NS = 4;
    Head H[NS];
    float Wr[NS][2][KS], Br[NS][2][KS];
    uint2 Rr[NS][2];                                                          // packed entries of the lane group's four hit slots
    f32x4 D[NS][2];
    auto read_head = [&](Head &h, int pair) {                                 // cell weight offsets + hit entries (LDS)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            h.woff[u] = item_woff[2 * pair + u];
            h.ent[u] = entries[(2 * pair + u) * 16 + hl];
            h.rows[u] = *reinterpret_cast<const uint2 *>(entries + (2 * pair + u) * 16 + 4 * kq);
        }
    };
    auto load_w = [&](float (&w)[KS], uint32_t woff) {                        // one item's weight block (global, L2-resident)
        const char *wp = wlane + __builtin_amdgcn_readfirstlane(woff);
        if constexpr (KS == 1) { w[0] = *reinterpret_cast<const float *>(wp); }
        else if constexpr (KS == 2) { const float2 t = *reinterpret_cast<const float2 *>(wp); w[0] = t.x; w[1] = t.y; }
        else {
#pragma unroll
            for (int q = 0; q < KS / 4; ++q) {
                const float4 t = reinterpret_cast<const float4 *>(wp)[q];
                w[4 * q] = t.x; w[4 * q + 1] = t.y; w[4 * q + 2] = t.z; w[4 * q + 3] = t.w;
            }
        }
    };
    auto load_b = [&](float (&bb)[KS], int ent) {                             // one item's neighbour encodings (LDS)
        const float *ep = encq + (ent & 255) * ELD;
        if constexpr (KS == 1) { bb[0] = ep[0]; }
        else if constexpr (KS == 2) { const float2 t = *reinterpret_cast<const float2 *>(ep); bb[0] = t.x; bb[1] = t.y; }
        else {
#pragma unroll
            for (int q = 0; q < KS / 4; ++q) {
                const float4 t = reinterpret_cast<const float4 *>(ep)[q];
                bb[4 * q] = t.x; bb[4 * q + 1] = t.y; bb[4 * q + 2] = t.z; bb[4 * q + 3] = t.w;
            }
        }
    };
    float *acol = acc + wave * 16 + hl;                                       // + row * CM_ACCLD
    auto row_ptr = [&](const uint2 &rows, int rr) -> float * {                // accumulator row of hit slot 4 kq + rr
        const uint32_t w = rr < 2 ? rows.x : rows.y;
        const uint32_t row = (rr & 1) ? (w >> 24) : ((w >> 8) & 0xffu);
        return acol + row * CM_ACCLD;
    };
    const int npairs = (((nitems + 1) >> 1) + NS - 1) / NS * NS;              // padded items add into the scratch rows
#pragma unroll
    for (int q = 0; q < NS; ++q) read_head(H[q], q);
    const uint2 scratch_rows = *reinterpret_cast<const uint2 *>(entries + nitems * 16 + 4 * kq);   // a padding item
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        load_w(Wr[0][u], H[0].woff[u]); load_w(Wr[1][u], H[1].woff[u]); load_w(Wr[2][u], H[2].woff[u]);
        load_b(Br[0][u], H[0].ent[u]);
        Rr[0][u] = H[0].rows[u];
        Rr[NS - 1][u] = scratch_rows; D[NS - 1][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    auto step = [&](auto R, int p) {
        constexpr int r = decltype(R)::value, r1 = (r + 1) % NS, r3 = (r + 3) % NS;
        constexpr int NM = 2 * KS;                                            // MFMAs of the step
        f32x4 d[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        float t0[4], t1[4];
        float *q0[4], *q1[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { q0[rr] = row_ptr(Rr[r3][0], rr); q1[rr] = row_ptr(Rr[r3][1], rr); }
        auto chunk = [&](int j) {
            switch (j) {
            case 0: if constexpr (!(ABL & 4)) {                                                          // pair p-1, item 0: read its 4 rows
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) t0[rr] = *q0[rr];
                    } break;
            case 1: load_b(Br[r1][0], H[r1].ent[0]); load_b(Br[r1][1], H[r1].ent[1]);                    // pair p+1 encodings
                    Rr[r1][0] = H[r1].rows[0]; Rr[r1][1] = H[r1].rows[1]; break;
            case 2: if constexpr (!(ABL & 8)) { load_w(Wr[r3][0], H[r3].woff[0]); load_w(Wr[r3][1], H[r3].woff[1]); } break;   // pair p+3 weights
            case 3: if constexpr (!(ABL & 4)) {
                        asm volatile("" : "+v"(t0[0]), "+v"(t0[1]), "+v"(t0[2]), "+v"(t0[3]));          // the adds stay HERE, far from the reads
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) *q0[rr] = t0[rr] + D[r3][0][rr];
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) t1[rr] = *q1[rr];                                 // item 1 is read after item 0's writes
                    } break;
            case 4: read_head(H[r], p + NS); break;
            case 6: if constexpr (!(ABL & 4)) {
                        asm volatile("" : "+v"(t1[0]), "+v"(t1[1]), "+v"(t1[2]), "+v"(t1[3]));
#pragma unroll
                        for (int rr = 0; rr < 4; ++rr) *q1[rr] = t1[rr] + D[r3][1][rr];
                    } break;
            default: break;
            }
        };
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            d[m & 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(Br[r][m & 1][m >> 1], Wr[r][m & 1][m >> 1], d[m & 1], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j * NM / 8 == m) chunk(j);
            __builtin_amdgcn_sched_barrier(0);                                // keep the interleave as written
        }
        asm volatile("" : "+a"(d[0]), "+a"(d[1]));                            // results stay in the accumulation registers until
        D[r][0] = d[0]; D[r][1] = d[1];                                       // the next step adds them to the tile
    };
    for (int p = 0; p < npairs; p += NS) {
        step(std::integral_constant<int, 0>{}, p);
        step(std::integral_constant<int, 1>{}, p + 1);
        step(std::integral_constant<int, 2>{}, p + 2);
        step(std::integral_constant<int, 3>{}, p + 3);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)                                               // the last pair's results
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { float *q = row_ptr(Rr[NS - 1][u], rr); *q += D[NS - 1][u][rr]; }
    __syncthreads();

    CM_T(4);
    // ---- epilogue: bias + activation, 128 rows x 64 columns ----
    for (int idx = tid; idx < CM_TE * (CM_OB / 4); idx += 256) {
        const int e = idx / (CM_OB / 4), q = idx - e * (CM_OB / 4);
        const int row = row0 + e, oo = ob * CM_OB + 4 * q;
        if (row >= a.M) continue;
        float4 t = *reinterpret_cast<const float4 *>(acc + e * CM_ACCLD + 4 * q);
        if (a.bias) {
            const float4 b = *reinterpret_cast<const float4 *>(a.bias + oo);
            t.x += b.x; t.y += b.y; t.z += b.z; t.w += b.w;
        }
        if (a.relu) { t.x = fmaxf(t.x, 0.0f); t.y = fmaxf(t.y, 0.0f); t.z = fmaxf(t.z, 0.0f); t.w = fmaxf(t.w, 0.0f); }
        *reinterpret_cast<float4 *>(a.out + (size_t)row * a.ldo + oo) = t;
    }
    CM_T(5);
}

bool cellmajor_supported(int C, int N1, int ncell, int max_scene, int ldv) {
    return (C == 4 || C == 8 || C == 16 || C == 32) && N1 % CM_OB == 0 && ncell >= 1 && ncell <= 512 &&
           max_scene >= 1 && max_scene <= CM_MAX_SCENE && ldv % 4 == 0 && cm_layout(ncell, C).total <= (size_t)160 * 1024 &&
           (size_t)ncell * C * N1 * 4 < ((size_t)1 << 31);
}

int launch_pool_embed_cellmajor(const SparseGridFuse &fg, const float *enc, int ldv, const int32_t *row_base, const float *Wp,
                                const float *bias, int M, int ncell, int C, int N1, int relu, float *out, int ldo,
                                hipStream_t s) {
    if (M <= 0) return 0;
    if (!cellmajor_supported(C, N1, ncell, fg.max_scene, ldv)) TNP_FAIL(-1, "cell-major pooling embedding: unsupported shape");
    if ((reinterpret_cast<uintptr_t>(enc) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) || ldo % 4 != 0 ||
        (bias && (reinterpret_cast<uintptr_t>(bias) & 15)))
        TNP_FAIL(-1, "cell-major pooling embedding: operands must be 16-byte aligned");
    CellMajorArgs a;
    a.obs2 = fg.obs2; a.row_base = row_base; a.row_end = fg.row_end; a.row_padded = fg.row_padded;
    a.enc = enc; a.ldv = ldv; a.Wp = Wp; a.bias = bias;
    a.M = M; a.ncell = ncell; a.G = fg.G; a.N1 = N1; a.out_blocks = N1 / CM_OB; a.relu = relu;
    a.item_cap = cm_item_cap(ncell);
    a.cell = fg.cell; a.half_x = fg.half_x; a.half_y = fg.half_y;
    a.out = out; a.ldo = ldo; a.winners_out = fg.winners_out;
    { const char *e = getenv("TNP_CM_ABL"); a.abl = e ? atoi(e) : 0; }
    a.dbg = g_cm_dbg;
    const CmLayout L = cm_layout(ncell, C);
    const int blocks = ((M + CM_TE - 1) / CM_TE) * a.out_blocks;
#define CM_LAUNCH_C(CC) { static bool set = false; if (!set) { TNP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>( \
        pool_embed_cellmajor_kernel<CC>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
        hipLaunchKernelGGL((pool_embed_cellmajor_kernel<CC>), dim3(blocks), dim3(256), L.total, s, a, L); }
#define CM_LAUNCH_ABL(AB) { static bool set = false; if (!set) { TNP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>( \
        pool_embed_cellmajor_kernel<16, AB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
        hipLaunchKernelGGL((pool_embed_cellmajor_kernel<16, AB>), dim3(blocks), dim3(256), L.total, s, a, L); }
    if (C == 16 && a.abl) { switch (a.abl) { case 1: CM_LAUNCH_ABL(1) break; case 3: CM_LAUNCH_ABL(3) break; case 4: CM_LAUNCH_ABL(4) break;
        case 8: CM_LAUNCH_ABL(8) break; case 12: CM_LAUNCH_ABL(12) break; default: CM_LAUNCH_ABL(2) break; } }
    else if (C == 4) CM_LAUNCH_C(4) else if (C == 8) CM_LAUNCH_C(8) else if (C == 16) CM_LAUNCH_C(16) else CM_LAUNCH_C(32)
    TNP_HIP(hipGetLastError());
    return 0;
}

}  // namespace tnp

extern "C" __attribute__((visibility("default"))) void tnp_cm_debug(void *buf) { tnp::g_cm_dbg = (long long *)buf; }
