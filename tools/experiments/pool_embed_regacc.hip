// NOT compiled into libtrajnet_hip.so -- see tools/experiments/README.md.  Register-accumulator variant of the sparse first
// layer (was a section of csrc/pool_embed_sparse.hip: uses its SparseArgs, pop_bit / sload_row helpers), followed by the
// launcher fragment that routed to it.
// ---------------------------------------------------------------------------------------------------------
// Register-accumulator kernel (default for C <= 16): 64 egos x 128 columns per workgroup, 16 waves = 8 cell groups x
// 2 column sets.  Same hit discovery as the cell-split kernel (winner tile in LDS, one ballot per cell with lane <->
// ego, scalar loads of the neighbour rows), but
//   * the accumulators of a wave -- 64 egos x its lane's column -- live in VGPRs (two 32-element vectors): the ego of a
//     hit is wave-uniform, so acc[ego] is a register read / write through s_set_gpr_idx (no LDS read-modify-write per
//     hit, no accumulator copies in LDS: the 128 KiB they took are gone);
//   * a weight register set W'[c][.][o] now serves the hits of 64 egos instead of 32: the L2 -> CU weight stream that
//     bounded the cell-split kernel (1 GB per launch at config 2) is halved;
//   * the 8 cell groups' partial sums are combined ONCE at the end, through LDS, 16 egos at a time, in fixed order
//     (group 0 + 1 + ... + 7, then bias and activation): deterministic.
// ---------------------------------------------------------------------------------------------------------
static long long *g_ra_dbg = nullptr;
#define RA_T(k) do { if (dbg && lane == 0) dbg[(blockIdx.x * 16 + wave) * 8 + (k)] = (long long)__builtin_readcyclecounter(); } while (0)
constexpr int RA_TE = 64;     // egos per workgroup (= one ballot)
constexpr int RA_OB = 128;    // output columns per workgroup
constexpr int RA_NQ = 8;      // cell groups (cell c belongs to group c % 8)
constexpr int RA_NCS = RA_OB / 64;
constexpr int RA_RED = 16;    // egos per round of the final reduction
typedef float f32x32 __attribute__((ext_vector_type(32)));

static size_t ra_smem_bytes(int ncell) {
    const size_t keys = (size_t)RA_TE * ncell * 4, red = (size_t)(RA_NQ - 1) * RA_RED * RA_OB * 4;
    return (((keys > red ? keys : red) + 15) & ~(size_t)15) + (((size_t)ncell * (RA_TE + 2) * 2 + 15) & ~(size_t)15);
}

template <int C, bool FG>
__global__ void __launch_bounds__(64 * RA_NQ * RA_NCS) pool_embed_regacc_kernel(const SparseArgs a) {
    constexpr int TE = RA_TE, OB = RA_OB, NQ = RA_NQ, NCS = RA_NCS, WLS = TE + 2, NTH = 64 * NQ * NCS;
    extern __shared__ __attribute__((aligned(16))) float rsm[];
    const size_t keys = (size_t)TE * a.ncell * 4, redb = (size_t)(NQ - 1) * RA_RED * OB * 4;
    int *wkey = reinterpret_cast<int *>(rsm);                                       // [TE][ncell]            (prologue)
    float *red = rsm;                                                               // [NQ-1][RA_RED][OB]     (epilogue)
    int16_t *wl = reinterpret_cast<int16_t *>(reinterpret_cast<char *>(rsm) + (((keys > redb ? keys : redb) + 15) & ~(size_t)15));   // [ncell][WLS]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cs = wave % NCS, q = wave / NCS;
    const int ob = blockIdx.x % a.out_blocks, tile = blockIdx.x / a.out_blocks;     // blocks b, b+8, .. share an XCD
    const int row0 = tile * TE;
    const int o = ob * OB + cs * 64 + lane;
    const unsigned ocu = (unsigned)(o < a.N1 ? o : a.N1 - 1);
    long long *dbg = a.dbg;
    RA_T(0);

    if constexpr (FG) {
        // Winner tile straight from the positions, exactly as grid_build_kernel / the cell-split kernel build it (IEEE fp32
        // cell arithmetic, LDS integer max on key = 2 j + in_range, cell-0 clobber by out-of-range / absent / padded
        // neighbours).  The per-ego scene geometry is fetched by 64 threads at once and the neighbour positions of a
        // wave's four egos are all in flight before the first is used: two global round trips for the whole tile.
        int *meta = reinterpret_cast<int *>(wl);                                    // [TE][4] lo, ns, pad, - ; [TE] float2 (dead before wl is written)
        float2 *mpos = reinterpret_cast<float2 *>(meta + 4 * TE);
        if (tid < TE) {
            const int row = row0 + tid;
            const bool ok = row < a.M;
            const int lo = ok ? a.row_base[row] : 0;
            meta[4 * tid] = lo;
            meta[4 * tid + 1] = ok ? a.row_end[row] - lo : 0;
            meta[4 * tid + 2] = ok ? a.row_padded[row] : 0;
            float2 p = ok ? reinterpret_cast<const float2 *>(a.obs2)[row] : make_float2(0.f, 0.f);
            if (p.x != p.x || p.y != p.y) { p.x = -500.0f; p.y = -500.0f; }
            mpos[tid] = p;
        }
        for (int idx = tid; idx < TE * a.ncell; idx += NTH) wkey[idx] = -1;
        __syncthreads();
        constexpr int EPW = TE / (NTH / 64);                                        // egos per wave
        const float fG = (float)a.G;
        float2 pj[EPW];
#pragma unroll
        for (int i = 0; i < EPW; ++i) {
            const int e = wave + (NTH / 64) * i;
            const int lo = meta[4 * e], ns = meta[4 * e + 1];
            pj[i] = lane < ns ? reinterpret_cast<const float2 *>(a.obs2)[lo + lane] : make_float2(0.f, 0.f);
        }
        auto vote = [&](int *wk, float2 pi, float2 p, int j) {
            if (p.x != p.x || p.y != p.y) { p.x = -500.0f; p.y = -500.0f; }
            const float ox = __fadd_rn(__fdiv_rn(__fsub_rn(p.x, pi.x), a.cell), a.half_x);
            const float oy = __fadd_rn(__fdiv_rn(__fsub_rn(p.y, pi.y), a.cell), a.half_y);
            const bool inr = !(ox < 0.0f) && !(ox >= fG) && !(oy < 0.0f) && !(oy >= fG);
            const int cellid = inr ? ((int)ox * a.G + (int)oy) : 0;
            atomicMax(&wk[cellid], 2 * j + (inr ? 1 : 0));
        };
#pragma unroll
        for (int i = 0; i < EPW; ++i) {
            const int e = wave + (NTH / 64) * i;
            const int lo = meta[4 * e], ns = meta[4 * e + 1], pad = meta[4 * e + 2], ki = row0 + e - lo;
            const float2 pi = mpos[e];
            int *wk = wkey + e * a.ncell;
            if (lane < ns && lane != ki) vote(wk, pi, pj[i], lane);
            for (int j = lane + 64; j < ns; j += 64)                                 // scenes of more than 64 tracks
                if (j != ki) vote(wk, pi, reinterpret_cast<const float2 *>(a.obs2)[lo + j], j);
            if (ns > 0 && ns < pad && lane == 0) atomicMax(&wk[0], 2 * (pad - 1));
        }
        __syncthreads();
        for (int idx = tid; idx < TE * a.ncell; idx += NTH) {
            const int e = idx / a.ncell, c = idx - e * a.ncell;
            const int row = row0 + e;
            const int k = wkey[idx];
            const int16_t v = (row < a.M && k >= 0 && (k & 1)) ? (int16_t)(k >> 1) : (int16_t)-1;
            wl[c * WLS + e] = v;
            if (a.winners_out && ob == 0 && row < a.M) a.winners_out[(size_t)row * a.ncell + c] = v;
        }
    } else {
        for (int idx = tid; idx < TE * a.ncell; idx += NTH) {
            const int e = idx / a.ncell, c = idx - e * a.ncell;
            const int row = row0 + e;
            wl[c * WLS + e] = row < a.M ? a.winners[(size_t)row * a.ncell + c] : (int16_t)-1;
        }
    }
    const int rb = a.row_base[min(row0 + lane, a.M - 1)];
    __syncthreads();

    RA_T(1);
    f32x32 accA, accB;                                                              // egos 0..31 / 32..63 of the tile, this lane's column
#pragma unroll
    for (int i = 0; i < 32; ++i) { accA[i] = 0.0f; accB[i] = 0.0f; }

    auto load_w = [&](float (&w)[C], int c) {
        const float *wb = a.Wp + (size_t)c * C * a.N1;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) w[ch] = (wb + (size_t)ch * a.N1)[ocu];
    };
    typedef float f2 __attribute__((ext_vector_type(2)));
    // even / odd channels accumulate in the two halves of one packed register: C/2 v_pk_fma_f32 with the weight pair
    // (w[2k], w[2k+1]) and the SGPR pair (e[2k], e[2k+1])
    auto fin = [&](const float (&w)[C], const typename SRow<C>::type &ev, float av) -> float {
        f2 p = {av, 0.0f};
#pragma unroll
        for (int k = 0; k < C / 2; ++k) {
            const f2 wk = {w[2 * k], w[2 * k + 1]};
            const f2 ek = {ev[2 * k], ev[2 * k + 1]};
            p = __builtin_elementwise_fma(wk, ek, p);
        }
        return p.x + p.y;
    };
    // the hits of one half of the tile (32 egos, bits of `m`) against the accumulator vector of that half
    auto half = [&](const float (&w)[C], f32x32 &acc, unsigned m, unsigned off, int lane0) {
        while (m & (m - 1u)) {                                                      // at least two hits left: both rows in flight
            const int b0 = __builtin_ctz(m); m &= m - 1u;
            const int b1 = __builtin_ctz(m); m &= m - 1u;
            typename SRow<C>::type e0, e1;
            sload_row<C>(e0, a.enc, (unsigned)__builtin_amdgcn_readlane((int)off, lane0 + b0));
            sload_row<C>(e1, a.enc, (unsigned)__builtin_amdgcn_readlane((int)off, lane0 + b1));
            const float a0 = acc[b0], a1 = acc[b1];
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(e0), "+s"(e1));
            acc[b0] = fin(w, e0, a0);
            acc[b1] = fin(w, e1, a1);
        }
        if (m) {
            const int b0 = __builtin_ctz(m);
            typename SRow<C>::type e0;
            sload_row<C>(e0, a.enc, (unsigned)__builtin_amdgcn_readlane((int)off, lane0 + b0));
            const float a0 = acc[b0];
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(e0));
            acc[b0] = fin(w, e0, a0);
        }
    };
    auto process = [&](float (&w)[C], int c) {
        const int wv = (int)wl[c * WLS + lane];                                     // lane <-> ego of the tile
        const unsigned long long mask = __ballot(wv >= 0);
        if (mask == 0ull) return;
#pragma unroll
        for (int ch = 0; ch < C; ++ch) asm("" : "+v"(w[ch]));                        // the weights are waited for HERE, once per cell
        const unsigned off = __umul24((unsigned)(rb + wv), (unsigned)(a.ldv * 4));   // byte offset of every lane's neighbour row
        half(w, accA, (unsigned)mask, off, 0);
        half(w, accB, (unsigned)(mask >> 32), off, 32);
    };
    // Cells of this wave's group that have a hit in the tile.  Group q takes cell NQ k + ((q - k) mod NQ) of every block k
    // of NQ consecutive cells (bit k): the rotation spreads the dense central columns of the grid over all groups
    // (cell = x * n + y, so a plain c mod NQ would give one group the busiest columns).
    const int nk = (a.ncell + NQ - 1) / NQ;                                          // <= 64: ncell <= 512
    auto cell_of = [&](int k) { return NQ * k + ((q - k) & (NQ - 1)); };
    unsigned long long occ;
    {
        bool any = false;
        const int c = cell_of(lane);
        if (lane < nk && c < a.ncell) {
            const int16_t *col = wl + c * WLS;
            for (int e = 0; e < TE; ++e) any |= col[e] >= 0;
        }
        occ = __ballot(any);
    }
    auto pop = [&]() -> int { if (!occ) return -1; const int k = __ffsll((long long)occ) - 1; occ &= occ - 1ull; return cell_of(k); };
    float wA[C], wB[C];
    int ca = pop();
    if (ca >= 0) load_w(wA, ca);
    while (ca >= 0) {
        const int cb = pop();
        if (cb >= 0) load_w(wB, cb);
        process(wA, ca);
        if (cb < 0) break;
        ca = pop();
        if (ca >= 0) load_w(wA, ca);
        process(wB, cb);
    }

    RA_T(2);
    // ---- the 8 cell groups' partial sums, 16 egos per round, fixed order; bias + activation; coalesced rows ----
    const float bias = (a.bias && o < a.N1) ? a.bias[ocu] : 0.0f;
#pragma unroll
    for (int r = 0; r < TE / RA_RED; ++r) {
        __syncthreads();                                                            // winner tile / previous round no longer read
        if (q > 0) {
#pragma unroll
            for (int e = 0; e < RA_RED; ++e) {
                const int g = RA_RED * r + e;
                red[((q - 1) * RA_RED + e) * OB + cs * 64 + lane] = g < 32 ? accA[g] : accB[g - 32];
            }
        }
        __syncthreads();
        if (q == 0) {
#pragma unroll
            for (int e = 0; e < RA_RED; ++e) {
                const int g = RA_RED * r + e;
                float v = g < 32 ? accA[g] : accB[g - 32];
#pragma unroll
                for (int qq = 1; qq < NQ; ++qq) v += red[((qq - 1) * RA_RED + e) * OB + cs * 64 + lane];
                v += bias;
                if (a.relu) v = fmaxf(v, 0.0f);
                const int row = row0 + g;
                if (row < a.M && o < a.N1) a.out[(size_t)row * a.ldo + o] = v;
            }
        }
    }
    RA_T(3);
}

bool regacc_supported(int C, int ncell) { return C <= 16 && ncell <= 64 * RA_NQ && ra_smem_bytes(ncell) <= (size_t)160 * 1024; }


/* launcher fragment (inside launch_pool_embed_sparse):
    const bool lean_rows = (size_t)M * ldv * sizeof(float) < ((size_t)1 << 32);   // 32-bit byte offsets of the neighbour rows
    if (lean_rows && regacc_supported(C, ncell)) {                                // register accumulators, 64-ego tiles
        a.out = out;
        a.S = 1; a.cps = ncell; a.ego_tiles = (M + RA_TE - 1) / RA_TE; a.out_blocks = (N1 + RA_OB - 1) / RA_OB;
        if (fg) {
            a.obs2 = fg->obs2; a.row_end = fg->row_end; a.row_padded = fg->row_padded; a.G = fg->G;
            a.cell = fg->cell; a.half_x = fg->half_x; a.half_y = fg->half_y; a.winners_out = fg->winners_out;
        }
        a.dbg = g_ra_dbg;
        const size_t rsmem = ra_smem_bytes(ncell);
        const int rblocks = a.ego_tiles * a.out_blocks;
#define RA_LAUNCH(CC, FGB) { static bool set = false; if (!set) { TNP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>( \
        pool_embed_regacc_kernel<CC, FGB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); set = true; } \
        hipLaunchKernelGGL((pool_embed_regacc_kernel<CC, FGB>), dim3(rblocks), dim3(64 * RA_NQ * RA_NCS), rsmem, s, a); }
#define RA_SWITCH(CC) { if (fg) RA_LAUNCH(CC, true) else RA_LAUNCH(CC, false) }
        if (C == 4) RA_SWITCH(4) else if (C == 8) RA_SWITCH(8) else RA_SWITCH(16)
        TNP_HIP(hipGetLastError());
        return 0;
    }
*/
