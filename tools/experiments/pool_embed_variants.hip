// Measured-and-not-adopted shapes of the sparse first layer (round 3).  NOT compiled into libtrajnet_hip.so: included by
// sparse_ablate.hip after the product source (which provides SparseArgs, SRow, sload_row, pop_bit, ra_f32x32 and the
// TNP_ABL hooks).  Numbers and the reasons they lose are in tools/experiments/README.md.
//   * pool_embed_regwide_kernel<C, 128, 1> / <C, 64, 2>: 8 waves with 128 accumulators per lane (128-ego tiles, or two
//     columns per lane);
//   * pool_embed_regshare_kernel<C>: 128-ego tiles on 16 waves, the two waves of a cell group share the group's weight
//     blocks through an LDS ring filled by LDS-DMA, byte winners voted through per-half-wave tables;
//   * pool_embed_regring_kernel<C, DB>: the product's tile and waves, every wave with a private two-slot LDS ring (LDS-DMA two
//     cells ahead) and one weight register set (DB = false) or two, the next visit's read issued mid-visit (DB = true).
// All five are bit-identical to the product kernel on every output element.
#pragma once

namespace tnp {

// ---------------------------------------------------------------------------------------------------------
// Wide register-accumulator kernel (round 3): 8 waves = 8 cell groups, TWO waves per SIMD with up to 256 VGPRs each, so a
// wave holds TE x CPL = 128 accumulators instead of 64 (same 64 Ki accumulator floats per CU as the kernel above, split
// among half as many waves).  Two shapes:
//   * TE = 128, CPL = 1: 128 egos x 64 columns per workgroup.  A weight register set W''[c][.][o] now serves the hits of 128
//     egos: the L2 -> CU weight stream (0.55 GB per launch at BASELINE config 2 = 16.7 us at the L2's 34.5 TB/s) halves;
//   * TE = 64, CPL = 2: 64 egos x 128 columns, two columns per lane.  Same stream as the kernel above, but a hit is one
//     scalar-pipe sequence (pop, readlane, row load, register-index switches) for 16 packed FMAs instead of 8.
// Everything else is the kernel above: winner keys built from the positions in the prologue (transposed [cell][ego], LDS
// integer max = "last writer in ascending j wins", out-of-range / absent / padded neighbours clobber cell 0 -- here ONE
// atomic per ego carries the largest such j instead of one per neighbour, which all hit the same LDS address), occupied
// cells only, quad-major weights through inline-asm loads with manual vmcnt, neighbour rows through scalar loads (two
// hits in flight per wait), partial sums of the 8 cell groups added through LDS in fixed order (group 0 + 1 + ... + 7:
// results do not depend on the tile shape's alignment -- but the association differs from nothing: the per-ego sum is the
// same function of the cell indices as in the kernel above, so both kernels agree bit for bit).
// ---------------------------------------------------------------------------------------------------------
constexpr int RW_NQ = 8, RW_RED = 32;
template <int TE> constexpr int rw_ks() { return TE + 1; }
static size_t rw_smem_bytes(int ncell, int TE, int OB) {
    const size_t keys = (((size_t)ncell * (TE + 1) * 4 + 15) & ~(size_t)15) + (size_t)ncell * 4 + 6 * (size_t)TE * 4;
    const size_t red = (size_t)RW_NQ * RW_RED * OB * 4;
    return keys > red ? keys : red;
}

template <int C, int TE, int CPL TNP_ABL_TPARAM>
__global__ void __launch_bounds__(64 * RW_NQ) pool_embed_regwide_kernel(const SparseArgs a) {
    constexpr int NQ = RW_NQ, NTH = 64 * NQ, NW = NQ, OB = 64 * CPL, KS = TE + 1, NH = TE / 32, NE64 = TE / 64;
    static_assert(TE == 64 || TE == 128, "tile");
    static_assert(TE * CPL == 128, "128 accumulators per lane");
    static_assert(C == 4 || C == 8 || C == 16, "channels");
    extern __shared__ __attribute__((aligned(16))) float wsm[];
    int *keyT = reinterpret_cast<int *>(wsm);                                        // [ncell][KS]
    int *socc = keyT + (((size_t)a.ncell * KS + 3) & ~(size_t)3);                   // [ncell] cell may have a hit in the tile
    int *sg = socc + a.ncell;                                                       // lo / ns / ki / pad [TE] each, then x / y
    float *sp = reinterpret_cast<float *>(sg + 4 * TE);
    float *red = wsm;                                                               // epilogue: [NQ][RW_RED][OB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int q = __builtin_amdgcn_readfirstlane(tid >> 6);                         // wave = cell group
    const int ob = blockIdx.x % a.out_blocks, tile = blockIdx.x / a.out_blocks;     // blocks b, b+8, .. share an XCD
    const int row0 = tile * TE;
#ifdef TNP_EXPERIMENT_HOOKS
    long long *dbg = nullptr;
    if constexpr (TNP_ABL(256)) dbg = reinterpret_cast<long long *>(const_cast<int16_t *>(a.winners));
#define RW_T(k) do { if constexpr (TNP_ABL(256)) { if (lane == 0) dbg[(blockIdx.x * 16 + q) * 8 + (k)] = (long long)__builtin_readcyclecounter(); } } while (0)
#else
#define RW_T(k) do { } while (0)
#endif
    RW_T(0);

    for (int c = tid; c < a.ncell; c += NTH) socc[c] = 0;
    {
        const int4 m1 = {-1, -1, -1, -1};
        const int nk4 = (a.ncell * KS + 3) / 4;
        for (int idx = tid; idx < nk4; idx += NTH) reinterpret_cast<int4 *>(keyT)[idx] = m1;
    }
    if (tid < TE) {                                                                 // scene geometry of the tile's egos
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
    RW_T(1);
    if constexpr (!TNP_ABL(32)) {
        // ---- votes: the reference's exact fp32 cell arithmetic (gridbased_pooling.py:276-288), key = 2 j + in_range ----
        constexpr int EU = TE / NW;                                                 // egos per wave: 16 / 8
        const float fG = (float)a.G;
        const int e0 = q * EU;
        const int ns_l = sg[TE + e0 + (lane & (EU - 1))];
        const bool small = __builtin_amdgcn_readfirstlane((int)(__ballot(ns_l > 32) == 0ull)) != 0;
        auto cell_of_pair = [&](float2 pj, float px, float py, bool &inr) -> int {
            if (pj.x != pj.x || pj.y != pj.y) { pj.x = -500.0f; pj.y = -500.0f; }
            const float ox = __fadd_rn(__fdiv_rn(__fsub_rn(pj.x, px), a.cell), a.half_x);
            const float oy = __fadd_rn(__fdiv_rn(__fsub_rn(pj.y, py), a.cell), a.half_y);
            inr = !(ox < 0.0f) && !(ox >= fG) && !(oy < 0.0f) && !(oy >= fG);
            return inr ? ((int)ox * a.G + (int)oy) : 0;
        };
        if (small) {
            // two egos per pass (lanes 0-31 / 32-63): every scene of this wave has at most 32 tracks.  All geometry reads
            // and neighbour positions of the wave's passes are in flight together.
            const int hi = lane >> 5, j = lane & 31;
            int lo_[EU / 2], ns_[EU / 2], ki_[EU / 2], pad_[EU / 2];
            float px_[EU / 2], py_[EU / 2];
            float2 pq[EU / 2];
#pragma unroll
            for (int p = 0; p < EU / 2; ++p) {
                const int e = e0 + 2 * p + hi;
                lo_[p] = sg[e]; ns_[p] = sg[TE + e]; ki_[p] = sg[2 * TE + e]; pad_[p] = sg[3 * TE + e];
                px_[p] = sp[e]; py_[p] = sp[TE + e];
            }
#pragma unroll
            for (int p = 0; p < EU / 2; ++p)
                pq[p] = reinterpret_cast<const float2 *>(a.obs2)[lo_[p] + (j < ns_[p] ? j : 0)];
#pragma unroll
            for (int p = 0; p < EU / 2; ++p) {
                const int e = e0 + 2 * p + hi;
                const bool valid = j < ns_[p] && j != ki_[p];
                bool inr;
                const int cellid = cell_of_pair(pq[p], px_[p], py_[p], inr);
                if (valid && inr) { atomicMax(&keyT[cellid * KS + e], 2 * j + 1); socc[cellid] = 1; }
                // neighbours that clobber cell 0 (out of range / absent; padded slots are the highest j of all): one vote
                const unsigned long long oor = __ballot(valid && !inr);
                const unsigned mh = hi ? (unsigned)(oor >> 32) : (unsigned)oor;
                int k0 = mh ? 2 * (31 - __builtin_clz(mh)) : -1;
                if (ns_[p] < pad_[p]) k0 = 2 * (pad_[p] - 1);
                if (j == 0 && k0 >= 0) atomicMax(&keyT[e], k0);
            }
        } else {
            for (int u = 0; u < EU; ++u) {
                const int e = e0 + u;
                const int lo = sg[e], ns = sg[TE + e], ki = sg[2 * TE + e], pad = sg[3 * TE + e];
                const float px = sp[e], py = sp[TE + e];
                for (int j = lane; j < ns; j += 64) {
                    bool inr;
                    const int cellid = cell_of_pair(reinterpret_cast<const float2 *>(a.obs2)[lo + j], px, py, inr);
                    if (j != ki) {
                        atomicMax(&keyT[cellid * KS + e], 2 * j + (inr ? 1 : 0));
                        if (inr) socc[cellid] = 1;
                    }
                }
                if (ns < pad && lane == 0) atomicMax(&keyT[e], 2 * (pad - 1));
            }
        }
    }
    __syncthreads();
    RW_T(2);
    if (a.winners_out && ob == 0) {                                                 // training: the winner table for the backward
        for (int e = q; e < TE; e += NW) {
            const int row = row0 + e;
            if (row >= a.M) continue;
            for (int c = lane; c < a.ncell; c += 64) {
                const int kq = keyT[c * KS + e];
                a.winners_out[(size_t)row * a.ncell + c] = (kq >= 0 && (kq & 1)) ? (int16_t)(kq >> 1) : (int16_t)-1;
            }
        }
    }
    int rb[NE64];
#pragma unroll
    for (int h = 0; h < NE64; ++h) { rb[h] = sg[64 * h + lane]; asm volatile("" : "+v"(rb[h])); }
    float bias_l[CPL];                                                              // fetched here: after the cell loop its latency would be exposed
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
        const int o = ob * OB + 64 * k + lane;
        bias_l[k] = (a.bias && o < a.N1) ? a.bias[o] : 0.0f;
        asm volatile("" : "+v"(bias_l[k]));                                         // landed before the loop (a compiler-placed vmcnt(0) inside it would drain the weight prefetch)
    }
    RW_T(3);

    // accumulators: vector v = 32 egos x one column of this lane; TE = 128: v <-> egos 32 v .. 32 v + 31;
    // TE = 64, CPL = 2: v = 2 * (ego half) + column
    ra_f32x32 acc[4];
#pragma unroll
    for (int v = 0; v < 4; ++v)
#pragma unroll
        for (int i = 0; i < 32; ++i) acc[v][i] = 0.0f;

    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    struct WS { f4 q[CPL][C / 4]; };
    const unsigned vl = lane * 16u;
    auto load_w = [&](WS &ws, int c) {
        if constexpr (TNP_ABL(1)) {
#pragma unroll
            for (int k = 0; k < CPL; ++k)
#pragma unroll
                for (int kq = 0; kq < C / 4; ++kq) ws.q[k][kq] = f4{1.0f, 2.0f, 3.0f, 4.0f};
        } else {
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                const int cb = min(ob * CPL + k, (a.N1 >> 6) - 1);                  // a column set past N1 re-reads the last block
                const float *wb = a.Wp + ((size_t)c * (a.N1 >> 6) + (size_t)cb) * (C * 64);
#pragma unroll
                for (int kq = 0; kq < C / 4; ++kq)
                    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(ws.q[k][kq]) : "v"(vl), "s"(wb), "i"(kq * 1024) : "memory");
            }
        }
    };
    // "this set has landed, the younger set may still be in flight": every load_w issues exactly CPL * C / 4 loads
    auto wait_w = [&](WS &ws) {
        if constexpr (TNP_ABL(1)) {
        } else {
            constexpr int NL = CPL * C / 4;
            static_assert(NL == 1 || NL == 2 || NL == 4 || NL == 8, "vmcnt immediate");
            if constexpr (NL == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if constexpr (NL == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (NL == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
#pragma unroll
            for (int k = 0; k < CPL; ++k)
#pragma unroll
                for (int kq = 0; kq < C / 4; ++kq) asm volatile("" : "+v"(ws.q[k][kq]));
        }
    };
    // one hit, one column: acc += sum_ch w[ch] * e[ch] (even / odd channels in the two halves of a packed register)
    auto fin = [&](const WS &w, int k, const typename SRow<C>::type &ev, float av) -> float {
        f2 p = {av, 0.0f};
#pragma unroll
        for (int i = 0; i < C / 2; ++i) {
            const f2 wk = {w.q[k][(2 * i) >> 2][(2 * i) & 3], w.q[k][(2 * i + 1) >> 2][(2 * i + 1) & 3]};
            const f2 ek = {ev[2 * i], ev[2 * i + 1]};
            p = __builtin_elementwise_fma(wk, ek, p);
        }
        return p.x + p.y;
    };
    // harness ablations of the hit chain: 4 = no neighbour-row loads, 8 = accumulator index fixed at 0 (no register-index
    // switches), 64 = no FMAs
    auto row = [&](typename SRow<C>::type &r, unsigned off, int ln) {
        if constexpr (TNP_ABL(4)) asm volatile("" : "=s"(r) : "s"(__builtin_amdgcn_readlane((int)off, ln)));
        else sload_row<C>(r, a.enc, (unsigned)__builtin_amdgcn_readlane((int)off, ln));
    };
    auto fin2 = [&](const WS &w, int k, const typename SRow<C>::type &ev, float av) -> float {
        if constexpr (TNP_ABL(64)) return av + ev[0];
        else return fin(w, k, ev, av);
    };
    // the hits of 32 egos (bits of `m`, lanes lane0 .. lane0 + 31 of `off`) against accumulator vectors v0 (column 0) and,
    // for CPL = 2, v0 + 1 (column 1).  Keep the exact read / chain / write shape (see the kernel above).
    auto half = [&](const WS &w, ra_f32x32 &accx, ra_f32x32 &accy, unsigned m, unsigned off, int lane0) {
        while (m & (m - 1u)) {                                                      // two hits: both rows in flight before the wait
            const int b0 = __builtin_ctz(m); m &= m - 1u;
            const int b1 = __builtin_ctz(m); m &= m - 1u;
            const int i0 = TNP_ABL(8) ? 0 : b0, i1 = TNP_ABL(8) ? 1 : b1;
            typename SRow<C>::type r0, r1;
            row(r0, off, lane0 + b0);
            row(r1, off, lane0 + b1);
            const float x0 = accx[i0], x1 = accx[i1];
            float y0 = 0.0f, y1 = 0.0f;
            if constexpr (CPL == 2) { y0 = accy[i0]; y1 = accy[i1]; }
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r0), "+s"(r1));
            accx[i0] = fin2(w, 0, r0, x0);
            accx[i1] = fin2(w, 0, r1, x1);
            if constexpr (CPL == 2) { accy[i0] = fin2(w, 1, r0, y0); accy[i1] = fin2(w, 1, r1, y1); }
        }
        if (m) {
            const int b0 = __builtin_ctz(m);
            const int i0 = TNP_ABL(8) ? 0 : b0;
            typename SRow<C>::type r0;
            row(r0, off, lane0 + b0);
            const float x0 = accx[i0];
            float y0 = 0.0f;
            if constexpr (CPL == 2) y0 = accy[i0];
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(r0));
            accx[i0] = fin2(w, 0, r0, x0);
            if constexpr (CPL == 2) accy[i0] = fin2(w, 1, r0, y0);
        }
    };
    auto process = [&](WS &w, int c) {
        int wv[NE64];
        unsigned long long mask[NE64];
#pragma unroll
        for (int h = 0; h < NE64; ++h) {
            const int kq = keyT[c * KS + 64 * h + lane];                            // lane <-> ego of the tile
            wv[h] = (kq >= 0 && (kq & 1)) ? (kq >> 1) : -1;
            mask[h] = __ballot(wv[h] >= 0);
        }
        wait_w(w);
        if constexpr (TNP_ABL(2)) { if (mask[0] == 1234567ull) acc[0][0] += w.q[0][0][0] + w.q[CPL - 1][C / 4 - 1][3]; return; }
#pragma unroll
        for (int h = 0; h < NE64; ++h) {
            const unsigned off = __umul24((unsigned)(rb[h] + wv[h]), (unsigned)(a.ldv * 4));
            if constexpr (CPL == 1) {
                half(w, acc[2 * h], acc[2 * h], (unsigned)mask[h], off, 0);
                half(w, acc[2 * h + 1], acc[2 * h + 1], (unsigned)(mask[h] >> 32), off, 32);
            } else {
                half(w, acc[0], acc[1], (unsigned)mask[h], off, 0);
                half(w, acc[2], acc[3], (unsigned)(mask[h] >> 32), off, 32);
            }
        }
    };
    // cells of this wave's group with a hit in the tile; group q takes cell NQ k + ((q - k) mod NQ) of every block k of NQ cells
    const int nk = (a.ncell + NQ - 1) / NQ;                                          // <= 64: ncell <= 512
    auto cell_of = [&](int k) { return NQ * k + ((q - k) & (NQ - 1)); };
    unsigned long long occ;
    {
        const int c = cell_of(lane);
        occ = __ballot(lane < nk && c < a.ncell && socc[c < a.ncell ? c : 0] != 0);
    }
    auto pop = [&]() -> int { if (!occ) return -1; const int k = pop_bit(occ); return cell_of(k); };
    WS wA, wB;
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
        if constexpr (!TNP_ABL(1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    RW_T(4);

    // ---- the 8 cell groups' partial sums, 32 (ego, column-set) rows per round: every wave leaves its partials in LDS, then
    //      wave w adds the 8 copies of rows w, w + 8, w + 16, w + 24 in fixed order (group 0 + 1 + ... + 7), bias + activation
    if constexpr (TNP_ABL(128)) { if (acc[0][3] + acc[3][5] == 1.234e30f) a.out[tid] = 0.0f; return; }
#pragma unroll
    for (int r = 0; r < NH; ++r) {                                                  // round r: egos 32 r .. 32 r + 31
        __syncthreads();                                                            // prologue data / previous round no longer read
#pragma unroll
        for (int k = 0; k < CPL; ++k)
#pragma unroll
            for (int e = 0; e < RW_RED; ++e)
                red[(q * RW_RED + e) * OB + 64 * k + lane] = acc[CPL == 1 ? r : 2 * r + k][e];
        __syncthreads();
#pragma unroll
        for (int h = 0; h < RW_RED / NW; ++h) {
            const int e = q + NW * h;
            const int row = row0 + RW_RED * r + e;
#pragma unroll
            for (int k = 0; k < CPL; ++k) {
                float v = red[(size_t)e * OB + 64 * k + lane];
#pragma unroll
                for (int qq = 1; qq < NQ; ++qq) v += red[((size_t)qq * RW_RED + e) * OB + 64 * k + lane];
                v += bias_l[k];
                if (a.relu) v = fmaxf(v, 0.0f);
                const int o = ob * OB + 64 * k + lane;
                if (row < a.M && o < a.N1) a.out[(size_t)row * a.ldo + o] = v;
            }
        }
    }
    RW_T(5);
#undef RW_T
}

bool regwide_supported(int C, int ncell, int TE, int OB) {
    return (C == 4 || C == 8 || C == 16) && ncell <= 64 * RW_NQ && rw_smem_bytes(ncell, TE, OB) <= (size_t)160 * 1024;
}

// ---------------------------------------------------------------------------------------------------------
// Shared-weight kernel (round 3): 128 egos x 64 columns per workgroup, 16 waves = 8 cell groups x 2 ego halves, 64 register
// accumulators per lane as in pool_embed_regacc_kernel (4 waves per SIMD: this loop is bound by instruction issue, it
// needs the waves), but the two waves of a cell group SHARE the group's weight blocks through LDS:
//   * the weights of a cell (C x 64 floats, contiguous in the quad-major copy) are fetched ONCE per workgroup by LDS-DMA
//     (global_load_lds_dwordx4, no VGPR round trip) into a three-slot ring per cell group -- the two waves take turns as
//     loader (visit v is loaded by the wave of half v & 1), two visits ahead -- and read by both waves with four
//     ds_read_b128: the L2 -> CU stream halves (0.55 -> 0.28 GB per launch at config 2) and the prefetch is two cells deep
//     without costing registers (one cell deep in the register-staged kernel, whose memory pipe idles during dense cells);
//   * hand-off through two LDS flags per wave: ready (my DMA of visit v has landed: s_waitcnt vmcnt(0), then the flag) and
//     done (I have read visit v's weights into registers); a loader overwrites a slot only when both waves are done with the
//     visit that used it three visits earlier.  The partner can be at most ~1.5 visits away; no workgroup barrier in the loop;
//   * winners are bytes [cell][ego] (scenes of at most 254 tracks): every half-wave votes one ego at a time into a PRIVATE
//     256-entry int table (integer max on key = 2 j + in_range, exactly the rule of the kernels above), reads the winners
//     back, writes the winning neighbour's byte and clears the entries it touched -- 34 KB instead of the 132 KB an int key
//     table for 128 egos would take, which is what makes room for the weight ring.
// Per ego the summation order is the one of the kernels above (cells of a group in ascending order, groups 0 + 1 + ... + 7).
// ---------------------------------------------------------------------------------------------------------
// LDS flag traffic of the shared-weight kernel as explicit DS instructions: a `volatile` C++ access through a pointer whose
// address space the compiler cannot prove becomes a FLAT access, and hipcc waits vmcnt(0) on those -- which would drain the
// LDS-DMA ring at every flag read.  `addr` = LDS byte address (low 32 bits of the shared-memory pointer).
__device__ __forceinline__ void lds_flag_store(unsigned addr, int v) {
    asm volatile("ds_write_b32 %0, %1" :: "v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ int lds_flag_load(unsigned addr) {
    int r;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    return __builtin_amdgcn_readfirstlane(r);
}

constexpr int RS_TE = 128, RS_OB = 64, RS_NQ = 8, RS_NH = 2, RS_SLOTS = 3, RS_WS = RS_TE + 4, RS_RED = 32;
static size_t rs_smem_bytes(int ncell, int C) {
    const size_t head = (size_t)ncell * RS_WS + (size_t)ncell * 4 + 6 * RS_TE * 4 + 4 * RS_NQ * RS_NH * 4;
    const size_t ring = (size_t)RS_NQ * RS_SLOTS * C * 64 * 4;
    const size_t votes = (size_t)RS_NQ * RS_NH * 2 * ncell * 4;
    const size_t red = (size_t)RS_NQ * RS_RED * RS_OB * 4;
    size_t body = ring > votes ? ring : votes;
    if (red > body) body = red;
    return ((head + 15) & ~(size_t)15) + body;
}

template <int C TNP_ABL_TPARAM>
__global__ void __launch_bounds__(64 * RS_NQ * RS_NH) pool_embed_regshare_kernel(const SparseArgs a) {
    constexpr int TE = RS_TE, OB = RS_OB, NQ = RS_NQ, NW = RS_NQ * RS_NH, NTH = 64 * NW, WS = RS_WS, SLOT = C * 64;   // floats per slot
    static_assert(C == 4 || C == 8 || C == 16, "channels");
    extern __shared__ __attribute__((aligned(16))) unsigned char ssm8[];
    unsigned char *win8 = ssm8;                                                      // [ncell][WS]: winning neighbour, 255 = none
    int *socc = reinterpret_cast<int *>(ssm8 + (size_t)a.ncell * WS);               // [ncell] cell may have a hit in the tile
    int *sg = socc + a.ncell;                                                       // lo / ns / ki / pad [TE] each, then x / y
    float *sp = reinterpret_cast<float *>(sg + 4 * TE);
    int *flag = sg + 6 * TE;                                                        // ready[q][h], done[q][h]
    const size_t head = (((size_t)a.ncell * WS + (size_t)a.ncell * 4 + 6 * TE * 4 + 4 * NW * 4) + 15) & ~(size_t)15;
    float *ring = reinterpret_cast<float *>(ssm8 + head);                           // [NQ][SLOTS][SLOT] (main loop)
    int *vtab = reinterpret_cast<int *>(ssm8 + head);                               // [NW][2][ncell] (prologue)
    float *red = reinterpret_cast<float *>(ssm8 + head);                            // [NQ][RS_RED][OB] (epilogue)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = wave >> 1, hh = wave & 1;                                         // cell group, ego half
    const int ob = blockIdx.x % a.out_blocks, tile = blockIdx.x / a.out_blocks;     // blocks b, b+8, .. share an XCD
    const int row0 = tile * TE;
#ifdef TNP_EXPERIMENT_HOOKS
    long long *dbg = nullptr;
    if constexpr (TNP_ABL(256)) dbg = reinterpret_cast<long long *>(const_cast<int16_t *>(a.winners));
#define RS_T(k) do { if constexpr (TNP_ABL(256)) { if (lane == 0) dbg[(blockIdx.x * 16 + wave) * 8 + (k)] = (long long)__builtin_readcyclecounter(); } } while (0)
#else
#define RS_T(k) do { } while (0)
#endif
    RS_T(0);

    // ---- init: winners = none, vote tables = -1, flags = -1, scene geometry of the tile's egos ----
    for (int c = tid; c < a.ncell; c += NTH) socc[c] = 0;
    {
        const int4 ff = {-1, -1, -1, -1};
        const int nw4 = (a.ncell * WS) / 16;                                         // WS % 4 == 0 and ncell % 4 == 0 (checked by the launcher)
        for (int idx = tid; idx < nw4; idx += NTH) reinterpret_cast<int4 *>(win8)[idx] = ff;
        const int nv4 = NW * 2 * a.ncell / 4;
        for (int idx = tid; idx < nv4; idx += NTH) reinterpret_cast<int4 *>(vtab)[idx] = ff;
    }
    if (tid < 4 * NW) flag[tid] = -1;
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
    RS_T(1);
    if constexpr (!TNP_ABL(32)) {
        // ---- votes: the reference's exact fp32 cell arithmetic (gridbased_pooling.py:276-288), key = 2 j + in_range ----
        constexpr int EU = TE / NW;                                                 // 8 egos per wave
        const float fG = (float)a.G;
        const int e0 = wave * EU;
        const int ns_l = sg[TE + e0 + (lane & (EU - 1))];
        const bool small = __builtin_amdgcn_readfirstlane((int)(__ballot(ns_l > 32) == 0ull)) != 0;
        auto cell_of_pair = [&](float2 pj, float px, float py, bool &inr) -> int {
            if (pj.x != pj.x || pj.y != pj.y) { pj.x = -500.0f; pj.y = -500.0f; }
            const float ox = __fadd_rn(__fdiv_rn(__fsub_rn(pj.x, px), a.cell), a.half_x);
            const float oy = __fadd_rn(__fdiv_rn(__fsub_rn(pj.y, py), a.cell), a.half_y);
            inr = !(ox < 0.0f) && !(ox >= fG) && !(oy < 0.0f) && !(oy >= fG);
            return inr ? ((int)ox * a.G + (int)oy) : 0;
        };
        if (small) {
            // two egos per pass (lanes 0-31 / 32-63), each half-wave with its own vote table
            const int hi = lane >> 5, j = lane & 31;
            int *tab = vtab + (size_t)(wave * 2 + hi) * a.ncell;
            int lo_[EU / 2], ns_[EU / 2], ki_[EU / 2], pad_[EU / 2];
            float px_[EU / 2], py_[EU / 2];
            float2 pq[EU / 2];
#pragma unroll
            for (int p = 0; p < EU / 2; ++p) {
                const int e = e0 + 2 * p + hi;
                lo_[p] = sg[e]; ns_[p] = sg[TE + e]; ki_[p] = sg[2 * TE + e]; pad_[p] = sg[3 * TE + e];
                px_[p] = sp[e]; py_[p] = sp[TE + e];
            }
#pragma unroll
            for (int p = 0; p < EU / 2; ++p)
                pq[p] = reinterpret_cast<const float2 *>(a.obs2)[lo_[p] + (j < ns_[p] ? j : 0)];
#pragma unroll
            for (int p = 0; p < EU / 2; ++p) {
                const int e = e0 + 2 * p + hi;
                const bool valid = j < ns_[p] && j != ki_[p];
                bool inr;
                const int cellid = cell_of_pair(pq[p], px_[p], py_[p], inr);
                const bool vin = valid && inr;
                // neighbours that clobber cell 0 (out of range / absent; padded slots are the highest j of all): one vote
                const unsigned long long oor = __ballot(valid && !inr);
                const unsigned mh = hi ? (unsigned)(oor >> 32) : (unsigned)oor;
                int k0 = mh ? 2 * (31 - __builtin_clz(mh)) : -1;
                if (ns_[p] < pad_[p]) k0 = 2 * (pad_[p] - 1);
                if (vin) atomicMax(&tab[cellid], 2 * j + 1);
                if (j == 0 && k0 >= 0) atomicMax(&tab[0], k0);
                const int w = vin ? tab[cellid] : -1;                                // LDS operations of a wave complete in order
                if (vin && w == 2 * j + 1) { win8[cellid * WS + e] = (unsigned char)j; socc[cellid] = 1; }
                if (vin) tab[cellid] = -1;                                           // leave the table clean for the next ego
                if (j == 0) tab[0] = -1;
            }
        } else {
            int *tab = vtab + (size_t)(wave * 2) * a.ncell;
            for (int u = 0; u < EU; ++u) {
                const int e = e0 + u;
                const int lo = sg[e], ns = sg[TE + e], ki = sg[2 * TE + e], pad = sg[3 * TE + e];
                const float px = sp[e], py = sp[TE + e];
                for (int j = lane; j < ns; j += 64) {
                    bool inr;
                    const int cellid = cell_of_pair(reinterpret_cast<const float2 *>(a.obs2)[lo + j], px, py, inr);
                    if (j != ki) atomicMax(&tab[cellid], 2 * j + (inr ? 1 : 0));
                }
                if (ns < pad && lane == 0) atomicMax(&tab[0], 2 * (pad - 1));
                for (int c = lane; c < a.ncell; c += 64) {                            // winners of this ego, table cleared
                    const int w = tab[c];
                    if (w >= 0) {
                        if (w & 1) { win8[c * WS + e] = (unsigned char)(w >> 1); socc[c] = 1; }
                        tab[c] = -1;
                    }
                }
            }
        }
    }
    __syncthreads();
    RS_T(2);
    if (a.winners_out && ob == 0) {                                                 // training: the winner table for the backward
        for (int e = wave; e < TE; e += NW) {
            const int row = row0 + e;
            if (row >= a.M) continue;
            for (int c = lane; c < a.ncell; c += 64) {
                const int w = win8[c * WS + e];
                a.winners_out[(size_t)row * a.ncell + c] = w == 255 ? (int16_t)-1 : (int16_t)w;
            }
        }
    }
    int rb = sg[64 * hh + lane];
    asm volatile("" : "+v"(rb));
    const int ocol = ob * OB + lane;
    float bias_l = (a.bias && ocol < a.N1) ? a.bias[ocol] : 0.0f;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias_l) :: "memory");                   // every compiler-tracked global load has landed: from here
    RS_T(3);                                                                         // on vmcnt counts LDS-DMA batches only

    ra_f32x32 accA, accB;                                                           // egos 64 hh + 0..31 / + 32..63, this lane's column
#pragma unroll
    for (int i = 0; i < 32; ++i) { accA[i] = 0.0f; accB[i] = 0.0f; }

    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    struct WS_ { f4 q[C / 4]; };
    float *myring = ring + (size_t)q * RS_SLOTS * SLOT;
    const int cbk = min(ob, (a.N1 >> 6) - 1);
    // LDS-DMA of the weights of cell c into slot s: C / 4 instructions of 1 KB (lane l: 16 bytes at l * 16), destination
    // lane-linear = the quad-major source order
    auto dma = [&](int c, int s) {
        if constexpr (!TNP_ABL(1)) {
            const float *gsrc = a.Wp + ((size_t)c * (a.N1 >> 6) + (size_t)cbk) * (C * 64) + lane * 4;
            const unsigned lds_dst = (unsigned)(uintptr_t)(myring + (size_t)s * SLOT);   // low 32 bits of an LDS pointer = its LDS byte address
            unsigned keep;
            // inline asm on purpose: hipcc counts a __builtin_amdgcn_global_load_lds as a VMEM operation it must wait for
            // (vmcnt(0)) before ANY later LDS read, which would drain the ring at every key / flag read.  M0 (the DMA's LDS
            // base) is written and restored inside the statement; the immediate offset applies to both addresses.
            if constexpr (C == 16)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                             "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\t"
                             "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
            else if constexpr (C == 8)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                             "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
            else
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %1, off\n\t"
                             "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
        }
    };
    auto fin = [&](const WS_ &w, const typename SRow<C>::type &ev, float av) -> float {
        f2 p = {av, 0.0f};
#pragma unroll
        for (int k = 0; k < C / 2; ++k) {
            const f2 wk = {w.q[(2 * k) >> 2][(2 * k) & 3], w.q[(2 * k + 1) >> 2][(2 * k + 1) & 3]};
            const f2 ek = {ev[2 * k], ev[2 * k + 1]};
            p = __builtin_elementwise_fma(wk, ek, p);
        }
        return p.x + p.y;
    };
    auto half = [&](const WS_ &w, ra_f32x32 &acc, unsigned m, unsigned off, int lane0) {
        while (m & (m - 1u)) {                                                      // two hits: both rows in flight before the wait
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
    // cells of this group with a hit in the TILE (both halves: the two waves of a group walk the same visit sequence);
    // group q takes cell NQ k + ((q - k) mod NQ) of every block k of NQ cells
    const int nk = (a.ncell + NQ - 1) / NQ;
    auto cell_of = [&](int k) { return NQ * k + ((q - k) & (NQ - 1)); };
    unsigned long long occ;
    {
        const int c = cell_of(lane);
        occ = __ballot(lane < nk && c < a.ncell && socc[c < a.ncell ? c : 0] != 0);
    }
    if constexpr (TNP_ABL(16)) occ = 0ull;
    const int nvis = __popcll(occ);
    // visit v -> cell: the v-th set bit of occ; the loader of visit v is the wave of half v & 1
    unsigned long long o_ld = occ;                                                   // bits not yet issued as DMA by EITHER wave (tracked identically)
    int v_ld = 0;                                                                    // next visit whose DMA has not been issued
    auto issue_upto = [&](int vmax) {                                                // issue my DMAs for visits v_ld .. vmax
        while (v_ld <= vmax && o_ld) {
            const int k = pop_bit(o_ld);
            if ((v_ld & 1) == hh) dma(cell_of(k), v_ld % RS_SLOTS);
            ++v_ld;
        }
    };
    const unsigned flag0 = (unsigned)(uintptr_t)flag;
    const unsigned ready_me = flag0 + 4u * (q * 2 + hh), ready_ot = flag0 + 4u * (q * 2 + (hh ^ 1));
    const unsigned done_me = flag0 + 4u * (2 * NW + q * 2 + hh), done_ot = flag0 + 4u * (2 * NW + q * 2 + (hh ^ 1));
    issue_upto(1);                                                                   // visits 0 and 1: slots free at start
    unsigned long long o_v = occ;
    for (int v = 0; v < nvis; ++v) {
        const int c = cell_of(pop_bit(o_v));
        // (1) the weights of visit v are in LDS
        if ((v & 1) == hh) {
            if constexpr (!TNP_ABL(1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // my only outstanding DMA batch is visit v's
            lds_flag_store(ready_me, v);
        } else {
            if constexpr (!TNP_ABL(1)) while (lds_flag_load(ready_ot) < v) __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        // (2) into registers
        WS_ w;
        {
            const f4 *src = reinterpret_cast<const f4 *>(myring + (size_t)(v % RS_SLOTS) * SLOT) + lane;
#pragma unroll
            for (int kq = 0; kq < C / 4; ++kq) {
                if constexpr (TNP_ABL(512)) w.q[kq] = f4{1.0f, 2.0f, 3.0f, (float)kq};   // harness: no LDS weight reads
                else w.q[kq] = src[kq * 64];
            }
        }
        // the keys of the cell while the weights arrive from LDS
        const int wv8 = win8[c * WS + 64 * hh + lane];
        const int wv = wv8 == 255 ? -1 : wv8;
        const unsigned long long mask = __ballot(wv >= 0);
#pragma unroll
        for (int kq = 0; kq < C / 4; ++kq) asm volatile("" : "+v"(w.q[kq]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // (3) I no longer need the slot of visit v
        lds_flag_store(done_me, v);
        // (4) prefetch visit v + 2 (same parity: mine to load iff v is) into the slot of visit v - 1, once the partner has read it
        if ((v & 1) == hh && v + 2 < nvis) {
            if constexpr (!TNP_ABL(1)) while (lds_flag_load(done_ot) < v - 1) __builtin_amdgcn_s_sleep(1);
        }
        issue_upto(v + 2);
        // (5) the hits of my 64 egos
        if constexpr (TNP_ABL(2)) { if (mask == 1234567ull) accA[0] += w.q[0][0] + w.q[C / 4 - 1][3]; continue; }
        const unsigned off = __umul24((unsigned)(rb + wv), (unsigned)(a.ldv * 4));
        half(w, accA, (unsigned)mask, off, 0);
        half(w, accB, (unsigned)(mask >> 32), off, 32);
    }
    RS_T(4);
    if constexpr (TNP_ABL(128)) { if (accA[3] + accB[5] == 1.234e30f) a.out[tid] = 0.0f; return; }

    // ---- the 8 cell groups' partial sums, 32 egos per round (round r: half r / 2, accumulator vector r % 2): the 8 waves
    //      of that half leave their partials in LDS, then wave w adds the copies of egos w and w + 16 in fixed order
    //      (group 0 + 1 + ... + 7), bias + activation, coalesced rows
#pragma unroll
    for (int r = 0; r < TE / RS_RED; ++r) {
        __syncthreads();                                                            // ring / previous round no longer read
        if ((r >> 1) == hh) {
#pragma unroll
            for (int e = 0; e < RS_RED; ++e) red[(q * RS_RED + e) * OB + lane] = (r & 1) ? accB[e] : accA[e];
        }
        __syncthreads();
#pragma unroll
        for (int h = 0; h < RS_RED / NW; ++h) {
            const int e = wave + NW * h;
            float v = red[(size_t)e * OB + lane];
#pragma unroll
            for (int qq = 1; qq < NQ; ++qq) v += red[((size_t)qq * RS_RED + e) * OB + lane];
            v += bias_l;
            if (a.relu) v = fmaxf(v, 0.0f);
            const int row = row0 + RS_RED * r + e;
            if (row < a.M && ocol < a.N1) a.out[(size_t)row * a.ldo + ocol] = v;
        }
    }
    RS_T(5);
#undef RS_T
}

bool regshare_supported(int C, int ncell, int n_max) {
    return (C == 4 || C == 8 || C == 16) && ncell <= 64 * RS_NQ && ncell % 4 == 0 && n_max <= 254 &&
           rs_smem_bytes(ncell, C) <= (size_t)160 * 1024;
}



// ---------------------------------------------------------------------------------------------------------
// Ring kernel: the product kernel's tile (64 egos x 128 columns, 16 waves = 8 cell groups x 2 column sets, 64 register
// accumulators per lane), but every wave streams its weight blocks through a PRIVATE two-slot LDS ring filled by LDS-DMA two
// cells ahead (no inter-wave hand-off: a wave reads only what it loaded itself), and holds ONE weight register set that it
// refills from LDS at the top of a visit.  The register-staged product kernel prefetches one cell ahead: its memory pipe runs
// dry while a wave works through a dense cell, and stream (16.5 us) and hits (14.6 us) add up to 25 us instead of overlapping.
// LDS: winners as bytes (17 KB, voted through per-half-wave tables as in the shared-weight kernel) + 16 x 2 x 4 KB of ring.
// ---------------------------------------------------------------------------------------------------------
constexpr int RG_TE = 64, RG_OB = 128, RG_NQ = 8, RG_NCS = 2, RG_WS = RG_TE + 4, RG_RED = 32;
static size_t rg_smem_bytes(int ncell, int C) {
    const size_t head = (((size_t)ncell * RG_WS + (size_t)ncell * 4 + 6 * RG_TE * 4) + 15) & ~(size_t)15;
    const size_t ring = (size_t)RG_NQ * RG_NCS * 2 * C * 64 * 4;
    const size_t votes = (size_t)RG_NQ * RG_NCS * 2 * ncell * 4;
    const size_t red = (size_t)RG_NQ * RG_RED * RG_OB * 4;
    size_t body = ring > votes ? ring : votes;
    if (red > body) body = red;
    return head + body;
}

template <int C, bool DB TNP_ABL_TPARAM>
__global__ void __launch_bounds__(64 * RG_NQ * RG_NCS) pool_embed_regring_kernel(const SparseArgs a) {
    constexpr int TE = RG_TE, OB = RG_OB, NQ = RG_NQ, NCS = RG_NCS, NW = NQ * NCS, NTH = 64 * NW, WS = RG_WS, SLOT = C * 64;
    static_assert(C == 4 || C == 8 || C == 16, "channels");
    extern __shared__ __attribute__((aligned(16))) unsigned char gsm8[];
    unsigned char *win8 = gsm8;                                                      // [ncell][WS]: winning neighbour, 255 = none
    int *socc = reinterpret_cast<int *>(gsm8 + (size_t)a.ncell * WS);
    int *sg = socc + a.ncell;
    float *sp = reinterpret_cast<float *>(sg + 4 * TE);
    const size_t head = (((size_t)a.ncell * WS + (size_t)a.ncell * 4 + 6 * TE * 4) + 15) & ~(size_t)15;
    float *ring = reinterpret_cast<float *>(gsm8 + head);                            // [NW][2][SLOT] (main loop)
    int *vtab = reinterpret_cast<int *>(gsm8 + head);                                // [NW][2][ncell] (prologue)
    float *red = reinterpret_cast<float *>(gsm8 + head);                             // [NQ][RG_RED][OB] (epilogue)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cs = wave % NCS, q = wave / NCS;
    const int ob = blockIdx.x % a.out_blocks, tile = blockIdx.x / a.out_blocks;
    const int row0 = tile * TE;
    long long *dbg = nullptr;
    if constexpr (TNP_ABL(256)) dbg = reinterpret_cast<long long *>(const_cast<int16_t *>(a.winners));
#define RG_T(k) do { if constexpr (TNP_ABL(256)) { if (lane == 0) dbg[(blockIdx.x * 16 + wave) * 8 + (k)] = (long long)__builtin_readcyclecounter(); } } while (0)
    RG_T(0);
    for (int c = tid; c < a.ncell; c += NTH) socc[c] = 0;
    {
        const int4 ff = {-1, -1, -1, -1};
        const int nw4 = (a.ncell * WS) / 16;
        for (int idx = tid; idx < nw4; idx += NTH) reinterpret_cast<int4 *>(win8)[idx] = ff;
        const int nv4 = NW * 2 * a.ncell / 4;
        for (int idx = tid; idx < nv4; idx += NTH) reinterpret_cast<int4 *>(vtab)[idx] = ff;
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
    RG_T(1);
    if constexpr (!TNP_ABL(32)) {
        constexpr int EU = TE / NW;                                                 // 4 egos per wave
        const float fG = (float)a.G;
        const int e0 = wave * EU;
        const int ns_l = sg[TE + e0 + (lane & (EU - 1))];
        const bool small = __builtin_amdgcn_readfirstlane((int)(__ballot(ns_l > 32) == 0ull)) != 0;
        auto cell_of_pair = [&](float2 pj, float px, float py, bool &inr) -> int {
            if (pj.x != pj.x || pj.y != pj.y) { pj.x = -500.0f; pj.y = -500.0f; }
            const float ox = __fadd_rn(__fdiv_rn(__fsub_rn(pj.x, px), a.cell), a.half_x);
            const float oy = __fadd_rn(__fdiv_rn(__fsub_rn(pj.y, py), a.cell), a.half_y);
            inr = !(ox < 0.0f) && !(ox >= fG) && !(oy < 0.0f) && !(oy >= fG);
            return inr ? ((int)ox * a.G + (int)oy) : 0;
        };
        if (small) {
            const int hi = lane >> 5, j = lane & 31;
            int *tab = vtab + (size_t)(wave * 2 + hi) * a.ncell;
            int lo_[EU / 2], ns_[EU / 2], ki_[EU / 2], pad_[EU / 2];
            float px_[EU / 2], py_[EU / 2];
            float2 pq[EU / 2];
#pragma unroll
            for (int p = 0; p < EU / 2; ++p) {
                const int e = e0 + 2 * p + hi;
                lo_[p] = sg[e]; ns_[p] = sg[TE + e]; ki_[p] = sg[2 * TE + e]; pad_[p] = sg[3 * TE + e];
                px_[p] = sp[e]; py_[p] = sp[TE + e];
            }
#pragma unroll
            for (int p = 0; p < EU / 2; ++p)
                pq[p] = reinterpret_cast<const float2 *>(a.obs2)[lo_[p] + (j < ns_[p] ? j : 0)];
#pragma unroll
            for (int p = 0; p < EU / 2; ++p) {
                const int e = e0 + 2 * p + hi;
                const bool valid = j < ns_[p] && j != ki_[p];
                bool inr;
                const int cellid = cell_of_pair(pq[p], px_[p], py_[p], inr);
                const bool vin = valid && inr;
                const unsigned long long oor = __ballot(valid && !inr);
                const unsigned mh = hi ? (unsigned)(oor >> 32) : (unsigned)oor;
                int k0 = mh ? 2 * (31 - __builtin_clz(mh)) : -1;
                if (ns_[p] < pad_[p]) k0 = 2 * (pad_[p] - 1);
                if (vin) atomicMax(&tab[cellid], 2 * j + 1);
                if (j == 0 && k0 >= 0) atomicMax(&tab[0], k0);
                const int w = vin ? tab[cellid] : -1;
                if (vin && w == 2 * j + 1) { win8[cellid * WS + e] = (unsigned char)j; socc[cellid] = 1; }
                if (vin) tab[cellid] = -1;
                if (j == 0) tab[0] = -1;
            }
        } else {
            int *tab = vtab + (size_t)(wave * 2) * a.ncell;
            for (int u = 0; u < EU; ++u) {
                const int e = e0 + u;
                const int lo = sg[e], ns = sg[TE + e], ki = sg[2 * TE + e], pad = sg[3 * TE + e];
                const float px = sp[e], py = sp[TE + e];
                for (int j = lane; j < ns; j += 64) {
                    bool inr;
                    const int cellid = cell_of_pair(reinterpret_cast<const float2 *>(a.obs2)[lo + j], px, py, inr);
                    if (j != ki) atomicMax(&tab[cellid], 2 * j + (inr ? 1 : 0));
                }
                if (ns < pad && lane == 0) atomicMax(&tab[0], 2 * (pad - 1));
                for (int c = lane; c < a.ncell; c += 64) {
                    const int w = tab[c];
                    if (w >= 0) {
                        if (w & 1) { win8[c * WS + e] = (unsigned char)(w >> 1); socc[c] = 1; }
                        tab[c] = -1;
                    }
                }
            }
        }
    }
    __syncthreads();
    RG_T(2);
    if (!TNP_ABL(1024) && a.winners_out && ob == 0) {
        for (int e = wave; e < TE; e += NW) {
            const int row = row0 + e;
            if (row >= a.M) continue;
            for (int c = lane; c < a.ncell; c += 64) {
                const int w = win8[c * WS + e];
                a.winners_out[(size_t)row * a.ncell + c] = w == 255 ? (int16_t)-1 : (int16_t)w;
            }
        }
    }
    int rb = sg[lane];
    asm volatile("" : "+v"(rb));
    const int col2 = 2 * lane;
    float2 bias2 = {0.0f, 0.0f};
    if (a.bias && ob * OB + col2 + 1 < a.N1) bias2 = *reinterpret_cast<const float2 *>(a.bias + ob * OB + col2);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(bias2) :: "memory");                    // from here on vmcnt counts LDS-DMA batches only
    RG_T(3);
    ra_f32x32 accA, accB;
#pragma unroll
    for (int i = 0; i < 32; ++i) { accA[i] = 0.0f; accB[i] = 0.0f; }

    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    struct WS_ { f4 q[C / 4]; };
    float *myring = ring + (size_t)wave * 2 * SLOT;
    const unsigned lds0 = (unsigned)(uintptr_t)myring, lds1 = lds0 + SLOT * 4;
    const char *wq_base = reinterpret_cast<const char *>(a.Wp) + (size_t)min(ob * NCS + cs, (a.N1 >> 6) - 1) * (C * 64 * 4);
    const unsigned wq_stride = (unsigned)(a.N1 >> 6) * (C * 64 * 4);
    const unsigned vl = lane * 16u;
    // LDS-DMA of the weights of cell c into slot `lds`: C / 4 instructions of 1 KB, destination lane-linear = the quad-major
    // source order; M0 is set inside the statement (nothing else in this kernel depends on M0)
    auto dma = [&](int c, unsigned lds) {
        if constexpr (!TNP_ABL(1)) {
            const char *wb = wq_base + __builtin_amdgcn_readfirstlane((unsigned)c * wq_stride);
            if constexpr (C == 16)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                             "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072"
                             :: "v"(vl), "s"(wb), "s"(lds) : "memory");
            else if constexpr (C == 8)
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024"
                             :: "v"(vl), "s"(wb), "s"(lds) : "memory");
            else
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(vl), "s"(wb), "s"(lds) : "memory");
        }
    };
    auto wait_older_dma = [&]() {                                                   // the older of the two batches in flight has landed
        if constexpr (!TNP_ABL(1)) {
            if constexpr (C == 16) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if constexpr (C == 8) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        }
    };
    auto fin = [&](const WS_ &w, const typename SRow<C>::type &ev, float av) -> float {
        f2 p = {av, 0.0f};
#pragma unroll
        for (int k = 0; k < C / 2; ++k) {
            const f2 wk = {w.q[(2 * k) >> 2][(2 * k) & 3], w.q[(2 * k + 1) >> 2][(2 * k + 1) & 3]};
            const f2 ek = {ev[2 * k], ev[2 * k + 1]};
            p = __builtin_elementwise_fma(wk, ek, p);
        }
        return p.x + p.y;
    };
    auto half = [&](const WS_ &w, ra_f32x32 &acc, unsigned m, unsigned off, int lane0) {
        while (m & (m - 1u)) {
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
    const int nk = (a.ncell + NQ - 1) / NQ;
    auto cell_of = [&](int k) { return NQ * k + ((q - k) & (NQ - 1)); };
    unsigned long long occ;
    {
        const int c = cell_of(lane);
        occ = __ballot(lane < nk && c < a.ncell && socc[c < a.ncell ? c : 0] != 0);
    }
    if constexpr (TNP_ABL(16)) occ = 0ull;
    int left = __popcll(occ);
    auto lds_to_regs = [&](WS_ &w, const float *slot) {
        const f4 *src = reinterpret_cast<const f4 *>(slot) + lane;
#pragma unroll
        for (int kq = 0; kq < C / 4; ++kq) w.q[kq] = src[kq * 64];
    };
    if constexpr (!DB) {
    // one visit: the slot's weights into registers, the next-but-one cell's DMA into the same slot, the hits
    auto visit = [&](const float *slot, unsigned lds, int c_dma, int key8) {
        wait_older_dma();
        WS_ w;
        lds_to_regs(w, slot);
        const int wv = key8 == 255 ? -1 : key8;
        const unsigned long long mask = __ballot(wv >= 0);
#pragma unroll
        for (int kq = 0; kq < C / 4; ++kq) asm volatile("" : "+v"(w.q[kq]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                           // the slot has been read: it may be refilled
        dma(c_dma, lds);
        if constexpr (TNP_ABL(2)) { if (mask == 1234567ull) accA[0] += w.q[0][0] + w.q[C / 4 - 1][3]; return; }
        const unsigned off = __umul24((unsigned)(rb + wv), (unsigned)(a.ldv * 4));
        half(w, accA, (unsigned)mask, off, 0);
        half(w, accB, (unsigned)(mask >> 32), off, 32);
    };
    if (left > 0) {
        // cells of visits v, v + 1 (cA, cB); every DMA is issued unconditionally (past the end it re-reads the last cell) so that
        // the vmcnt arithmetic is exact
        int cA = cell_of(pop_bit(occ)), cB = cA;
        if (left > 1) cB = cell_of(pop_bit(occ));
        dma(cA, lds0);
        dma(cB, lds1);
        int kA = win8[cA * WS + lane], kB;
        while (true) {
            int cn = cB;
            if (left > 2) cn = cell_of(pop_bit(occ));
            kB = win8[cB * WS + lane];
            visit(myring, lds0, cn, kA);
            if (--left == 0) break;
            cA = cn;                                                                // cell of visit v + 2
            int cm = cA;
            if (left > 2) cm = cell_of(pop_bit(occ));
            kA = win8[cA * WS + lane];
            visit(myring + SLOT, lds1, cm, kB);
            if (--left == 0) break;
            cB = cm;
        }
        if constexpr (!TNP_ABL(1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    } else {
    // double-buffered registers: the weights of visit v + 1 go from their slot into the second register set between the two
    // halves of visit v (their LDS latency hides behind the second half's hits); the slot of visit v is refilled with visit
    // v + 2 at the top of visit v (its reads finished a visit ago)
    auto visit2 = [&](WS_ &w, WS_ &wn, const float *slot_next, unsigned lds_cur, int c_dma, int key8) {
#pragma unroll
        for (int kq = 0; kq < C / 4; ++kq) asm volatile("" : "+v"(w.q[kq]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                           // this visit's registers (read a visit ago) and key row
        dma(c_dma, lds_cur);                                                         // visit v + 2 into the slot visit v came from
        const int wv = key8 == 255 ? -1 : key8;
        const unsigned long long mask = __ballot(wv >= 0);
        const unsigned off = __umul24((unsigned)(rb + wv), (unsigned)(a.ldv * 4));
        if constexpr (!TNP_ABL(2)) half(w, accA, (unsigned)mask, off, 0);
        wait_older_dma();                                                           // visit v + 1's DMA (issued a visit ago) has landed
        lds_to_regs(wn, slot_next);
        if constexpr (TNP_ABL(2)) { if (mask == 1234567ull) accA[0] += w.q[0][0] + w.q[C / 4 - 1][3]; return; }
        half(w, accB, (unsigned)(mask >> 32), off, 32);
    };
    if (left > 0) {
        int cA = cell_of(pop_bit(occ)), cB = cA;
        if (left > 1) cB = cell_of(pop_bit(occ));
        dma(cA, lds0);
        dma(cB, lds1);
        WS_ wA, wB;
        wait_older_dma();
        lds_to_regs(wA, myring);
        int kA = win8[cA * WS + lane], kB;
        while (true) {
            int cn = cB;
            if (left > 2) cn = cell_of(pop_bit(occ));
            kB = win8[cB * WS + lane];
            visit2(wA, wB, myring + SLOT, lds0, cn, kA);
            if (--left == 0) break;
            cA = cn;
            int cm = cA;
            if (left > 2) cm = cell_of(pop_bit(occ));
            kA = win8[cA * WS + lane];
            visit2(wB, wA, myring, lds1, cm, kB);
            if (--left == 0) break;
            cB = cm;
        }
        if constexpr (!TNP_ABL(1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    }
    RG_T(4);
    if constexpr (TNP_ABL(128)) { if (accA[3] + accB[5] == 1.234e30f) a.out[tid] = 0.0f; return; }
#pragma unroll
    for (int r = 0; r < TE / RG_RED; ++r) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < RG_RED; ++e)
            red[(q * RG_RED + e) * OB + cs * 64 + lane] = r == 0 ? accA[e] : accB[e];
        __syncthreads();
#pragma unroll
        for (int h = 0; h < RG_RED / NW; ++h) {
            const int e = wave + NW * h;
            float2 v = *reinterpret_cast<const float2 *>(red + (size_t)e * OB + col2);
#pragma unroll
            for (int qq = 1; qq < NQ; ++qq) {
                const float2 t = *reinterpret_cast<const float2 *>(red + ((size_t)qq * RG_RED + e) * OB + col2);
                v.x += t.x; v.y += t.y;
            }
            v.x += bias2.x; v.y += bias2.y;
            if (a.relu) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); }
            const int row = row0 + RG_RED * r + e;
            if (row < a.M && ob * OB + col2 + 1 < a.N1) *reinterpret_cast<float2 *>(a.out + (size_t)row * a.ldo + ob * OB + col2) = v;
        }
    }
    RG_T(5);
#undef RG_T
}

}  // namespace tnp
