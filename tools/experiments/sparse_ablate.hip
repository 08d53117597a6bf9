// Timing harness for the sparse first layer (csrc/pool_embed_sparse.hip): runs the product kernel and its timing
// ablations (template parameter ABL) on a synthetic config-2 crowd, stand-alone (no torch).  Build + run:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Itrajnetplusplusbaselines_amd/csrc \
//       -Itools/experiments tools/experiments/sparse_ablate.hip -o tools/experiments/sparse_ablate && tools/experiments/sparse_ablate
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include <random>
#define TNP_EXPERIMENT_HOOKS 1   // compiles the timing ablations / clock stamps of the product kernels
#include "pool_embed_sparse.hip"
#include "pool_embed_variants.hip"

namespace tnp { void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
bool take_dispatch_events(hipEvent_t *, hipEvent_t *) { return false; } }   // the library's profiling hook (lstm_seq.hip)

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <typename T> T *dev(const std::vector<T> &h) {
    T *d; CK(hipMalloc(&d, h.size() * sizeof(T))); CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d;
}

typedef void (*kern_t)(const tnp::SparseArgs);

static float time_kernel(kern_t k, const tnp::SparseArgs &a, int blocks, size_t smem, int reps, int threads = 1024) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), smem, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), smem, 0, a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipGetLastError());
    return ms * 1000.0f / reps;
}

int main(int argc, char **argv) {
    const int scenes = 64, agents = 32, M = scenes * agents, G = 16, ncell = G * G, C = 16, N1 = 1024;
    const float tstep = argc > 1 ? atof(argv[1]) : 10.0f;
    const int only = argc > 2 ? atoi(argv[2]) : 0;                                  // 3: only the register-accumulator kernels
    setvbuf(stdout, nullptr, _IONBF, 0);
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-4.0f, 4.0f); std::normal_distribution<float> Nrm(0.0f, 1.0f);
    std::vector<float> obs(M * 2), enc((size_t)M * C), W((size_t)ncell * C * N1), bias(N1);
    for (int i = 0; i < M; ++i) for (int d = 0; d < 2; ++d) obs[2 * i + d] = U(rng) + 0.3f * Nrm(rng) * tstep;
    for (auto &v : enc) v = Nrm(rng);
    for (auto &v : W) v = 0.02f * Nrm(rng);
    for (auto &v : bias) v = 0.1f * Nrm(rng);
    std::vector<int32_t> rb(M), re(M), rp(M);
    for (int i = 0; i < M; ++i) { rb[i] = i / agents * agents; re[i] = rb[i] + agents; rp[i] = agents; }
    tnp::SparseArgs a{};
    a.winners = nullptr; a.enc = dev(enc); a.ldv = C; a.row_base = dev(rb); a.Wp = dev(W); a.bias = dev(bias);
    a.M = M; a.ncell = ncell; a.C = C; a.N1 = N1; a.S = 1; a.cps = ncell; a.ego_tiles = M / tnp::TL_TE; a.out_blocks = N1 / tnp::TL_OB;
    float *out; CK(hipMalloc(&out, (size_t)M * N1 * 4)); a.out = out; a.ldo = N1; a.relu = 1;
    a.obs2 = dev(obs); a.row_end = dev(re); a.row_padded = dev(rp); a.G = G; a.cell = 0.6f; a.half_x = a.half_y = G / 2.0f;
    a.winners_out = nullptr;
    const size_t smem = (size_t)tnp::TL_NQ * tnp::TL_TE * tnp::TL_OB * 4 + (((size_t)ncell * (tnp::TL_TE + 2) * 2 + 15) & ~(size_t)15) + ncell * 4;
    const int blocks = a.ego_tiles * a.out_blocks;
    // hits on the host (for the per-hit figures)
    long hits = 0;
    for (int i = 0; i < M; ++i) {
        std::vector<char> occ(ncell, 0);
        for (int j = rb[i]; j < re[i]; ++j) {
            if (j == i) continue;
            const float ox = (obs[2 * j] - obs[2 * i]) / 0.6f + 8.0f, oy = (obs[2 * j + 1] - obs[2 * i + 1]) / 0.6f + 8.0f;
            if (ox >= 0 && ox < G && oy >= 0 && oy < G) occ[(int)ox * G + (int)oy] = 1;
        }
        for (char o : occ) hits += o;
    }
    printf("M %d hits/ego %.2f blocks %d smem %zu\n", M, (double)hits / M, blocks, smem);
    // host reference of every 37th ego (last writer in ascending j wins a cell; out-of-range neighbours clobber cell 0)
    auto check = [&](const std::vector<float> &y, const char *name) {
        double md = 0;
        for (int i = 0; i < M; i += (getenv("FULLCHECK") ? 1 : 37)) {
            std::vector<int> win(ncell, -1);
            for (int j = rb[i]; j < re[i]; ++j) {
                if (j == i) continue;
                const float ox = (obs[2 * j] - obs[2 * i]) / 0.6f + 8.0f, oy = (obs[2 * j + 1] - obs[2 * i + 1]) / 0.6f + 8.0f;
                const bool inr = !(ox < 0) && !(ox >= G) && !(oy < 0) && !(oy >= G);
                win[inr ? (int)ox * G + (int)oy : 0] = inr ? j : -1;
            }
            for (int o = 0; o < N1; o += (getenv("FULLCHECK") ? 1 : 5)) {
                double acc = bias[o];
                for (int c = 0; c < ncell; ++c) if (win[c] >= 0)
                    for (int ch = 0; ch < C; ++ch) acc += (double)W[((size_t)c * C + ch) * N1 + o] * enc[(size_t)win[c] * C + ch];
                md = fmax(md, fabs(fmax(acc, 0.0) - y[(size_t)i * N1 + o]));
            }
        }
        printf("   %s: max |y - host reference| %.3g\n", name, md);
    };
    struct V { const char *name; kern_t k; };
#define KV(abl) (kern_t)tnp::pool_embed_cellsplit_kernel<16, true, true, abl>
    const V vs[] = {{"product", KV(0)}, {"no weight loads (1)", KV(1)}, {"no hits (2)", KV(2)}, {"no weights, no hits (3)", KV(3)},
                    {"no row loads (4)", KV(4)}, {"no acc rmw (8)", KV(8)}, {"no row loads, no rmw (12)", KV(12)},
                    {"no weights, no row loads, no rmw (13)", KV(13)}, {"no cell loop (16)", KV(16)}};
    std::vector<float> ref((size_t)M * N1), got((size_t)M * N1);
    for (const V &v : vs) {
        if (only == 3 && v.k != vs[0].k) continue;
        const float us = time_kernel(v.k, a, blocks, smem, 50);
        printf("%-42s %8.2f us\n", v.name, us);
        if (v.k == vs[0].k) { CK(hipMemcpy(ref.data(), out, ref.size() * 4, hipMemcpyDeviceToHost)); check(ref, "product"); }
    }
    {   // 64 egos x 128 columns, 4 cell groups x 2 column sets = 8 waves, int8 winner tile
        tnp::SparseArgs b = a; b.ego_tiles = M / 64; b.out_blocks = N1 / 128;
        const size_t smem2 = (size_t)4 * 64 * 128 * 4 + (((size_t)ncell * (64 + 4) + 15) & ~(size_t)15) + ncell * 4;
#define KW(abl) (kern_t)tnp::pool_embed_cellsplit_kernel<16, true, true, abl, 64, 128, 4, int8_t>
        const V ws[] = {{"64x128 8 waves", KW(0)}, {"64x128 no weight loads", KW(1)}, {"64x128 no hits", KW(2)}, {"64x128 no cell loop", KW(16)}};
        for (const V &v : ws) {
            if (only == 3) continue;
            const float us = time_kernel(v.k, b, b.ego_tiles * b.out_blocks, smem2, 50, 512);
            printf("%-42s %8.2f us\n", v.name, us);
            if (v.k == ws[0].k) {
                CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
                double md = 0; for (size_t i = 0; i < got.size(); ++i) md = fmax(md, fabs((double)got[i] - ref[i]));
                printf("   max |diff| vs product %.3g\n", md);
            }
        }
    }
    {   // register accumulators: 64 egos x 128 columns, 8 cell groups x 2 column sets = 16 waves
        tnp::SparseArgs b = a; b.ego_tiles = M / 64; b.out_blocks = N1 / 128;
        const size_t smem3 = tnp::ra_smem_bytes(ncell);
#define KR(abl) (kern_t)tnp::pool_embed_regacc_kernel<16, true, false, abl>
#define KQ(abl) (kern_t)tnp::pool_embed_regacc_kernel<16, true, true, abl>
        const V ws[] = {{"regacc", KR(0)}, {"regacc no weight loads", KR(1)}, {"regacc no hits", KR(2)}, {"regacc no cell loop", KR(16)},
                        {"regacc no loop, no votes", KR(48)}, {"regacc no loop, no epilogue", KR(144)}, {"regacc nothing (176)", KR(176)}};
        for (const V &v : ws) {
            const float us = time_kernel(v.k, b, b.ego_tiles * b.out_blocks, smem3, 50, 1024);
            printf("%-42s %8.2f us\n", v.name, us);
            if (v.k == ws[0].k) { CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost)); check(got, "regacc"); }
        }
    }
    tnp::SparseArgs a2 = a; a2.ego_tiles = M / 64; a2.out_blocks = N1 / 128;
    {   // quad-major weights [cell][N1/64][C/4][64][4] (what the sequence driver hands over)
        std::vector<float> W4(W.size());
        for (int c = 0; c < ncell; ++c) for (int ch = 0; ch < C; ++ch) for (int o = 0; o < N1; ++o)
            W4[((((size_t)c * (N1 / 64) + o / 64) * (C / 4) + ch / 4) * 64 + o % 64) * 4 + ch % 4] = W[((size_t)c * C + ch) * N1 + o];
        tnp::SparseArgs b = a2; b.Wp = dev(W4);
        const int nb = b.ego_tiles * b.out_blocks;
        const V ws[] = {{"regacc quad-major", KQ(0)}, {"regacc quad-major no weight loads", KQ(1)}, {"regacc quad-major no hits", KQ(2)},
                        {"regacc quad-major no cell loop", KQ(16)}};
        for (const V &v : ws) {
            printf("%-42s %8.2f us\n", v.name, time_kernel(v.k, b, nb, tnp::ra_smem_bytes(ncell), 50, 1024));
            if (v.k == ws[0].k) { CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost)); check(got, "regacc quad-major"); }
        }
        {   // determinism: repeated launches must agree bit for bit, with and without the winner table being written
            const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 20;
            int16_t *wout; CK(hipMalloc(&wout, (size_t)M * ncell * 2));
            std::vector<float> first(got.size()), again(got.size());
            struct T { const char *name; kern_t k; tnp::SparseArgs args; bool wo; };
            T ts[] = {{"quad-major", KQ(0), b, false}, {"quad-major + winners_out", KQ(0), b, true}, {"cell-major + winners_out", KR(0), a2, true}};
            int ti = -1;
            for (T &t : ts) {
                ++ti;
                if (getenv("ONLY") && atoi(getenv("ONLY")) != ti) continue;
                int bad = 0;
                t.args.winners_out = t.wo ? wout : nullptr;
                const bool cs = t.k == KV(0);
                for (int rep = 0; rep < reps; ++rep) {
                    CK(hipMemsetAsync(out, 0xff, got.size() * 4));
                    if (cs) hipLaunchKernelGGL(t.k, dim3(blocks), dim3(1024), smem, 0, t.args);
                    else hipLaunchKernelGGL(t.k, dim3(nb), dim3(1024), tnp::ra_smem_bytes(ncell), 0, t.args);
                    CK(hipMemcpy(rep ? again.data() : first.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
                    if (rep && memcmp(first.data(), again.data(), got.size() * 4)) {
                        size_t n = 0, f = 0; std::vector<char> rows(M, 0);
                        for (size_t i = 0; i < got.size(); ++i) if (memcmp(&first[i], &again[i], 4)) { if (!n) f = i; ++n; rows[i / N1] = 1; }
                        int nr = 0; for (char r : rows) nr += r;
                        if (bad < 4) printf("   rep %d: %zu elements in %d rows differ, first at row %zu col %zu (%g vs %g)\n", rep, n, nr, f / N1, f % N1, first[f], again[f]);
                        ++bad;
                    }
                }
                printf("   %-36s %d of %d repeats differ from the first\n", t.name, bad, reps - 1);
            }
        }
        {   // cost of folding track_prepare into the prologue: the product kernel + a prepare-like phase
            std::vector<float> hh((size_t)M * 128), w21(21 * 128);
            for (auto &v : hh) v = Nrm(rng);
            for (auto &v : w21) v = 0.1f * Nrm(rng);
            float *dh = dev(hh), *dw21 = dev(w21), *de; CK(hipMalloc(&de, (size_t)M * 21 * 4));
            const float *ptrs[3] = {dh, dw21, de};
            const float **dptrs; CK(hipMalloc(&dptrs, sizeof(ptrs))); CK(hipMemcpy(dptrs, ptrs, sizeof(ptrs), hipMemcpyHostToDevice));
            tnp::SparseArgs pa = b; pa.winners_out = reinterpret_cast<int16_t *>(dptrs);
            printf("%-42s %8.2f us\n", "regacc quad-major + prepare-like prologue", time_kernel(KQ(1024), pa, nb, tnp::ra_smem_bytes(ncell), 50, 1024));
            printf("%-42s %8.2f us\n", "regacc quad-major (again)", time_kernel(KQ(0), b, nb, tnp::ra_smem_bytes(ncell), 50, 1024));
        }
        {   // wide register-accumulator kernels (round 3): 8 waves, 128 accumulators per lane
            std::vector<float> refq(got);                                           // output of the quad-major 64 x 128 kernel
            CK(hipMemset(out, 0, got.size() * 4));
            time_kernel(KQ(0), b, nb, tnp::ra_smem_bytes(ncell), 1, 1024);
            CK(hipMemcpy(refq.data(), out, refq.size() * 4, hipMemcpyDeviceToHost));
#define KW1(abl) (kern_t)tnp::pool_embed_regwide_kernel<16, 128, 1, abl>
#define KW2(abl) (kern_t)tnp::pool_embed_regwide_kernel<16, 64, 2, abl>
            struct WV { const char *name; kern_t k; int TE, OB; };
            const WV wv[] = {{"wide 128x64", KW1(0), 128, 64}, {"wide 128x64 no weight loads", KW1(1), 128, 64}, {"wide 128x64 no hits", KW1(2), 128, 64},
                             {"wide 128x64 no w, no row loads (5)", KW1(5), 128, 64}, {"wide 128x64 no w, fixed acc index (9)", KW1(9), 128, 64},
                             {"wide 128x64 no w, no FMAs (65)", KW1(65), 128, 64}, {"wide 128x64 no w, no rows, fixed idx (13)", KW1(13), 128, 64},
                             {"wide 128x64 no w, no rows/idx/FMA (77)", KW1(77), 128, 64},
                             {"wide 128x64 no cell loop", KW1(16), 128, 64}, {"wide 128x64 no loop, no votes", KW1(48), 128, 64},
                             {"wide 128x64 no loop, no epilogue", KW1(144), 128, 64}, {"wide 128x64 nothing (176)", KW1(176), 128, 64},
                             {"wide 64x128 2 columns / lane", KW2(0), 64, 128}, {"wide 64x128 no weight loads", KW2(1), 64, 128},
                             {"wide 64x128 no hits", KW2(2), 64, 128},
                             {"wide 64x128 no w, no row loads (5)", KW2(5), 64, 128}, {"wide 64x128 no w, fixed acc index (9)", KW2(9), 64, 128},
                             {"wide 64x128 no w, no FMAs (65)", KW2(65), 64, 128}, {"wide 64x128 no w, no rows/idx/FMA (77)", KW2(77), 64, 128}, {"wide 64x128 no cell loop", KW2(16), 64, 128},
                             {"wide 64x128 no loop, no votes", KW2(48), 64, 128}, {"wide 64x128 nothing (176)", KW2(176), 64, 128}};
            for (const WV &v : wv) {
                tnp::SparseArgs w = b; w.ego_tiles = M / v.TE; w.out_blocks = N1 / v.OB;
                const size_t sm = tnp::rw_smem_bytes(ncell, v.TE, v.OB);
                CK(hipMemset(out, 0, got.size() * 4));
                printf("%-42s %8.2f us\n", v.name, time_kernel(v.k, w, w.ego_tiles * w.out_blocks, sm, 50, 512));
                if (v.k == KW1(0) || v.k == KW2(0)) {
                    CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
                    check(got, v.name);
                    printf("   bit-identical to the 64 x 128 kernel: %s\n", memcmp(got.data(), refq.data(), got.size() * 4) ? "NO" : "yes");
                    int bad = 0;
                    std::vector<float> again(got.size());
                    for (int rep = 0; rep < 20; ++rep) {
                        CK(hipMemsetAsync(out, 0xff, got.size() * 4));
                        hipLaunchKernelGGL(v.k, dim3(w.ego_tiles * w.out_blocks), dim3(512), sm, 0, w);
                        CK(hipMemcpy(again.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
                        bad += memcmp(again.data(), got.data(), got.size() * 4) != 0;
                    }
                    printf("   %d of 20 repeats differ\n", bad);
                    // phase stamps
                    long long *dbgw; CK(hipMalloc(&dbgw, (size_t)w.ego_tiles * w.out_blocks * 16 * 8 * 8));
                    CK(hipMemset(dbgw, 0, (size_t)w.ego_tiles * w.out_blocks * 16 * 8 * 8));
                    tnp::SparseArgs ws = w; ws.winners = reinterpret_cast<const int16_t *>(dbgw);
                    kern_t ks = v.k == KW1(0) ? KW1(256) : KW2(256);
                    time_kernel(ks, ws, w.ego_tiles * w.out_blocks, sm, 3, 512);
                    const int nbw = w.ego_tiles * w.out_blocks;
                    std::vector<long long> hh((size_t)nbw * 16 * 8); CK(hipMemcpy(hh.data(), dbgw, hh.size() * 8, hipMemcpyDeviceToHost));
                    const char *names[] = {"init + geometry", "votes", "winners_out", "main loop", "epilogue"};
                    for (int ph = 0; ph < 5; ++ph) {
                        double mn = 0, mx = 0;
                        for (int wg = 0; wg < nbw; ++wg) {
                            long long lo = 1ll << 62, hi = 0;
                            for (int x = 0; x < 8; ++x) { const long long d = hh[((size_t)wg * 16 + x) * 8 + ph + 1] - hh[((size_t)wg * 16 + x) * 8 + ph]; lo = std::min(lo, d); hi = std::max(hi, d); }
                            mn += lo; mx += hi;
                        }
                        printf("   phase %-16s clocks per wave: min %8.0f max %8.0f (mean over workgroups; 100 MHz units x clock ratio)\n", names[ph], mn / nbw, mx / nbw);
                    }
                    CK(hipFree(dbgw));
                }
            }
        }
        {   // ring kernel (round 3): product tile, per-wave two-slot LDS ring filled by LDS-DMA two cells ahead
            std::vector<float> refq(got.size());
            CK(hipMemset(out, 0, got.size() * 4));
            time_kernel(KQ(0), b, nb, tnp::ra_smem_bytes(ncell), 1, 1024);
            CK(hipMemcpy(refq.data(), out, refq.size() * 4, hipMemcpyDeviceToHost));
#define KG(abl) (kern_t)tnp::pool_embed_regring_kernel<16, false, abl>
#define KG2(abl) (kern_t)tnp::pool_embed_regring_kernel<16, true, abl>
            struct GV { const char *name; kern_t k; };
            const GV gv[] = {{"ring 64x128", KG(0)}, {"ring 64x128 no DMA (1)", KG(1)}, {"ring 64x128 no hits (2)", KG(2)},
                             {"ring 64x128 no DMA, no hits (3)", KG(3)}, {"ring 64x128 no cell loop (16)", KG(16)},
                             {"ring 64x128 no loop, no votes (48)", KG(48)}, {"ring 64x128 nothing (176)", KG(176)},
                             {"ring2 (two register sets)", KG2(0)}, {"ring2 no DMA (1)", KG2(1)}, {"ring2 no hits (2)", KG2(2)},
                             {"ring2 no DMA, no hits (3)", KG2(3)}};
            const size_t sm = tnp::rg_smem_bytes(ncell, 16);
            printf("ring kernel: %d workgroups, %zu bytes of LDS\n", nb, sm);
            for (const GV &v : gv) {
                CK(hipMemset(out, 0, got.size() * 4));
                printf("%-42s %8.2f us\n", v.name, time_kernel(v.k, b, nb, sm, 50, 1024));
                if (v.k == KG(0) || v.k == KG2(0)) {
                    CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
                    check(got, v.name);
                    size_t nd = 0; for (size_t i = 0; i < got.size(); ++i) nd += memcmp(&got[i], &refq[i], 4) != 0;
                    printf("   bit-identical to the product kernel: %s (%zu elements differ)\n", nd ? "NO" : "yes", nd);
                    int bad = 0;
                    const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 60;
                    std::vector<float> again(got.size());
                    for (int rep = 0; rep < reps; ++rep) {
                        CK(hipMemsetAsync(out, 0xff, got.size() * 4));
                        hipLaunchKernelGGL(v.k, dim3(nb), dim3(1024), sm, 0, b);
                        CK(hipMemcpy(again.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
                        bad += memcmp(again.data(), got.data(), got.size() * 4) != 0;
                    }
                    printf("   %d of %d repeats differ\n", bad, reps);
                    {
                        int16_t *w1, *w2; CK(hipMalloc(&w1, (size_t)M * ncell * 2)); CK(hipMalloc(&w2, (size_t)M * ncell * 2));
                        tnp::SparseArgs x = b; x.winners_out = w1; hipLaunchKernelGGL(v.k, dim3(nb), dim3(1024), sm, 0, x);
                        tnp::SparseArgs y = b; y.winners_out = w2; hipLaunchKernelGGL(KQ(0), dim3(nb), dim3(1024), tnp::ra_smem_bytes(ncell), 0, y);
                        std::vector<int16_t> h1((size_t)M * ncell), h2((size_t)M * ncell);
                        CK(hipMemcpy(h1.data(), w1, h1.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), w2, h2.size() * 2, hipMemcpyDeviceToHost));
                        printf("   winner tables equal: %s\n", memcmp(h1.data(), h2.data(), h1.size() * 2) ? "NO" : "yes");
                        CK(hipFree(w1)); CK(hipFree(w2));
                    }
                }
            }
        }
        {   // shared-weight kernel (round 3): 128 egos x 64 columns, 16 waves, weights through an LDS ring (LDS-DMA)
            std::vector<float> refq(got.size());
            CK(hipMemset(out, 0, got.size() * 4));
            time_kernel(KQ(0), b, nb, tnp::ra_smem_bytes(ncell), 1, 1024);
            CK(hipMemcpy(refq.data(), out, refq.size() * 4, hipMemcpyDeviceToHost));
#define KS(abl) (kern_t)tnp::pool_embed_regshare_kernel<16, abl>
            struct SV { const char *name; kern_t k; };
            const SV sv[] = {{"shared 128x64", KS(0)}, {"shared 128x64 no DMA / sync (1)", KS(1)}, {"shared 128x64 no hits (2)", KS(2)},
                             {"shared 128x64 no DMA, no hits (3)", KS(3)}, {"shared 128x64 no DMA, no LDS weight reads (513)", KS(513)},
                             {"shared 128x64 no DMA/LDS reads/hits (515)", KS(515)},
                             {"shared 128x64 no cell loop (16)", KS(16)}, {"shared 128x64 no loop, no votes (48)", KS(48)},
                             {"shared 128x64 no loop, no epilogue (144)", KS(144)}, {"shared 128x64 nothing (176)", KS(176)}};
            tnp::SparseArgs w = b; w.ego_tiles = M / 128; w.out_blocks = N1 / 64;
            const size_t sm = tnp::rs_smem_bytes(ncell, 16);
            const int nbw = w.ego_tiles * w.out_blocks;
            printf("shared-weight kernel: %d workgroups, %zu bytes of LDS\n", nbw, sm);
            for (const SV &v : sv) {
                CK(hipMemset(out, 0, got.size() * 4));
                printf("%-42s %8.2f us\n", v.name, time_kernel(v.k, w, nbw, sm, 50, 1024));
                if (v.k == KS(0)) {
                    CK(hipMemcpy(got.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
                    check(got, v.name);
                    size_t nd = 0; for (size_t i = 0; i < got.size(); ++i) nd += memcmp(&got[i], &refq[i], 4) != 0;
                    printf("   bit-identical to the 64 x 128 kernel: %s (%zu elements differ)\n", nd ? "NO" : "yes", nd);
                    int bad = 0;
                    const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 40;
                    std::vector<float> again(got.size());
                    for (int rep = 0; rep < reps; ++rep) {
                        CK(hipMemsetAsync(out, 0xff, got.size() * 4));
                        hipLaunchKernelGGL(v.k, dim3(nbw), dim3(1024), sm, 0, w);
                        CK(hipMemcpy(again.data(), out, got.size() * 4, hipMemcpyDeviceToHost));
                        bad += memcmp(again.data(), got.data(), got.size() * 4) != 0;
                    }
                    printf("   %d of %d repeats differ\n", bad, reps);
                    {   // winners_out equals the 64 x 128 kernel's
                        int16_t *w1, *w2; CK(hipMalloc(&w1, (size_t)M * ncell * 2)); CK(hipMalloc(&w2, (size_t)M * ncell * 2));
                        tnp::SparseArgs x = w; x.winners_out = w1; hipLaunchKernelGGL(v.k, dim3(nbw), dim3(1024), sm, 0, x);
                        tnp::SparseArgs y = b; y.winners_out = w2; hipLaunchKernelGGL(KQ(0), dim3(nb), dim3(1024), tnp::ra_smem_bytes(ncell), 0, y);
                        std::vector<int16_t> h1((size_t)M * ncell), h2((size_t)M * ncell);
                        CK(hipMemcpy(h1.data(), w1, h1.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), w2, h2.size() * 2, hipMemcpyDeviceToHost));
                        printf("   winner tables equal: %s\n", memcmp(h1.data(), h2.data(), h1.size() * 2) ? "NO" : "yes");
                        CK(hipFree(w1)); CK(hipFree(w2));
                    }
                    long long *dbgw; CK(hipMalloc(&dbgw, (size_t)nbw * 16 * 8 * 8)); CK(hipMemset(dbgw, 0, (size_t)nbw * 16 * 8 * 8));
                    tnp::SparseArgs ws = w; ws.winners = reinterpret_cast<const int16_t *>(dbgw);
                    time_kernel(KS(256), ws, nbw, sm, 3, 1024);
                    std::vector<long long> hh((size_t)nbw * 16 * 8); CK(hipMemcpy(hh.data(), dbgw, hh.size() * 8, hipMemcpyDeviceToHost));
                    const char *names[] = {"init + geometry", "votes", "winners_out", "main loop", "epilogue"};
                    for (int ph = 0; ph < 5; ++ph) {
                        double mn = 0, mx = 0;
                        for (int wg = 0; wg < nbw; ++wg) {
                            long long lo = 1ll << 62, hi = 0;
                            for (int x = 0; x < 16; ++x) { const long long d = hh[((size_t)wg * 16 + x) * 8 + ph + 1] - hh[((size_t)wg * 16 + x) * 8 + ph]; lo = std::min(lo, d); hi = std::max(hi, d); }
                            mn += lo; mx += hi;
                        }
                        printf("   phase %-16s clocks per wave: min %8.0f max %8.0f (mean over workgroups)\n", names[ph], mn / nbw, mx / nbw);
                    }
                    CK(hipFree(dbgw));
                }
            }
        }
        long long *dbg; CK(hipMalloc(&dbg, (size_t)nb * 16 * 8 * 8)); CK(hipMemset(dbg, 0, (size_t)nb * 16 * 8 * 8));
        b.winners = reinterpret_cast<const int16_t *>(dbg);
        time_kernel(KQ(256), b, nb, tnp::ra_smem_bytes(ncell), 3, 1024);
        std::vector<long long> h((size_t)nb * 16 * 8); CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        double tot = 0, ml = 0; for (int w = 0; w < nb; ++w) { long long lo = 1ll << 62, hi = 0, m = 0; for (int v = 0; v < 16; ++v) { lo = std::min(lo, h[((size_t)w * 16 + v) * 8]); hi = std::max(hi, h[((size_t)w * 16 + v) * 8 + 5]); m = std::max(m, h[((size_t)w * 16 + v) * 8 + 4] - h[((size_t)w * 16 + v) * 8 + 3]); } tot += hi - lo; ml += m; }
        printf("quad-major weights: workgroup residency %.0f clocks, main loop (slowest wave) %.0f\n", tot / nb, ml / nb);
    }
    {   // per-wave phase stamps of the register-accumulator kernel
        tnp::SparseArgs b = a; b.ego_tiles = M / 64; b.out_blocks = N1 / 128;
        const int nb = b.ego_tiles * b.out_blocks;
        long long *dbg; CK(hipMalloc(&dbg, (size_t)nb * 16 * 8 * 8)); CK(hipMemset(dbg, 0, (size_t)nb * 16 * 8 * 8));
        b.winners = reinterpret_cast<const int16_t *>(dbg);
        time_kernel(KR(256), b, nb, tnp::ra_smem_bytes(ncell), 3, 1024);
        std::vector<long long> h((size_t)nb * 16 * 8); CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        const char *names[] = {"stage geometry + key init", "pairs", "conversion", "main loop", "epilogue"};
        for (int ph = 0; ph < 5; ++ph) {
            std::vector<double> med, mx, mn;
            for (int w = 0; w < nb; ++w) {
                std::vector<long long> d;
                for (int v = 0; v < 16; ++v) d.push_back(h[((size_t)w * 16 + v) * 8 + ph + 1] - h[((size_t)w * 16 + v) * 8 + ph]);
                std::sort(d.begin(), d.end());
                med.push_back(d[8]); mx.push_back(d[15]); mn.push_back(d[0]);
            }
            auto mean = [](const std::vector<double> &x) { double t = 0; for (double v : x) t += v; return t / x.size(); };
            printf("phase %-28s clocks per wave: min %8.0f median %8.0f max %8.0f (mean over workgroups)\n", names[ph], mean(mn), mean(med), mean(mx));
        }
        double tot = 0; for (int w = 0; w < nb; ++w) { long long lo = 1ll << 62, hi = 0; for (int v = 0; v < 16; ++v) { lo = std::min(lo, h[((size_t)w * 16 + v) * 8]); hi = std::max(hi, h[((size_t)w * 16 + v) * 8 + 5]); } tot += hi - lo; }
        printf("workgroup residency %.0f clocks (100 MHz shader clock units: x24 for 2.4 GHz cycles?)\n", tot / nb);
    }
    return 0;
}
