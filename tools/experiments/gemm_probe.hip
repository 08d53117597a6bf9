// Tile-configuration probe for the two small GEMMs of the recurrent step (csrc/gemm_f32_mfma.hip), stand-alone:
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Itrajnetplusplusbaselines_amd/csrc tools/experiments/gemm_probe.hip -o tools/experiments/gemm_probe
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <algorithm>
#include <random>
#include "gemm_f32_mfma.hip"

namespace tnp { void set_error(const char *fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); } }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
template <typename T> T *dev(const std::vector<T> &h) { T *d; CK(hipMalloc(&d, h.size() * sizeof(T))); CK(hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice)); return d; }

template <typename F> static float time_us(F f, int reps = 100) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) f();
    CK(hipDeviceSynchronize()); CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); CK(hipGetLastError());
    return ms * 1000.0f / reps;
}

int main() {
    using namespace tnp;
    setvbuf(stdout, nullptr, _IONBF, 0);
    std::mt19937 rng(3); std::normal_distribution<float> N01(0.f, 1.f);
    const int M = 2048;
    {   // second embedding layer: [2048,1024] x [256,1024]^T, bias + relu
        const int N = 256, K = 1024;
        std::vector<float> A((size_t)M * K), B((size_t)N * K), bias(N);
        for (auto &v : A) v = fmaxf(N01(rng), 0.f); for (auto &v : B) v = 0.03f * N01(rng); for (auto &v : bias) v = 0.1f * N01(rng);
        GemmArgs g{}; g.A1 = dev(A); g.lda1 = K; g.K1 = K; g.B1 = dev(B); g.ldb1 = K; g.bias1 = dev(bias); g.M = M; g.N = N; g.vec_ok = 1;
        float *C; CK(hipMalloc(&C, (size_t)M * N * 4)); g.C = C; g.ldc = N; g.relu = 1;
        std::vector<float> ref((size_t)M * N), got((size_t)M * N);
        auto run = [&](const char *name, auto fn) {
            CK(hipMemset(C, 0, (size_t)M * N * 4));
            const float us = time_us([&] { fn(g, (hipStream_t)0); });
            CK(hipMemcpy(got.data(), C, got.size() * 4, hipMemcpyDeviceToHost));
            static bool first = true; double md = 0;
            if (first) { ref = got; first = false; } else for (size_t i = 0; i < got.size(); ++i) md = fmax(md, fabs((double)got[i] - ref[i]));
            printf("linear 2048x256x1024  %-28s %7.2f us  %6.1f TFLOP/s  maxdiff %.2g\n", name, us, 2.0 * M * N * K / us * 1e-6, md);
        };
        run("pipe<1,2,2,1,32> (product)", [](GemmArgs g, hipStream_t s) { return launch_pipe<1, 2, 2, 1, 32, EPI_BIAS>(g, s); });
        run("pipe<1,2,4,1,16>", [](GemmArgs g, hipStream_t s) { return launch_pipe<1, 2, 4, 1, 16, EPI_BIAS>(g, s); });
        run("pipe<1,1,4,2,16>", [](GemmArgs g, hipStream_t s) { return launch_pipe<1, 1, 4, 2, 16, EPI_BIAS>(g, s); });
        run("pipe<1,1,4,1,16> 32x32", [](GemmArgs g, hipStream_t s) { return launch_pipe<1, 1, 4, 1, 16, EPI_BIAS>(g, s); });
        run("pipe<1,1,2,1,32> 32x32", [](GemmArgs g, hipStream_t s) { return launch_pipe<1, 1, 2, 1, 32, EPI_BIAS>(g, s); });
        run("pipe<1,1,4,1,32> 32x32", [](GemmArgs g, hipStream_t s) { return launch_pipe<1, 1, 4, 1, 32, EPI_BIAS>(g, s); });
        run("pipe<1,1,8,1,16> 32x32", [](GemmArgs g, hipStream_t s) { return launch_pipe<1, 1, 8, 1, 16, EPI_BIAS>(g, s); });
        run("pipe<1,2,2,1,16>", [](GemmArgs g, hipStream_t s) { return launch_pipe<1, 2, 2, 1, 16, EPI_BIAS>(g, s); });
        run("pipe<2,1,2,1,32> 64x32", [](GemmArgs g, hipStream_t s) { return launch_pipe<2, 1, 2, 1, 32, EPI_BIAS>(g, s); });
        run("fast<1,2,2,1,32>", [](GemmArgs g, hipStream_t s) { return launch_fast<1, 2, 2, 1, 32, EPI_BIAS>(g, s); });
        {   // phase stamps of the product configuration
            constexpr int WM = 1, WN = 2, WK = 2, AN = 1, BK = 32;
            GemmArgs d = g; d.tiles_m = M / 32; d.tiles_n = N / 64;
            const int nb = d.tiles_m * d.tiles_n, nw = WM * WN * WK;
            long long *dbg; CK(hipMalloc(&dbg, (size_t)nb * nw * 64)); CK(hipMemset(dbg, 0, (size_t)nb * nw * 64));
            d.gates_out = reinterpret_cast<float *>(dbg);
            auto k = gemm_nt_pipe<WM, WN, WK, AN, BK, EPI_BIAS, false, 1, false, 1>;   // the product's interleaved schedule
            const size_t smem = (size_t)3 * WK * (32 * WM + 32 * WN * AN) * (BK + 4) * 4;
            CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(nb), dim3(64 * nw), smem, 0, d);
            CK(hipDeviceSynchronize());
            std::vector<long long> h((size_t)nb * nw * 8); CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
            const char *names[] = {"address setup + first loads issued", "first tile in LDS (load latency)", "main loop", "split-K reduction", "epilogue stores"};
            for (int ph = 0; ph < 5; ++ph) {
                double sum = 0, mx = 0; long cnt = 0;
                for (size_t w = 0; w < (size_t)nb * nw; ++w) { const long long a0 = h[w * 8 + ph], a1 = h[w * 8 + ph + 1]; if (a1 && a0) { sum += a1 - a0; mx = fmax(mx, (double)(a1 - a0)); ++cnt; } }
                printf("   phase %-38s mean %7.0f max %7.0f clocks (%ld waves)\n", names[ph], cnt ? sum / cnt : 0, mx, cnt);
            }
            long long lo = 1ll << 62, hi = 0; double res = 0;
            for (int b = 0; b < nb; ++b) { long long l = 1ll << 62, hh = 0; for (int v = 0; v < nw; ++v) for (int q = 0; q < 6; ++q) { const long long t = h[((size_t)b * nw + v) * 8 + q]; if (t) { l = std::min(l, t); hh = std::max(hh, t); } } res += hh - l; lo = std::min(lo, l); hi = std::max(hi, hh); }
            printf("   workgroup residency mean %.0f clocks; first start to last end %lld clocks\n", res / nb, hi - lo);
        }
        {
            auto k2 = gemm_nt_pipe<1, 2, 2, 1, 32, EPI_BIAS, false, 0, true>;
            GemmArgs d = g; d.tiles_m = M / 32; d.tiles_n = N / 64;
            const size_t smem = (size_t)3 * 2 * 96 * 36 * 4;
            CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            printf("linear 2048x256x1024  two accumulator chains        %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k2, dim3(d.tiles_m * d.tiles_n), dim3(256), smem, 0, d); }));
            d.prio = 1;
            printf("linear 2048x256x1024  two accumulator chains, prio  %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k2, dim3(d.tiles_m * d.tiles_n), dim3(256), smem, 0, d); }));
        }
        {
            auto k2 = gemm_nt_pipe<1, 2, 2, 1, 32, EPI_BIAS, false, 0, false, 1>;
            GemmArgs d = g; d.tiles_m = M / 32; d.tiles_n = N / 64;
            const size_t smem = (size_t)3 * 2 * 96 * 36 * 4;
            CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            CK(hipMemset(C, 0, (size_t)M * N * 4));
            printf("linear 2048x256x1024  interleaved schedule          %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k2, dim3(d.tiles_m * d.tiles_n), dim3(256), smem, 0, d); }));
            d.prio = 1;
            printf("linear 2048x256x1024  interleaved schedule, prio    %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k2, dim3(d.tiles_m * d.tiles_n), dim3(256), smem, 0, d); }));
            CK(hipMemcpy(got.data(), C, got.size() * 4, hipMemcpyDeviceToHost));
            double md = 0; for (size_t i = 0; i < got.size(); ++i) md = fmax(md, fabs((double)got[i] - ref[i]));
            printf("   interleaved vs product: max diff %.3g\n", md);
        }
        {
            GemmArgs d = g; d.tiles_m = M / 32; d.tiles_n = N / 64; d.prio = 1;
            const size_t smem = (size_t)3 * 2 * 96 * 36 * 4;
            auto t = [&](const char *name, auto k2) {
                CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
                printf("linear interleaved+prio ablation: %-30s %7.2f us\n", name, time_us([&] { hipLaunchKernelGGL(k2, dim3(d.tiles_m * d.tiles_n), dim3(256), smem, 0, d); }));
            };
            t("no global loads in loop", gemm_nt_pipe<1, 2, 2, 1, 32, EPI_BIAS, false, 0, false, 1, 1>);
            t("no LDS writes in loop", gemm_nt_pipe<1, 2, 2, 1, 32, EPI_BIAS, false, 0, false, 1, 2>);
            t("no barrier in loop", gemm_nt_pipe<1, 2, 2, 1, 32, EPI_BIAS, false, 0, false, 1, 4>);
            t("no fragment reads in loop", gemm_nt_pipe<1, 2, 2, 1, 32, EPI_BIAS, false, 0, false, 1, 8>);
            t("MFMAs only (15)", gemm_nt_pipe<1, 2, 2, 1, 32, EPI_BIAS, false, 0, false, 1, 15>);
            t("no loads, writes, barrier (7)", gemm_nt_pipe<1, 2, 2, 1, 32, EPI_BIAS, false, 0, false, 1, 7>);
        }
        g.prio = 1;
        run("pipe<1,2,2,1,32> prio", [](GemmArgs g, hipStream_t s) { return launch_pipe<1, 2, 2, 1, 32, EPI_BIAS>(g, s); });
    }
    {   // LSTM gates: [x | h] [2048, 320 + 128] x [W_ih | W_hh]^T -> 4 x 128 gates + cell update
        const int H = 128, I = 320;
        std::vector<float> X((size_t)M * I), Hh((size_t)M * H), Cc((size_t)M * H), Wih((size_t)4 * H * I), Whh((size_t)4 * H * H), b1(4 * H), b2(4 * H);
        for (auto &v : X) v = fmaxf(N01(rng), 0.f); for (auto &v : Hh) v = 0.5f * N01(rng); for (auto &v : Cc) v = 0.5f * N01(rng);
        for (auto &v : Wih) v = 0.05f * N01(rng); for (auto &v : Whh) v = 0.05f * N01(rng); for (auto &v : b1) v = 0.1f * N01(rng); for (auto &v : b2) v = 0.1f * N01(rng);
        std::vector<uint8_t> mask(M, 1);
        GemmArgs g{}; g.A1 = dev(X); g.lda1 = I; g.K1 = I; g.A2 = dev(Hh); g.lda2 = H; g.K2 = H; g.B1 = dev(Wih); g.ldb1 = I; g.B2 = dev(Whh); g.ldb2 = H;
        g.bias1 = dev(b1); g.bias2 = dev(b2); g.M = M; g.N = 4 * H; g.vec_ok = 1; g.H = H; g.h_in = g.A2; g.c_in = dev(Cc); g.mask = dev(mask);
        float *ho, *co; CK(hipMalloc(&ho, (size_t)M * H * 4)); CK(hipMalloc(&co, (size_t)M * H * 4)); g.h_out = ho; g.c_out = co;
        auto run = [&](const char *name, auto fn) {
            const float us = time_us([&] { fn(g, (hipStream_t)0); });
            printf("gates 2048x512x448     %-28s %7.2f us  %6.1f TFLOP/s\n", name, us, 2.0 * M * 4 * H * (I + H) / us * 1e-6);
        };
        run("pipe<1,1,4,4,16> (product)", [](GemmArgs g, hipStream_t s) { return launch_pipe<1, 1, 4, 4, 16, EPI_LSTM>(g, s); });
        g.prio = 1;
        run("pipe<1,1,4,4,16> prio", [](GemmArgs g, hipStream_t s) { return launch_pipe<1, 1, 4, 4, 16, EPI_LSTM>(g, s); });
        g.prio = 0;
        {   // phase stamps of the PRODUCT gates kernel (gate split: four waves own one gate block each over the whole K)
            constexpr int WM = 1, WN = 4, WK = 1, AN = 1, BK = 32;
            GemmArgs d = g; d.tiles_m = M / 32; d.tiles_n = H / 32;
            const int nb = d.tiles_m * d.tiles_n, nw = WM * WN * WK;
            long long *dbg; CK(hipMalloc(&dbg, (size_t)nb * nw * 64)); CK(hipMemset(dbg, 0, (size_t)nb * nw * 64));
            d.C = reinterpret_cast<float *>(dbg);
            auto k = gemm_nt_pipe<WM, WN, WK, AN, BK, EPI_LSTM, true, 1, false, 1>;
            const size_t smem = (size_t)3 * WK * (32 * WM + 32 * WN * AN) * (BK + 4) * 4;
            CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(nb), dim3(64 * nw), smem, 0, d);
            CK(hipDeviceSynchronize());
            std::vector<long long> h((size_t)nb * nw * 8); CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
            const char *names[] = {"address setup + first loads issued", "first tile in LDS (load latency)", "main loop", "gate exchange + LSTM epilogue"};
            const int from[] = {0, 1, 2, 3}, to[] = {1, 2, 3, 5};
            for (int ph = 0; ph < 4; ++ph) {
                double sum = 0, mx = 0; long cnt = 0;
                for (size_t w = 0; w < (size_t)nb * nw; ++w) { const long long a0 = h[w * 8 + from[ph]], a1 = h[w * 8 + to[ph]]; if (a1 && a0) { sum += a1 - a0; mx = fmax(mx, (double)(a1 - a0)); ++cnt; } }
                printf("   product gates <1,4,1,1,32> phase %-38s mean %7.0f max %7.0f clocks (%ld waves)\n", names[ph], cnt ? sum / cnt : 0, mx, cnt);
            }
            long long lo = 1ll << 62, hi = 0; double res = 0;
            for (int b = 0; b < nb; ++b) { long long l = 1ll << 62, hh = 0; for (int v = 0; v < nw; ++v) for (int q = 0; q < 6; ++q) { const long long t = h[((size_t)b * nw + v) * 8 + q]; if (t) { l = std::min(l, t); hh = std::max(hh, t); } } res += hh - l; lo = std::min(lo, l); hi = std::max(hi, hh); }
            printf("   product gates: workgroup residency mean %.0f clocks; first start to last end %lld clocks\n", res / nb, hi - lo);
        }
        {
            constexpr int WM = 1, WN = 1, WK = 4, AN = 4, BK = 16;
            GemmArgs d = g; d.tiles_m = M / 32; d.tiles_n = H / 32;
            const int nb = d.tiles_m * d.tiles_n, nw = WM * WN * WK;
            long long *dbg; CK(hipMalloc(&dbg, (size_t)nb * nw * 64)); CK(hipMemset(dbg, 0, (size_t)nb * nw * 64));
            d.C = reinterpret_cast<float *>(dbg);
            auto k = gemm_nt_pipe<WM, WN, WK, AN, BK, EPI_LSTM, true, 1>;
            const size_t smem = (size_t)3 * WK * (32 * WM + 32 * WN * AN) * (BK + 4) * 4;
            CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k, dim3(nb), dim3(64 * nw), smem, 0, d);
            CK(hipDeviceSynchronize());
            std::vector<long long> h((size_t)nb * nw * 8); CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
            const char *names[] = {"address setup + first loads issued", "first tile in LDS (load latency)", "main loop", "reduction + LSTM epilogue"};
            const int from[] = {0, 1, 2, 3}, to[] = {1, 2, 3, 5};
            for (int ph = 0; ph < 4; ++ph) {
                double sum = 0, mx = 0; long cnt = 0;
                for (size_t w = 0; w < (size_t)nb * nw; ++w) { const long long a0 = h[w * 8 + from[ph]], a1 = h[w * 8 + to[ph]]; if (a1 && a0) { sum += a1 - a0; mx = fmax(mx, (double)(a1 - a0)); ++cnt; } }
                printf("   gates phase %-38s mean %7.0f max %7.0f clocks (%ld waves)\n", names[ph], cnt ? sum / cnt : 0, mx, cnt);
            }
        }
        {
            auto k2 = gemm_nt_pipe<1, 1, 4, 4, 16, EPI_LSTM, true, 0, false, 1>;
            GemmArgs d = g; d.tiles_m = M / 32; d.tiles_n = H / 32;
            const size_t smem = (size_t)3 * 4 * 160 * 20 * 4;
            CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            printf("gates 2048x512x448     interleaved schedule          %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k2, dim3(d.tiles_m * d.tiles_n), dim3(256), smem, 0, d); }));
        }
        {   // gate split: four waves own one gate block each over the whole K
            GemmArgs d = g; d.tiles_m = M / 32; d.tiles_n = H / 32; d.prio = 1;
            auto t = [&](const char *name, auto k2, size_t smem) {
                CK(hipFuncSetAttribute(reinterpret_cast<const void *>(k2), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
                CK(hipMemset(ho, 0, (size_t)M * H * 4));
                printf("gates 2048x512x448     %-28s %7.2f us\n", name, time_us([&] { hipLaunchKernelGGL(k2, dim3(d.tiles_m * d.tiles_n), dim3(256), smem, 0, d); }));
            };
            t("gate split BK 32", gemm_nt_pipe<1, 4, 1, 1, 32, EPI_LSTM, true, 0, false, 1>, (size_t)3 * 160 * 36 * 4);
            t("gate split BK 32, 2 chains", gemm_nt_pipe<1, 4, 1, 1, 32, EPI_LSTM, true, 0, true, 1>, (size_t)3 * 160 * 36 * 4);
        }
        // reference check of h_out against the host (of the LAST variant run above)
        std::vector<float> hh((size_t)M * H); CK(hipMemcpy(hh.data(), ho, hh.size() * 4, hipMemcpyDeviceToHost));
        double md = 0;
        for (int m = 0; m < M; m += 61) for (int u = 0; u < H; u += 7) {
            double pre[4];
            for (int gt = 0; gt < 4; ++gt) {
                double a = b1[gt * H + u] + b2[gt * H + u];
                for (int k = 0; k < I; ++k) a += (double)X[(size_t)m * I + k] * Wih[((size_t)gt * H + u) * I + k];
                for (int k = 0; k < H; ++k) a += (double)Hh[(size_t)m * H + k] * Whh[((size_t)gt * H + u) * H + k];
                pre[gt] = a;
            }
            auto sg = [](double x) { return 1.0 / (1.0 + exp(-x)); };
            const double cn = sg(pre[1]) * Cc[(size_t)m * H + u] + sg(pre[0]) * tanh(pre[2]);
            md = fmax(md, fabs(sg(pre[3]) * tanh(cn) - hh[(size_t)m * H + u]));
        }
        printf("   gates: max |h - host reference| %.3g\n", md);
    }
    return 0;
}
