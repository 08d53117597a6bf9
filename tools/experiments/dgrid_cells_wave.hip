// Round 6, measured negative: the sparse first layer's cell gradient with ONE wave per (cell, K slice) and the weight slice in
// registers, for small batches (batch_size 8, ~310 tracks per step).  Bit-identical to dgrid_cells_xcd_kernel (lstm_bwd.hip) and
// SLOWER: 14.5 us per launch against 9.5 (tools/diag/small_train_ab.sh).  The mean of 12 hits per cell hides what sets the
// launch's length: the few central cells of the grid collect most of the hits (a scene's neighbours are close by), a single wave
// walks their 5-8 groups of 16 one after the other, and four waves per workgroup take them two at a time.  Requesting the
// first list entries / dy rows ahead of the LDS staging in the four-wave kernel (one round trip less on paper) was slower too
// (9.8 -> 10.5 us): the idle waves of the ~90 % of workgroups with a single group then hold their slots until their own loads
// return.  Drop-in for launch_dgrid_cells_xcd: `if (N1 <= 1024 && M <= 512) hipLaunchKernelGGL(dgrid_cells_xcd_wave_kernel,
// dim3(ncell * 8), dim3(64), 0, s, <the same arguments>)`.
// The same product for SMALL batches (batch_size 8: ~12 hits per cell and step): one WAVE per (cell, slice), the weight slice
// in 32 registers per lane (the loads the staging loop above does, kept), no LDS and no barrier.  With four waves per workgroup
// three of them had nothing to do but stage weights and leave, and the launch was the chain count -> weights -> barrier ->
// list -> dy -> MFMA; here the weights travel beside the list / dy requests.  Same MFMA order per group: bit-identical.
__global__ void __launch_bounds__(64) dgrid_cells_xcd_wave_kernel(const float *__restrict__ dy, int ldy, const float *__restrict__ Wc,
                                                                  const int2 *__restrict__ list, const int32_t *__restrict__ count,
                                                                  int R, int nseg, int seg, int M, int C, int ncell, int N1,
                                                                  float *__restrict__ dcell8) {
    const int x = blockIdx.x & 7, c = blockIdx.x >> 3;
    const int cnt = count[c * nseg + seg];
    if (cnt <= 0) return;
    const int ngroups = (cnt + 15) >> 4;
    const int2 *L = list + (size_t)c * R + (size_t)seg * M;
    const int lane = threadIdx.x, row = lane & 15, kq = lane >> 4;
    const int ncol = N1 >> 3, T = ncol >> 4, k0 = x * ncol;                   // T <= 8 (N1 <= 1024)
    const int row_off = seg * M;
    float *out = dcell8 + (size_t)x * M * ncell * C;
    auto rows_of = [&](int g, int &rl, int (&ro)[4]) {
        const int idx = g * 16 + row;
        rl = L[(g < ngroups && idx < cnt) ? idx : 0].x - row_off;
#pragma unroll
        for (int v = 0; v < 4; ++v) { const int id2 = g * 16 + 4 * kq + v; ro[v] = L[(g < ngroups && id2 < cnt) ? id2 : 0].x - row_off; }
    };
    auto load_a = [&](float4 (&a)[8], int rl) {
        const float *src = dy + (size_t)rl * ldy + k0 + 4 * kq;
#pragma unroll
        for (int u = 0; u < 8; ++u) a[u] = *reinterpret_cast<const float4 *>(src + 16 * (u < T ? u : 0));
    };
    int rl0, ro0[4], rl1, ro1[4];
    rows_of(0, rl0, ro0);
    rows_of(1, rl1, ro1);
    float4 b[8];
    {
        const float *wsrc = Wc + ((size_t)c * C + (row < C ? row : 0)) * N1 + k0 + 4 * kq;
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = *reinterpret_cast<const float4 *>(wsrc + 16 * (u < T ? u : 0));
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (row >= C || u >= T) b[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 a0[8], a1[8];
    load_a(a0, rl0);
    for (int g = 0; g < ngroups; ++g) {
        int rl2, ro2[4];
        rows_of(g + 2, rl2, ro2);
        if (g + 1 < ngroups) load_a(a1, rl1);
        floatx4 acc = floatx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (u < T) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u].x, b[u].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u].y, b[u].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u].z, b[u].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[u].w, b[u].w, acc, 0, 0, 0);
            }
#pragma unroll
        for (int v = 0; v < 4; ++v)
            if (g * 16 + 4 * kq + v < cnt && row < C) out[((size_t)ro0[v] * ncell + c) * C + row] = acc[v];
#pragma unroll
        for (int u = 0; u < 8; ++u) a0[u] = a1[u];
        rl0 = rl1; rl1 = rl2;
#pragma unroll
        for (int v = 0; v < 4; ++v) { ro0[v] = ro1[v]; ro1[v] = ro2[v]; }
    }
}

