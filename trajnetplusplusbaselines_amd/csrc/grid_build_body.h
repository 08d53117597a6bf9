// The body of grid_build_kernel (pool_grid.hip) with the staging of the scene's positions left to the caller: the stand-alone
// kernel reads them from the obs1 / obs2 buffers, the merged prepare + grid launch of the sequence driver (lstm_seq.hip)
// forms them from the previous step's state itself.  Included by both translation units.
#pragma once
#include "tnp_internal.h"

namespace tnp {

#define TNP_GRID_EGOS 4
#define TNP_GRID_MAX_TRACKS 1024

__device__ __forceinline__ float nan_to_num_dev(float v) {
    if (v != v) return 0.0f;
    if (__builtin_isinf(v)) return v > 0.0f ? 3.402823466e+38f : -3.402823466e+38f;
    return v;
}

// stage(start, ns, pos, vel): fills pos[0 .. ns) (absent -> the (-500, -500) sentinel) and, for the directional grid, vel[0 .. ns)
template <int EGOS = TNP_GRID_EGOS, class Stage>
__device__ __forceinline__ void grid_build_body(const GridArgs &a, const int bx, const int s, float *gsm, Stage stage) {
    const int start = a.scene_start[s];
    const int ns = a.scene_start[s + 1] - start;
    const int ego0 = bx * EGOS;
    const int pad = a.scene_slots ? a.scene_slots[s] : a.n_max;   // slots the reference pads this scene to
    if (ego0 >= ns) return;  // uniform for the workgroup

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = a.n, ncell = G * G, C = a.C;
    // LDS carve: pos[ns] (float2) | vel[ns] (float2) | vals[ns*C] | win[4][ncell] (int)
    float2 *pos = reinterpret_cast<float2 *>(gsm);
    float2 *vel = pos + ns;
    float *vals = reinterpret_cast<float *>(vel + ns);
    int *win_all = reinterpret_cast<int *>(vals + (a.type == TNP_POOL_SOCIAL ? ((ns * C + 3) & ~3) : 0));
    int *win = win_all + wave * ncell;

    stage(start, ns, pos, vel);
    if (a.type == TNP_POOL_SOCIAL)
        for (int q = tid; q < ns * C; q += 256) {
            const int j = q / C, c = q - j * C;
            vals[q] = a.values[(size_t)(start + j) * a.ldv + c];
        }

    const float fG = (float)G;
    for (int e = 0; e < EGOS / 4; ++e) {
        const int ki = ego0 + e * 4 + wave;
        const bool active = ki < ns;
        for (int c = lane; c < ncell; c += 64) win[c] = -1;
        __syncthreads();
        if (active) {
            const float2 pi = pos[ki];
            for (int j = lane; j < ns; j += 64) {
                if (j == ki) continue;  // diagonal removed, :259-263
                const float2 pj = pos[j];
                const float ox = __fadd_rn(__fdiv_rn(__fsub_rn(pj.x, pi.x), a.cell), a.half_x);  // :276
                const float oy = __fadd_rn(__fdiv_rn(__fsub_rn(pj.y, pi.y), a.cell), a.half_y);
                const bool inr = !(ox < 0.0f) && !(ox >= fG) && !(oy < 0.0f) && !(oy >= fG);  // :278-279
                const int cell = inr ? ((int)ox * G + (int)oy) : 0;                            // :281-287
                atomicMax(&win[cell], 2 * j + (inr ? 1 : 0));
            }
            // slots the reference pads this scene with (lstm/lstm.py:29-40): absent, highest j, cell 0
            if (ns < pad && lane == 0) atomicMax(&win[0], 2 * (pad - 1));
        }
        __syncthreads();
        if (active) {
            const size_t row = (size_t)(start + ki);
            if (a.winners) {
                for (int c = lane; c < ncell; c += 64) {
                    const int w = win[c];
                    a.winners[row * ncell + c] = (w >= 0 && (w & 1)) ? (int16_t)(w >> 1) : (int16_t)-1;
                }
            }
            if (a.grid) {
                float *out = a.grid + row * (size_t)a.ldg;
                const int F = C * ncell;
                const float2 vi = (a.type == TNP_POOL_DIRECTIONAL) ? vel[ki] : make_float2(0.f, 0.f);
                auto value_of = [&](int w, int c) -> float {
                    if (!(w >= 0 && (w & 1))) return a.constant;
                    const int j = w >> 1;
                    if (a.type == TNP_POOL_OCCUPANCY) return 1.0f;                                    // :266-267
                    if (a.type == TNP_POOL_DIRECTIONAL) {
                        const float2 vj = vel[j];
                        return nan_to_num_dev(c == 0 ? __fsub_rn(vj.x, vi.x) : __fsub_rn(vj.y, vi.y));  // :131-140
                    }
                    return vals[j * C + c];                                                            // :160-167
                };
                if (a.vec4) {  // 4 consecutive cells of one channel per lane: 1 KiB coalesced wave stores
                    for (int f = lane * 4; f < F; f += 256) {
                        const int c = f / ncell, cell = f - c * ncell;
                        const int4 w4 = *reinterpret_cast<const int4 *>(win + cell);
                        float4 v;
                        v.x = value_of(w4.x, c); v.y = value_of(w4.y, c); v.z = value_of(w4.z, c); v.w = value_of(w4.w, c);
                        *reinterpret_cast<float4 *>(out + f) = v;
                    }
                } else {
                    for (int f = lane; f < F; f += 64) {
                        const int c = f / ncell, cell = f - c * ncell;
                        out[f] = value_of(win[cell], c);
                    }
                }
            }
        }
        __syncthreads();
    }
}

// what the scene's positions become in LDS: pos (NaN -> sentinel, lstm/gridbased_pooling.py:247-249), vel = obs2 - obs1 (:127, NaN kept)
__device__ __forceinline__ void grid_stage_track(int type, float2 p1, float2 p2, int j, float2 *pos, float2 *vel) {
    float2 p = p2;
    if (p.x != p.x || p.y != p.y) { p.x = -500.0f; p.y = -500.0f; }
    pos[j] = p;
    if (type == TNP_POOL_DIRECTIONAL) vel[j] = make_float2(__fsub_rn(p2.x, p1.x), __fsub_rn(p2.y, p1.y));
}

}  // namespace tnp
