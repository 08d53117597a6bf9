// Batched classical predictors on MI355X: social force, ORCA and constant-velocity Kalman rollouts for many scenes
// at once (reference wrappers classical/socialforce.py:10-111, classical/orca.py:10-134, classical/kalman.py:6-73,
// which simulate ONE scene per call through third-party CPU libraries).  The arithmetic lives in classical_core.h
// (shared with the host oracle); this file is the GPU execution: one workgroup per scene, one lane per agent, the
// scene's state staged in LDS for the all-pairs force / neighbour search, synchronous (Jacobi) state updates
// separated by workgroup barriers, the 8 simulator sub-steps per output frame fused in one launch.
// Compiled with -ffp-contract=off so that the float32 ORCA arithmetic is bit-identical to the host restatement.
#include "tnp_internal.h"
#include "classical_core.h"
#include <stdlib.h>

#define TNP_MAX_DEVICES 64
#define ORCA_LP3_SLOTS 16      /* lanes of a wave that run linearProgram3 at a time (their lines live in LDS) */

namespace tnp {

// ---- social force: state0 [M][6] = x, y, vx, vy, gx, gy (float64); out [n_out][M][2] --------------------------------
// sf_agent_step_terms (classical_core.h) with the five fp64 divides per ordered pair -- three "-b / sigma", two "/ delta" -- as
// multiplications by reciprocals formed once per launch: 60 of the ~364 vector instructions a pair costs (round 6: 95 -> 82 ms
// at 4096 x 128).  A quotient and the product with the rounded reciprocal differ in the last bit; the host execution keeps the
// literal formula and the two agree to 1e-11 as before (the exp / sqrt implementations of the two sides differ anyway).
// sqrt for the squared distances of a crowd (metres: never denormal, never near overflow): v_rsq_f64 + the usual two coupled
// Newton steps and a final residual correction, WITHOUT the library's range scaling (two v_ldexp, a compare and the selects
// around them: 5 of its ~16 instructions, ten square roots per ordered pair).  Zero maps to zero; NaN / negative inputs give NaN
// as sqrt does.
__device__ __forceinline__ double sf_sqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g);
    h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    return x == 0.0 ? 0.0 : g;
}

// semi-minor axis b of the elliptical potential at displacement (rx, ry) (sf_potential: V = v0 exp(-b / sigma))
__device__ __forceinline__ double sf_b_gpu(double rx, double ry, const sf_agent_terms *tb) {
    const double n1 = sf_sqrt(rx * rx + ry * ry);
    const double sx = rx - tb->dex, sy = ry - tb->dey;
    const double n2 = sf_sqrt(sx * sx + sy * sy);
    const double in_sqrt = (n1 + n2) * (n1 + n2) - tb->d2;
    return 0.5 * sf_sqrt(in_sqrt);
}

// exp(d) - 1 for the |d| <= 1/16 of a finite-difference step (delta / sigma = 1/300 times a gradient of order one): the Taylor
// polynomial to d^9 / 9! (remainder < 1e-17 relative), Horner form.
__device__ __forceinline__ double sf_expm1_small(double d) {
    double p = 1.0 / 362880.0;
    p = __builtin_fma(p, d, 1.0 / 40320.0);
    p = __builtin_fma(p, d, 1.0 / 5040.0);
    p = __builtin_fma(p, d, 1.0 / 720.0);
    p = __builtin_fma(p, d, 1.0 / 120.0);
    p = __builtin_fma(p, d, 1.0 / 24.0);
    p = __builtin_fma(p, d, 1.0 / 6.0);
    p = __builtin_fma(p, d, 0.5);
    p = __builtin_fma(p, d, 1.0);
    return p * d;
}

__device__ __forceinline__ void sf_agent_step_gpu(int a, int n, const double *st, const sf_agent_terms *terms, double initial_speed,
                                                  double max_speed, const sf_params *p, double neg_inv_sigma, double *vx_new,
                                                  double *vy_new) {
    const double *sa = st + 7 * a;
    const double eax = terms[a].ex, eay = terms[a].ey;
    const double tau = sa[6];
    double Fx = 1.0 / tau * (initial_speed * eax - sa[2]);
    double Fy = 1.0 / tau * (initial_speed * eay - sa[3]);
    const double delta = 1e-3, inv_delta = 1.0 / delta;
    double sum_x = 0.0, sum_y = 0.0;
    // Two neighbours per trip in STAGES: the finite difference of both (three elliptic distances, an exp, two polynomials each --
    // straight-line code, so the two chains interleave and fill each other's fp64 latencies), then the rare literal form where
    // a step crosses the potential's edge, then the field-of-view weights.  The sums are still taken in neighbour order.
    // V(r + delta) - V(r) = V(r) (exp((b' - b) / -sigma) - 1): ONE exp per pair instead of three, and the difference without the
    // cancellation of two nearly equal potentials (the host execution subtracts them as the package does; 1e-12 apart).
    struct Diff { double arg, v, dx_, dy_, dvdx, dvdy; };
    auto diff = [&](int b, Diff &d) {
        const double *sb = st + 7 * b;
        const sf_agent_terms *tb = terms + b;
        const double rx = sa[0] - sb[0], ry = sa[1] - sb[1];
        const double b0 = sf_b_gpu(rx, ry, tb);
        d.arg = b0 * neg_inv_sigma;
        d.v = p->v0 * exp(d.arg);
        d.dx_ = (sf_b_gpu(rx + delta, ry, tb) - b0) * neg_inv_sigma;
        d.dy_ = (sf_b_gpu(rx, ry + delta, tb) - b0) * neg_inv_sigma;
        d.dvdx = d.v * sf_expm1_small(d.dx_) * inv_delta;
        d.dvdy = d.v * sf_expm1_small(d.dy_) * inv_delta;
    };
    auto edge = [&](Diff &d) {                 // |argument| beyond the polynomial's range: the literal difference of two potentials
        if (!(fabs(d.dx_) <= 0.0625 && fabs(d.dy_) <= 0.0625)) {
            d.dvdx = (p->v0 * exp(d.arg + d.dx_) - d.v) * inv_delta;
            d.dvdy = (p->v0 * exp(d.arg + d.dy_) - d.v) * inv_delta;
        }
    };
    auto weigh = [&](const Diff &d, double &cx, double &cy) {
        const double fx = -1.0 * d.dvdx, fy = -1.0 * d.dvdy;
        const double in_sight = (eax * (-fx) + eay * (-fy)) > sf_sqrt(fx * fx + fy * fy) * p->cosphi;
        const double w = in_sight ? 1.0 : p->out_of_view;
        cx = w * fx; cy = w * fy;
    };
    int b = 0;
    for (; b + 1 < n; b += 2) {
        Diff d0, d1;
        diff(b, d0);
        diff(b + 1, d1);
        edge(d0);
        edge(d1);
        double c0x, c0y, c1x, c1y;
        weigh(d0, c0x, c0y);
        weigh(d1, c1x, c1y);
        if (b != a) { sum_x += c0x; sum_y += c0y; }
        if (b + 1 != a) { sum_x += c1x; sum_y += c1y; }
    }
    if (b < n && b != a) {
        Diff d0;
        double c0x, c0y;
        diff(b, d0);
        edge(d0);
        weigh(d0, c0x, c0y);
        sum_x += c0x; sum_y += c0y;
    }
    Fx += sum_x; Fy += sum_y;
    const double wx = sa[2] + p->delta_t * Fx, wy = sa[3] + p->delta_t * Fy;
    const double ws = sqrt(wx * wx + wy * wy);
    const double factor = fmin(1.0, max_speed / ws);
    *vx_new = wx * factor; *vy_new = wy * factor;
}

__global__ void __launch_bounds__(256) sf_rollout_kernel(const double *state0, const int32_t *scene_start, int M,
                                                         int n_steps, int sample_every, double tau, sf_params prm,
                                                         double max_speed_mult, double *out) {
    extern __shared__ __attribute__((aligned(16))) double sfs[];
    const int s = blockIdx.x;
    const int lo = scene_start[s], ns = scene_start[s + 1] - lo;
    double *st = sfs;                    // [ns][7]
    double *nv = st + (size_t)ns * 7;    // [ns][2]
    double *isp = nv + (size_t)ns * 2;   // [ns] initial speeds
    sf_agent_terms *terms = reinterpret_cast<sf_agent_terms *>(isp + ns);   // [ns] per-agent terms of the current state
    for (int a = threadIdx.x; a < ns; a += blockDim.x) {
        const double *src = state0 + (size_t)(lo + a) * 6;
        for (int k = 0; k < 6; ++k) st[a * 7 + k] = src[k];
        st[a * 7 + 6] = tau;
        isp[a] = sqrt(src[2] * src[2] + src[3] * src[3]);
        sf_terms(st + a * 7, &prm, &terms[a]);
    }
    __syncthreads();
    int n_out = 0;
    const double neg_inv_sigma = -1.0 / prm.sigma;
    for (int step = 0; step < n_steps; ++step) {
        for (int a = threadIdx.x; a < ns; a += blockDim.x)
            sf_agent_step_gpu(a, ns, st, terms, isp[a], max_speed_mult * isp[a], &prm, neg_inv_sigma, &nv[2 * a], &nv[2 * a + 1]);
        __syncthreads();
        for (int a = threadIdx.x; a < ns; a += blockDim.x) {
            st[a * 7 + 0] += nv[2 * a] * prm.delta_t;
            st[a * 7 + 1] += nv[2 * a + 1] * prm.delta_t;
            st[a * 7 + 2] = nv[2 * a];
            st[a * 7 + 3] = nv[2 * a + 1];
            sf_terms(st + a * 7, &prm, &terms[a]);   // direction / d e / d^2 of the NEW state, once per agent instead of per pair
            if (step % sample_every == 0) {   // reference keeps the states after steps 0, 8, 16, ... (socialforce.py:95)
                out[((size_t)n_out * M + lo + a) * 2 + 0] = st[a * 7 + 0];
                out[((size_t)n_out * M + lo + a) * 2 + 1] = st[a * 7 + 1];
            }
        }
        if (step % sample_every == 0) ++n_out;
        __syncthreads();
    }
}

// ---- ORCA: pos0/vel0 [M][2] float32, goals [M][2] and speed [M] float64 (the wrapper's numpy side) --------------------
__global__ void __launch_bounds__(256) orca_rollout_kernel(const float *pos0, const float *vel0, const double *goals,
                                                           const double *speed, const float *max_speed,
                                                           const int32_t *scene_start, int M, int n_iter, int sample_every,
                                                           orca_params prm, float *out, int *nbr_dbg) {
    extern __shared__ __attribute__((aligned(16))) float ors[];
    const int s = blockIdx.x;
    const int lo = scene_start[s], ns = scene_start[s + 1] - lo;
    float *pos = ors;                   // [ns][2]
    float *vel = pos + (size_t)ns * 2;  // [ns][2]
    float *nvl = vel + (size_t)ns * 2;  // [ns][2]
    float *prf = nvl + (size_t)ns * 2;  // [ns][2] preferred velocity (0 before the first step, orca.py:99-119)
    for (int a = threadIdx.x; a < ns; a += blockDim.x) {
        pos[2 * a] = pos0[2 * (lo + a)]; pos[2 * a + 1] = pos0[2 * (lo + a) + 1];
        vel[2 * a] = vel0[2 * (lo + a)]; vel[2 * a + 1] = vel0[2 * (lo + a) + 1];
        prf[2 * a] = 0.0f; prf[2 * a + 1] = 0.0f;
    }
    __syncthreads();
    int n_out = 0;
    for (int count = 1; count <= n_iter; ++count) {
        for (int a = threadIdx.x; a < ns; a += blockDim.x) {
            int *dbg = (nbr_dbg && count == 1) ? nbr_dbg + (size_t)(lo + a) * ORCA_MAX_NEIGHBORS : nullptr;
            orca_agent_new_velocity(a, ns, pos, vel, prf[2 * a], prf[2 * a + 1], max_speed[lo + a], &prm, &nvl[2 * a],
                                    &nvl[2 * a + 1], dbg);
        }
        __syncthreads();
        for (int a = threadIdx.x; a < ns; a += blockDim.x) {
            vel[2 * a] = nvl[2 * a]; vel[2 * a + 1] = nvl[2 * a + 1];          // Agent::update
            pos[2 * a] += vel[2 * a] * prm.time_step; pos[2 * a + 1] += vel[2 * a + 1] * prm.time_step;
            if (count % sample_every == 0) {                                    // orca.py:107
                out[((size_t)n_out * M + lo + a) * 2 + 0] = pos[2 * a];
                out[((size_t)n_out * M + lo + a) * 2 + 1] = pos[2 * a + 1];
            }
            // preferred velocity towards the goal, capped at the initial speed (orca.py:111-119), in float64
            const double px = (double)pos[2 * a], py = (double)pos[2 * a + 1];
            const double gx = goals[2 * (lo + a)], gy = goals[2 * (lo + a) + 1];
            const double dx = gx - px, dy = gy - py;
            const double dist = sqrt(dx * dx + dy * dy);
            float pvx, pvy;
            if (dist < 0.05) { pvx = 0.0f; pvy = 0.0f; }
            else {
                const double sp = speed[lo + a];
                if (dist > sp) { pvx = (float)(sp * dx / dist); pvy = (float)(sp * dy / dist); }
                else { pvx = (float)dx; pvy = (float)dy; }
            }
            prf[2 * a] = pvx; prf[2 * a + 1] = pvy;
        }
        if (count % sample_every == 0) ++n_out;
        __syncthreads();
    }
}

// ---- ORCA, register form (round 6) -----------------------------------------------------------------------------------
// The generic orca_agent_new_velocity above is CPU code run one lane per agent: its neighbour list, half-planes and LP
// projections are dynamically indexed private arrays (scratch memory: 7.3e7 VMEM reads per launch) and every lane runs the
// insertion-shift loop whenever any lane of the wave has a candidate in range -- 58 k vector instructions per wave and
// simulator step for ~3 k of useful work (profiles/archive/round3_d_pmc_classical.md).  Same arithmetic, GPU shape:
//   * neighbour search in two phases per 64 candidates: an in-range bit per candidate (positions broadcast from LDS), then
//     every lane walks ITS OWN set bits in ascending order and inserts into a sorted list of MN (dist^2, index) REGISTER
//     pairs by a branch-free compare network -- the list insertAgentNeighbor builds (the MN smallest by (dist^2, visiting
//     order), ties behind their equals, range shrinking to the last entry once full), entry for entry;
//   * half-planes and linearProgram1/2 fully unrolled over MN compile-time slots: no indexed array, no scratch;
//   * linearProgram3 (dense crowds) keeps the generic code, on LDS copies of the lines of the few lanes that need it.
// MN must equal max_neighbors (the wrapper always passes RVO2's 10); other values take the generic kernel.
template <int MN>
__device__ __forceinline__ bool orca_lp1_reg(const orca_line (&L)[MN], const int line_no, float radius, float ox, float oy,
                                             float *rx, float *ry) {
    // linearProgram1, directionOpt = false (classical_core.h: orca_lp1, same expressions in the same order)
    const orca_line Ln = L[line_no];
    const float dot = Ln.px * Ln.dx + Ln.py * Ln.dy;
    const float disc = dot * dot + radius * radius - (Ln.px * Ln.px + Ln.py * Ln.py);
    if (disc < 0.0f) return false;
    const float sq = sqrtf(disc);
    float t_left = -dot - sq, t_right = -dot + sq;
    bool feasible = true;
#pragma unroll
    for (int i = 0; i < MN; ++i) {
        if (i < line_no && feasible) {
            const float den = orca_det(Ln.dx, Ln.dy, L[i].dx, L[i].dy);
            const float num = orca_det(L[i].dx, L[i].dy, Ln.px - L[i].px, Ln.py - L[i].py);
            if (fabsf(den) <= ORCA_EPSILON) {
                if (num < 0.0f) feasible = false;
            } else {
                const float t = num / den;
                if (den >= 0.0f) t_right = fminf(t_right, t);
                else t_left = fmaxf(t_left, t);
                if (t_left > t_right) feasible = false;
            }
        }
    }
    if (!feasible) return false;
    const float t = Ln.dx * (ox - Ln.px) + Ln.dy * (oy - Ln.py);
    if (t < t_left) { *rx = Ln.px + t_left * Ln.dx; *ry = Ln.py + t_left * Ln.dy; }
    else if (t > t_right) { *rx = Ln.px + t_right * Ln.dx; *ry = Ln.py + t_right * Ln.dy; }
    else { *rx = Ln.px + t * Ln.dx; *ry = Ln.py + t * Ln.dy; }
    return true;
}

template <int MN>
__device__ __forceinline__ void orca_new_velocity_reg(int a, int ns, const float *pos, const float *vel, float prefx, float prefy,
                                                      float max_speed, const orca_params &p, float *nvx, float *nvy, int *nbr_out,
                                                      orca_line *slots /* this WAVE's ORCA_LP3_SLOTS x 2 MN lines of LDS */) {
    float nd[MN];
    int nbr[MN];
#pragma unroll
    for (int i = 0; i < MN; ++i) { nd[i] = INFINITY; nbr[i] = -1; }
    const float nd2 = p.neighbor_dist * p.neighbor_dist;
    const float ax = pos[2 * a], ay = pos[2 * a + 1];
    for (int base = 0; base < ns; base += 64) {
        unsigned long long m = 0ull;
        const int lim = min(64, ns - base);
        for (int j = 0; j < lim; ++j) {                          // wave-uniform: pos[b] is one LDS broadcast
            const int b = base + j;
            const float ddx = ax - pos[2 * b], ddy = ay - pos[2 * b + 1];
            const float dist_sq = ddx * ddx + ddy * ddy;
            if (b != a && dist_sq < nd2) m |= 1ull << j;
        }
        while (m) {                                              // this lane's candidates, ascending
            const int b = base + __builtin_ctzll(m);
            m &= m - 1ull;
            const float ddx = ax - pos[2 * b], ddy = ay - pos[2 * b + 1];
            const float dist_sq = ddx * ddx + ddy * ddy;
            if (dist_sq < fminf(nd2, nd[MN - 1])) {              // rangeSq: neighborDist^2 until the list is full, then its last entry
#pragma unroll
                for (int i = MN - 1; i >= 1; --i) {
                    const bool shift = dist_sq < nd[i - 1];
                    const bool place = !shift && dist_sq < nd[i];
                    nbr[i] = shift ? nbr[i - 1] : (place ? b : nbr[i]);
                    nd[i] = shift ? nd[i - 1] : (place ? dist_sq : nd[i]);
                }
                const bool place0 = dist_sq < nd[0];
                nbr[0] = place0 ? b : nbr[0];
                nd[0] = place0 ? dist_sq : nd[0];
            }
        }
    }
    int cnt = 0;
#pragma unroll
    for (int i = 0; i < MN; ++i) cnt += nbr[i] >= 0 ? 1 : 0;
    if (nbr_out) {
#pragma unroll
        for (int i = 0; i < ORCA_MAX_NEIGHBORS; ++i) nbr_out[i] = i < MN ? nbr[i < MN ? i : 0] : -1;
    }
    orca_line L[MN];
    const float inv_th = 1.0f / p.time_horizon;
    const float vx = vel[2 * a], vy = vel[2 * a + 1];
#pragma unroll
    for (int k = 0; k < MN; ++k) {
        L[k].px = 0.0f; L[k].py = 0.0f; L[k].dx = 0.0f; L[k].dy = 0.0f;
        if (k < cnt) {
            const int b = nbr[k];
            const float rpx = pos[2 * b] - ax, rpy = pos[2 * b + 1] - ay;
            const float rvx = vx - vel[2 * b], rvy = vy - vel[2 * b + 1];
            L[k] = orca_make_line(rpx, rpy, rvx, rvy, vx, vy, inv_th, &p);
        }
    }
    // linearProgram2, directionOpt = false (orca_lp2)
    float rx, ry;
    if (prefx * prefx + prefy * prefy > max_speed * max_speed) {
        const float inv = 1.0f / sqrtf(prefx * prefx + prefy * prefy);
        rx = prefx * inv * max_speed; ry = prefy * inv * max_speed;
    } else { rx = prefx; ry = prefy; }
    int fail = cnt;
#pragma unroll
    for (int i = 0; i < MN; ++i) {
        if (i < fail) {                                          // (fail == cnt until a line fails, then the loop is over)
            if (orca_det(L[i].dx, L[i].dy, L[i].px - rx, L[i].py - ry) > 0.0f) {
                const float tx = rx, ty = ry;
                if (!orca_lp1_reg<MN>(L, i, max_speed, prefx, prefy, &rx, &ry)) { rx = tx; ry = ty; fail = i; }
            }
        }
    }
    // linearProgram3 (infeasible half-planes: dense crowds -- at BASELINE config 5 some lane of a wave needs it in a third of the
    // wave-steps, and then usually several at once: jams are local; without it the kernel takes 7.3 ms): the generic code on a
    // copy of the lines in LDS.  The lanes that need it take one of ORCA_LP3_SLOTS per-wave slots (rank among the needing lanes:
    // ballot + mbcnt; more than ORCA_LP3_SLOTS of them go in rounds), so the dynamically indexed `lines[]` / `proj[]` of the
    // generic code are LDS accesses and the kernel has no scratch memory.  Measured at 4096 x 128 (round 6): private arrays
    // (scratch) 14.4 ms; 4 slots 18.5 (rounds: a divergent linearProgram3 pass costs the same for one lane as for sixteen);
    // 16 slots 13.5; 32 slots 13.9; one copy per THREAD 22.6 (40 KB per workgroup: half of the resident waves).
    const bool need = fail < cnt;
    const unsigned long long nm = __ballot(need);
    if (nm) {
        const int rank = __builtin_amdgcn_mbcnt_hi((unsigned)(nm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)nm, 0u));
        const int total = __popcll(nm);
        for (int base = 0; base < total; base += ORCA_LP3_SLOTS) {
            if (need && rank >= base && rank < base + ORCA_LP3_SLOTS) {
                orca_line *sp = slots + (rank - base) * 2 * MN;
#pragma unroll
                for (int k = 0; k < MN; ++k) sp[k] = L[k];
                orca_lp3_buf(sp, cnt, fail, max_speed, &rx, &ry, sp + MN);
            }
        }
    }
    *nvx = rx; *nvy = ry;
}

template <int MN>
__global__ void __launch_bounds__(256) orca_rollout_reg_kernel(const float *pos0, const float *vel0, const double *goals,
                                                               const double *speed, const float *max_speed,
                                                               const int32_t *scene_start, int M, int n_iter, int sample_every,
                                                               orca_params prm, float *out, int *nbr_dbg) {
    extern __shared__ __attribute__((aligned(16))) float ors[];
    const int s = blockIdx.x;
    const int lo = scene_start[s], ns = scene_start[s + 1] - lo;
    float *pos = ors;                   // [ns][2]
    float *vel = pos + (size_t)ns * 2;  // [ns][2]
    float *nvl = vel + (size_t)ns * 2;  // [ns][2]
    float *prf = nvl + (size_t)ns * 2;  // [ns][2] preferred velocity (0 before the first step, orca.py:99-119)
    orca_line *slots = reinterpret_cast<orca_line *>(prf + (((size_t)ns * 2 + 3) & ~(size_t)3)) + (size_t)(threadIdx.x >> 6) * ORCA_LP3_SLOTS * 2 * MN;
    for (int a = threadIdx.x; a < ns; a += blockDim.x) {
        pos[2 * a] = pos0[2 * (lo + a)]; pos[2 * a + 1] = pos0[2 * (lo + a) + 1];
        vel[2 * a] = vel0[2 * (lo + a)]; vel[2 * a + 1] = vel0[2 * (lo + a) + 1];
        prf[2 * a] = 0.0f; prf[2 * a + 1] = 0.0f;
    }
    __syncthreads();
    int n_out = 0;
    for (int count = 1; count <= n_iter; ++count) {
        for (int a = threadIdx.x; a < ns; a += blockDim.x) {
            int *dbg = (nbr_dbg && count == 1) ? nbr_dbg + (size_t)(lo + a) * ORCA_MAX_NEIGHBORS : nullptr;
            orca_new_velocity_reg<MN>(a, ns, pos, vel, prf[2 * a], prf[2 * a + 1], max_speed[lo + a], prm, &nvl[2 * a],
                                      &nvl[2 * a + 1], dbg, slots);
        }
        __syncthreads();
        for (int a = threadIdx.x; a < ns; a += blockDim.x) {
            vel[2 * a] = nvl[2 * a]; vel[2 * a + 1] = nvl[2 * a + 1];          // Agent::update
            pos[2 * a] += vel[2 * a] * prm.time_step; pos[2 * a + 1] += vel[2 * a + 1] * prm.time_step;
            if (count % sample_every == 0) {                                    // orca.py:107
                out[((size_t)n_out * M + lo + a) * 2 + 0] = pos[2 * a];
                out[((size_t)n_out * M + lo + a) * 2 + 1] = pos[2 * a + 1];
            }
            // preferred velocity towards the goal, capped at the initial speed (orca.py:111-119), in float64
            const double px = (double)pos[2 * a], py = (double)pos[2 * a + 1];
            const double gx = goals[2 * (lo + a)], gy = goals[2 * (lo + a) + 1];
            const double dx = gx - px, dy = gy - py;
            const double dist = sqrt(dx * dx + dy * dy);
            float pvx, pvy;
            if (dist < 0.05) { pvx = 0.0f; pvy = 0.0f; }
            else {
                const double sp = speed[lo + a];
                if (dist > sp) { pvx = (float)(sp * dx / dist); pvy = (float)(sp * dy / dist); }
                else { pvx = (float)dx; pvy = (float)dy; }
            }
            prf[2 * a] = pvx; prf[2 * a + 1] = pvy;
        }
        if (count % sample_every == 0) ++n_out;
        __syncthreads();
    }
}

// ---- Kalman: one lane per track; obs [n_tracks][T][2] float64, z [n_tracks][n_samples][n_steps][6] ---------------------
// One wave per SIMD on purpose: the EM pass keeps ~310 registers of 4 x 4 double matrices live; capping the allocation for 2 / 3 /
// 4 waves per SIMD spills them to scratch memory and costs 7.1 / 14.2 / 22.6 ms against 5.8 (BASELINE config 5, round 6) -- the
// 4 x 4 products carry enough independent fp64 FMAs to keep the pipe of one wave busy.
__global__ void __launch_bounds__(64, 1) kalman_kernel(const double *obs, int n_tracks, int T, int n_iter, int n_steps,
                                                       int n_samples, const double *z, double q0, double r0, double *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_tracks) return;
    const double *o = obs + (size_t)i * T * 2;
    kf_model md;
    for (int k = 0; k < 16; ++k) { md.Q[k] = (k % 5 == 0) ? q0 : 0.0; md.P0[k] = (k % 5 == 0) ? 1.0 : 0.0; }
    md.R[0] = r0; md.R[1] = 0.0; md.R[2] = 0.0; md.R[3] = r0;
    md.m0[0] = o[0]; md.m0[1] = 0.0; md.m0[2] = o[1]; md.m0[3] = 0.0;     // kalman.py:31
    double x_last[4];
    kf_em_smooth(o, T, n_iter, &md, x_last);
    kf_sample_mean(&md, x_last, n_steps, n_samples, z + (size_t)i * n_samples * n_steps * 6, out + (size_t)i * n_steps * 2);
}

}  // namespace tnp

extern "C" TNP_API int tnp_sf_rollout(const double *state0, const int32_t *scene_start, int B, int M, int n_max,
                                      int n_steps, int sample_every, double tau, double v0, double sigma, double delta_t,
                                      double *out, void *stream) {
    if (B <= 0 || M <= 0) return 0;
    sf_params p;
    p.delta_t = delta_t; p.v0 = v0; p.sigma = sigma;
    p.cosphi = cos(200.0 / 2.0 / 180.0 * M_PI);      // FieldOfView(twophi = 200 degrees)
    p.out_of_view = 0.5;
    const size_t smem = (size_t)n_max * (10 + 6) * sizeof(double);
    if (smem > 160 * 1024) TNP_FAIL(-1, "tnp_sf_rollout: %d agents in one scene exceed the LDS-staged limit", n_max);
    int dev = 0;
    TNP_HIP(hipGetDevice(&dev));
    static size_t attr[TNP_MAX_DEVICES] = {0};                               // per DEVICE (one process may drive several)
    if (dev >= TNP_MAX_DEVICES || smem > attr[dev]) {
        TNP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tnp::sf_rollout_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        if (dev < TNP_MAX_DEVICES) attr[dev] = smem;
    }
    const int threads = n_max <= 64 ? 64 : (n_max <= 128 ? 128 : 256);
    hipLaunchKernelGGL(tnp::sf_rollout_kernel, dim3(B), dim3(threads), smem, (hipStream_t)stream, state0, scene_start, M,
                       n_steps, sample_every, tau, p, 1.3, out);
    TNP_HIP(hipGetLastError());
    return 0;
}

extern "C" TNP_API int tnp_orca_rollout(const float *pos0, const float *vel0, const double *goals, const double *speed,
                                        const float *max_speed, const int32_t *scene_start, int B, int M, int n_max,
                                        int n_iter, int sample_every, float time_step, float neighbor_dist,
                                        int max_neighbors, float time_horizon, float radius, float *out, int *nbr_dbg,
                                        void *stream) {
    if (B <= 0 || M <= 0) return 0;
    if (max_neighbors > ORCA_MAX_NEIGHBORS) TNP_FAIL(-1, "tnp_orca_rollout: max_neighbors %d > %d", max_neighbors, ORCA_MAX_NEIGHBORS);
    orca_params p;
    p.time_step = time_step; p.neighbor_dist = neighbor_dist; p.time_horizon = time_horizon; p.radius = radius;
    p.max_neighbors = max_neighbors;
    const int threads = n_max <= 64 ? 64 : (n_max <= 128 ? 128 : 256);
    int dev = 0;
    TNP_HIP(hipGetDevice(&dev));
    if (max_neighbors == 10) {           // RVO2's / the wrapper's value (classical/orca.py:95): the register form
        constexpr int MN = 10;
        const size_t smem = (((size_t)n_max * 8 + 3) & ~(size_t)3) * sizeof(float) + (size_t)(threads / 64) * ORCA_LP3_SLOTS * 2 * MN * sizeof(orca_line);
        if (smem > 160 * 1024) TNP_FAIL(-1, "tnp_orca_rollout: %d agents in one scene exceed the LDS-staged limit", n_max);
        static size_t attr[TNP_MAX_DEVICES] = {0};                           // per DEVICE: the attribute belongs to the device's code object
        if (dev >= TNP_MAX_DEVICES || smem > attr[dev]) {
            TNP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tnp::orca_rollout_reg_kernel<MN>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            if (dev < TNP_MAX_DEVICES) attr[dev] = smem;
        }
        hipLaunchKernelGGL(tnp::orca_rollout_reg_kernel<MN>, dim3(B), dim3(threads), smem, (hipStream_t)stream, pos0, vel0, goals,
                           speed, max_speed, scene_start, M, n_iter, sample_every, p, out, nbr_dbg);
        TNP_HIP(hipGetLastError());
        return 0;
    }
    const size_t smem = (size_t)n_max * 8 * sizeof(float);
    if (smem > 160 * 1024) TNP_FAIL(-1, "tnp_orca_rollout: %d agents in one scene exceed the LDS-staged limit", n_max);
    static size_t attr[TNP_MAX_DEVICES] = {0};
    if (dev >= TNP_MAX_DEVICES || smem > attr[dev]) {
        TNP_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(tnp::orca_rollout_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        if (dev < TNP_MAX_DEVICES) attr[dev] = smem;
    }
    hipLaunchKernelGGL(tnp::orca_rollout_kernel, dim3(B), dim3(threads), smem, (hipStream_t)stream, pos0, vel0, goals, speed,
                       max_speed, scene_start, M, n_iter, sample_every, p, out, nbr_dbg);
    TNP_HIP(hipGetLastError());
    return 0;
}

extern "C" TNP_API int tnp_kalman_predict(const double *obs, int n_tracks, int T, int n_iter, int n_steps, int n_samples,
                                          const double *z, double transition_var, double observation_var, double *out,
                                          void *stream) {
    if (n_tracks <= 0) return 0;
    if (T < 2 || T > KF_MAX_T) TNP_FAIL(-1, "tnp_kalman_predict: observation length %d not in 2..%d", T, KF_MAX_T);
    hipLaunchKernelGGL(tnp::kalman_kernel, dim3((n_tracks + 63) / 64), dim3(64), 0, (hipStream_t)stream, obs, n_tracks, T,
                       n_iter, n_steps, n_samples, z, transition_var, observation_var, out);
    TNP_HIP(hipGetLastError());
    return 0;
}

// ---- classical.constant_velocity.predict (classical/constant_velocity.py:4-20) ---------------------------------------
namespace tnp {
__global__ void constant_velocity_kernel(const double *last, const double *prev, int N2, int n_predict, double *out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= N2) return;
    const double l = last[q], v = l - prev[q];
    // this file is built with -ffp-contract=off: last + t * v keeps numpy's two roundings (bit-exact float64)
    for (int t = 1; t <= n_predict; ++t) out[(size_t)(t - 1) * N2 + q] = l + (double)t * v;
}
}  // namespace tnp

extern "C" TNP_API int tnp_constant_velocity(const double *last, const double *prev, int N, int n_predict, double *out,
                                     void *stream) {
    if (N <= 0 || n_predict <= 0) return 0;
    const int N2 = 2 * N;
    hipLaunchKernelGGL(tnp::constant_velocity_kernel, dim3((N2 + 255) / 256), dim3(256), 0, (hipStream_t)stream, last,
                       prev, N2, n_predict, out);
    TNP_HIP(hipGetLastError());
    return 0;
}
