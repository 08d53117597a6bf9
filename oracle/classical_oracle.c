/*
 * oracle/classical_oracle.c -- TEST INFRASTRUCTURE ONLY (see trajnet_oracle.c for the rules).
 *
 * Host execution of the classical predictors' arithmetic core.  PARITY UNPINNED: the reference only wraps the
 * third-party packages socialforce, rvo2 (RVO2) and pykalman (classical/socialforce.py:6-8,89-93,
 * classical/orca.py:4,90-119, classical/kalman.py:2,40-60), none of which is vendored, pinned or installed, and no
 * reference test touches classical/  -- there is nothing to pin against.  The formulae are a
 * restatement of the published algorithms and live in ONE place, trajnetplusplusbaselines_amd/csrc/classical_core.h,
 * compiled here by gcc for the host and by hipcc for gfx950: these functions therefore check the GPU EXECUTION
 * (batching over scenes, LDS staging, lane-parallel neighbour search, synchronous updates, sampling cadence), not the
 * formulae; tests/test_classical.py adds formula-independent invariants (constant-velocity limit, mirror symmetry,
 * ORCA collision-freeness, Kalman on noise-free lines) and compares this core with oracle/classical_numpy.py, a second,
 * independent numpy restatement of the same published algorithms that does not include the header.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../trajnetplusplusbaselines_amd/csrc/classical_core.h"

#define ORC_API __attribute__((visibility("default")))

/* socialforce.Simulator(...).step() x n_steps; classical/socialforce.py:84-95 keeps states 0, 8, 16, ... */
ORC_API void orc_sf_rollout(const double *state0, const int32_t *scene_start, int B, int M, int n_steps, int sample_every,
                            double tau, double v0, double sigma, double delta_t, double *out) {
    sf_params p;
    p.delta_t = delta_t; p.v0 = v0; p.sigma = sigma; p.cosphi = cos(200.0 / 2.0 / 180.0 * M_PI); p.out_of_view = 0.5;
#pragma omp parallel for schedule(dynamic, 1)
    for (int s = 0; s < B; ++s) {
        const int lo = scene_start[s], ns = scene_start[s + 1] - lo;
        double *st = (double *)malloc(sizeof(double) * (size_t)ns * 10 + sizeof(sf_agent_terms) * (size_t)ns);
        double *nv = st + (size_t)ns * 7, *isp = nv + (size_t)ns * 2;
        sf_agent_terms *terms = (sf_agent_terms *)(isp + ns);
        for (int a = 0; a < ns; ++a) {
            for (int k = 0; k < 6; ++k) st[a * 7 + k] = state0[(size_t)(lo + a) * 6 + k];
            st[a * 7 + 6] = tau;
            isp[a] = sqrt(st[a * 7 + 2] * st[a * 7 + 2] + st[a * 7 + 3] * st[a * 7 + 3]);
        }
        int n_out = 0;
        for (int step = 0; step < n_steps; ++step) {
            for (int a = 0; a < ns; ++a) sf_terms(st + a * 7, &p, &terms[a]);
            for (int a = 0; a < ns; ++a) sf_agent_step_terms(a, ns, st, terms, isp[a], 1.3 * isp[a], &p, &nv[2 * a], &nv[2 * a + 1]);
            for (int a = 0; a < ns; ++a) {
                st[a * 7 + 0] += nv[2 * a] * delta_t; st[a * 7 + 1] += nv[2 * a + 1] * delta_t;
                st[a * 7 + 2] = nv[2 * a]; st[a * 7 + 3] = nv[2 * a + 1];
                if (step % sample_every == 0) {
                    out[((size_t)n_out * M + lo + a) * 2 + 0] = st[a * 7 + 0];
                    out[((size_t)n_out * M + lo + a) * 2 + 1] = st[a * 7 + 1];
                }
            }
            if (step % sample_every == 0) ++n_out;
        }
        free(st);
    }
}

/* rvo2 doStep() x n_iter with the wrapper's preferred-velocity update; classical/orca.py:90-119 */
ORC_API void orc_orca_rollout(const float *pos0, const float *vel0, const double *goals, const double *speed,
                              const float *max_speed, const int32_t *scene_start, int B, int M, int n_iter,
                              int sample_every, float time_step, float neighbor_dist, int max_neighbors,
                              float time_horizon, float radius, float *out, int *nbr_dbg) {
    orca_params p;
    p.time_step = time_step; p.neighbor_dist = neighbor_dist; p.time_horizon = time_horizon; p.radius = radius;
    p.max_neighbors = max_neighbors;
#pragma omp parallel for schedule(dynamic, 1)
    for (int s = 0; s < B; ++s) {
        const int lo = scene_start[s], ns = scene_start[s + 1] - lo;
        float *pos = (float *)malloc(sizeof(float) * (size_t)ns * 8);
        float *vel = pos + (size_t)ns * 2, *nvl = vel + (size_t)ns * 2, *prf = nvl + (size_t)ns * 2;
        for (int a = 0; a < ns; ++a) {
            pos[2 * a] = pos0[2 * (lo + a)]; pos[2 * a + 1] = pos0[2 * (lo + a) + 1];
            vel[2 * a] = vel0[2 * (lo + a)]; vel[2 * a + 1] = vel0[2 * (lo + a) + 1];
            prf[2 * a] = 0.0f; prf[2 * a + 1] = 0.0f;
        }
        int n_out = 0;
        for (int count = 1; count <= n_iter; ++count) {
            for (int a = 0; a < ns; ++a) {
                int *dbg = (nbr_dbg && count == 1) ? nbr_dbg + (size_t)(lo + a) * ORCA_MAX_NEIGHBORS : NULL;
                orca_agent_new_velocity(a, ns, pos, vel, prf[2 * a], prf[2 * a + 1], max_speed[lo + a], &p, &nvl[2 * a],
                                        &nvl[2 * a + 1], dbg);
            }
            for (int a = 0; a < ns; ++a) {
                vel[2 * a] = nvl[2 * a]; vel[2 * a + 1] = nvl[2 * a + 1];
                pos[2 * a] += vel[2 * a] * time_step; pos[2 * a + 1] += vel[2 * a + 1] * time_step;
                if (count % sample_every == 0) {
                    out[((size_t)n_out * M + lo + a) * 2 + 0] = pos[2 * a];
                    out[((size_t)n_out * M + lo + a) * 2 + 1] = pos[2 * a + 1];
                }
                const double px = (double)pos[2 * a], py = (double)pos[2 * a + 1];
                const double dx = goals[2 * (lo + a)] - px, dy = goals[2 * (lo + a) + 1] - py;
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
        }
        free(pos);
    }
}

/* pykalman KalmanFilter.em -> smooth -> mean of sampled observation sequences; classical/kalman.py:40-60 */
ORC_API void orc_kalman_predict(const double *obs, int n_tracks, int T, int n_iter, int n_steps, int n_samples,
                                const double *z, double q0, double r0, double *out) {
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n_tracks; ++i) {
        const double *o = obs + (size_t)i * T * 2;
        kf_model md;
        for (int k = 0; k < 16; ++k) { md.Q[k] = (k % 5 == 0) ? q0 : 0.0; md.P0[k] = (k % 5 == 0) ? 1.0 : 0.0; }
        md.R[0] = r0; md.R[1] = 0.0; md.R[2] = 0.0; md.R[3] = r0;
        md.m0[0] = o[0]; md.m0[1] = 0.0; md.m0[2] = o[1]; md.m0[3] = 0.0;
        double x_last[4];
        kf_em_smooth(o, T, n_iter, &md, x_last);
        kf_sample_mean(&md, x_last, n_steps, n_samples, z + (size_t)i * n_samples * n_steps * 6, out + (size_t)i * n_steps * 2);
    }
}

/* ---- single-step entry points for the driving stubs of oracle/classical_stubs.py (backend "core"): the reference's
 * wrappers call step() / doStep() / em() / sample() themselves, one call at a time (classical/socialforce.py:91,
 * classical/orca.py:102, classical/kalman.py:47-55) ---- */

/* one socialforce.Simulator.step(): st [n][7] updated in place, isp [n] = initial speeds */
ORC_API void orc_sf_step(double *st, const double *isp, int ns, double v0, double sigma, double delta_t) {
    sf_params p;
    p.delta_t = delta_t; p.v0 = v0; p.sigma = sigma; p.cosphi = cos(200.0 / 2.0 / 180.0 * M_PI); p.out_of_view = 0.5;
    sf_agent_terms *terms = (sf_agent_terms *)malloc(sizeof(sf_agent_terms) * (size_t)ns + sizeof(double) * (size_t)ns * 2);
    double *nv = (double *)(terms + ns);
    for (int a = 0; a < ns; ++a) sf_terms(st + a * 7, &p, &terms[a]);
    for (int a = 0; a < ns; ++a) sf_agent_step_terms(a, ns, st, terms, isp[a], 1.3 * isp[a], &p, &nv[2 * a], &nv[2 * a + 1]);
    for (int a = 0; a < ns; ++a) {
        st[a * 7 + 0] += nv[2 * a] * delta_t; st[a * 7 + 1] += nv[2 * a + 1] * delta_t;
        st[a * 7 + 2] = nv[2 * a]; st[a * 7 + 3] = nv[2 * a + 1];
    }
    free(terms);
}

/* one rvo2 doStep(): pos / vel [n][2] updated in place from the preferred velocities prf [n][2] */
ORC_API void orc_orca_step(float *pos, float *vel, const float *prf, const float *max_speed, int ns, float time_step,
                           float neighbor_dist, int max_neighbors, float time_horizon, float radius) {
    orca_params p;
    p.time_step = time_step; p.neighbor_dist = neighbor_dist; p.time_horizon = time_horizon; p.radius = radius;
    p.max_neighbors = max_neighbors;
    float *nvl = (float *)malloc(sizeof(float) * (size_t)ns * 2);
    for (int a = 0; a < ns; ++a)
        orca_agent_new_velocity(a, ns, pos, vel, prf[2 * a], prf[2 * a + 1], max_speed[a], &p, &nvl[2 * a], &nvl[2 * a + 1], NULL);
    for (int a = 0; a < ns; ++a) {
        vel[2 * a] = nvl[2 * a]; vel[2 * a + 1] = nvl[2 * a + 1];
        pos[2 * a] += vel[2 * a] * time_step; pos[2 * a + 1] += vel[2 * a + 1] * time_step;
    }
    free(nvl);
}

/* KalmanFilter.em(X) followed by smooth(X)[-1]: model [46] = Q(16) R(4) m0(4) P0(16) + 6 spare, x_last [4] */
ORC_API void orc_kalman_em(const double *obs, int T, int n_iter, double q0, double r0, double *model, double *x_last) {
    kf_model md;
    for (int k = 0; k < 16; ++k) { md.Q[k] = (k % 5 == 0) ? q0 : 0.0; md.P0[k] = (k % 5 == 0) ? 1.0 : 0.0; }
    md.R[0] = r0; md.R[1] = 0.0; md.R[2] = 0.0; md.R[3] = r0;
    md.m0[0] = obs[0]; md.m0[1] = 0.0; md.m0[2] = obs[1]; md.m0[3] = 0.0;
    kf_em_smooth(obs, T, n_iter, &md, x_last);
    for (int k = 0; k < 16; ++k) { model[k] = md.Q[k]; model[24 + k] = md.P0[k]; }
    for (int k = 0; k < 4; ++k) { model[16 + k] = md.R[k]; model[20 + k] = md.m0[k]; }
}

/* one KalmanFilter.sample(n_steps, initial_state=x0): observations [n_steps][2] from draws z [n_steps][6] */
ORC_API void orc_kalman_sample(const double *model, const double *x0, int n_steps, const double *z, double *out) {
    kf_model md;
    for (int k = 0; k < 16; ++k) { md.Q[k] = model[k]; md.P0[k] = model[24 + k]; }
    for (int k = 0; k < 4; ++k) { md.R[k] = model[16 + k]; md.m0[k] = model[20 + k]; }
    kf_sample_mean(&md, x0, n_steps, 1, z, out);
}
