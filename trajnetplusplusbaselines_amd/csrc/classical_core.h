/* Arithmetic core of the classical predictors (social force, ORCA, constant-velocity Kalman), written once in
 * plain C so that hipcc (device, gfx950) and gcc (oracle/classical_oracle.c, host) compile the same formulae.
 *
 * The reference only wraps three un-vendored third-party packages (classical/socialforce.py:6-8,89-93,
 * classical/orca.py:4,90-119, classical/kalman.py:2,40-60); their sources are not in the reference tree and are not
 * installed, so what follows restates the PUBLISHED algorithms those wrappers call:
 *   - Helbing & Molnar social force as implemented by svenkreiss/socialforce 0.1 (numpy era): desired-velocity
 *     relaxation, pedestrian-pedestrian potential V = v0*exp(-b/sigma) with the elliptical b, gradient by forward
 *     finite difference (delta = 1e-3), 200-degree field of view (weight 0.5 outside), speed cap 1.3x initial speed,
 *     explicit Euler;  float64.
 *   - ORCA of van den Berg et al. as in the RVO2 library v2.0 (snape/RVO2): k-nearest neighbours within
 *     neighborDist, one half-plane per neighbour (cut-off circle / legs / collision case), linearProgram1/2/3 with
 *     RVO_EPSILON = 1e-5;  float32.
 *   - Linear-Gaussian state-space EM / RTS smoothing as in pykalman's standard KalmanFilter (em_vars = transition
 *     covariance, observation covariance, initial state mean and covariance; 10 iterations);  float64.
 * Parity for these rows is UNPINNED (no reference test or golden vector exists, the Kalman predictor is stochastic).
 */
#ifndef TNP_CLASSICAL_CORE_H
#define TNP_CLASSICAL_CORE_H

#include <math.h>
#include <stddef.h>

#if defined(__HIPCC__)
#define TNP_HD __host__ __device__ static inline
#else
#define TNP_HD static inline
#endif

/* ============================================================================================
 * Social force.  state row = [x, y, vx, vy, gx, gy, tau]
 * ========================================================================================== */
typedef struct { double delta_t, v0, sigma, cosphi, out_of_view; } sf_params;

/* Terms of agent b that every pair (a, b) of a simulation step needs and that depend on b's state only: desired direction
 * e_b, d e_b and d^2 with d = delta_t * |v_b|.  The GPU kernel computes them once per agent and step (LDS table) instead
 * of once per ordered pair; the products are formed exactly as the per-pair expressions were ((delta_t * speed_b) * e_b),
 * so hoisting them changes no result bit. */
typedef struct { double ex, ey, dex, dey, d2, pad; } sf_agent_terms;

TNP_HD void sf_terms(const double *sb /* state row */, const sf_params *p, sf_agent_terms *t) {
    const double speed_b = sqrt(sb[2] * sb[2] + sb[3] * sb[3]);
    double gx = sb[4] - sb[0], gy = sb[5] - sb[1];
    const double gn = sqrt(gx * gx + gy * gy);
    t->ex = gx / gn; t->ey = gy / gn;
    const double d = p->delta_t * speed_b;
    t->dex = d * t->ex; t->dey = d * t->ey; t->d2 = d * d; t->pad = 0.0;
}

/* V(r_ab) for the pair (a, b): r = r_a - r_b (+ finite-difference offset), b's terms from sf_terms */
TNP_HD double sf_potential(double rx, double ry, const sf_agent_terms *tb, const sf_params *p) {
    const double n1 = sqrt(rx * rx + ry * ry);
    const double sx = rx - tb->dex, sy = ry - tb->dey;
    const double n2 = sqrt(sx * sx + sy * sy);
    const double in_sqrt = (n1 + n2) * (n1 + n2) - tb->d2;
    const double b = 0.5 * sqrt(in_sqrt);
    return p->v0 * exp(-b / p->sigma);
}

/* social force on agent a from the OLD state of all n agents of the scene (terms[b] = sf_terms of that state); returns the
 * new velocity (capped) */
TNP_HD void sf_agent_step_terms(int a, int n, const double *st /* [n][7] */, const sf_agent_terms *terms, double initial_speed,
                                double max_speed, const sf_params *p, double *vx_new, double *vy_new) {
    const double *sa = st + 7 * a;
    const double eax = terms[a].ex, eay = terms[a].ey;   /* desired direction e_a */
    const double tau = sa[6];
    double Fx = 1.0 / tau * (initial_speed * eax - sa[2]);
    double Fy = 1.0 / tau * (initial_speed * eay - sa[3]);
    const double delta = 1e-3;
    double sum_x = 0.0, sum_y = 0.0;
    for (int b = 0; b < n; ++b) {
        if (b == a) continue;                          /* diagonal: zero force, zero weight */
        const double *sb = st + 7 * b;
        const sf_agent_terms *tb = terms + b;
        const double rx = sa[0] - sb[0], ry = sa[1] - sb[1];
        const double v = sf_potential(rx, ry, tb, p);
        const double dvdx = (sf_potential(rx + delta, ry, tb, p) - v) / delta;
        const double dvdy = (sf_potential(rx, ry + delta, tb, p) - v) / delta;
        const double fx = -1.0 * dvdx, fy = -1.0 * dvdy;     /* f_ab = -grad V */
        /* field of view: e_a . (-f_ab) > |f_ab| cos(phi) */
        const double in_sight = (eax * (-fx) + eay * (-fy)) > sqrt(fx * fx + fy * fy) * p->cosphi;
        const double w = in_sight ? 1.0 : p->out_of_view;
        sum_x += w * fx; sum_y += w * fy;
    }
    Fx += sum_x; Fy += sum_y;
    const double wx = sa[2] + p->delta_t * Fx, wy = sa[3] + p->delta_t * Fy;
    const double ws = sqrt(wx * wx + wy * wy);
    const double factor = fmin(1.0, max_speed / ws);
    *vx_new = wx * factor; *vy_new = wy * factor;
}

/* ============================================================================================
 * ORCA (RVO2 v2.0).  float32 throughout.
 * ========================================================================================== */
#define ORCA_MAX_NEIGHBORS 16
#define ORCA_EPSILON 0.00001f

typedef struct { float px, py, dx, dy; } orca_line;     /* point, direction */
typedef struct { float time_step, neighbor_dist, time_horizon, radius; int max_neighbors; } orca_params;

TNP_HD float orca_det(float ax, float ay, float bx, float by) { return ax * by - ay * bx; }

TNP_HD int orca_lp1(const orca_line *lines, int line_no, float radius, float ox, float oy, int direction_opt,
                    float *rx, float *ry) {
    const orca_line *L = &lines[line_no];
    const float dot = L->px * L->dx + L->py * L->dy;
    const float disc = dot * dot + radius * radius - (L->px * L->px + L->py * L->py);
    if (disc < 0.0f) return 0;
    const float sq = sqrtf(disc);
    float t_left = -dot - sq, t_right = -dot + sq;
    for (int i = 0; i < line_no; ++i) {
        const float den = orca_det(L->dx, L->dy, lines[i].dx, lines[i].dy);
        const float num = orca_det(lines[i].dx, lines[i].dy, L->px - lines[i].px, L->py - lines[i].py);
        if (fabsf(den) <= ORCA_EPSILON) {
            if (num < 0.0f) return 0;
            continue;
        }
        const float t = num / den;
        if (den >= 0.0f) t_right = fminf(t_right, t);
        else t_left = fmaxf(t_left, t);
        if (t_left > t_right) return 0;
    }
    if (direction_opt) {
        if (ox * L->dx + oy * L->dy > 0.0f) { *rx = L->px + t_right * L->dx; *ry = L->py + t_right * L->dy; }
        else { *rx = L->px + t_left * L->dx; *ry = L->py + t_left * L->dy; }
    } else {
        const float t = L->dx * (ox - L->px) + L->dy * (oy - L->py);
        if (t < t_left) { *rx = L->px + t_left * L->dx; *ry = L->py + t_left * L->dy; }
        else if (t > t_right) { *rx = L->px + t_right * L->dx; *ry = L->py + t_right * L->dy; }
        else { *rx = L->px + t * L->dx; *ry = L->py + t * L->dy; }
    }
    return 1;
}

TNP_HD int orca_lp2(const orca_line *lines, int n_lines, float radius, float ox, float oy, int direction_opt,
                    float *rx, float *ry) {
    if (direction_opt) { *rx = ox * radius; *ry = oy * radius; }
    else if (ox * ox + oy * oy > radius * radius) {
        const float inv = 1.0f / sqrtf(ox * ox + oy * oy);      /* normalize(): multiply by the reciprocal */
        *rx = ox * inv * radius; *ry = oy * inv * radius;
    } else { *rx = ox; *ry = oy; }
    for (int i = 0; i < n_lines; ++i) {
        if (orca_det(lines[i].dx, lines[i].dy, lines[i].px - *rx, lines[i].py - *ry) > 0.0f) {
            const float tx = *rx, ty = *ry;
            if (!orca_lp1(lines, i, radius, ox, oy, direction_opt, rx, ry)) { *rx = tx; *ry = ty; return i; }
        }
    }
    return n_lines;
}

/* proj: ORCA_MAX_NEIGHBORS lines of scratch (the GPU kernel hands LDS, the host a local array) */
TNP_HD void orca_lp3_buf(const orca_line *lines, int n_lines, int begin_line, float radius, float *rx, float *ry, orca_line *proj) {
    float distance = 0.0f;
    for (int i = begin_line; i < n_lines; ++i) {
        if (orca_det(lines[i].dx, lines[i].dy, lines[i].px - *rx, lines[i].py - *ry) > distance) {
            int np = 0;
            for (int j = 0; j < i; ++j) {
                orca_line l;
                const float determinant = orca_det(lines[i].dx, lines[i].dy, lines[j].dx, lines[j].dy);
                if (fabsf(determinant) <= ORCA_EPSILON) {
                    if (lines[i].dx * lines[j].dx + lines[i].dy * lines[j].dy > 0.0f) continue;
                    l.px = 0.5f * (lines[i].px + lines[j].px); l.py = 0.5f * (lines[i].py + lines[j].py);
                } else {
                    const float t = orca_det(lines[j].dx, lines[j].dy, lines[i].px - lines[j].px, lines[i].py - lines[j].py) / determinant;
                    l.px = lines[i].px + t * lines[i].dx; l.py = lines[i].py + t * lines[i].dy;
                }
                const float ddx = lines[j].dx - lines[i].dx, ddy = lines[j].dy - lines[i].dy;
                const float inv = 1.0f / sqrtf(ddx * ddx + ddy * ddy);
                l.dx = ddx * inv; l.dy = ddy * inv;
                proj[np++] = l;
            }
            const float tx = *rx, ty = *ry;
            if (orca_lp2(proj, np, radius, -lines[i].dy, lines[i].dx, 1, rx, ry) < np) { *rx = tx; *ry = ty; }
            distance = orca_det(lines[i].dx, lines[i].dy, lines[i].px - *rx, lines[i].py - *ry);
        }
    }
}

TNP_HD void orca_lp3(const orca_line *lines, int n_lines, int begin_line, float radius, float *rx, float *ry) {
    orca_line proj[ORCA_MAX_NEIGHBORS];
    orca_lp3_buf(lines, n_lines, begin_line, radius, rx, ry, proj);
}

/* Agent::computeNewVelocity, the half-plane of ONE neighbour: rp = its position - ours, rv = our velocity - its velocity,
 * (vx, vy) our velocity.  Shared by the generic form below and the register form of csrc/classical.hip. */
TNP_HD orca_line orca_make_line(float rpx, float rpy, float rvx, float rvy, float vx, float vy, float inv_th,
                                const orca_params *p) {
    const float dist_sq = rpx * rpx + rpy * rpy;
    const float cr = p->radius + p->radius;
    const float cr_sq = cr * cr;
    orca_line l;
    float ux, uy;
    if (dist_sq > cr_sq) {
        const float wx = rvx - inv_th * rpx, wy = rvy - inv_th * rpy;
        const float wlen_sq = wx * wx + wy * wy;
        const float dot1 = wx * rpx + wy * rpy;
        if (dot1 < 0.0f && dot1 * dot1 > cr_sq * wlen_sq) {
            const float wlen = sqrtf(wlen_sq);
            const float iw = 1.0f / wlen;
            const float uwx = wx * iw, uwy = wy * iw;
            l.dx = uwy; l.dy = -uwx;
            const float s = cr * inv_th - wlen;
            ux = s * uwx; uy = s * uwy;
        } else {
            const float leg = sqrtf(dist_sq - cr_sq);
            const float ids = 1.0f / dist_sq;
            if (orca_det(rpx, rpy, wx, wy) > 0.0f) {
                l.dx = (rpx * leg - rpy * cr) * ids; l.dy = (rpx * cr + rpy * leg) * ids;
            } else {
                l.dx = -((rpx * leg + rpy * cr) * ids); l.dy = -((-rpx * cr + rpy * leg) * ids);
            }
            const float dot2 = rvx * l.dx + rvy * l.dy;
            ux = dot2 * l.dx - rvx; uy = dot2 * l.dy - rvy;
        }
    } else {
        const float inv_ts = 1.0f / p->time_step;
        const float wx = rvx - inv_ts * rpx, wy = rvy - inv_ts * rpy;
        const float wlen = sqrtf(wx * wx + wy * wy);
        const float iw = 1.0f / wlen;
        const float uwx = wx * iw, uwy = wy * iw;
        l.dx = uwy; l.dy = -uwx;
        const float s = cr * inv_ts - wlen;
        ux = s * uwx; uy = s * uwy;
    }
    l.px = vx + 0.5f * ux; l.py = vy + 0.5f * uy;
    return l;
}

/* new velocity of agent a from the OLD positions / velocities of the n agents of its scene.
 * nbr_out (optional, ORCA_MAX_NEIGHBORS ints, -1 padded) receives the neighbour indices in processing order. */
TNP_HD void orca_agent_new_velocity(int a, int n, const float *pos, const float *vel, float prefx, float prefy,
                                    float max_speed, const orca_params *p, float *nvx, float *nvy, int *nbr_out) {
    /* Agent::computeNeighbors / insertAgentNeighbor: the max_neighbors closest agents within neighbor_dist, sorted */
    int nbr[ORCA_MAX_NEIGHBORS];
    float nd[ORCA_MAX_NEIGHBORS];
    int cnt = 0;
    float range_sq = p->neighbor_dist * p->neighbor_dist;
    const float ax = pos[2 * a], ay = pos[2 * a + 1];
    for (int b = 0; b < n; ++b) {
        if (b == a) continue;
        const float ddx = ax - pos[2 * b], ddy = ay - pos[2 * b + 1];
        const float dist_sq = ddx * ddx + ddy * ddy;
        if (dist_sq < range_sq) {
            if (cnt < p->max_neighbors) ++cnt;
            int i = cnt - 1;
            while (i != 0 && dist_sq < nd[i - 1]) { nd[i] = nd[i - 1]; nbr[i] = nbr[i - 1]; --i; }
            nd[i] = dist_sq; nbr[i] = b;
            if (cnt == p->max_neighbors) range_sq = nd[cnt - 1];
        }
    }
    if (nbr_out) for (int i = 0; i < ORCA_MAX_NEIGHBORS; ++i) nbr_out[i] = i < cnt ? nbr[i] : -1;
    /* Agent::computeNewVelocity */
    orca_line lines[ORCA_MAX_NEIGHBORS];
    const float inv_th = 1.0f / p->time_horizon;
    const float vx = vel[2 * a], vy = vel[2 * a + 1];
    for (int k = 0; k < cnt; ++k) {
        const int b = nbr[k];
        const float rpx = pos[2 * b] - ax, rpy = pos[2 * b + 1] - ay;
        const float rvx = vx - vel[2 * b], rvy = vy - vel[2 * b + 1];
        const orca_line l = orca_make_line(rpx, rpy, rvx, rvy, vx, vy, inv_th, p);
        lines[k] = l;
    }
    float rx, ry;
    const int fail = orca_lp2(lines, cnt, max_speed, prefx, prefy, 0, &rx, &ry);
    if (fail < cnt) orca_lp3(lines, cnt, fail, max_speed, &rx, &ry);
    *nvx = rx; *nvy = ry;
}

/* ============================================================================================
 * Constant-velocity Kalman filter with EM (pykalman-style), one track.  State [x, vx, y, vy].
 * ========================================================================================== */
#define KF_MAX_T 16

typedef struct { double Q[16], R[4], m0[4], P0[16]; } kf_model;

TNP_HD void kf_mat4_mul(const double *A, const double *B, double *C) {        /* C = A B (4x4) */
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        double s = 0.0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 4 + j]; C[i * 4 + j] = s; }
}
TNP_HD void kf_mat4_mul_t(const double *A, const double *B, double *C) {      /* C = A B^T */
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) {
        double s = 0.0; for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[j * 4 + k]; C[i * 4 + j] = s; }
}
TNP_HD int kf_inv4(const double *M, double *inv) {                            /* Gauss-Jordan with partial pivoting */
    double a[4][8];
    for (int i = 0; i < 16; ++i) inv[i] = 0.0;          /* singular input: zero result */
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { a[i][j] = M[i * 4 + j]; a[i][4 + j] = (i == j) ? 1.0 : 0.0; }
    for (int c = 0; c < 4; ++c) {
        int piv = c; double best = fabs(a[c][c]);
        for (int r = c + 1; r < 4; ++r) if (fabs(a[r][c]) > best) { best = fabs(a[r][c]); piv = r; }
        if (best == 0.0) return 0;
        /* the pivot row is swapped in by comparing every candidate row's index with `piv`: all array indices stay
         * compile-time constants once the loops are unrolled (a[piv][j] made the GPU keep `a` in scratch memory) */
        for (int r = c + 1; r < 4; ++r)
            if (r == piv) for (int j = 0; j < 8; ++j) { double t = a[c][j]; a[c][j] = a[r][j]; a[r][j] = t; }
        const double d = a[c][c];
        for (int j = 0; j < 8; ++j) a[c][j] /= d;
        for (int r = 0; r < 4; ++r) if (r != c) { const double f = a[r][c]; for (int j = 0; j < 8; ++j) a[r][j] -= f * a[c][j]; }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) inv[i * 4 + j] = a[i][4 + j];
    return 1;
}

/* transition A = [[1,1,0,0],[0,1,0,0],[0,0,1,1],[0,0,0,1]], observation C picks x (0) and y (2) */
TNP_HD void kf_A_mul(const double *x, double *y) { y[0] = x[0] + x[1]; y[1] = x[1]; y[2] = x[2] + x[3]; y[3] = x[3]; }
/* Products with the constant-velocity transition matrix written out: a general 4 x 4 product with A is 64 multiply-adds of which
 * 40 multiply by the literal 0 and 24 by the literal 1; x * 1 is x, x * 0 is 0 and s + 0 is s for the finite values a filter
 * holds, so the sums below are the values the general product returns (a -0 becomes +0), at 8 additions instead of 64 fmas.
 * Six of the ten products of a filter / smoother step pair are of this kind (round 6: 6.4 -> 4.x ms for 524 288 tracks). */
TNP_HD void kf_AX(const double *X, double *out) {         /* A X:   rows (x0 + x1, x1, x2 + x3, x3) */
    for (int j = 0; j < 4; ++j) {
        out[0 * 4 + j] = X[0 * 4 + j] + X[1 * 4 + j]; out[1 * 4 + j] = X[1 * 4 + j];
        out[2 * 4 + j] = X[2 * 4 + j] + X[3 * 4 + j]; out[3 * 4 + j] = X[3 * 4 + j];
    }
}
TNP_HD void kf_XAt(const double *X, double *out) {        /* X A^T: columns (c0 + c1, c1, c2 + c3, c3) */
    for (int i = 0; i < 4; ++i) {
        out[i * 4 + 0] = X[i * 4 + 0] + X[i * 4 + 1]; out[i * 4 + 1] = X[i * 4 + 1];
        out[i * 4 + 2] = X[i * 4 + 2] + X[i * 4 + 3]; out[i * 4 + 3] = X[i * 4 + 3];
    }
}
TNP_HD void kf_APAt(const double *P, double *out) {       /* A P A^T */
    double T[16];
    kf_AX(P, T);
    kf_XAt(T, out);
}

/* forward filter; obs [T][2] -> filtered moments xf [T][4], Pf [T][16] */
TNP_HD void kf_filter(const kf_model *md, const double *obs, int T, double *xf, double *Pf) {
    double xpt[4], Ppt[16];
    for (int t = 0; t < T; ++t) {
        double *xft = xf + 4 * t, *Pft = Pf + 16 * t;
        if (t == 0) { for (int i = 0; i < 4; ++i) xpt[i] = md->m0[i]; for (int i = 0; i < 16; ++i) Ppt[i] = md->P0[i]; }
        else {
            kf_A_mul(xf + 4 * (t - 1), xpt);
            kf_APAt(Pf + 16 * (t - 1), Ppt);
            for (int i = 0; i < 16; ++i) Ppt[i] += md->Q[i];
        }
        /* S = C P C^T + R (2x2), K = P C^T S^-1 (4x2) */
        const double S00 = Ppt[0] + md->R[0], S01 = Ppt[2] + md->R[1], S10 = Ppt[8] + md->R[2], S11 = Ppt[10] + md->R[3];
        const double det = S00 * S11 - S01 * S10;
        const double i00 = S11 / det, i01 = -S01 / det, i10 = -S10 / det, i11 = S00 / det;
        double K[8];
        for (int i = 0; i < 4; ++i) {
            const double pc0 = Ppt[i * 4 + 0], pc1 = Ppt[i * 4 + 2];
            K[i * 2 + 0] = pc0 * i00 + pc1 * i10; K[i * 2 + 1] = pc0 * i01 + pc1 * i11;
        }
        const double r0 = obs[2 * t] - xpt[0], r1 = obs[2 * t + 1] - xpt[2];
        for (int i = 0; i < 4; ++i) xft[i] = xpt[i] + K[i * 2] * r0 + K[i * 2 + 1] * r1;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j)      /* P - K C P */
            Pft[i * 4 + j] = Ppt[i * 4 + j] - (K[i * 2] * Ppt[0 * 4 + j] + K[i * 2 + 1] * Ppt[2 * 4 + j]);
    }
}

/* EM (n_iter iterations) on Q, R, m0, P0, then a final filter pass; returns the last smoothed state (= the last FILTERED
 * state: the RTS recursion starts from it) in x_last.
 *
 * Round 6: the RTS smoother runs backwards from t = T - 1 and the M-step sums are accumulated AS the smoothed moments appear
 * (descending t) instead of in a second loop over stored xs / Ps / Lg arrays: per track only xf and Pf (20 doubles per frame)
 * live across the passes instead of 56 -- on the GPU that is the difference between 7.4 KB and 2.6 KB of private memory per
 * lane (profiles/round6_pmc_classical.md).  The sums are the same terms in the opposite order (last-bit differences). */
TNP_HD void kf_em_smooth(const double *obs, int T, int n_iter, kf_model *md, double *x_last) {
    double xf[KF_MAX_T * 4], Pf[KF_MAX_T * 16];
    for (int it = 0; it < n_iter; ++it) {
        kf_filter(md, obs, T, xf, Pf);
        double xn[4], Pn[16];                      /* smoothed moments of step t + 1 */
        for (int i = 0; i < 4; ++i) xn[i] = xf[4 * (T - 1) + i];
        for (int i = 0; i < 16; ++i) Pn[i] = Pf[16 * (T - 1) + i];
        double R[4], Q[16];
        {   /* observation covariance, term of t = T - 1 */
            const double e0 = obs[2 * (T - 1)] - xn[0], e1 = obs[2 * (T - 1) + 1] - xn[2];
            R[0] = e0 * e0 + Pn[0]; R[1] = e0 * e1 + Pn[2]; R[2] = e1 * e0 + Pn[8]; R[3] = e1 * e1 + Pn[10];
        }
        for (int i = 0; i < 16; ++i) Q[i] = 0.0;
        for (int t = T - 2; t >= 0; --t) {
            double xpt[4], Ppt[16], inv[16], PAt[16], L[16], xt[4], Pt[16];
            /* predicted moments of step t + 1 from the filtered ones of step t, smoother gain L_t */
            kf_A_mul(xf + 4 * t, xpt);
            kf_APAt(Pf + 16 * t, Ppt);
            for (int i = 0; i < 16; ++i) Ppt[i] += md->Q[i];
            kf_inv4(Ppt, inv);
            kf_XAt(Pf + 16 * t, PAt);                      /* P_f A^T */
            kf_mat4_mul(PAt, inv, L);
            double d[4];
            for (int i = 0; i < 4; ++i) d[i] = xn[i] - xpt[i];
            for (int i = 0; i < 4; ++i) { double sacc = xf[4 * t + i]; for (int k = 0; k < 4; ++k) sacc += L[i * 4 + k] * d[k]; xt[i] = sacc; }
            double D[16], LD[16], LDLt[16];
            for (int i = 0; i < 16; ++i) D[i] = Pn[i] - Ppt[i];
            kf_mat4_mul(L, D, LD);
            kf_mat4_mul_t(LD, L, LDLt);
            for (int i = 0; i < 16; ++i) Pt[i] = Pf[16 * t + i] + LDLt[i];
            /* M-step terms of this t */
            const double e0 = obs[2 * t] - xt[0], e1 = obs[2 * t + 1] - xt[2];
            R[0] += e0 * e0 + Pt[0]; R[1] += e0 * e1 + Pt[2]; R[2] += e1 * e0 + Pt[8]; R[3] += e1 * e1 + Pt[10];
            double ax[4], err[4], APA[16], pair[16], VA[16];
            kf_A_mul(xt, ax);
            for (int i = 0; i < 4; ++i) err[i] = xn[i] - ax[i];
            kf_APAt(Pt, APA);
            kf_mat4_mul_t(Pn, L, pair);                    /* Cov(x_{t+1}, x_t) = P^s_{t+1} L_t^T */
            kf_XAt(pair, VA);                              /* V_{t+1,t} A^T */
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j)
                Q[i * 4 + j] += err[i] * err[j] + APA[i * 4 + j] + Pn[i * 4 + j] - VA[i * 4 + j] - VA[j * 4 + i];
            for (int i = 0; i < 4; ++i) xn[i] = xt[i];
            for (int i = 0; i < 16; ++i) Pn[i] = Pt[i];
        }
        for (int i = 0; i < 4; ++i) md->R[i] = R[i] / (double)T;
        if (T > 1) for (int i = 0; i < 16; ++i) md->Q[i] = Q[i] / (double)(T - 1);
        for (int i = 0; i < 4; ++i) md->m0[i] = xn[i];
        for (int i = 0; i < 16; ++i) md->P0[i] = Pn[i];
    }
    kf_filter(md, obs, T, xf, Pf);
    for (int i = 0; i < 4; ++i) x_last[i] = xf[4 * (T - 1) + i];
}

/* lower Cholesky factor of a symmetric PSD n x n matrix (n <= 4), zero pivots tolerated */
TNP_HD void kf_chol(const double *M, int n, double *L) {
    for (int i = 0; i < n * n; ++i) L[i] = 0.0;
    for (int j = 0; j < n; ++j) {
        double s = M[j * n + j];
        for (int k = 0; k < j; ++k) s -= L[j * n + k] * L[j * n + k];
        const double d = s > 0.0 ? sqrt(s) : 0.0;
        L[j * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double v = M[i * n + j];
            for (int k = 0; k < j; ++k) v -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = d > 0.0 ? v / d : 0.0;
        }
    }
}

/* mean of n_samples sampled observation sequences (pykalman KalmanFilter.sample from initial_state), given standard
 * normal draws z [n_samples][n_steps][6] (4 state-noise + 2 observation-noise components per step).
 * out [n_steps][2] = averaged observations (step 0 = the initial state, no transition noise). */
TNP_HD void kf_sample_mean(const kf_model *md, const double *x0, int n_steps, int n_samples, const double *z, double *out) {
    double LQ[16], LR[4];
    kf_chol(md->Q, 4, LQ);
    kf_chol(md->R, 2, LR);
    for (int t = 0; t < n_steps * 2; ++t) out[t] = 0.0;
    for (int s = 0; s < n_samples; ++s) {
        double x[4] = {x0[0], x0[1], x0[2], x0[3]};
        for (int t = 0; t < n_steps; ++t) {
            const double *zt = z + ((size_t)s * n_steps + t) * 6;
            if (t > 0) {
                double ax[4];
                kf_A_mul(x, ax);
                for (int i = 0; i < 4; ++i) { double n = 0.0; for (int k = 0; k <= i; ++k) n += LQ[i * 4 + k] * zt[k]; x[i] = ax[i] + n; }
            }
            out[2 * t + 0] += x[0] + LR[0] * zt[4];
            out[2 * t + 1] += x[2] + LR[2] * zt[4] + LR[3] * zt[5];
        }
    }
    for (int t = 0; t < n_steps * 2; ++t) out[t] /= (double)n_samples;
}

#endif /* TNP_CLASSICAL_CORE_H */
