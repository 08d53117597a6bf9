/*
 * oracle/trajnet_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C, CPU restatement of the reference's LSTM + grid-pooling hot path
 * (vita-epfl/trajnetplusplusbaselines).  Nothing in the product package
 * (trajnetplusplusbaselines_amd/) may import, link or call this file; it is
 * used by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg as
 * the CHECKER, never as the thing that is shipped or measured as the product.
 *
 * Pinned against the real reference: oracle/gen_golden.py imports the Python
 * reference from /root/reference (in the build container), runs it on seeded
 * inputs and stores inputs + outputs under tests/golden/; tests/test_oracle_*.py
 * check this restatement against those files (cell ids bit-exact, floats to
 * 2e-5) and against the reference's own adapted known-answer vectors
 * (reference tests/test_pooling.py:9-99, tests/test_lstm_modules.py:5-14).
 *
 * Each function cites the reference file:line it restates (paths relative to
 * the reference checkout, package trajnetbaselines/).
 *
 * Arithmetic is IEEE fp32 without contraction (compile with -ffp-contract=off)
 * so that the integer cell ids match the reference's CPU result bit for bit.
 * Dense layers accumulate sequentially over k per output (axpy form); the
 * reference uses a blocked BLAS, so float outputs agree to rounding only.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

enum { ORC_OCCUPANCY = 0, ORC_DIRECTIONAL = 1, ORC_SOCIAL = 2, ORC_NOPOOL = -1,
       ORC_NN = 4,        /* NearestNeighborMLP, lstm/non_gridbased_pooling.py:64-147 */
       ORC_HIDDENMLP = 5, /* HiddenStateMLPPooling, lstm/non_gridbased_pooling.py:150-239 */
       ORC_ATTNMLP = 6,   /* AttentionMLPPooling, lstm/non_gridbased_pooling.py:242-351 */
       ORC_NNLSTM = 7,    /* NearestNeighborLSTM, lstm/non_gridbased_pooling.py:354-455 */
       ORC_TRAJ = 8       /* TrajectronPooling, lstm/non_gridbased_pooling.py:457-538 */ };

/* torch.nan_to_num defaults (lstm/gridbased_pooling.py:140,166): nan->0, +-inf->+-FLT_MAX */
static inline float nan_to_num_f(float v) {
    if (v != v) return 0.0f;
    if (isinf(v)) return v > 0 ? FLT_MAX : -FLT_MAX;
    return v;
}

static inline float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

/* ------------------------------------------------------------------------- *
 * Cell indexing: lstm/gridbased_pooling.py:245-288 (GridBasedPooling.occupancy)
 *   obs      [B,N,2]   positions, NaN = absent / padded slot (NOT modified)
 *   oi       [B,N,N-1] flat cell id  ox*G + oy   (0 for out-of-range)
 *   in_range [B,N,N-1] 1 if 0 <= oij < G on both axes
 *   G = n*pool_size, cell = float32(cell_side / pool_size),
 *   half_x = G/2, half_y = G/2 (front: 0)   (:273-276)
 * ------------------------------------------------------------------------- */
ORC_API void orc_cell_ids(const float *obs, int B, int N, int G, float cell,
                          float half_x, float half_y, int64_t *oi, uint8_t *in_range) {
    if (N < 2) return;
    float *p = (float *)malloc(sizeof(float) * (size_t)N * 2);
    for (int b = 0; b < B; ++b) {
        /* :247-249  mask = isnan(obs).any(-1); obs[mask] = -500 */
        for (int j = 0; j < N; ++j) {
            float x = obs[((size_t)b * N + j) * 2 + 0], y = obs[((size_t)b * N + j) * 2 + 1];
            if (x != x || y != y) { x = -500.0f; y = -500.0f; }
            p[2 * j] = x; p[2 * j + 1] = y;
        }
        for (int i = 0; i < N; ++i) {
            int jj = 0;
            for (int j = 0; j < N; ++j) {
                if (j == i) continue; /* :259-263 diagonal removed, j ascending */
                /* :257-258 relative = unfolded - obs.unsqueeze(2)  -> pos[j] - pos[i] */
                float rx = p[2 * j] - p[2 * i];
                float ry = p[2 * j + 1] - p[2 * i + 1];
                /* :276 oij = relative / (cell_side/pool_size) + n*pool_size/2  (fp32 div, fp32 add) */
                float ox = rx / cell + half_x;
                float oy = ry / cell + half_y;
                /* :278-279 */
                int viol = (ox < 0.0f) + (ox >= (float)G) + (oy < 0.0f) + (oy >= (float)G);
                size_t o = ((size_t)b * N + i) * (N - 1) + jj;
                if (viol == 0) {
                    /* :284,287  .long() truncation; oi = ox*G + oy */
                    oi[o] = (int64_t)ox * G + (int64_t)oy;
                    in_range[o] = 1;
                } else {
                    oi[o] = 0; /* :281 oij[~range_mask] = 0 */
                    in_range[o] = 0;
                }
                ++jj;
            }
        }
    }
    free(p);
}

/* avg_pool2d on one [Hin,Win] plane, count_include_pad=True, ceil_mode=False
 * (torch CPU kernel semantics: sum over the valid window, divide by the
 * padded-window size).  Used for blur (:297-301) and lp_pool2d (:303). */
static void avg_pool2d_plane(const float *in, int Hin, int Win, int k, int stride, int pad,
                             float *out, int Hout, int Wout) {
    for (int oh = 0; oh < Hout; ++oh)
        for (int ow = 0; ow < Wout; ++ow) {
            int ih0 = oh * stride - pad, iw0 = ow * stride - pad;
            int ih1 = ih0 + k, iw1 = iw0 + k;
            if (ih1 > Hin + pad) ih1 = Hin + pad;
            if (iw1 > Win + pad) iw1 = Win + pad;
            int pool_size = (ih1 - ih0) * (iw1 - iw0);
            if (ih0 < 0) ih0 = 0;
            if (iw0 < 0) iw0 = 0;
            if (ih1 > Hin) ih1 = Hin;
            if (iw1 > Win) iw1 = Win;
            float sum = 0.0f;
            for (int ih = ih0; ih < ih1; ++ih)
                for (int iw = iw0; iw < iw1; ++iw) sum += in[ih * Win + iw];
            out[oh * Wout + ow] = sum / (float)pool_size;
        }
}

/* ------------------------------------------------------------------------- *
 * Grid build: GridBasedPooling.{occupancies,directional,social}+occupancy
 *   (lstm/gridbased_pooling.py:112-170, 227-305)
 *   obs1, obs2 [B,N,2] (prev / current positions, NaN = absent)
 *   values     social: enc [B,N,C] = hidden_dim_encoding(nan_to_num(hidden))
 *              (the reference encodes the unfolded [B,N,N-1,H] tensor, :160-167;
 *               row-wise identical); ignored otherwise
 *   grid out   [B*N, C, n, n]
 * ------------------------------------------------------------------------- */
ORC_API int orc_grid(int type, const float *obs1, const float *obs2, const float *values,
                     int B, int N, int C, int n, int pool_size, int blur_size,
                     double cell_side, float constant, int front, float *grid) {
    const int G = n * pool_size;
    const size_t plane = (size_t)G * G;
    /* :252-253 single-track shortcut: constant grid (shape [1,C,n,n] in the
     * reference; we fill all B*N rows, identical for the only legal case B=1) */
    if (N == 1) {
        for (size_t i = 0; i < (size_t)B * C * n * n; ++i) grid[i] = constant;
        return 0;
    }
    float cell = (float)(cell_side / (double)pool_size);
    float half_x = (float)((double)G / 2.0), half_y = front ? 0.0f : half_x;
    int64_t *oi = (int64_t *)malloc(sizeof(int64_t) * (size_t)B * N * (N - 1));
    uint8_t *inr = (uint8_t *)malloc((size_t)B * N * (N - 1));
    orc_cell_ids(obs2, B, N, G, cell, half_x, half_y, oi, inr);

    float *occ = (float *)malloc(sizeof(float) * plane * C);      /* [G*G][C] (:290) */
    float *occ_t = (float *)malloc(sizeof(float) * plane * C);    /* [C][G][G] (:294-295) */
    float *blur = (float *)malloc(sizeof(float) * (plane + 4 * (size_t)G + 4) * 4);
    float val[64];
    if (C > 64) return -1;
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < N; ++i) {
            for (size_t q = 0; q < plane * C; ++q) occ[q] = constant;
            int jj = 0;
            for (int j = 0; j < N; ++j) {
                if (j == i) continue;
                size_t o = ((size_t)b * N + i) * (N - 1) + jj;
                if (!inr[o]) {
                    for (int c = 0; c < C; ++c) val[c] = constant; /* :282 */
                } else if (type == ORC_OCCUPANCY) {
                    val[0] = 1.0f; /* :266-267 */
                } else if (type == ORC_DIRECTIONAL) {
                    /* :127-140 vel = obs2-obs1 ; relative = vel[j]-vel[i] ; nan_to_num */
                    for (int c = 0; c < 2; ++c) {
                        float vj = obs2[((size_t)b * N + j) * 2 + c] - obs1[((size_t)b * N + j) * 2 + c];
                        float vi = obs2[((size_t)b * N + i) * 2 + c] - obs1[((size_t)b * N + i) * 2 + c];
                        val[c] = nan_to_num_f(vj - vi);
                    }
                } else { /* social :160-167 */
                    for (int c = 0; c < C; ++c) val[c] = values[((size_t)b * N + j) * C + c];
                }
                /* :293 occ[arange, oi] = other_values : last writer wins, j ascending */
                for (int c = 0; c < C; ++c) occ[(size_t)oi[o] * C + c] = val[c];
                ++jj;
            }
            for (int c = 0; c < C; ++c)
                for (size_t q = 0; q < plane; ++q) occ_t[c * plane + q] = occ[q * C + c];
            float *dst = grid + ((size_t)b * N + i) * C * n * n;
            for (int c = 0; c < C; ++c) {
                const float *src = occ_t + c * plane;
                int Gb = G;
                if (blur_size != 1) { /* :297-301 */
                    int pad = blur_size / 2;
                    Gb = G + 2 * pad - blur_size + 1;
                    avg_pool2d_plane(src, G, G, blur_size, 1, pad, blur, Gb, Gb);
                    src = blur;
                }
                /* :303 lp_pool2d(x, 1, pool_size) = sign(a)*relu(|a|)*k*k, a = avg_pool2d(x,k) */
                int no = (Gb - pool_size) / pool_size + 1;
                if (no != n) return -2;
                float *a = blur + plane + 2 * (size_t)G + 2;
                avg_pool2d_plane(src, Gb, Gb, pool_size, pool_size, 0, a, no, no);
                for (int q = 0; q < no * no; ++q) {
                    float av = a[q];
                    float sg = (av > 0.0f) ? 1.0f : ((av < 0.0f) ? -1.0f : 0.0f);
                    float ab = fabsf(av);
                    dst[c * n * n + q] = (sg * (ab > 0.0f ? ab : 0.0f)) * (float)(pool_size * pool_size);
                }
            }
        }
    free(oi); free(inr); free(occ); free(occ_t); free(blur);
    return 0;
}

/* ------------------------------------------------------------------------- *
 * torch.nn.Linear (+ optional ReLU): y[m,:] = act(b + x[m,:] @ W^T), W [Nout,K].
 * axpy form, sequential in k per output; exact zeros of x are skipped (adds
 * of +-0 do not change a finite accumulator), which makes the sparse grid
 * layer cheap for the cpu_baseline timing.
 * ------------------------------------------------------------------------- */
/* core: Wt is the transposed weight [K,Nout] */
ORC_API void orc_linear_wt(const float *x, int M, int K, const float *Wt, const float *bias, int Nout,
                           int relu, float *y) {
#pragma omp parallel for schedule(dynamic, 4)
    for (int m = 0; m < M; ++m) {
        float *acc = y + (size_t)m * Nout;
        for (int o = 0; o < Nout; ++o) acc[o] = bias ? bias[o] : 0.0f;
        for (int k = 0; k < K; ++k) {
            float xv = x[(size_t)m * K + k];
            if (xv == 0.0f) continue;
            const float *w = Wt + (size_t)k * Nout;
            for (int o = 0; o < Nout; ++o) acc[o] += xv * w[o];
        }
        if (relu)
            for (int o = 0; o < Nout; ++o) acc[o] = acc[o] > 0.0f ? acc[o] : 0.0f;
    }
}

ORC_API void orc_linear(const float *x, int M, int K, const float *W, const float *bias, int Nout,
                        int relu, float *y) {
    float *Wt = (float *)malloc(sizeof(float) * (size_t)K * Nout);
    for (int o = 0; o < Nout; ++o)
        for (int k = 0; k < K; ++k) Wt[(size_t)k * Nout + o] = W[(size_t)o * K + k];
    orc_linear_wt(x, M, K, Wt, bias, Nout, relu, y);
    free(Wt);
}

/* W if no cached transpose is supplied, else the cached [K,Nout] copy */
static void linear_any(const float *x, int M, int K, const float *W, const float *Wt, const float *bias,
                       int Nout, int relu, float *y) {
    if (Wt) orc_linear_wt(x, M, K, Wt, bias, Nout, relu, y);
    else orc_linear(x, M, K, W, bias, Nout, relu, y);
}

/* InputEmbedding.forward, lstm/modules.py:24-30 : relu(Linear(vel*scale)) ++ two zero tag columns */
ORC_API void orc_input_embedding(const float *vel, int M, const float *W, const float *b, int E,
                                 float scale, float *out) {
    for (int m = 0; m < M; ++m) {
        float vx = vel[2 * m] * scale, vy = vel[2 * m + 1] * scale;
        for (int o = 0; o < E - 2; ++o) {
            float a = b[o];
            a += vx * W[2 * o];
            a += vy * W[2 * o + 1];
            out[(size_t)m * E + o] = a > 0.0f ? a : 0.0f;
        }
        out[(size_t)m * E + E - 2] = 0.0f;
        out[(size_t)m * E + E - 1] = 0.0f;
    }
}

/* torch.nn.LSTMCell (lstm/lstm.py:84-85,154): gate order i,f,g,o */
static void lstm_cell_any(const float *x, int M, int I, const float *h, const float *c, int H,
                          const float *Wih, const float *Whh, const float *WihT, const float *WhhT,
                          const float *bih, const float *bhh, float *h_out, float *c_out) {
    float *gi = (float *)malloc(sizeof(float) * (size_t)M * 4 * H);
    float *gh = (float *)malloc(sizeof(float) * (size_t)M * 4 * H);
    linear_any(x, M, I, Wih, WihT, bih, 4 * H, 0, gi);
    linear_any(h, M, H, Whh, WhhT, bhh, 4 * H, 0, gh);
    for (int m = 0; m < M; ++m)
        for (int u = 0; u < H; ++u) {
            size_t r = (size_t)m * 4 * H;
            float ig = sigmoid_f(gi[r + u] + gh[r + u]);
            float fg = sigmoid_f(gi[r + H + u] + gh[r + H + u]);
            float gg = tanhf(gi[r + 2 * H + u] + gh[r + 2 * H + u]);
            float og = sigmoid_f(gi[r + 3 * H + u] + gh[r + 3 * H + u]);
            float cn = fg * c[(size_t)m * H + u] + ig * gg;
            c_out[(size_t)m * H + u] = cn;
            h_out[(size_t)m * H + u] = og * tanhf(cn);
        }
    free(gi); free(gh);
}

ORC_API void orc_lstm_cell(const float *x, int M, int I, const float *h, const float *c, int H,
                           const float *Wih, const float *Whh, const float *bih, const float *bhh,
                           float *h_out, float *c_out) {
    lstm_cell_any(x, M, I, h, c, H, Wih, Whh, NULL, NULL, bih, bhh, h_out, c_out);
}

/* Hidden2Normal.forward, lstm/modules.py:56-64 */
ORC_API void orc_hidden2normal(const float *h, int M, int H, const float *W, const float *b, float *normal) {
    orc_linear(h, M, H, W, b, 5, 0, normal);
    for (int m = 0; m < M; ++m) {
        float *nr = normal + (size_t)m * 5;
        nr[2] = 0.01f + 0.2f * sigmoid_f(nr[2]);
        nr[3] = 0.01f + 0.2f * sigmoid_f(nr[3]);
        nr[4] = 0.7f * sigmoid_f(nr[4]);
    }
}

/* ------------------------------------------------------------------------- *
 * Model description (all weights in PyTorch layout [out,in], fp32).
 * ------------------------------------------------------------------------- */
typedef struct {
    int E;            /* embedding_dim (64) */
    int H;            /* hidden_dim (128) */
    int goal_flag;    /* lstm/lstm.py:73-76 */
    int goal_dim;
    int pool_type;    /* ORC_NOPOOL / OCCUPANCY / DIRECTIONAL / SOCIAL / NN / HIDDENMLP */
    int n;            /* cells per side */
    int C;            /* pooling_dim */
    int P;            /* pool out_dim */
    int n_layers;     /* embedding MLP depth: 1,2,3 */
    int dims[4];      /* dims[0]=C*n*n, dims[n_layers]=P */
    int front;
    int pool_size;
    int blur_size;
    float constant;
    double cell_side;
    const float *We, *be;     /* input_embedding.input_embeddings.0 [E-2,2] */
    const float *Wg, *bg;     /* goal_embedding.input_embeddings.0 [goal_dim-2,2] */
    const float *enc_Wih, *enc_Whh, *enc_bih, *enc_bhh;
    const float *dec_Wih, *dec_Whh, *dec_bih, *dec_bhh;
    const float *Wn, *bn;     /* hidden2normal.linear [5,H] */
    const float *Wh, *bh;     /* pool.hidden_dim_encoding [C,H] (social) */
    const float *Wp[3], *bp[3]; /* pool.embedding.{0,2,4} */
    /* optional cached transposes [in,out] of the big matrices (speed only; may be NULL) */
    const float *WpT[3];
    const float *enc_WihT, *enc_WhhT, *dec_WihT, *dec_WhhT;
    /* AttentionMLPPooling only: wq, wk, wv [D,D] (no bias), MultiheadAttention in_proj_weight [3D,D] / in_proj_bias [3D],
     * out_proj weight [D,D] / bias [D] */
    const float *att_wq, *att_wk, *att_wv, *att_in_w, *att_in_b, *att_out_w, *att_out_b;
    /* NearestNeighborLSTM / TrajectronPooling only: interaction-encoder pool_lstm (LSTMCell(P -> Hp)) and hidden2pool
     * Linear(Hp -> P) */
    int Hp;
    const float *pl_Wih, *pl_Whh, *pl_bih, *pl_bhh, *pl_Wo, *pl_bo;
    int pool_to_hidden; /* LSTM(pool_to_input=False): the interaction vector is ADDED to the hidden state (lstm/lstm.py:150-151) */
} orc_model;

/* One row of torch.nn.Linear (+ReLU) without the OpenMP region / transposed copy of orc_linear: same axpy order
 * (sequential in k per output, zeros skipped), for the per-pair embeddings of the non-grid modules. */
static void small_linear(const float *x, int K, const float *W, const float *bias, int Nout, int relu, float *y) {
    for (int o = 0; o < Nout; ++o) {
        float acc = bias ? bias[o] : 0.0f;
        for (int k = 0; k < K; ++k) {
            if (x[k] == 0.0f) continue;
            acc += x[k] * W[(size_t)o * K + k];
        }
        y[o] = (relu && !(acc > 0.0f)) ? 0.0f : acc;
    }
}

/* NearestNeighborMLP.forward (lstm/non_gridbased_pooling.py:98-147) on the padded [B,N,2] tensors.
 * Model fields: n = neighbours kept, C = input_dim (4, or 2 with no_vel), Wp[0] [P/n, C], bp[0], P = out_dim.
 * For every ego: distances to the other slots (NaN -> 1000, :132-133), the n nearest in ascending distance
 * (torch.topk of the negated distance, :136-139; ties between absent neighbours are irrelevant because their
 * attributes become 0 after nan_to_num, :142), attributes [rel pos | rel vel] with NaN -> 0, zero rows when the
 * scene has fewer than n other slots (:134-137), Linear(C -> P/n) + ReLU per neighbour, concatenated (:145-147). */
static void pool_nn_forward(const orc_model *md, const float *obs1, const float *obs2, int B, int N, float *out) {
    const int n = md->n, C = md->C, d = md->P / md->n;
    float *dist = (float *)malloc(sizeof(float) * (size_t)(N > 1 ? N - 1 : 1));
    int *order = (int *)malloc(sizeof(int) * (size_t)(N > 1 ? N - 1 : 1));
    for (int b = 0; b < B; ++b) {
        const float *p1 = obs1 + (size_t)b * N * 2, *p2 = obs2 + (size_t)b * N * 2;
        for (int i = 0; i < N; ++i) {
            int cnt = 0;
            for (int j = 0; j < N; ++j) {
                if (j == i) continue;
                float dx = p2[2 * j] - p2[2 * i], dy = p2[2 * j + 1] - p2[2 * i + 1];
                float dd = sqrtf(dx * dx + dy * dy);               /* torch.norm, :132 */
                dist[cnt] = (dd != dd) ? 1000.0f : dd;             /* :133 */
                order[cnt] = j;
                ++cnt;
            }
            /* stable selection of the n smallest distances */
            for (int k = 0; k < cnt && k < n; ++k) {
                int best = k;
                for (int q = k + 1; q < cnt; ++q) if (dist[q] < dist[best]) best = q;
                float td = dist[best]; int tj = order[best];
                for (int q = best; q > k; --q) { dist[q] = dist[q - 1]; order[q] = order[q - 1]; }
                dist[k] = td; order[k] = tj;
            }
            float *o = out + ((size_t)b * N + i) * md->P;
            for (int k = 0; k < n; ++k) {
                float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                if (k < cnt) {
                    int j = order[k];
                    a[0] = nan_to_num_f(p2[2 * j] - p2[2 * i]);
                    a[1] = nan_to_num_f(p2[2 * j + 1] - p2[2 * i + 1]);
                    if (C == 4) { /* rel_directional, :25-38 */
                        a[2] = nan_to_num_f((p2[2 * j] - p1[2 * j]) - (p2[2 * i] - p1[2 * i]));
                        a[3] = nan_to_num_f((p2[2 * j + 1] - p1[2 * j + 1]) - (p2[2 * i + 1] - p1[2 * i + 1]));
                    }
                }
                small_linear(a, C, md->Wp[0], md->bp[0], d, 1, o + (size_t)k * d);
            }
        }
    }
    free(dist); free(order);
}

/* HiddenStateMLPPooling.forward (lstm/non_gridbased_pooling.py:196-239).  Model fields: dims[0] = mlp_dim_spatial,
 * dims[1] = mlp_dim_vel, dims[2] = mlp_dim_hidden (= C); Wp[0]/bp[0] spatial_embedding [ms,2], Wp[1]/bp[1]
 * vel_embedding [mv,2], Wh/bh hidden_embedding [mh,H], Wp[2]/bp[2] out_projection [P, ms+mh+mv].
 * Per (ego i, slot j) INCLUDING j == i: ReLU(Linear) of the relative position (fill -100 where NaN, :53-61),
 * of slot j's hidden state (fill -100 where NaN) and of 4x the relative velocity; max over j (:238); projection. */
static void pool_hiddenmlp_forward(const orc_model *md, const float *hidden, const float *obs1, const float *obs2,
                                   int B, int N, float *out) {
    const int ms = md->dims[0], mv = md->dims[1], mh = md->dims[2], H = md->H;
    const int D = ms + mh + mv;
    float *hemb = (float *)malloc(sizeof(float) * (size_t)(N * (mh > 0 ? mh : 1)));
    float *pooled = (float *)malloc(sizeof(float) * (size_t)D);
    float *e = (float *)malloc(sizeof(float) * (size_t)D);
    for (int b = 0; b < B; ++b) {
        const float *p1 = obs1 + (size_t)b * N * 2, *p2 = obs2 + (size_t)b * N * 2;
        for (int j = 0; j < N && mh > 0; ++j) {
            const float *hj = hidden + ((size_t)b * N + j) * H;
            int nan = 0;
            for (int k = 0; k < H; ++k) nan |= (hj[k] != hj[k]);
            if (nan) for (int k = 0; k < mh; ++k) hemb[(size_t)j * mh + k] = -100.0f;
            else small_linear(hj, H, md->Wh, md->bh, mh, 1, hemb + (size_t)j * mh);
        }
        for (int i = 0; i < N; ++i) {
            for (int k = 0; k < D; ++k) pooled[k] = -INFINITY;
            for (int j = 0; j < N; ++j) {
                float r[2] = { p2[2 * j] - p2[2 * i], p2[2 * j + 1] - p2[2 * i + 1] };          /* rel_obs, :13-22 */
                if (r[0] != r[0] || r[1] != r[1]) for (int k = 0; k < ms; ++k) e[k] = -100.0f;
                else small_linear(r, 2, md->Wp[0], md->bp[0], ms, 1, e);
                for (int k = 0; k < mh; ++k) e[ms + k] = hemb[(size_t)j * mh + k];              /* :222-227, order :227,234 */
                if (mv > 0) {
                    float v[2] = { ((p2[2 * j] - p1[2 * j]) - (p2[2 * i] - p1[2 * i])) * 4.0f,
                                   ((p2[2 * j + 1] - p1[2 * j + 1]) - (p2[2 * i + 1] - p1[2 * i + 1])) * 4.0f };
                    if (v[0] != v[0] || v[1] != v[1]) for (int k = 0; k < mv; ++k) e[ms + mh + k] = -100.0f;
                    else small_linear(v, 2, md->Wp[1], md->bp[1], mv, 1, e + ms + mh);
                }
                for (int k = 0; k < D; ++k) if (e[k] > pooled[k]) pooled[k] = e[k];             /* torch.max, :238 */
            }
            small_linear(pooled, D, md->Wp[2], md->bp[2], md->P, 0, out + ((size_t)b * N + i) * md->P);
        }
    }
    free(hemb); free(pooled); free(e);
}

/* AttentionMLPPooling.forward (lstm/non_gridbased_pooling.py:297-351).  Model fields as for HIDDENMLP (dims[0..2],
 * Wp[0]/bp[0] spatial, Wp[1]/bp[1] vel, Wh/bh hidden, Wp[2]/bp[2] out_projection), `constant` = fill_value (-10),
 * plus att_*.  Per ego i the sequence is ALL N slots of the padded scene (padded / absent slots included, there is
 * no key padding mask): embedding of slot j relative to i (fill_value where NaN, 0 for a NaN hidden state, :319-335),
 * wq / wk / wv, single-head torch.nn.MultiheadAttention (in_proj, softmax(q k^T / sqrt(D)) v, out_proj), of which only
 * the output at sequence position j == i is kept (:347-350), then out_projection. */
static void pool_attnmlp_forward(const orc_model *md, const float *hidden, const float *obs1, const float *obs2,
                                 int B, int N, float *out) {
    const int ms = md->dims[0], mv = md->dims[1], mh = md->dims[2], H = md->H;
    const int D = ms + mh + mv;
    const float fill = md->constant;
    const float scale = 1.0f / sqrtf((float)D);
#pragma omp parallel for schedule(dynamic, 1)
    for (int bi = 0; bi < B * N; ++bi) {
        const int b = bi / N, i = bi - b * N;
        const float *p1 = obs1 + (size_t)b * N * 2, *p2 = obs2 + (size_t)b * N * 2;
        float *emb = (float *)malloc(sizeof(float) * (size_t)N * D);
        float *kk = (float *)malloc(sizeof(float) * (size_t)N * D);
        float *vv = (float *)malloc(sizeof(float) * (size_t)N * D);
        float *t = (float *)malloc(sizeof(float) * (size_t)D);
        float *q = (float *)malloc(sizeof(float) * (size_t)D);
        float *sc = (float *)malloc(sizeof(float) * (size_t)N);
        float *ao = (float *)malloc(sizeof(float) * (size_t)D);
        for (int j = 0; j < N; ++j) {
            float *e = emb + (size_t)j * D;
            float r[2] = { p2[2 * j] - p2[2 * i], p2[2 * j + 1] - p2[2 * i + 1] };
            if (r[0] != r[0] || r[1] != r[1]) for (int k = 0; k < ms; ++k) e[k] = fill;
            else small_linear(r, 2, md->Wp[0], md->bp[0], ms, 1, e);
            if (mh > 0) {
                const float *hj = hidden + ((size_t)b * N + j) * H;
                int nan = 0;
                for (int k = 0; k < H; ++k) nan |= (hj[k] != hj[k]);
                if (nan) for (int k = 0; k < mh; ++k) e[ms + k] = 0.0f;                       /* fill_value=0, :327 */
                else small_linear(hj, H, md->Wh, md->bh, mh, 1, e + ms);
            }
            if (mv > 0) {
                float v[2] = { ((p2[2 * j] - p1[2 * j]) - (p2[2 * i] - p1[2 * i])) * 4.0f,
                               ((p2[2 * j + 1] - p1[2 * j + 1]) - (p2[2 * i + 1] - p1[2 * i + 1])) * 4.0f };
                if (v[0] != v[0] || v[1] != v[1]) for (int k = 0; k < mv; ++k) e[ms + mh + k] = fill;
                else small_linear(v, 2, md->Wp[1], md->bp[1], mv, 1, e + ms + mh);
            }
            small_linear(e, D, md->att_wk, NULL, D, 0, t);                                      /* key = wk(emb), :341 */
            small_linear(t, D, md->att_in_w + (size_t)D * D, md->att_in_b + D, D, 0, kk + (size_t)j * D);
            small_linear(e, D, md->att_wv, NULL, D, 0, t);                                      /* value, :342 */
            small_linear(t, D, md->att_in_w + (size_t)2 * D * D, md->att_in_b + 2 * D, D, 0, vv + (size_t)j * D);
        }
        small_linear(emb + (size_t)i * D, D, md->att_wq, NULL, D, 0, t);                        /* query at position i */
        small_linear(t, D, md->att_in_w, md->att_in_b, D, 0, q);
        float mx = -INFINITY;
        for (int j = 0; j < N; ++j) {
            float a = 0.0f;
            for (int k = 0; k < D; ++k) a += (q[k] * scale) * kk[(size_t)j * D + k];            /* q scaled, then q k^T */
            sc[j] = a;
            if (a > mx) mx = a;
        }
        float den = 0.0f;
        for (int j = 0; j < N; ++j) { sc[j] = expf(sc[j] - mx); den += sc[j]; }
        for (int k = 0; k < D; ++k) ao[k] = 0.0f;
        for (int j = 0; j < N; ++j) {
            const float a = sc[j] / den;
            for (int k = 0; k < D; ++k) ao[k] += a * vv[(size_t)j * D + k];
        }
        small_linear(ao, D, md->att_out_w, md->att_out_b, D, 0, t);                             /* out_proj */
        small_linear(t, D, md->Wp[2], md->bp[2], md->P, 0, out + (size_t)bi * md->P);           /* out_projection, :351 */
        free(emb); free(kk); free(vv); free(t); free(q); free(sc); free(ao);
    }
}

/* State of the interaction-encoder LSTM of NearestNeighborLSTM / TrajectronPooling: one (h, c) pair per PADDED slot,
 * zeroed by pool.reset() at the start of every LSTM.forward (lstm/lstm.py:213-216, non_gridbased_pooling.py:385-389). */
static float *g_pool_h = NULL, *g_pool_c = NULL;
static size_t g_pool_rows = 0;
static void pool_state_reset(void) { free(g_pool_h); free(g_pool_c); g_pool_h = g_pool_c = NULL; g_pool_rows = 0; }
static void pool_state_need(size_t rows, int Hp) {
    if (g_pool_h && g_pool_rows == rows) return;
    pool_state_reset();
    g_pool_h = (float *)calloc(rows * Hp, sizeof(float));
    g_pool_c = (float *)calloc(rows * Hp, sizeof(float));
    g_pool_rows = rows;
}

static void pool_lstm_tail(const orc_model *md, const float *feat, size_t rows, float *out);

/* NearestNeighborLSTM.forward (:391-455): the NearestNeighborMLP features (always with velocities), then the
 * interaction-encoder LSTMCell over ALL padded slots (no presence mask, :450) and hidden2pool. */
static void pool_nnlstm_forward(const orc_model *md, const float *obs1, const float *obs2, int B, int N, float *out) {
    size_t rows = (size_t)B * N;
    float *feat = (float *)malloc(sizeof(float) * rows * md->P);
    pool_nn_forward(md, obs1, obs2, B, N, feat);
    pool_lstm_tail(md, feat, rows, out);
    free(feat);
}

/* TrajectronPooling.forward (:499-538): per visible slot [pos, vel] of the slot ++ the SUM of [pos, vel] over all
 * other visible slots OF THE WHOLE BATCH (one_cold over states_vis, :525-527 -- not per scene), Linear(8 -> P) + ReLU;
 * invisible slots get a zero row (:521,529); then the interaction-encoder LSTMCell over all slots and hidden2pool. */
static void pool_traj_forward(const orc_model *md, const float *obs1, const float *obs2, int B, int N, float *out) {
    size_t rows = (size_t)B * N;
    float *st = (float *)malloc(sizeof(float) * rows * 4);
    uint8_t *vis = (uint8_t *)malloc(rows);
    for (size_t r = 0; r < rows; ++r) {
        st[4 * r] = obs2[2 * r]; st[4 * r + 1] = obs2[2 * r + 1];
        st[4 * r + 2] = obs2[2 * r] - obs1[2 * r]; st[4 * r + 3] = obs2[2 * r + 1] - obs1[2 * r + 1];
        int nan = 0;
        for (int k = 0; k < 4; ++k) nan |= (st[4 * r + k] != st[4 * r + k]);
        vis[r] = !nan;
    }
    float *feat = (float *)calloc(rows * md->P, sizeof(float));
    for (size_t i = 0; i < rows; ++i) {
        if (!vis[i]) continue;
        float x[8] = { st[4 * i], st[4 * i + 1], st[4 * i + 2], st[4 * i + 3], 0.0f, 0.0f, 0.0f, 0.0f };
        for (size_t k = 0; k < rows; ++k) {
            if (k == i || !vis[k]) continue;
            for (int q = 0; q < 4; ++q) x[4 + q] += st[4 * k + q];
        }
        small_linear(x, 8, md->Wp[0], md->bp[0], md->P, 1, feat + i * md->P);
    }
    pool_lstm_tail(md, feat, rows, out);
    free(st); free(vis); free(feat);
}

/* GridBasedPooling.forward (lstm/gridbased_pooling.py:94-110) on the padded
 * [B,N,*] tensors produced by generate_pooling_inputs (lstm/lstm.py:25-42).
 * need[B*N] marks rows whose embedding is consumed (lstm/lstm.py:146); the
 * MLP is row-wise, so skipping the others changes nothing. out [B*N,P]. */
static void pool_forward(const orc_model *md, const float *hidden, const float *obs1, const float *obs2,
                         int B, int N, const uint8_t *need, float *out, float *grid_dbg) {
    if (md->pool_type == ORC_NN) { pool_nn_forward(md, obs1, obs2, B, N, out); return; }
    if (md->pool_type == ORC_HIDDENMLP) { pool_hiddenmlp_forward(md, hidden, obs1, obs2, B, N, out); return; }
    if (md->pool_type == ORC_ATTNMLP) { pool_attnmlp_forward(md, hidden, obs1, obs2, B, N, out); return; }
    if (md->pool_type == ORC_NNLSTM) { pool_nnlstm_forward(md, obs1, obs2, B, N, out); return; }
    if (md->pool_type == ORC_TRAJ) { pool_traj_forward(md, obs1, obs2, B, N, out); return; }
    const int Fin = md->C * md->n * md->n;
    size_t rows = (size_t)B * N;
    float *enc = NULL;
    if (md->pool_type == ORC_SOCIAL) {
        float *hz = (float *)malloc(sizeof(float) * rows * md->H);
        for (size_t q = 0; q < rows * md->H; ++q) hz[q] = nan_to_num_f(hidden[q]); /* :166 */
        enc = (float *)malloc(sizeof(float) * rows * md->C);
        orc_linear(hz, (int)rows, md->H, md->Wh, md->bh, md->C, 0, enc); /* :167 */
        free(hz);
    }
    float *grid = (float *)malloc(sizeof(float) * rows * Fin);
    orc_grid(md->pool_type, obs1, obs2, enc, B, N, md->C, md->n, md->pool_size, md->blur_size,
             md->cell_side, md->constant, md->front, grid);
    if (grid_dbg) memcpy(grid_dbg, grid, sizeof(float) * rows * Fin);
    /* compact needed rows, run the MLP (:308-335), scatter back */
    int cnt = 0;
    for (size_t r = 0; r < rows; ++r) cnt += need[r] ? 1 : 0;
    float *a = (float *)malloc(sizeof(float) * (size_t)(cnt > 0 ? cnt : 1) * Fin);
    int q = 0;
    for (size_t r = 0; r < rows; ++r)
        if (need[r]) memcpy(a + (size_t)(q++) * Fin, grid + r * Fin, sizeof(float) * Fin);
    int din = Fin;
    for (int l = 0; l < md->n_layers; ++l) {
        int dout = md->dims[l + 1];
        float *y = (float *)malloc(sizeof(float) * (size_t)(cnt > 0 ? cnt : 1) * dout);
        linear_any(a, cnt, din, md->Wp[l], md->WpT[l], md->bp[l], dout, 1, y);
        free(a); a = y; din = dout;
    }
    q = 0;
    for (size_t r = 0; r < rows; ++r) {
        if (need[r]) memcpy(out + r * md->P, a + (size_t)(q++) * md->P, sizeof(float) * md->P);
        else for (int p = 0; p < md->P; ++p) out[r * md->P + p] = NAN;
    }
    free(a); free(grid); free(enc);
}

/* pool_lstm + hidden2pool of the stateful interaction encoders (non_gridbased_pooling.py:450-455, 531-538) */
static void pool_lstm_tail(const orc_model *md, const float *feat, size_t rows, float *out) {
    const int Hp = md->Hp;
    pool_state_need(rows, Hp);
    float *ho = (float *)malloc(sizeof(float) * rows * Hp);
    float *co = (float *)malloc(sizeof(float) * rows * Hp);
    lstm_cell_any(feat, (int)rows, md->P, g_pool_h, g_pool_c, Hp, md->pl_Wih, md->pl_Whh, NULL, NULL, md->pl_bih, md->pl_bhh,
                  ho, co);
    memcpy(g_pool_h, ho, sizeof(float) * rows * Hp);
    memcpy(g_pool_c, co, sizeof(float) * rows * Hp);
    orc_linear(ho, (int)rows, Hp, md->pl_Wo, md->pl_bo, md->P, 0, out);
    free(ho); free(co);
}

/* LSTM.step, lstm/lstm.py:91-168, on dense state h,c [M,H] (the reference keeps
 * per-track lists, :207-210; rows of absent tracks are left untouched = frozen).
 * normal [M,5] : NaN rows for absent tracks (:158). */
static void lstm_step(const orc_model *md, int decoder, float *h, float *c, const float *obs1,
                      const float *obs2, const float *goals, const int64_t *split, int B, int M,
                      float *normal, float *grid_dbg) {
    const int H = md->H, E = md->E;
    const int GD = md->goal_flag ? md->goal_dim : 0;
    const int P = (md->pool_type != ORC_NOPOOL) ? md->P : 0;
    const int I = E + GD + (md->pool_to_hidden ? 0 : P);
    uint8_t *mask = (uint8_t *)malloc(M);
    int cnt = 0;
    for (int m = 0; m < M; ++m) { /* :118 */
        float a = obs1[2 * m], b2 = obs2[2 * m];
        mask[m] = (a == a) && (b2 == b2);
        cnt += mask[m];
    }
    float *pooled = NULL; /* [B*N,P] */
    int N = 0;
    if (md->pool_type != ORC_NOPOOL) {
        /* generate_pooling_inputs, lstm/lstm.py:25-42 */
        for (int s = 0; s < B; ++s) { int ns = (int)(split[s + 1] - split[s]); if (ns > N) N = ns; }
        size_t rows = (size_t)B * N;
        float *cur = (float *)malloc(sizeof(float) * rows * 2);
        float *prev = (float *)malloc(sizeof(float) * rows * 2);
        float *hid = (float *)malloc(sizeof(float) * rows * H);
        uint8_t *need = (uint8_t *)calloc(rows, 1);
        for (size_t q = 0; q < rows * 2; ++q) { cur[q] = NAN; prev[q] = NAN; }
        for (size_t q = 0; q < rows * H; ++q) hid[q] = NAN;
        for (int s = 0; s < B; ++s) {
            int ns = (int)(split[s + 1] - split[s]);
            for (int k = 0; k < ns; ++k) {
                size_t r = (size_t)s * N + k; int m = (int)split[s] + k;
                cur[2 * r] = obs2[2 * m]; cur[2 * r + 1] = obs2[2 * m + 1];
                prev[2 * r] = obs1[2 * m]; prev[2 * r + 1] = obs1[2 * m + 1];
                memcpy(hid + r * H, h + (size_t)m * H, sizeof(float) * H); /* previous-step hidden, :26 */
                need[r] = mask[m];
            }
        }
        pooled = (float *)malloc(sizeof(float) * rows * P);
        pool_forward(md, hid, prev, cur, B, N, need, pooled, grid_dbg);
        free(cur); free(prev); free(hid); free(need);
    }
    /* compact present rows (:121-129,146-149) */
    int Mp = cnt > 0 ? cnt : 1;
    float *x = (float *)malloc(sizeof(float) * (size_t)Mp * I);
    float *hc = (float *)malloc(sizeof(float) * (size_t)Mp * H);
    float *cc = (float *)malloc(sizeof(float) * (size_t)Mp * H);
    float *ho = (float *)malloc(sizeof(float) * (size_t)Mp * H);
    float *co = (float *)malloc(sizeof(float) * (size_t)Mp * H);
    float *nm = (float *)malloc(sizeof(float) * (size_t)Mp * 5);
    float *emb = (float *)malloc(sizeof(float) * (size_t)(E > GD ? E : GD));
    int q = 0, s = 0;
    for (int m = 0; m < M; ++m) {
        while (s + 1 < B && m >= split[s + 1]) ++s;
        if (!mask[m]) continue;
        float vel[2] = { obs2[2 * m] - obs1[2 * m], obs2[2 * m + 1] - obs1[2 * m + 1] }; /* :127 */
        orc_input_embedding(vel, 1, md->We, md->be, E, 4.0f, emb);                          /* :129 */
        memcpy(x + (size_t)q * I, emb, sizeof(float) * E);
        if (md->goal_flag) { /* :132-139 */
            float dx = obs2[2 * m] - goals[2 * m], dy = obs2[2 * m + 1] - goals[2 * m + 1];
            float nf = sqrtf(dx * dx + dy * dy);
            float gd[2] = { dx / nf, dy / nf };
            if (nf == 0.0f) { gd[0] = 0.0f; gd[1] = 0.0f; }
            orc_input_embedding(gd, 1, md->Wg, md->bg, md->goal_dim, 4.0f, emb);
            memcpy(x + (size_t)q * I + E, emb, sizeof(float) * GD);
        }
        memcpy(hc + (size_t)q * H, h + (size_t)m * H, sizeof(float) * H);
        if (pooled) {
            size_t r = (size_t)s * N + (m - (int)split[s]);
            if (!md->pool_to_hidden) memcpy(x + (size_t)q * I + E + GD, pooled + r * P, sizeof(float) * P); /* :146,149 */
            else for (int k = 0; k < H; ++k) hc[(size_t)q * H + k] += pooled[r * P + k];               /* :151, P == H */
        }
        memcpy(cc + (size_t)q * H, c + (size_t)m * H, sizeof(float) * H);
        ++q;
    }
    if (cnt > 0) {
        if (decoder) lstm_cell_any(x, cnt, I, hc, cc, H, md->dec_Wih, md->dec_Whh, md->dec_WihT, md->dec_WhhT, md->dec_bih, md->dec_bhh, ho, co);
        else lstm_cell_any(x, cnt, I, hc, cc, H, md->enc_Wih, md->enc_Whh, md->enc_WihT, md->enc_WhhT, md->enc_bih, md->enc_bhh, ho, co);
        orc_hidden2normal(ho, cnt, H, md->Wn, md->bn, nm); /* :155 */
    }
    q = 0;
    for (int m = 0; m < M; ++m) { /* :158-166 */
        if (!mask[m]) { for (int k = 0; k < 5; ++k) normal[(size_t)m * 5 + k] = NAN; continue; }
        memcpy(h + (size_t)m * H, ho + (size_t)q * H, sizeof(float) * H);
        memcpy(c + (size_t)m * H, co + (size_t)q * H, sizeof(float) * H);
        memcpy(normal + (size_t)m * 5, nm + (size_t)q * 5, sizeof(float) * 5);
        ++q;
    }
    free(mask); free(pooled); free(x); free(hc); free(cc); free(ho); free(co); free(nm); free(emb);
}

/* One public step for step-level parity tests (state in/out, optional grid dump). */
/* Stand-alone pooling module call on padded [B,N,*] tensors (module-level parity tests). */
ORC_API void orc_pool_module(const orc_model *md, const float *hidden, const float *obs1, const float *obs2, int B, int N,
                             float *out) {
    if (md->pool_type == ORC_NN) pool_nn_forward(md, obs1, obs2, B, N, out);
    else if (md->pool_type == ORC_HIDDENMLP) pool_hiddenmlp_forward(md, hidden, obs1, obs2, B, N, out);
    else if (md->pool_type == ORC_ATTNMLP) pool_attnmlp_forward(md, hidden, obs1, obs2, B, N, out);
    else if (md->pool_type == ORC_NNLSTM) { pool_state_reset(); pool_nnlstm_forward(md, obs1, obs2, B, N, out); pool_state_reset(); }
    else if (md->pool_type == ORC_TRAJ) { pool_state_reset(); pool_traj_forward(md, obs1, obs2, B, N, out); pool_state_reset(); }
}

ORC_API void orc_lstm_step(const orc_model *md, int decoder, float *h, float *c, const float *obs1,
                           const float *obs2, const float *goals, const int64_t *split, int B, int M,
                           float *normal, float *grid_dbg) {
    lstm_step(md, decoder, h, c, obs1, obs2, goals, split, B, M, normal, grid_dbg);
}

/* ------------------------------------------------------------------------- *
 * LSTM.forward, lstm/lstm.py:170-264.
 *   observed [T_obs,M,2]; truth [T_dec,M,2] or NULL (n_predict mode, T_dec = n_predict-1)
 *   rel_pred [T_obs-1+T_dec, M, 5]
 *   pred     [npos, M, 2], npos = T_obs-1+T_dec (+1 when T_obs == 2, :222-223)
 *   returns npos
 * ------------------------------------------------------------------------- */
ORC_API int orc_lstm_forward(const orc_model *md, const float *observed, int T_obs, int M,
                             const float *goals, const int64_t *split, int B, const float *truth,
                             int T_dec, float *rel_pred, float *pred) {
    const int H = md->H;
    const size_t F = (size_t)M * 2;
    float *h = (float *)calloc((size_t)M * H, sizeof(float)); /* :207-210 */
    float *c = (float *)calloc((size_t)M * H, sizeof(float));
    pool_state_reset();   /* pool.reset(), lstm/lstm.py:213-216 */
    int npos = 0, nnorm = 0;
    if (T_obs == 2) { memcpy(pred, observed + F, sizeof(float) * F); npos = 1; } /* :222-223 */
    /* encoder :226-232 */
    for (int t = 1; t < T_obs; ++t) {
        const float *o1 = observed + (size_t)(t - 1) * F, *o2 = observed + (size_t)t * F;
        float *nr = rel_pred + (size_t)nnorm * M * 5;
        lstm_step(md, 0, h, c, o1, o2, goals, split, B, M, nr, NULL);
        float *ps = pred + (size_t)npos * F;
        for (int m = 0; m < M; ++m) { ps[2 * m] = o2[2 * m] + nr[5 * m]; ps[2 * m + 1] = o2[2 * m + 1] + nr[5 * m + 1]; }
        ++nnorm; ++npos;
    }
    /* decoder :235-255. pt[0] = copy of observed[-1], pt[k>=1] = copy of truth[k-1] or None */
    float *pt_prev = (float *)malloc(sizeof(float) * F);
    float *pt_cur = (float *)malloc(sizeof(float) * F);
    memcpy(pt_prev, observed + (size_t)(T_obs - 1) * F, sizeof(float) * F);
    int prev_is_none = 0;
    for (int k = 0; k < T_dec; ++k) {
        const float *pos_m2 = pred + (size_t)(npos - 2) * F, *pos_m1 = pred + (size_t)(npos - 1) * F;
        /* obs1 :241-245 */
        if (prev_is_none) memcpy(pt_prev, pos_m2, sizeof(float) * F);
        else for (int s = 0; s < B; ++s) { size_t p = (size_t)split[s]; pt_prev[2 * p] = pos_m2[2 * p]; pt_prev[2 * p + 1] = pos_m2[2 * p + 1]; }
        /* obs2 :246-250 */
        if (truth == NULL) memcpy(pt_cur, pos_m1, sizeof(float) * F);
        else {
            memcpy(pt_cur, truth + (size_t)k * F, sizeof(float) * F);
            for (int s = 0; s < B; ++s) { size_t p = (size_t)split[s]; pt_cur[2 * p] = pos_m1[2 * p]; pt_cur[2 * p + 1] = pos_m1[2 * p + 1]; }
        }
        float *nr = rel_pred + (size_t)nnorm * M * 5;
        lstm_step(md, 1, h, c, pt_prev, pt_cur, goals, split, B, M, nr, NULL);
        float *ps = pred + (size_t)npos * F;
        for (int m = 0; m < M; ++m) { ps[2 * m] = pt_cur[2 * m] + nr[5 * m]; ps[2 * m + 1] = pt_cur[2 * m + 1] + nr[5 * m + 1]; }
        ++nnorm; ++npos;
        /* next iteration's obs1 is this iteration's (already primary-patched) obs2 */
        memcpy(pt_prev, pt_cur, sizeof(float) * F);
        prev_is_none = (truth == NULL);
    }
    free(h); free(c); free(pt_prev); free(pt_cur);
    pool_state_reset();
    return npos;
}

/* classical/constant_velocity.py:4-20 : x_last + t*(x_last - x_prev), t = 1..n_predict (float64) */
ORC_API void orc_constant_velocity(const double *xy, int T, int N, int n_predict, double *out) {
    const double *last = xy + (size_t)(T - 1) * N * 2, *prev = xy + (size_t)(T - 2) * N * 2;
    for (int t = 1; t <= n_predict; ++t)
        for (int q = 0; q < N * 2; ++q) {
            double v = last[q] - prev[q];
            out[(size_t)(t - 1) * N * 2 + q] = last[q] + (double)t * v;
        }
}

ORC_API int orc_abi_version(void) { return 1; }

/* OpenMP thread count of the following calls (bench.py's cpu_baseline times the port at several counts and reports
 * the best one with the count it was measured at); returns the previous maximum.  No-op without OpenMP. */
#ifdef _OPENMP
#include <omp.h>
ORC_API int orc_set_threads(int n) { int prev = omp_get_max_threads(); if (n > 0) omp_set_num_threads(n); return prev; }
#else
ORC_API int orc_set_threads(int n) { (void)n; return 1; }
#endif

/* ------------------------------------------------------------------------- *
 * Losses, lstm/loss.py.  inputs [T,M,5], targets [T,M,2]; primaries = rows split[s].
 * ------------------------------------------------------------------------- */
static float gaussian_2d_f(const float *p, float s1, float s2, float rho, const float *x) {
    /* lstm/loss.py:23-50 */
    float norm1 = x[0] - p[0], norm2 = x[1] - p[1];
    float s1s2 = s1 * s2;
    float q1 = norm1 / s1, q2 = norm2 / s2;
    float z = q1 * q1 + q2 * q2 - 2.0f * rho * norm1 * norm2 / s1s2;
    float omr = 1.0f - rho * rho;
    float num = expf(-z / (2.0f * omr));
    float den = 6.283185307179586f * s1s2 * sqrtf(omr);
    return num / den;
}

/* mode 0: PredictionLoss.forward (:52-91); mode 1: L2Loss.forward (:107-135).
 * keep_batch_dim: out [B] (mean over time; L2 also over the 2 coordinates), else out [1] (mean over everything).
 * The result is multiplied by `multiplier` (1 / 100). */
ORC_API void orc_primary_loss(int mode, const float *inputs, const float *targets, const int64_t *split, int B, int T,
                              int M, float bg, int keep_batch_dim, float multiplier, float *out) {
    double tot = 0.0;
    for (int s = 0; s < B; ++s) {
        double acc = 0.0;
        for (int t = 0; t < T; ++t) {
            const float *in = inputs + ((size_t)t * M + split[s]) * 5;
            const float *tg = targets + ((size_t)t * M + split[s]) * 2;
            float v;
            if (mode == 0) {
                float g_bg = gaussian_2d_f(in, 3.0f, 3.0f, 0.0f, tg);          /* :73-76 */
                float g = gaussian_2d_f(in, in[2], in[3], in[4], tg);
                v = -logf(0.01f + bg * g_bg + (0.99f - bg) * g);              /* :78-82 */
            } else {
                float d0 = in[0] - tg[0], d1 = in[1] - tg[1];
                v = 0.5f * (d0 * d0 + d1 * d1);                              /* mean over the 2 coordinates */
            }
            acc += v;
        }
        if (keep_batch_dim) out[s] = (float)(acc / T) * multiplier;
        tot += acc;
    }
    if (!keep_batch_dim) out[0] = (float)(tot / ((double)T * B)) * multiplier;
}

/* CollisionLoss, lstm/loss.py:138-162 (predictions [T,M,ld], first two columns) */
ORC_API float orc_collision_loss(const float *pred, int ld, const int64_t *split, int B, int T, int M, float col_wt,
                                 float col_distance) {
    double loss = 0.0;
    for (int s = 0; s < B; ++s) {
        int lo = (int)split[s], hi = (int)split[s + 1];
        double a = 0.0;
        for (int t = 0; t < T; ++t)
            for (int j = lo + 1; j < hi; ++j) {
                const float *pp = pred + ((size_t)t * M + lo) * ld, *pn = pred + ((size_t)t * M + j) * ld;
                float px = pp[0] != pp[0] ? -1000.0f : pp[0], py = pp[1] != pp[1] ? -1000.0f : pp[1];
                float nx = pn[0] != pn[0] ? -1000.0f : pn[0], ny = pn[1] != pn[1] ? -1000.0f : pn[1];
                float dx = px - nx, dy = py - ny;
                float d = sqrtf(dx * dx + dy * dy);
                if (d <= col_distance) a += 1.0f - d / col_distance;
            }
        loss += col_wt * a;
    }
    return (float)loss;
}
