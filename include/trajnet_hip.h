/*
 * trajnet_hip.h -- C ABI of libtrajnet_hip.so, the MI355X (gfx950) implementation of the
 * TrajNet++ baselines hot path: the per-timestep recurrent step of trajnetbaselines.lstm,
 * its grid-based interaction pooling, and the classical rollouts.
 *
 * The reference (vita-epfl/trajnetplusplusbaselines) is pure Python/PyTorch and has no FFI;
 * each entry point below names the reference interface it replaces (file:line relative to
 * the reference's trajnetbaselines/ package).  INTEGRATION.md shows the ctypes binding a
 * reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless the name ends in _host;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only enqueue
 *     work on it -- no host synchronisation, no allocation -- so they are graph-capturable;
 *   - all floating point is IEEE fp32, cell indices are exact (IEEE division, truncation);
 *   - return value 0 = success, <0 = error; tnp_last_error() gives the message
 *     (the Python host layer raises RuntimeError / ValueError with it);
 *   - weights use the PyTorch layout [out_features, in_features], LSTM gate order i,f,g,o,
 *     i.e. the tensors of the reference's state_dict are passed in place.
 */
#ifndef TRAJNET_HIP_H
#define TRAJNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TNP_ABI_VERSION 7
#define TNP_API __attribute__((visibility("default")))

/* pooling types: GridBasedPooling(type_=...)  lstm/gridbased_pooling.py:16-19,55-66 */
#define TNP_POOL_NONE        (-1)
#define TNP_POOL_OCCUPANCY   0
#define TNP_POOL_DIRECTIONAL 1
#define TNP_POOL_SOCIAL      2
#define TNP_POOL_NN          4   /* NearestNeighborMLP, lstm/non_gridbased_pooling.py:64-147   */
#define TNP_POOL_HIDDENMLP   5   /* HiddenStateMLPPooling, lstm/non_gridbased_pooling.py:150-239 */
#define TNP_POOL_ATTNMLP     6   /* AttentionMLPPooling, lstm/non_gridbased_pooling.py:242-351   */
#define TNP_POOL_NNLSTM      7   /* NearestNeighborLSTM, lstm/non_gridbased_pooling.py:354-455   */
#define TNP_POOL_TRAJ        8   /* TrajectronPooling, lstm/non_gridbased_pooling.py:457-538      */

TNP_API int tnp_abi_version(void);
TNP_API const char *tnp_last_error(void);

/* -------------------------------------------------------------------------------------------
 * Scene index.  The reference describes a batch by `batch_split` (int64 [B+1], lstm/lstm.py:
 * 170-187); the kernels use int32 scene starts plus a per-track primary flag.
 *   scene_start [B+1] int32 (device), primary_flag [M] uint8 (device, written)
 * ----------------------------------------------------------------------------------------- */
TNP_API int tnp_mark_primaries(const int32_t *scene_start, int B, int M, uint8_t *primary_flag, void *stream);

/* -------------------------------------------------------------------------------------------
 * Grid build: GridBasedPooling.occupancy / occupancies / directional / social
 * (lstm/gridbased_pooling.py:112-170, 227-305), pool_size = blur_size = 1.
 *   obs1, obs2   [rows,2]   previous / current positions, NaN = absent track
 *   values       [rows,ldv] social: hidden_dim_encoding(hidden) per track (C columns used)
 *   scene_start  [B+1]      rows of scene s are [scene_start[s], scene_start[s+1])
 *   n_max        width of the reference's padded [B, n_max, .] pooling tensors (lstm/lstm.py:29): the largest
 *                entry of scene_slots, or -- with scene_slots == NULL -- the number of slots EVERY scene is
 *                padded to (at least the largest scene; a sharded batch passes the global maximum)
 *   scene_slots  [B] int32 or NULL: slots the reference pads scene s to (>= its track count).  The padded
 *                slots are absent neighbours with the highest index, so a scene with scene_slots[s] > its
 *                track count gets the reference's cell-0 clobber and a scene with
 *                scene_slots[s] == its track count does not -- the latter is what one call per scene
 *                (lstm/lstm.py:291, the evaluator) computes, the former what a ragged training batch does
 *   cell         float32(cell_side / pool_size); half_x/half_y = n/2 (front: half_y = 0)
 *   grid         [rows,ldg] out, feature = c*n*n + cell_x*n + cell_y   (may be NULL)
 *   winners      [rows,n*n] out, int16: scene-local index of the neighbour that owns the
 *                cell, -1 = background `constant`                       (may be NULL)
 * ----------------------------------------------------------------------------------------- */
TNP_API int tnp_pool_grid_forward(int type, const float *obs1, const float *obs2, const float *values, int ldv,
                          const int32_t *scene_start, int B, int n_max, const int32_t *scene_slots, int n, int C,
                          float cell, float half_x, float half_y, float constant, float *grid, int ldg,
                          int16_t *winners, void *stream);

/* Cells of all ordered (ego, neighbour) pairs, for the backward pass of the grid scatter (autograd of
 * `occ[arange, oi] = other_values`, lstm/gridbased_pooling.py:293: every in-range neighbour receives the gradient of
 * its cell).  row_base / row_count [M]: first row and size of the row's scene; out [M, n_max] int32, -1 = no cell. */
TNP_API int tnp_pool_pair_cells(const float *obs2, const int32_t *row_base, const int32_t *row_count, int M, int n_max,
                        int n, float cell, float half_x, float half_y, int32_t *out, void *stream);
/* The same table as autograd sees it (what the training path uses).  The grid leaves occupancy() through
 * lp_pool2d(x, 1, 1) = sign(x) * relu(|x|) (lstm/gridbased_pooling.py:304), whose derivative at x == 0 is zero: a cell whose
 * final value is exactly 0 passes no gradient -- cell (0, 0) whenever its last writer is an out-of-range / absent / padded
 * neighbour and constant == 0 (they write `constant` there, :281-282), so the genuine neighbours in that cell get nothing.
 * cells [M, n_max]: the raw cell or -1 (no gradient); winner [M, n_max] (may be NULL): slot of the last writer of the
 * pair's cell (whose value is the cell's value; -2: the cell holds a non-zero `constant`, -1 with cells == -1).
 * row_padded [M] (NULL: pad_default for every row) = slots the row's scene is padded to (lstm/lstm.py:29);
 * raw_scratch [M, n_max] int32, a buffer different from `cells`. */
TNP_API int tnp_pool_pair_cells_autograd(const float *obs2, const int32_t *row_base, const int32_t *row_count,
                                         const int32_t *row_padded, int pad_default, int M, int n_max, int n, float cell,
                                         float half_x, float half_y, float constant, int32_t *raw_scratch, int32_t *cells,
                                         int32_t *winner, void *stream);

/* -------------------------------------------------------------------------------------------
 * Dense layer on the matrix cores: torch.nn.Linear (+ReLU) as used by the grid embedding
 * MLPs (lstm/gridbased_pooling.py:308-335).   C[M,N] = act(A[M,K] @ W[N,K]^T + bias)
 * fp32-in / fp32-accumulate MFMA (v_mfma_f32_32x32x2_f32): exact fp32 products.
 *   variant: 0 = automatic tile selection; 12 / 24 / 25 / 26 pin one of the fast kernels' tile shapes
 * ----------------------------------------------------------------------------------------- */
TNP_API int tnp_linear_forward(const float *A, int lda, const float *W, int ldw, const float *bias, float *C,
                       int ldc, int M, int N, int K, int relu, int variant, void *stream);

/* -------------------------------------------------------------------------------------------
 * Sparse first layer of the grid-embedding MLP on the winner table (social pooling, constant 0):
 *   out[i,o] = act(bias[o] + sum over occupied cells c of sum_ch W_cell_major[c][ch][o] * values[j(i,c)][ch])
 * == torch.nn.Linear on the dense grid (lstm/gridbased_pooling.py:107-109), 8x fewer multiply-adds.
 *   winners   [M, ncell] int16 from tnp_pool_grid_forward; values [M, ldv]; row_base [M] int32 =
 *   first row of the row's scene (tnp_row_base); W_cell_major [ncell][C][N1]; C in {4,8,16,32}
 *   workspace: tnp_pool_embed_sparse_workspace_bytes(M, N1, ncell) bytes (0 for grids of up to 440 cells;
 *   partial sums of the cell-range fallback beyond that)
 * ----------------------------------------------------------------------------------------- */
TNP_API size_t tnp_pool_embed_sparse_workspace_bytes(int M, int N1, int ncell);
TNP_API int tnp_row_base(const int32_t *scene_start, int B, int32_t *row_base, void *stream);
TNP_API int tnp_pool_embed_sparse_forward(const int16_t *winners, const float *values, int ldv,
                                  const int32_t *row_base, const float *W_cell_major, const float *bias,
                                  int M, int ncell, int C, int N1, int relu, float *out, int ldo,
                                  void *workspace, size_t workspace_bytes, void *stream);

/* -------------------------------------------------------------------------------------------
 * Non-grid interaction modules on concatenated tracks (scene s = rows scene_start[s]..scene_start[s+1]).
 *   tnp_pool_nn_forward: NearestNeighborMLP.forward (lstm/non_gridbased_pooling.py:98-147):
 *     out[i, k*d:(k+1)*d] = ReLU(W [d,in_dim] . attr(i, k-th nearest other track) + bias), k < n_sel;
 *     attr = [rel pos | rel vel] (in_dim 4) or rel pos (in_dim 2), NaN -> 0, missing neighbours -> 0.
 *   tnp_pool_hiddenmlp_forward: HiddenStateMLPPooling.forward (:196-239) up to the max-pool:
 *     pooled[i] = max over the slots j of the scene (incl. i) of
 *       [ReLU(W_spatial.(p_j - p_i) + b) | hidden_emb[j] | ReLU(W_vel.4(v_j - v_i) + b)], fill -100 where NaN;
 *     hidden_emb [M, mh] = Linear(H -> mh) of the slots' hidden state, ReLU applied here if hidden_emb_relu
 *     (else the caller passes ReLU'd / -100-masked values); the out_projection is a tnp_linear_forward call.
 * ----------------------------------------------------------------------------------------- */
TNP_API int tnp_pool_nn_forward(const float *obs1, const float *obs2, const int32_t *scene_start, int B,
                                int n_sel, int in_dim, const float *W, const float *bias, int d,
                                float *out, int ldo, void *stream);
TNP_API int tnp_pool_hiddenmlp_forward(const float *obs1, const float *obs2, const float *hidden_emb, int ldh,
                                       int hidden_emb_relu, const int32_t *scene_start, int B, int ms, int mv,
                                       int mh, const float *W_spatial, const float *b_spatial,
                                       const float *W_vel, const float *b_vel, float *pooled, int ldp,
                                       void *stream);
/* TrajectronPooling features (lstm/non_gridbased_pooling.py:513-529): out[i] = ReLU(W [P,8] . [pos_i, vel_i,
 * sum over the other visible tracks of the batch of (pos, vel)] + bias), zero rows for invisible tracks;
 * scratch4 = 4 doubles of device memory */
TNP_API int tnp_pool_traj_forward(const float *obs1, const float *obs2, int M, const float *W, const float *bias,
                                  int P, float *out, int ldo, double *scratch4, void *stream);
/* AttentionMLPPooling.forward (lstm/non_gridbased_pooling.py:297-351) in two kernels around three tnp_linear_forward
 * calls (see tnp_lstm_model.Wx):
 *   tnp_pool_attn_self : e_self[i] = embedding of slot i relative to itself ([ReLU(b_spatial) | hidden_emb[i] |
 *                        ReLU(b_vel)], fill where the ego's position / velocity is NaN)            -> [M, D]
 *   tnp_pool_attn_pair : a_ij = softmax_j((u[i,0:D] . e_ij + u[i,D]) / sqrt(D)) over ALL n_max slots of the padded
 *                        scene (slots beyond the scene's tracks count as padded: fill / 0 / fill; scene_slots as for
 *                        tnp_pool_grid_forward: per-scene slot counts, NULL = n_max), ebar[i] = sum_j a_ij e_ij */
TNP_API int tnp_pool_attn_self(const float *obs1, const float *obs2, const float *hidden_emb, int ldh,
                               int hidden_emb_relu, int M, int ms, int mv, int mh, const float *b_spatial,
                               const float *b_vel, float fill, float *e_self, int lde, void *stream);
TNP_API int tnp_pool_attn_pair(const float *obs1, const float *obs2, const float *hidden_emb, int ldh,
                               int hidden_emb_relu, const int32_t *scene_start, int B, int n_max,
                               const int32_t *scene_slots, int ms, int mv, int mh, const float *W_spatial, const float *b_spatial, const float *W_vel,
                               const float *b_vel, float fill, const float *u, int ldu, float *ebar, int lde,
                               void *stream);

/* -------------------------------------------------------------------------------------------
 * Model descriptor of trajnetbaselines.lstm.LSTM (lstm/lstm.py:45-89) with its
 * GridBasedPooling (lstm/gridbased_pooling.py:15-92).  All pointers are device pointers to
 * the parameters in PyTorch layout (may alias nn.Parameter storage; nothing is copied).
 * ----------------------------------------------------------------------------------------- */
typedef struct tnp_lstm_model {
    int32_t E;            /* embedding_dim (64)                                   */
    int32_t H;            /* hidden_dim (128), multiple of 32                     */
    int32_t goal_flag;    /* lstm/lstm.py:73-76                                    */
    int32_t goal_dim;
    int32_t pool_type;    /* TNP_POOL_*                                           */
    int32_t n;            /* grid cells per side (TNP_POOL_NN: neighbours kept)   */
    int32_t C;            /* pooling_dim: 1 / 2 / latent_dim (NN: input_dim 2|4; HIDDENMLP: mlp_dim_hidden) */
    int32_t P;            /* pool.out_dim                                         */
    int32_t n_layers;     /* embedding MLP depth 1..3 (one_/two_/three_layer)     */
    int32_t dims[4];      /* dims[0] = C*n*n ... dims[n_layers] = P; HIDDENMLP: {mlp_dim_spatial, mlp_dim_vel,
                             mlp_dim_hidden}, Wp/bp[0] spatial, [1] vel, [2] out_projection, Wh/bh hidden_embedding;
                             NN: Wp/bp[0] = embedding Linear(C -> P/n) */
    float cell;           /* float32(cell_side / pool_size)                       */
    float half_x, half_y; /* n/2 ; front=True: half_y = 0                         */
    float constant;       /* background value of the grid                         */
    const float *We, *be; /* input_embedding.input_embeddings.0   [E-2,2], [E-2]  */
    const float *Wg, *bg; /* goal_embedding.input_embeddings.0                    */
    const float *enc_Wih, *enc_Whh, *enc_bih, *enc_bhh; /* encoder LSTMCell       */
    const float *dec_Wih, *dec_Whh, *dec_bih, *dec_bhh; /* decoder LSTMCell       */
    const float *Wn, *bn; /* hidden2normal.linear [5,H], [5]                      */
    const float *Wh, *bh; /* pool.hidden_dim_encoding [C,H], [C] (social)         */
    const float *Wp[3];   /* pool.embedding.{0,2,4}.weight                        */
    const float *bp[3];   /* pool.embedding.{0,2,4}.bias                          */
    const float *Wp0_cell_major; /* optional [n*n][C][dims[1]] copy of Wp[0] (W'[c][ch][o] =
                                    Wp[0][o][ch*n*n + c]); enables the sparse first layer
                                    for social pooling with constant == 0; NULL = dense   */
    int32_t variant;      /* bits 0-7 / 8-15: tile selection of the dense embedding GEMM (12, 24) / the gates GEMM (5, 20, 21) -- 0 = automatic; bit 16: force the dense first
                             embedding layer; bit 17: LSTM(pool_to_input=False) -- the interaction vector (P == H) is
                             added to the hidden operand of the LSTMCell instead of concatenated to its input */
    /* TNP_POOL_ATTNMLP only (fields as for HIDDENMLP, `constant` = fill_value): the linear maps around the single-head
     * attention folded on the host, D = mlp_dim:
     *   Wx[0] [D,D], bx[0] [D]  query  q = (in_proj_q . wq) e_ii + in_proj_bias_q
     *   Wx[1] [D+4,D]           rows 0..D-1 = (in_proj_k . wk)^T, row D = in_proj_bias_k, rows D+1.. = 0:
     *                            u = Wx[1] q gives score_ij = (u[0:D] . e_ij + u[D]) / sqrt(D)
     *   Wx[2] [P,D], bx[2] [P]  out_projection . out_proj . (in_proj_v . wv) applied to sum_j a_ij e_ij (+ biases) */
    /* TNP_POOL_NNLSTM / TNP_POOL_TRAJ (stateful interaction encoders; tnp_lstm_forward only): Wp/bp[0] = embedding
     * Linear(4 -> P/n) resp. Linear(8 -> P), dims[0] = Hp, Wx/bx[0] = pool_lstm.weight_ih/bias_ih [4Hp,P],
     * Wx/bx[1] = pool_lstm.weight_hh/bias_hh [4Hp,Hp], Wx/bx[2] = hidden2pool [P,Hp] */
    const float *Wx[3];
    const float *bx[3];
    const float *Wp0_quad_major; /* optional second copy of Wp[0] next to Wp0_cell_major, laid out for the register-
                                    accumulator sparse kernel: [n*n][dims[1]/64][C/4][64][4],
                                    W''[c][o/64][ch/4][o%64][ch%4] = Wp[0][o][ch*n*n + c]  (needs dims[1] % 64 == 0 and
                                    C % 4 == 0; a wave's C x 64 weights of a cell are one contiguous 4 C x 64-byte block).
                                    NULL = that kernel reads Wp0_cell_major (slower: two scalar address ops per channel) */
    int32_t pool_size;    /* GridBasedPooling(pool_size=, blur_size=) (lstm/gridbased_pooling.py:297-304; ABI 7): the grid is   */
    int32_t blur_size;    /* built with n * pool_size cells per side (`cell`, `half_*` are the FINE grid's), blurred by
                             avg_pool2d(blur_size, stride 1, padding blur_size / 2, count_include_pad) and summed over
                             pool_size x pool_size windows (lp_pool2d, p = 1) down to n x n before the embedding MLP.
                             0 or 1 = off (the trainer's defaults).  Inference only (tnp_lstm_forward / _step); the training
                             entry points refuse a model with either set. */
} tnp_lstm_model;

/* bytes of scratch HBM tnp_lstm_forward / tnp_lstm_step need for M tracks in B scenes */
TNP_API size_t tnp_lstm_workspace_bytes(const tnp_lstm_model *model, int M, int B);

/* -------------------------------------------------------------------------------------------
 * LSTM.forward (lstm/lstm.py:170-264): T_obs-1 encoder steps + T_dec decoder steps.
 *   observed    [T_obs,M,2]
 *   goals       [M,2] (read only when model->goal_flag)
 *   scene_start [B+1] int32, primary_flag [M] uint8 (tnp_mark_primaries), n_max / scene_slots as for
 *               tnp_pool_grid_forward (NULL: every scene padded to n_max, the reference's batch semantics)
 *   truth       [T_dec,M,2] teacher-forcing frames (prediction_truth) or NULL = n_predict
 *               mode with T_dec = n_predict-1
 *   rel_pred    [T_obs-1+T_dec, M, 5] out (mu_x, mu_y, sigma_x, sigma_y, rho), NaN = absent
 *   pred        [npos, M, 2] out, npos = T_obs-1+T_dec (+1 if T_obs == 2, lstm/lstm.py:222)
 *   workspace   tnp_lstm_workspace_bytes() bytes, 256-byte aligned
 * ----------------------------------------------------------------------------------------- */
TNP_API int tnp_lstm_forward(const tnp_lstm_model *model, const float *observed, int T_obs, int M,
                     const float *goals, const int32_t *scene_start, const uint8_t *primary_flag,
                     int B, int n_max, const int32_t *scene_slots, const float *truth, int T_dec, float *rel_pred,
                     float *pred, void *workspace, size_t workspace_bytes, void *stream);

/* -------------------------------------------------------------------------------------------
 * Same sequence with the S-GAN hooks (sgan/sgan.py):
 *   noise interface of LSTMGenerator (adding_noise, :200-221): after the encoder,
 *       h <- [relu(W_ctx h + b_ctx) | noise]  for every track  (W_ctx [H-noise_dim, H], noise [noise_dim]);
 *   h_final [M,H]: hidden state after the last step (LSTMDiscriminator scores the primaries' rows, :564-576;
 *       run with truth = NULL, T_dec = 0 over the concatenated observed + predicted frames).
 * ----------------------------------------------------------------------------------------- */
typedef struct tnp_lstm_extras {
    const float *W_ctx, *b_ctx;   /* mlp_decoder_context.0 */
    const float *noise;           /* [noise_dim], one vector shared by all tracks (or [groups, noise_dim], below) */
    int32_t noise_dim;            /* 0 = no noise interface */
    int32_t noise_group_tracks;   /* 0: one vector; g > 0: tracks [i*g, (i+1)*g) carry vector i -- k generator samples
                                     batched as k replicas of the scenes (sgan/sgan.py:78-100 runs them one by one) */
    float *h_final;               /* optional out [M,H] */
    /* Loss fused into the sequence (PredictionLoss / L2Loss on the primaries, lstm/loss.py:52-135; what
     * Trainer.val_batch evaluates on rel_outputs[-pred_length:], lstm/trainer.py:296-309): while the kernel that finishes a
     * step still holds the step's normal (mu, sigma, rho) in registers, it evaluates the per-element loss of every PRIMARY
     * row against loss_targets and stores it -- the [T, M, 5] normals are not read back.
     *   loss_targets [loss_steps, M, 2]  targets of the LAST loss_steps outputs (NULL = no fused loss)
     *   loss_values  [loss_steps, M]     out: entry [t, scene_start[s]] = loss of scene s at step t (other rows untouched);
     *                                    tnp_primary_loss_reduce turns it into the mean / per-scene means
     *   loss_mode 0 = PredictionLoss (loss_background_rate), 1 = L2 (sum of the two squared errors) */
    const float *loss_targets;
    float *loss_values;
    int32_t loss_steps, loss_mode;
    float loss_background_rate;
    /* VAE (vae/vae.py:89-107, add_noise): after the last encoder step the hidden state is multiplied element by element
     * with h_scale [M,H] (= vae_decoder(z)), the cell state is kept; NULL = no such hook.  Exclusive with the noise interface.
     * (k modes = k replicas of the scenes, each replica's rows carrying its own multiplier.) */
    const float *h_scale;
} tnp_lstm_extras;
TNP_API int tnp_lstm_forward_ex(const tnp_lstm_model *model, const float *observed, int T_obs, int M,
                        const float *goals, const int32_t *scene_start, const uint8_t *primary_flag,
                        int B, int n_max, const int32_t *scene_slots, const float *truth, int T_dec, float *rel_pred,
                        float *pred, void *workspace, size_t workspace_bytes, const tnp_lstm_extras *extras,
                        void *stream);

/* -------------------------------------------------------------------------------------------
 * LSTM.step (lstm/lstm.py:91-168) on dense state: one masked recurrent step.
 *   decoder     0 = encoder cell, 1 = decoder cell
 *   h_in, c_in  [M,H] state before the step;  h_out, c_out [M,H] after (rows of absent
 *               tracks are copied through = frozen); in/out must not alias
 *   normal      [M,5] out, NaN rows for absent tracks
 * ----------------------------------------------------------------------------------------- */
TNP_API int tnp_lstm_step(const tnp_lstm_model *model, int decoder, const float *h_in, const float *c_in,
                  const float *obs1, const float *obs2, const float *goals, const int32_t *scene_start,
                  int B, int M, int n_max, const int32_t *scene_slots, float *h_out, float *c_out, float *normal,
                  void *workspace, size_t workspace_bytes, void *stream);

/* -------------------------------------------------------------------------------------------
 * Training support (loss.backward() through LSTM.forward, lstm/trainer.py:229-269).
 *   tnp_lstm_step_train: tnp_lstm_step that also keeps what the backward pass needs, written straight into the
 *   caller's buffers (any pointer may be NULL): X [M, I] the LSTMCell input (embedding | goal | interaction vector),
 *   act[l] [M, dims[l+1]] the ReLU outputs of the embedding MLP layers before the last, gates [M, 4H] the
 *   post-activation i, f, g, o of the present rows, enc [M, C] the social encoding, nn_attrs the inputs of
 *   NearestNeighborMLP's embedding.
 *   The backward kernels below are the pointwise / gather parts of the reverse sweep; every contraction is a
 *   tnp_linear_forward call on (transposed) operands.
 *   tnp_h2n_backward:      d(normal) -> d(Linear output) of Hidden2Normal (lstm/modules.py:56-64) and
 *                          dh_tot = dh_in + dlin . Wn; rows of absent tracks get dlin = 0
 *   tnp_lstm_cell_backward: torch.nn.LSTMCell backward from the saved gates: dG [M,4H] (pre-activation gradients,
 *                          gate order i,f,g,o), dc_prev, and dh_pass = gradient that bypasses the cell for absent
 *                          tracks (their state is copied through, lstm/lstm.py:158-166)
 *   tnp_relu_mask:         out = dy * (act > 0) on column slices (leading dimensions given)
 *   tnp_scaled_diff:       out = nan_to_num(a - b) * scale over n elements (torch.nan_to_num defaults) -- the input
 *                          embedding's operand 4 (obs2 - obs1) of the weight gradients (lstm/lstm.py:127-131 under autograd)
 *   tnp_social_scatter_backward: d(social encoding)[j] = sum over the egos i of j's scene of dgrid[i, :, cell(i,j)]
 *                          with cells from tnp_pool_pair_cells_autograd (every in-range neighbour of a cell whose value
 *                          is not the constant 0, SURVEY.md 8a quirk 4 + lp_pool2d's zero derivative at 0)
 * ----------------------------------------------------------------------------------------- */
typedef struct tnp_step_saves {
    float *X;
    float *act[2];
    float *gates;
    float *enc;
    float *nn_attrs;   /* TNP_POOL_NN: [M, n, input_dim] gathered neighbour attributes */
    int16_t *winners;  /* sparse first layer (tnp_lstm_sparse_first_layer() == 1): [M, n*n] winner of every cell, -1 = empty */
} tnp_step_saves;
TNP_API int tnp_lstm_step_train(const tnp_lstm_model *model, int decoder, const float *h_in, const float *c_in,
                                const float *obs1, const float *obs2, const float *goals,
                                const int32_t *scene_start, int B, int M, int n_max, const int32_t *scene_slots,
                                float *h_out, float *c_out, float *normal, const tnp_step_saves *saves,
                                void *workspace, size_t workspace_bytes, void *stream);
/* Whole training forward in one call: tnp_lstm_forward_ex that leaves what the backward sweep needs in the caller's
 * buffers, all [steps = T_obs-1+T_dec] x M rows, contiguous: h_all / c_all [steps+1, M, H] (state before step s and
 * after the last one), X_all [steps, M, I] (LSTMCell input), act_all[l] (ReLU output of embedding layer l), gates_all
 * [steps, M, 4H] (post-activation i,f,g,o), enc_all [steps, M, C] (social encodings), nn_attrs_all, winners_all
 * [steps, M, n*n] (see tnp_step_saves), obs1_all / obs2_all [steps, M, 2] (the positions every step ran on).  With the
 * S-GAN noise interface h_all[T_obs-1] holds [ReLU(W_ctx h + b_ctx) | z] and h_clean [M, H] the encoder state h. */
typedef struct tnp_train_saves {
    float *h_all, *c_all, *X_all;
    float *act_all[2];
    float *gates_all, *enc_all, *nn_attrs_all;
    int16_t *winners_all;
    float *obs1_all, *obs2_all;
    float *h_clean;
    /* TNP_POOL_NNLSTM / TNP_POOL_TRAJ (stateful interaction encoders): pool_lstm state before step s and after the last
     * one ph_all / pc_all [steps+1, M, Hp], its post-activation gates pgates_all [steps, M, 4Hp]; act_all[0] holds the
     * pool_lstm's input features [steps, M, P]; traj_in_all [steps, M, 8] the Trajectron embedding's inputs */
    float *ph_all, *pc_all, *pgates_all, *traj_in_all;
    float *pvec_all;   /* LSTM(pool_to_input=False): [steps, M, H] interaction vector before it is added to the hidden state */
} tnp_train_saves;
TNP_API int tnp_lstm_forward_train(const tnp_lstm_model *model, const float *observed, int T_obs, int M, const float *goals,
                                   const int32_t *scene_start, const uint8_t *primary_flag, int B, int n_max,
                                   const int32_t *scene_slots, const float *truth, int T_dec, float *rel_pred, float *pred, void *workspace,
                                   size_t workspace_bytes, const tnp_lstm_extras *extras, const tnp_train_saves *saves,
                                   void *stream);
/* The reverse sweep over steps s_hi .. s_lo (inclusive, descending) of a sequence run by tnp_lstm_forward_train: per step
 * tnp_h2n_backward, tnp_lstm_cell_backward, the data-gradient GEMMs against the transposed weights, ReLU masks, the
 * embedding MLP / grid scatter / social encoding backward -- what autograd does for one LSTM.step (lstm/lstm.py:91-168).
 * Leaves the stacked operands of the deferred weight-gradient GEMMs in the caller's [steps, M, .] buffers and the state
 * gradient in dh / dc.  Callers split the sweep where host logic intervenes (the S-GAN noise hook). */
typedef struct tnp_bwd_sweep {
    const tnp_lstm_model *model;     /* geometry + hidden2normal weights (NULL Wn: no output head, S-GAN discriminator) */
    const tnp_train_saves *saves;
    int32_t S, M, B, n_max, n_enc;   /* steps, tracks, scenes, largest scene, encoder steps (the rest use the decoder cell) */
    int32_t pos_offset;              /* d_pred row of step s is s + pos_offset (1 when T_obs == 2) */
    int32_t nn_pool;                 /* 1: TNP_POOL_NN (only its embedding has parameters) */
    int32_t social_sparse;           /* 1: first-layer backward from the ego lists (tnp_pair_ego_lists) */
    int32_t directional_in;          /* 1: gradient of the directional grid with respect to the velocities (input_grad) */
    int32_t h_override_step;         /* step whose output state is h_override instead of h_all[step+1] (-1: none) */
    const float *h_override;
    const int32_t *scene_start;
    const int32_t *scene_slots;      /* [B] or NULL, as handed to tnp_lstm_forward_train (AttentionMLPPooling's padded slots) */
    const float *d_rel, *d_pred;     /* upstream gradients [S,M,5] / [S+pos_offset,M,2], either may be NULL */
    const float *wT_enc, *wT_dec;    /* [I+H, 4H] = [W_ih^T ; W_hh^T] of the two cells (wT_dec NULL when no decoder step) */
    const float *layT[3];            /* W_l^T [in_l, out_l] of embedding layer l; l = 0 only for the dense first-layer paths */
    const float *whT;                /* social: hidden_dim_encoding.weight^T [H, C] */
    const float *w_cell_major;       /* social_sparse */
    const int32_t *row_base, *row_count;            /* [M] first row / size of each track's scene */
    const int32_t *cells_all, *ego_list, *ego_count; /* cells_all [S,M,n_max]: tnp_pool_pair_cells_autograd over the stacked steps
                                                      * (social, directional_in); social_sparse: ego lists [n*n,S*M,2], [n*n,S] */
    float *dlin_all, *dG_all, *de_all, *dgoal_all;  /* [S,M,5], [S,M,4H], [S,M,E-2], [S,M,goal_dim-2] */
    float *dy_all[3];                /* [S,M,dims[l+1]] gradient of embedding layer l's pre-activation */
    float *denc_all, *dnn_all;       /* [S,M,C] social; [S,M,P] TNP_POOL_NN */
    float *dvel_pool_all;            /* directional_in: [S,M,2] */
    float *grid_all;                 /* dense first-layer weight gradient: [S,M,dims[0]] recomputed grids (else NULL) */
    float *dh, *dc;                  /* [M,H] in: gradient of the state after step s_hi; out: of the state before s_lo */
    /* TNP_POOL_HIDDENMLP (HiddenStateMLPPooling): layT[0] = out_projection.weight^T [mlp_dim, P], whT =
     * hidden_embedding.0.weight^T [H, mlp_dim_hidden]; dy_all[0] [S,M,P] gradient of the interaction vector, denc_all
     * [S,M,mlp_dim_hidden] of the hidden embedding's pre-activation, hm_G_all [S,M,ms+mv] / hm_R_all [S,M,ms+mv,2]
     * routed gradients and winning inputs of the spatial | velocity embeddings (tnp_pool_hiddenmlp_backward);
     * saves->act_all[0] = pooled [S,M,mlp_dim], saves->enc_all = hidden embedding pre-activations */
    int32_t hidden_mlp;
    float *hm_G_all, *hm_R_all;
    /* TNP_POOL_ATTNMLP (AttentionMLPPooling, folded maps in model->Wx / bx): layT[0] = Wfin^T [mlp_dim, P], whT as above,
     * at_WuT [mlp_dim, mlp_dim+4] = Wx[1]^T, at_WqT [mlp_dim, mlp_dim] = Wx[0]^T; stacked outputs for the weight
     * gradients: dy_all[0] [S,M,P], denc_all, at_eself_all / at_q_all / at_dq_all / at_ebar_all [S,M,mlp_dim],
     * at_du_all [S,M,mlp_dim+4], at_A_all [S,M,ms+mv,3] */
    int32_t attention;
    const float *at_WuT, *at_WqT;
    float *at_eself_all, *at_q_all, *at_dq_all, *at_ebar_all, *at_du_all, *at_A_all;
    /* TNP_POOL_NNLSTM / TNP_POOL_TRAJ (BPTT through the interaction encoder's pool_lstm, whose state the reference
     * carries across steps without detaching): st_pwT [P+Hp, 4Hp] = [pool_lstm.weight_ih^T ; weight_hh^T], st_h2pT
     * [Hp, P] = hidden2pool.weight^T, st_zeros [M,2] any finite buffer (the pool_lstm updates every track);
     * stacked outputs dy_all[0] [S,M,P] (gradient of the interaction vector), st_dG_all [S,M,4Hp], st_dfeat_all
     * [S,M,P] (gradient of the embedding's pre-activation); st_dph / st_dpc [M,Hp] in/out like dh / dc */
    int32_t stateful;
    const float *st_pwT, *st_h2pT, *st_zeros;
    float *st_dG_all, *st_dfeat_all, *st_dph, *st_dpc;
    const int32_t *cellwin_all;      /* directional_in: [S,M,n_max] winner table of tnp_pool_pair_cells_autograd, or NULL */
    int32_t *hm_wslot_all;           /* TNP_POOL_HIDDENMLP, optional out: [S,M,ms+mv] winning rows (position gradients), or NULL */
    float *at_posrec_all;            /* TNP_POOL_ATTNMLP, optional out: [S,M,n_max,4] per-pair position-gradient records, or NULL */
} tnp_bwd_sweep;
/* sizeof() of the structs of this header as the library was compiled (which: 0 tnp_lstm_model, 1 tnp_lstm_extras,
 * 2 tnp_step_saves, 3 tnp_train_saves, 4 tnp_bwd_sweep; 0 for any other value) -- lets a binding check its mirrors */
TNP_API size_t tnp_abi_sizeof(int which);
TNP_API size_t tnp_lstm_backward_scratch_bytes(const tnp_bwd_sweep *sweep);
TNP_API int tnp_lstm_backward_sweep(const tnp_bwd_sweep *sweep, int s_hi, int s_lo, void *scratch, size_t scratch_bytes,
                                    void *stream);
TNP_API int tnp_h2n_backward(const float *h_out, const float *Wn, const float *bn, const float *d_normal,
                             const float *d_pos, const float *obs1, const float *obs2, const float *dh_in, int M, int H,
                             float *dlin, float *dh_tot, void *stream);
/* tnp_h2n_backward followed by tnp_lstm_cell_backward in one launch (dh_tot stays in registers) */
TNP_API int tnp_h2n_cell_backward(const float *h_out, const float *Wn, const float *bn, const float *d_normal,
                                  const float *d_pos, const float *obs1, const float *obs2, const float *dh_in,
                                  const float *gates, const float *c_prev, const float *dc, int M, int H, float *dlin,
                                  float *dG, float *dc_prev, float *dh_pass, void *stream);
TNP_API int tnp_lstm_cell_backward(const float *gates, const float *c_prev, const float *dh_tot, const float *dc,
                                   const float *obs1, const float *obs2, int M, int H, float *dG, float *dc_prev,
                                   float *dh_pass, void *stream);
TNP_API int tnp_scaled_diff(const float *a, const float *b, long n, float scale, float *out, void *stream);
TNP_API int tnp_relu_mask(const float *dy, int ld_dy, const float *act, int ld_act, int M, int N, float *out, int ld_out,
                          void *stream);
TNP_API int tnp_social_scatter_backward(const float *dgrid, int ldg, const int32_t *cells, const int32_t *row_base,
                                        const int32_t *row_count, int M, int n_max, int C, int ncell, float *denc,
                                        void *stream);
/* Sparse backward of the first grid-embedding layer (social pooling; replaces one dense [M,N1]x[N1,C*n*n] GEMM per step
 * and the [N1, S*M]x[S*M, C*n*n] weight-gradient GEMM of autograd through gridbased_pooling.py:160-167 + the embedding
 * Linear, lstm/gridbased_pooling.py:60-75).  w_cell_major / dw_cell_major: [n*n][C][N1], W'[c][ch][o] = W[o][ch*n*n+c].
 *   tnp_lstm_sparse_first_layer: 1 when the step kernels of this model run the first layer from the winner table
 *   tnp_pair_ego_lists:       cells [R = steps*M][n_max] (tnp_pool_pair_cells over the stacked steps) -> per step and
 *                             cell the egos with an in-range neighbour in that cell: list [n*n][R][2] (ego row, -),
 *                             count [n*n][steps]; occ [R][n*n] bytes and occ_t [n*n][R] are scratch
 *   tnp_social_dgrid_cells:   dcell[i][c][ch] = dy[i, :] . W'[c][ch][:] for the listed (ego i, cell c) of one step
 *                             (16 egos x 16 channels per v_mfma_f32_16x16x4_f32; other entries of dcell untouched)
 *   tnp_social_scatter_backward_cells: denc[j, ch] = sum_{i: cell(i,j) >= 0} dcell[i][cell(i,j)][ch]
 *                             (the three together = dense dgrid GEMM + tnp_social_scatter_backward)
 *   tnp_sparse_hits_build:    winners [R = steps*M][n*n] (saved by tnp_lstm_step_train) -> per-cell hit lists
 *                             list [n*n][R][2] = (ego row, encoding row), count [n*n]; hit_t [n*n][R] is scratch
 *   tnp_sparse_wgrad:         dW'[c][ch][o] = sum over the hits (r, e) of cell c of dy[r, o] * enc[e, ch] */
TNP_API int tnp_lstm_sparse_first_layer(const tnp_lstm_model *model, int M);
TNP_API int tnp_pair_ego_lists(const int32_t *cells, int R, int M, int n_max, int ncell, uint8_t *occ, int32_t *occ_t,
                               int32_t *list, int32_t *count, void *stream);
TNP_API int tnp_social_dgrid_cells(const float *dy, int ldy, const float *w_cell_major, const int32_t *list,
                                   const int32_t *count, int R, int step, int M, int C, int ncell, int N1, float *dcell,
                                   void *stream);
TNP_API int tnp_social_scatter_backward_cells(const float *dcell, const int32_t *cells, const int32_t *row_base,
                                              const int32_t *row_count, int M, int n_max, int C, int ncell, float *denc,
                                              void *stream);
TNP_API int tnp_sparse_hits_build(const int16_t *winners, const int32_t *row_base, int R, int M, int ncell, int32_t *hit_t,
                                  int32_t *list, int32_t *count, void *stream);
TNP_API int tnp_sparse_wgrad(const float *dy, int ldy, const float *enc, int lde, const int32_t *list, const int32_t *count,
                             int R, int C, int ncell, int N1, float *dw_cell_major, void *stream);
/* Weight gradient of a Linear over the stacked steps: dw[Mo, No] = dy[K, Mo]^T @ x[K, No] (+ dbias[Mo] = column sums of
 * dy, NULL to skip) -- what autograd accumulates for weight / bias over the steps of LSTM.forward.  Both operands are
 * read as stored (no transposes), K is split across workgroups and reduced in a fixed order (csrc/gemm_wgrad.hip). */
TNP_API size_t tnp_wgrad_workspace_bytes(int Mo, int No, int K);
TNP_API int tnp_wgrad(const float *dy, int ld_dy, const float *x, int ld_x, int K, int Mo, int No, float *dw, int ld_dw,
                      float *dbias, void *workspace, size_t workspace_bytes, void *stream);

/* Several weight-gradient contractions in ONE launch (+ one reduce launch): every problem keeps the split / summation
 * plan of its stand-alone tnp_wgrad call (bit-identical results); `problems` is a HOST array of device pointers.  An
 * optimisation step of Social-LSTM has eight contractions, five of them tiny: grouped they fill the chip together instead
 * of paying ~35 us of launches each. */
typedef struct tnp_wgrad_problem {
    const float *dy; int ld_dy;      /* [K, Mo] gradient rows  */
    const float *x;  int ld_x;       /* [K, No] input rows     */
    int K, Mo, No;
    float *dw; int ld_dw;            /* [Mo, No] out           */
    float *dbias;                    /* [Mo] column sums of dy, or NULL */
} tnp_wgrad_problem;                 /* No == 0 with dbias: the column sums only (x and dw are not read) */
TNP_API size_t tnp_wgrad_grouped_workspace_bytes(const tnp_wgrad_problem *problems, int n);
TNP_API int tnp_wgrad_grouped(const tnp_wgrad_problem *problems, int n, void *workspace, size_t workspace_bytes, void *stream);
/* Backward of HiddenStateMLPPooling's max-pool (lstm/non_gridbased_pooling.py:196-239 under autograd): d_pooled [M, ldp]
 * (gradient of the max-pooled [spatial | hidden | velocity] vector) is routed to the winning slot of every (ego,
 * dimension): G [M, ms+mv] and R [M, ms+mv, 2] (routed gradient and the winner's input of the two Linear(2 -> dim)
 * embeddings -- tnp_colsum_prod turns them into weight / bias gradients), d_hidden_emb_pre [M, mh] (gradient of the
 * hidden embedding's pre-activation, gathered per track); winner_scratch [M, mh] int32.
 * tnp_colsum_prod: dW[k, c] = sum_rows G[row, k] R[row, k, c], db[k] = sum_rows G[row, k]; with G == NULL, R is
 * [rows, cols, 3] = per-row (sum g r_x, sum g r_y, sum g) and the three planes are summed over the rows.
 *
 * Backward of AttentionMLPPooling's pair kernel (lstm/non_gridbased_pooling.py:297-351 under autograd, linear maps folded
 * as in tnp_lstm_model.Wx): from d_ebar [M, ldd] (gradient of sum_j a_ij e_ij) and u [M, ldu]
 *   tnp_pool_attn_pair_backward: du [M, ldu] (gradient of u, column D = the bias term), A3 [M, ms+mv, 3] (per ego and
 *       spatial | velocity unit: sum over the slots of g r_x, g r_y, g), dEh [M, n_max, mh] (gradient of every slot's
 *       hidden embedding as seen by ego i), ebar [M, lde] (recomputed forward output);
 *   tnp_pool_attn_self_backward: de_self [M, ldd] (gradient of the ego's own embedding through the query) is added to
 *       A3's bias plane and, with dEh gathered per track, gives d_hidden_emb_pre [M, mh]; dself_scratch [M, mh]. */
TNP_API int tnp_pool_attn_pair_backward(const float *obs1, const float *obs2, const float *hidden_emb_pre, int ldh,
                                        const int32_t *scene_start, int B, int n_max, const int32_t *scene_slots, int ms,
                                        int mv, int mh, const float *W_spatial, const float *b_spatial, const float *W_vel,
                                        const float *b_vel, float fill, const float *u, int ldu, const float *d_ebar, int ldd, float *du, float *A3,
                                        float *dEh, float *ebar, int lde, float *pos_rec, void *stream);
TNP_API int tnp_pool_attn_self_backward(const float *obs1, const float *obs2, const float *hidden_emb_pre, int ldh,
                                        const int32_t *row_base, const int32_t *row_count, int M, int n_max, int ms, int mv,
                                        int mh, const float *b_spatial, const float *b_vel, const float *de_self, int ldd,
                                        const float *dEh, float *A3, float *dself_scratch, float *d_hidden_emb_pre,
                                        void *stream);
TNP_API int tnp_pool_hiddenmlp_backward(const float *obs1, const float *obs2, const float *hidden_emb_pre, int ldh,
                                        const int32_t *scene_start, const int32_t *row_base, const int32_t *row_count, int B,
                                        int M, int ms, int mv, int mh, const float *W_spatial, const float *b_spatial,
                                        const float *W_vel, const float *b_vel, const float *d_pooled, int ldp, float *G,
                                        float *R, float *d_hidden_emb_pre, int32_t *winner_scratch, int32_t *winner_slots,
                                        void *stream);
/* Gradients with respect to the POSITIONS through the non-grid interaction modules -- needed when the frames fed to the
 * sequence carry gradient (the S-GAN discriminator scoring the generator's prediction, sgan/sgan.py:512-576 under autograd;
 * the reference trainer gives the discriminator a copy of the generator's module, sgan/trainer.py:590-592).  Rows are the
 * stacked steps (R = steps * M) with their scenes in row_base / row_count [R]; d_obs1 / d_obs2 [R, 2] out.
 *   tnp_pool_nn_pos_backward: NearestNeighborMLP / NearestNeighborLSTM features (lstm/non_gridbased_pooling.py:98-147): the
 *     top-n selection carries no gradient, the gathered [rel pos | rel vel] attributes do (NaN components excluded);
 *     d_pre [R, n*d] = gradient of the embedding's pre-activation, W [d, input_dim]; sel_scratch [R, n] int32,
 *     ga_scratch [R, n, 4].
 *   tnp_pool_hiddenmlp_pos_backward: HiddenStateMLPPooling (:196-239): the max-pool routes dimension kk's gradient G [R, ms+mv]
 *     to the slot winner_slots [R, ms+mv] (row index RELATIVE TO ITS OWN STEP's first row when M_step > 0 -- what the
 *     backward sweep leaves in hm_wslot_all -- or -1; tnp_pool_hiddenmlp_backward's optional last output). */
TNP_API int tnp_pool_nn_pos_backward(const float *obs1, const float *obs2, const int32_t *row_base, const int32_t *row_count,
                                     int R, int n_max, int n_sel, int in_dim, const float *W, int d, const float *d_pre, int ldd,
                                     int32_t *sel_scratch, float *ga_scratch, float *d_obs1, float *d_obs2, void *stream);
/*   tnp_pool_pair_pos_gather: AttentionMLPPooling (:297-351): tnp_pool_attn_pair_backward's optional `pos_rec` [R, n_max, 4] holds
 *     per (ego, slot) the gradient of the pair's relative position and relative velocity; gathered per row here. */
TNP_API int tnp_pool_pair_pos_gather(const float *pair_records, const int32_t *row_base, const int32_t *row_count, int R, int n_max,
                                     float *d_obs1, float *d_obs2, void *stream);
TNP_API int tnp_pool_hiddenmlp_pos_backward(const float *G, const int32_t *winner_slots, const float *W_spatial,
                                            const float *W_vel, const int32_t *row_base, const int32_t *row_count, int R,
                                            int M_step, int ms, int mv, float *d_obs1, float *d_obs2, void *stream);
TNP_API size_t tnp_colsum_prod_workspace_bytes(long rows, int cols);
TNP_API int tnp_colsum_prod(const float *G, const float *R, long rows, int cols, float *dW, float *db, void *workspace,
                            size_t workspace_bytes, void *stream);
/* The two re-laid-out copies of the first grid-embedding layer W [N1, C n n] (torch.nn.Linear weight,
 * lstm/gridbased_pooling.py:308-335) that tnp_lstm_model carries, in one launch:
 *   w_cell_major [ncell][C][N1]:               W'[c][ch][o]                      = W[o][ch ncell + c]
 *   w_quad_major [ncell][N1/64][C/4][64][4]:   W''[c][o/64][ch/4][o%64][ch%4]    = W[o][ch ncell + c]   (NULL: not wanted;
 *                                              needs N1 % 64 == 0 and C % 4 == 0)
 * Rebuild both whenever the parameter changes (every optimiser step while training). */
TNP_API int tnp_pool_embed_weight_layouts(const float *W, int ldw, int N1, int C, int ncell, float *w_cell_major,
                                          float *w_quad_major, void *stream);
/* out [cols, rows] = in [rows, cols]^T (LDS-tiled; operands of the weight-gradient GEMMs) */
TNP_API int tnp_transpose(const float *in, int ld_in, int rows, int cols, float *out, int ld_out, void *stream);
/* the same for several matrices in ONE launch (the backward sweep transposes six weight matrices per optimisation step:
 * six launches of a few microseconds of work each) */
typedef struct tnp_transpose_problem {
    const float *in; int ld_in;      /* [rows, cols] */
    int rows, cols;
    float *out; int ld_out;          /* [cols, rows] */
} tnp_transpose_problem;
TNP_API int tnp_transpose_grouped(const tnp_transpose_problem *problems, int n, void *stream);
/* gradient of the directional grid's values (v_j - v_i, lstm/gridbased_pooling.py:118-143) with respect to the tracks'
 * velocities: dvel [M,2]; needed when the positions fed to the sequence carry gradient (S-GAN discriminator input).
 * cells / winner: tnp_pool_pair_cells_autograd's tables (winner may be NULL: no per-channel zero-value mask) */
TNP_API int tnp_directional_scatter_backward(const float *dgrid, int ldg, const int32_t *cells, const int32_t *winner,
                                             const int32_t *row_base, const int32_t *row_count, const float *obs1,
                                             const float *obs2, int M, int n_max, int ncell, float *dvel, void *stream);

/* -------------------------------------------------------------------------------------------
 * Losses on the primaries (rows scene_start[s]) of a batch, lstm/loss.py:
 *   mode 0  PredictionLoss.forward (:52-91): -log(0.01 + bg*N(x;mu,3,3,0) + (0.99-bg)*N(x;mu,sigma,rho))
 *   mode 1  L2Loss.forward (:107-135): squared error of (mu_x, mu_y); scale carries the x100 (and the 1/2 of the
 *           mean over the two coordinates)
 *   inputs [T,M,5], targets [T,M,2]; out: [1] mean (keep_batch_dim = 0) or [B] per-scene mean over time;
 *   values_ws: T*B floats of scratch.
 * tnp_primary_loss_backward: d_inputs [T,M,5] = d(out)/d(inputs) . grad_out (grad_out [1] or [B] as `out`); analytic
 *   derivatives of the expressions above, rows of non-primaries are zero -- what loss.backward() hands to LSTM.forward.
 * tnp_collision_loss_forward = CollisionLoss (:138-162) over predictions [T,M,ld] (first two columns), out [1].
 * tnp_collision_loss_backward: d_predictions [T,M,ld] = d(out)/d(predictions) . grad_out[0] -- non-zero only in the
 *   primaries' rows (the neighbours are detached, :155): -col_wt/col_distance * sum over the colliding neighbours of the
 *   unit vector from the neighbour to the primary; NaN coordinates (overwritten with -1000 in place, :148) and
 *   coincident points (torch.norm's subgradient) get zero.
 * ----------------------------------------------------------------------------------------- */
TNP_API int tnp_primary_loss_forward(int mode, const float *inputs, const float *targets, const int32_t *scene_start,
                             int B, int T, int M, float background_rate, int keep_batch_dim, float scale,
                             float *values_ws, float *out, void *stream);
/* reduction of the fused loss (tnp_lstm_extras.loss_values, [T, ld] row-indexed) exactly as tnp_primary_loss_forward
 * reduces its values: out [1] or [B]; values_ws T*B floats */
TNP_API int tnp_primary_loss_reduce(const float *row_values, int ld, const int32_t *scene_start, int B, int T,
                            int keep_batch_dim, float scale, float *values_ws, float *out, void *stream);
TNP_API int tnp_primary_loss_backward(int mode, const float *inputs, const float *targets, const int32_t *scene_start,
                              int B, int T, int M, float background_rate, int keep_batch_dim, float scale,
                              const float *grad_out, float *d_inputs, void *stream);
TNP_API int tnp_collision_loss_forward(const float *predictions, int ld, const int32_t *scene_start, int B, int T,
                               int M, float col_wt, float col_distance, float *partial_ws, float *out, void *stream);
TNP_API int tnp_collision_loss_backward(const float *predictions, int ld, const int32_t *scene_start, int B, int T,
                                int M, float col_wt, float col_distance, const float *grad_out, float *d_predictions,
                                void *stream);

/* -------------------------------------------------------------------------------------------
 * Optimiser step of the reference's trainer (lstm/trainer.py:497, sgan/trainer.py:540-546:
 * torch.optim.Adam(params, lr, weight_decay), amsgrad off): ONE launch over all parameter tensors
 * instead of torch's nine per-operation multi-tensor launches; same single-tensor formulas,
 * one rounding per torch operation.  Tensors whose gradient is None are simply not listed (torch
 * skips them, weight decay included).  `tensors` is a HOST array of device pointers; `step` is the
 * 1-based step count AFTER the increment, as in torch's state['step'].  Hyper-parameters are doubles, as torch holds
 * them (python floats): 1 - beta2 formed from a float32 beta2 would already differ from torch's by 1.3e-5 relative.
 * ----------------------------------------------------------------------------------------- */
typedef struct tnp_adam_tensor {
    float *param;          /* [n] updated in place */
    const float *grad;     /* [n] */
    float *exp_avg;        /* [n] first moment, updated in place */
    float *exp_avg_sq;     /* [n] second moment, updated in place */
    int64_t n;
} tnp_adam_tensor;
TNP_API int tnp_adam_step(const tnp_adam_tensor *tensors, int n_tensors, int step, double lr, double beta1, double beta2,
                          double eps, double weight_decay, void *stream);

/* Measurement hooks (bench.py's roofline leg) are declared in trajnet_hip_profile.h: they are not part of the drop-in
 * boundary. */

/* -------------------------------------------------------------------------------------------
 * classical.constant_velocity.predict (classical/constant_velocity.py:4-20), batched:
 *   last, prev [N,2] float64 -> out [n_predict,N,2] float64
 * ----------------------------------------------------------------------------------------- */
TNP_API int tnp_constant_velocity(const double *last, const double *prev, int N, int n_predict, double *out,
                          void *stream);

/* -------------------------------------------------------------------------------------------
 * Batched classical rollouts (one workgroup per scene).  The reference wrappers simulate one scene per call through
 * un-vendored CPU libraries (socialforce, rvo2, pykalman); initial states are built on the host exactly as the
 * wrappers do (classical/socialforce.py:15-72, classical/orca.py:14-79) and handed over flat with scene_start.
 *   tnp_sf_rollout      socialforce.Simulator(...).step() x n_steps, state after every sample_every-th step
 *                       (classical/socialforce.py:84-95).  state0 [M,6] f64 = x,y,vx,vy,goal_x,goal_y -> out [n_out,M,2]
 *   tnp_orca_rollout    rvo2 doStep() x n_iter with the wrapper's preferred-velocity update (classical/orca.py:90-119),
 *                       float32 simulator state, float64 goals / speeds -> out [n_out,M,2] f32;
 *                       nbr_dbg (optional) [M,16] int32: neighbour indices of the first step (index parity checks)
 *   tnp_kalman_predict  pykalman KalmanFilter(CV model).em(n_iter) -> smooth -> mean of n_samples sampled observation
 *                       sequences (classical/kalman.py:40-60); obs [n_tracks,T,2] f64, z standard normal draws
 *                       [n_tracks,n_samples,n_steps,6] f64 -> out [n_tracks,n_steps,2] (row 0 = last smoothed state)
 * ----------------------------------------------------------------------------------------- */
TNP_API int tnp_sf_rollout(const double *state0, const int32_t *scene_start, int B, int M, int n_max, int n_steps,
                   int sample_every, double tau, double v0, double sigma, double delta_t, double *out, void *stream);
TNP_API int tnp_orca_rollout(const float *pos0, const float *vel0, const double *goals, const double *speed,
                     const float *max_speed, const int32_t *scene_start, int B, int M, int n_max, int n_iter,
                     int sample_every, float time_step, float neighbor_dist, int max_neighbors, float time_horizon,
                     float radius, float *out, int *nbr_dbg, void *stream);
TNP_API int tnp_kalman_predict(const double *obs, int n_tracks, int T, int n_iter, int n_steps, int n_samples,
                       const double *z, double transition_var, double observation_var, double *out, void *stream);

/* -------------------------------------------------------------------------------------------
 * Evaluator feed, host side (no GPU work; csrc/ndjson_io.cpp).  Replaces, for the batched predictor feed
 * data.predict_dataset, what the reference does per row in Python:
 *   tnp_ndjson_parse        trajnetplusplustools.Reader's json.loads per line (lstm/trajnet_evaluator.py:29-36,
 *                           evaluator/trajnet_evaluator.py:25-36): a TrajNet++ .ndjson buffer -> track columns
 *                           (frame, pedestrian, x, y; rows with a non-null "prediction_number" skipped) and scene
 *                           columns (id, primary, start, end), `cap` entries each.  Returns 0, or -(line number)
 *                           of the first line it does not take (non-integer frame / id fields, malformed JSON):
 *                           the caller falls back to its general reader.
 *   tnp_format_predictions  evaluator/write_utils.py:42-81 through trajnetplusplustools.writers.trajnet: the lines
 *                           of a batch's prediction file -- per scene one scene record (fps 2.5, tag 0), per mode
 *                           the primary's rows then every neighbour's, coordinates as repr(round(float(x), 2)).
 *                           pred [n_modes][pred_length][M][2] float64; scene s = columns split[s] .. split[s+1]
 *                           (the first is its primary); first_frame[s] = frame of the first predicted row.
 *                           Returns the bytes written or -1 if `cap` (tnp_format_predictions_bound) is too small.
 * ----------------------------------------------------------------------------------------- */
TNP_API int64_t tnp_ndjson_parse(const char *buf, size_t n, int64_t cap, int64_t *t_frame, int64_t *t_ped, double *t_x,
                                 double *t_y, int64_t *n_tracks, int64_t *s_id, int64_t *s_ped, int64_t *s_start,
                                 int64_t *s_end, int64_t *n_scenes);
TNP_API int64_t tnp_format_predictions(const double *pred, int n_modes, int pred_length, int64_t M, int64_t n_scenes,
                                       const int64_t *split, const int64_t *ped, const int64_t *scene_id,
                                       const int64_t *first_frame, const int64_t *frame_diff, const int64_t *scene_start,
                                       const int64_t *scene_end, char *out, size_t cap);
TNP_API size_t tnp_format_predictions_bound(int n_modes, int pred_length, int64_t M, int64_t n_scenes);

#ifdef __cplusplus
}
#endif
#endif /* TRAJNET_HIP_H */
