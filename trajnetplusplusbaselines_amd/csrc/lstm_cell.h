// torch.nn.LSTMCell's pointwise part (reference lstm/lstm.py:154) as the epilogue of the gates GEMMs -- shared by
// gemm_f32_mfma.hip (32 x 32 x 2 tiles) and gemm_skinny.hip (16 x 16 x 4 tiles for small batches), so that both evaluate the
// same expressions in the same order.
#pragma once
#include "tnp_internal.h"

namespace tnp {

// LSTMCell activations on the hardware transcendental units: sigmoid(x) = rcp(1 + exp(-x)) with v_exp_f32 / v_rcp_f32
// (1 ulp each), tanh(x) = 2 sigmoid(2x) - 1 (absolute error ~1e-7).  libm's expf / tanhf and an IEEE divide are 20 precise
// transcendentals per lane = 6 k of the gates kernel's 32 k cycles; with these the step is 1 us shorter (1.6 % of the forward)
// and the distance to the oracle does not move: max |pred - oracle| at config 2 full size 2.9e-6 against 2.4e-6 (summation
// order dominates both; tests/parity_margin.py), all parity tests unchanged.  -DTNP_PRECISE_GATES restores libm.
#ifndef TNP_PRECISE_GATES
__device__ __forceinline__ float sigmoidf_acc(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_acc(float x) { return fmaf(2.0f, sigmoidf_acc(2.0f * x), -1.0f); }
#else
__device__ __forceinline__ float sigmoidf_acc(float x) { return 1.0f / (1.0f + expf(-x)); }
__device__ __forceinline__ float tanhf_acc(float x) { return tanhf(x); }
#endif

// One (track, unit) of torch.nn.LSTMCell from the four gate pre-activations (bias included).
__device__ __forceinline__ void lstm_cell_store(const GemmArgs &g, int row, int unit, float pi, float pf, float pg,
                                                float po) {
    if (row >= g.M) return;
    const size_t o = (size_t)row * g.H + unit;
    if (g.mask[row]) {
        const float ig = sigmoidf_acc(pi);
        const float fg = sigmoidf_acc(pf);
        const float gg = tanhf_acc(pg);
        const float og = sigmoidf_acc(po);
        const float cn = fg * g.c_in[o] + ig * gg;
        g.c_out[o] = cn;
        g.h_out[o] = og * tanhf_acc(cn);
        if (g.gates_out) {
            float *go = g.gates_out + (size_t)row * 4 * g.H + unit;
            go[0] = ig; go[g.H] = fg; go[2 * g.H] = gg; go[3 * g.H] = og;
        }
    } else {  // absent track: state frozen (reference lstm/lstm.py:118-124,158-166)
        g.c_out[o] = g.c_in[o];
        g.h_out[o] = g.h_in[o];
    }
}

// same, with c_in / mask of the (row, unit) fetched by the caller before its main loop
__device__ __forceinline__ void lstm_cell_store_pf(const GemmArgs &g, int row, int unit, float pi, float pf, float pg,
                                                   float po, float c_prev, int present) {
    if (row >= g.M) return;
    const size_t o = (size_t)row * g.H + unit;
    if (present) {
        const float ig = sigmoidf_acc(pi);
        const float fg = sigmoidf_acc(pf);
        const float gg = tanhf_acc(pg);
        const float og = sigmoidf_acc(po);
        const float cn = fg * c_prev + ig * gg;
        g.c_out[o] = cn;
        g.h_out[o] = og * tanhf_acc(cn);
        if (g.gates_out) {
            float *go = g.gates_out + (size_t)row * 4 * g.H + unit;
            go[0] = ig; go[g.H] = fg; go[2 * g.H] = gg; go[3 * g.H] = og;
        }
    } else {
        g.c_out[o] = c_prev;
        g.h_out[o] = g.h_in[o];
    }
}

}  // namespace tnp
