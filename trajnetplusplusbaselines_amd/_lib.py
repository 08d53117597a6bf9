"""ctypes binding of libtrajnet_hip.so (include/trajnet_hip.h) -- the only way the Python host layer reaches
the HIP kernels.  There is NO CPU fallback: if the library is missing the import of any op fails loudly.

PyTorch is used only as plumbing here: device memory (tensor.data_ptr()), the current HIP stream and
torch.distributed.  Every entry point enqueues on ``torch.cuda.current_stream()`` and returns immediately.
"""
import ctypes
import numbers
import os
import threading

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libtrajnet_hip.so')
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'trajnet_hip.h')

POOL_NONE, POOL_OCCUPANCY, POOL_DIRECTIONAL, POOL_SOCIAL = -1, 0, 1, 2
POOL_NN, POOL_HIDDENMLP, POOL_ATTNMLP, POOL_NNLSTM, POOL_TRAJ = 4, 5, 6, 7, 8
ABI_VERSION = 7   # TNP_ABI_VERSION of include/trajnet_hip.h this binding was written against
POOL_TYPES = {None: POOL_NONE, 'occupancy': POOL_OCCUPANCY, 'directional': POOL_DIRECTIONAL, 'social': POOL_SOCIAL}

_fp = ctypes.c_void_p


class LstmModel(ctypes.Structure):
    """mirror of ``struct tnp_lstm_model`` (include/trajnet_hip.h)"""
    _fields_ = [
        ('E', ctypes.c_int32), ('H', ctypes.c_int32), ('goal_flag', ctypes.c_int32), ('goal_dim', ctypes.c_int32),
        ('pool_type', ctypes.c_int32), ('n', ctypes.c_int32), ('C', ctypes.c_int32), ('P', ctypes.c_int32),
        ('n_layers', ctypes.c_int32), ('dims', ctypes.c_int32 * 4),
        ('cell', ctypes.c_float), ('half_x', ctypes.c_float), ('half_y', ctypes.c_float), ('constant', ctypes.c_float),
        ('We', _fp), ('be', _fp), ('Wg', _fp), ('bg', _fp),
        ('enc_Wih', _fp), ('enc_Whh', _fp), ('enc_bih', _fp), ('enc_bhh', _fp),
        ('dec_Wih', _fp), ('dec_Whh', _fp), ('dec_bih', _fp), ('dec_bhh', _fp),
        ('Wn', _fp), ('bn', _fp), ('Wh', _fp), ('bh', _fp),
        ('Wp', _fp * 3), ('bp', _fp * 3),
        ('Wp0_cell_major', _fp),
        ('variant', ctypes.c_int32),
        ('Wx', _fp * 3), ('bx', _fp * 3),
        ('Wp0_quad_major', _fp),
        ('pool_size', ctypes.c_int32), ('blur_size', ctypes.c_int32),
    ]


class LstmExtras(ctypes.Structure):
    """mirror of ``struct tnp_lstm_extras``"""
    _fields_ = [('W_ctx', _fp), ('b_ctx', _fp), ('noise', _fp), ('noise_dim', ctypes.c_int32), ('noise_group_tracks', ctypes.c_int32),
                ('h_final', _fp), ('loss_targets', _fp), ('loss_values', _fp), ('loss_steps', ctypes.c_int32),
                ('loss_mode', ctypes.c_int32), ('loss_background_rate', ctypes.c_float), ('h_scale', _fp)]


_LIB = None


def exported_symbols_in_header():
    """Names of every entry point the headers under include/ declare (trajnet_hip.h = the drop-in boundary,
    trajnet_hip_profile.h = the measurement hooks; used by the ABI test)."""
    import glob
    import re
    text = ''
    for path in sorted(glob.glob(os.path.join(os.path.dirname(HEADER_PATH), '*.h'))):
        with open(path) as f:
            text += f.read()
    return sorted(set(re.findall(r'TNP_API\s+[\w\s\*]+?\b(tnp_\w+)\s*\(', text)))


def lib():
    """Load the shared library (once).  Raises if it has not been built -- there is no fallback."""
    global _LIB
    if _LIB is not None:
        return _LIB
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'libtrajnet_hip.so not found at %s: build it with `python -c "import __graft_entry__ as g; g.build()"` '
            '(hipcc --offload-arch=gfx950). The MI355X path has no CPU fallback.' % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    L.tnp_abi_version.restype = ctypes.c_int
    L.tnp_last_error.restype = ctypes.c_char_p
    L.tnp_lstm_workspace_bytes.restype = ctypes.c_size_t
    L.tnp_lstm_workspace_bytes.argtypes = [ctypes.POINTER(LstmModel), ctypes.c_int, ctypes.c_int]
    L.tnp_mark_primaries.argtypes = [_fp, ctypes.c_int, ctypes.c_int, _fp, _fp]
    L.tnp_pool_grid_forward.argtypes = [ctypes.c_int, _fp, _fp, _fp, ctypes.c_int, _fp, ctypes.c_int, ctypes.c_int, _fp,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                        ctypes.c_float, _fp, ctypes.c_int, _fp, _fp]
    L.tnp_linear_forward.argtypes = [_fp, ctypes.c_int, _fp, ctypes.c_int, _fp, _fp, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp]
    L.tnp_lstm_forward.argtypes = [ctypes.POINTER(LstmModel), _fp, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp,
                                   ctypes.c_int, ctypes.c_int, _fp, _fp, ctypes.c_int, _fp, _fp, _fp, ctypes.c_size_t, _fp]
    L.tnp_lstm_forward_ex.argtypes = [ctypes.POINTER(LstmModel), _fp, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp,
                                      ctypes.c_int, ctypes.c_int, _fp, _fp, ctypes.c_int, _fp, _fp, _fp, ctypes.c_size_t,
                                      ctypes.POINTER(LstmExtras), _fp]
    L.tnp_lstm_forward_train.argtypes = [ctypes.POINTER(LstmModel), _fp, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp,
                                         ctypes.c_int, ctypes.c_int, _fp, _fp, ctypes.c_int, _fp, _fp, _fp, ctypes.c_size_t,
                                         ctypes.POINTER(LstmExtras), _fp, _fp]
    L.tnp_abi_sizeof.restype = ctypes.c_size_t
    L.tnp_abi_sizeof.argtypes = [ctypes.c_int]
    L.tnp_lstm_backward_scratch_bytes.restype = ctypes.c_size_t
    L.tnp_lstm_backward_scratch_bytes.argtypes = [_fp]
    L.tnp_lstm_backward_sweep.argtypes = [_fp, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_size_t, _fp]
    L.tnp_lstm_step.argtypes = [ctypes.POINTER(LstmModel), ctypes.c_int, _fp, _fp, _fp, _fp, _fp, _fp, ctypes.c_int,
                                ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, _fp, _fp, ctypes.c_size_t, _fp]
    L.tnp_adam_step.argtypes = [_fp, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double,
                                ctypes.c_double, _fp]
    L.tnp_profile_begin.argtypes = [ctypes.c_int]
    L.tnp_profile_read.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)]
    L.tnp_profile_dispatch_timed.argtypes = []
    L.tnp_tuning_set.argtypes = [ctypes.c_char_p, ctypes.c_long]
    L.tnp_ndjson_parse.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int64] + [_fp] * 4 + [ctypes.POINTER(ctypes.c_int64)] + \
                                  [_fp] * 4 + [ctypes.POINTER(ctypes.c_int64)]
    L.tnp_ndjson_parse.restype = ctypes.c_int64
    L.tnp_format_predictions.argtypes = [_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64] + [_fp] * 7 + \
                                        [ctypes.c_char_p, ctypes.c_size_t]
    L.tnp_format_predictions.restype = ctypes.c_int64
    L.tnp_format_predictions_bound.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64]
    L.tnp_format_predictions_bound.restype = ctypes.c_size_t
    L.tnp_constant_velocity.argtypes = [_fp, _fp, ctypes.c_int, ctypes.c_int, _fp, _fp]
    L.tnp_pool_embed_sparse_workspace_bytes.restype = ctypes.c_size_t
    L.tnp_pool_embed_sparse_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.tnp_row_base.argtypes = [_fp, ctypes.c_int, _fp, _fp]
    L.tnp_pool_nn_forward.argtypes = [_fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp, ctypes.c_int,
                                      _fp, ctypes.c_int, _fp]
    L.tnp_lstm_step_train.argtypes = [ctypes.POINTER(LstmModel), ctypes.c_int, _fp, _fp, _fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, _fp, _fp, _fp, _fp, _fp, _fp, ctypes.c_size_t, _fp]
    L.tnp_h2n_backward.argtypes = [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp]
    L.tnp_h2n_cell_backward.argtypes = [_fp] * 11 + [ctypes.c_int, ctypes.c_int] + [_fp] * 5
    L.tnp_lstm_cell_backward.argtypes = [_fp, _fp, _fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, _fp]
    L.tnp_scaled_diff.argtypes = [_fp, _fp, ctypes.c_long, ctypes.c_float, _fp, _fp]
    L.tnp_relu_mask.argtypes = [_fp, ctypes.c_int, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_int, _fp]
    L.tnp_social_scatter_backward.argtypes = [_fp, ctypes.c_int, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, _fp, _fp]
    L.tnp_lstm_sparse_first_layer.argtypes = [ctypes.POINTER(LstmModel), ctypes.c_int]
    L.tnp_pair_ego_lists.argtypes = [_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, _fp, _fp]
    L.tnp_social_dgrid_cells.argtypes = [_fp, ctypes.c_int, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp]
    L.tnp_social_scatter_backward_cells.argtypes = [_fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                    ctypes.c_int, _fp, _fp]
    L.tnp_sparse_hits_build.argtypes = [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, _fp]
    L.tnp_sparse_wgrad.argtypes = [_fp, ctypes.c_int, _fp, ctypes.c_int, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, _fp, _fp]
    L.tnp_wgrad_workspace_bytes.restype = ctypes.c_size_t
    L.tnp_wgrad_workspace_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.tnp_wgrad_grouped_workspace_bytes.restype = ctypes.c_size_t
    L.tnp_wgrad_grouped_workspace_bytes.argtypes = [_fp, ctypes.c_int]
    L.tnp_wgrad_grouped.argtypes = [_fp, ctypes.c_int, _fp, ctypes.c_size_t, _fp]
    L.tnp_wgrad.argtypes = [_fp, ctypes.c_int, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_int,
                            _fp, _fp, ctypes.c_size_t, _fp]
    L.tnp_pool_hiddenmlp_backward.argtypes = [_fp, _fp, _fp, ctypes.c_int, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, _fp, _fp, ctypes.c_int, _fp, _fp, _fp, _fp,
                                              _fp, _fp]
    L.tnp_pool_nn_pos_backward.argtypes = [_fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp,
                                           ctypes.c_int, _fp, ctypes.c_int, _fp, _fp, _fp, _fp, _fp]
    L.tnp_pool_hiddenmlp_pos_backward.argtypes = [_fp, _fp, _fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, _fp, _fp, _fp]
    L.tnp_colsum_prod_workspace_bytes.restype = ctypes.c_size_t
    L.tnp_colsum_prod_workspace_bytes.argtypes = [ctypes.c_long, ctypes.c_int]
    L.tnp_colsum_prod.argtypes = [_fp, _fp, ctypes.c_long, ctypes.c_int, _fp, _fp, _fp, ctypes.c_size_t, _fp]
    L.tnp_pool_attn_pair_backward.argtypes = [_fp, _fp, _fp, ctypes.c_int, _fp, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, _fp, ctypes.c_float, _fp, ctypes.c_int,
                                              _fp, ctypes.c_int, _fp, _fp, _fp, _fp, ctypes.c_int, _fp, _fp]
    L.tnp_pool_pair_pos_gather.argtypes = [_fp, _fp, _fp, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp]
    L.tnp_pool_attn_self_backward.argtypes = [_fp, _fp, _fp, ctypes.c_int, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, ctypes.c_int, _fp, _fp, _fp, _fp, _fp]
    L.tnp_transpose.argtypes = [_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_int, _fp]
    L.tnp_directional_scatter_backward.argtypes = [_fp, ctypes.c_int, _fp, _fp, _fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int, _fp, _fp]
    L.tnp_pool_traj_forward.argtypes = [_fp, _fp, ctypes.c_int, _fp, _fp, ctypes.c_int, _fp, ctypes.c_int, _fp, _fp]
    L.tnp_pool_attn_self.argtypes = [_fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int, _fp, _fp, ctypes.c_float, _fp, ctypes.c_int, _fp]
    L.tnp_pool_attn_pair.argtypes = [_fp, _fp, _fp, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_int, ctypes.c_int, _fp,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, _fp, ctypes.c_float, _fp,
                                     ctypes.c_int, _fp, ctypes.c_int, _fp]
    L.tnp_pool_hiddenmlp_forward.argtypes = [_fp, _fp, _fp, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_int, ctypes.c_int, _fp, _fp, _fp, _fp, _fp, ctypes.c_int, _fp]
    L.tnp_pool_embed_sparse_forward.argtypes = [_fp, _fp, ctypes.c_int, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp, ctypes.c_int, _fp,
                                                ctypes.c_size_t, _fp]
    L.tnp_primary_loss_backward.argtypes = [ctypes.c_int, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_float, ctypes.c_int, ctypes.c_float, _fp, _fp, _fp]
    L.tnp_primary_loss_forward.argtypes = [ctypes.c_int, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_float, ctypes.c_int, ctypes.c_float, _fp, _fp, _fp]
    L.tnp_primary_loss_reduce.argtypes = [_fp, ctypes.c_int, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float, _fp, _fp, _fp]
    L.tnp_collision_loss_forward.argtypes = [_fp, ctypes.c_int, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_float, ctypes.c_float, _fp, _fp, _fp]
    L.tnp_collision_loss_backward.argtypes = [_fp, ctypes.c_int, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_float, ctypes.c_float, _fp, _fp, _fp]
    L.tnp_sf_rollout.argtypes = [_fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_double, ctypes.c_double, ctypes.c_double, ctypes.c_double, _fp, _fp]
    L.tnp_orca_rollout.argtypes = [_fp, _fp, _fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                   ctypes.c_int, ctypes.c_float, ctypes.c_float, ctypes.c_int, ctypes.c_float,
                                   ctypes.c_float, _fp, _fp, _fp]
    L.tnp_kalman_predict.argtypes = [_fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _fp,
                                     ctypes.c_double, ctypes.c_double, _fp, _fp]
    L.tnp_pool_pair_cells.argtypes = [_fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_float,
                                      ctypes.c_float, ctypes.c_float, _fp, _fp]
    L.tnp_pool_pair_cells_autograd.argtypes = [_fp, _fp, _fp, _fp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                               ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, _fp, _fp, _fp, _fp]
    if L.tnp_abi_version() != ABI_VERSION:
        raise RuntimeError('libtrajnet_hip.so ABI version mismatch')
    _LIB = L
    return L


def tuning_set(key, value):
    """Tile-selection knob of the native library (include/trajnet_hip_profile.h: tests and probes only)."""
    check(lib().tnp_tuning_set(key.encode(), int(value)), 'tnp_tuning_set')


def check(rc, what):
    if rc != 0:
        msg = lib().tnp_last_error().decode('utf-8', 'replace')
        raise RuntimeError('%s failed (%d): %s' % (what, rc, msg))


def raw_stream():
    """The CURRENT device's current HIP stream as an integer.  (``torch.cuda.current_stream().cuda_stream`` builds a Stream
    object through five layers of device-index helpers: 9 us a call, 32 calls per optimisation step = 0.28 ms of a
    batch_size-8 step's 2.3 ms of host time -- tools/diag/small_batch_workload.py train_hostprofile.)"""
    return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())


def stream_ptr():
    return ctypes.c_void_p(raw_stream())


def ptr(t):
    """Device pointer of a tensor (or NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def require_device(t, what):
    if not t.is_cuda:
        raise RuntimeError('%s must live on a ROCm device (got %s): the MI355X path has no CPU fallback'
                           % (what, t.device))


class _Staging(threading.local):
    """A small ring of REUSED pinned host buffers per thread for asynchronous host-to-device copies.  ``tensor.pin_memory()``
    asks torch's caching host allocator for a block per call; with ever-new sizes (one scene per call, a new batch_split per
    step) that is a hipHostMalloc of about a millisecond for every size class it has not seen -- the evaluator's per-scene call
    went from 1.2 to 2.4 ms.  Thirty-two slots (a host that runs a step or two ahead of the GPU never waits for one) of at least 64 KB, grown to the next power of two on demand; a slot is reused only after
    the event recorded behind its last copy has completed."""

    def __init__(self):
        self.slots, self.events, self.k = [None] * 32, [None] * 32, 0


_staging = _Staging()
#: largest copy that goes through the ring: its pinned footprint is bounded by 32 slots x 4 MB = 128 MB per thread
_STAGING_MAX_SLOT = 4 << 20


def h2d_async(t, device):
    """contiguous host tensor -> device tensor through a reused pinned buffer, without blocking the host"""
    t = t.contiguous()
    nbytes = t.numel() * t.element_size()
    if nbytes == 0 or torch.device(device).type != 'cuda':
        return t.to(device)
    if nbytes > _STAGING_MAX_SLOT:
        # big one-off inputs (a 4096 x 128 classical batch is 88 MB) must not grow the ring: 32 slots of 128 MB each would pin
        # 4 GB per thread for good.  torch's caching host allocator recycles the block once the copy has completed.
        with torch.cuda.device(device):
            return t.pin_memory().to(device, non_blocking=True)
    s = _staging
    k = s.k
    s.k = (k + 1) % len(s.slots)
    ev = s.events[k]
    if ev is not None:
        ev.synchronize()
    else:
        ev = s.events[k] = torch.cuda.Event()          # one event per slot, re-recorded: no construction per copy
    buf = s.slots[k]
    if buf is None or buf.numel() < nbytes:
        buf = s.slots[k] = torch.empty(max(1 << 16, 1 << (nbytes - 1).bit_length()), dtype=torch.uint8, pin_memory=True)
    view = buf[:nbytes].view(t.dtype).view(t.shape)
    view.copy_(t)
    device = torch.device(device)
    if device.index is None or device.index == torch._C._cuda_getDevice():
        # (already the current device, the usual case: `with torch.cuda.device(...)` and the Stream object behind a bare
        # Event.record() are ~10 us each, five copies per optimisation step)
        out = view.to(device, non_blocking=True)
        ev.record(_stream_object())
    else:
        with torch.cuda.device(device):
            out = view.to(device, non_blocking=True)
            ev.record()
    return out


_stream_objs = {}


def _stream_object():
    """torch Stream object of the current device's current stream, cached by its raw handle"""
    raw = raw_stream()
    key = (torch._C._cuda_getDevice(), raw)
    so = _stream_objs.get(key)
    if so is None:
        if len(_stream_objs) > 64:
            _stream_objs.clear()
        so = _stream_objs[key] = torch.cuda.current_stream()
    return so


def f32c(t, device=None):
    """fp32, contiguous, on `device` (H2D copy if the caller handed a host tensor: converted on the host first, then ONE
    asynchronous copy from pinned memory -- a pageable .to(device) blocks the host until the stream has drained, which keeps a
    loop over batches from ever running ahead of the GPU)."""
    if device is not None and t.device != device:
        if t.device.type == 'cpu' and torch.device(device).type == 'cuda' and not (t.requires_grad and torch.is_grad_enabled()):
            # (a host tensor that requires grad keeps its autograd edge: the differentiable .to(device) below)
            t = t.detach()
            if t.dtype != torch.float32:
                t = t.float()
            return h2d_async(t, device)
        t = t.to(device)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class StreamMark(object):
    """Remembers on which stream a cached device object was produced: ``join()`` makes the CURRENT stream wait for that point
    when it is a different stream (nothing otherwise, the common case).  Lets caches that are filled by kernels -- the
    re-laid-out weight copies, the scene index tables -- be used by forward passes on several streams."""

    def __init__(self):
        # the CURRENT device's current stream: the one _lib.stream_ptr() hands to every native launch
        self.device = None
        self.stream = torch.cuda.current_stream()
        self.raw = self.stream.cuda_stream
        self.event = torch.cuda.Event()
        self.event.record(self.stream)

    def join(self):
        if raw_stream() == self.raw:
            return          # the common case: same stream, ordered anyway
        if torch.cuda.is_current_stream_capturing():
            return          # hipGraph capture (LSTM._forward_graphed): the capture begins after a device synchronisation
        cur = torch.cuda.current_stream(self.device)
        if cur != self.stream:
            cur.wait_event(self.event)


class SceneIndex(object):
    """int32 scene starts + per-track primary flags + the reference's padded slot counts (lstm/lstm.py:29).

    ``pad_to`` says how many slots the reference would pad every scene to -- the padded slots are absent neighbours
    with the highest index, which clobber cell (0, 0) of a grid and enter AttentionMLPPooling's
    softmax:
      None      the largest scene of THIS batch_split (what ``generate_pooling_inputs`` does for the batch it is given);
      int       that many slots for every scene (>= the largest scene): a shard of a bigger batch passes the global
                maximum, so the shard computes what the unsharded batch would;
      'scene'   every scene on its own, no padded slots: what one reference call per scene computes (the evaluator,
                lstm/lstm.py:291);
      sequence  per-scene slot counts.
    ``n_max`` is the largest slot count, ``slots`` the per-scene int32 device tensor (None = n_max for all)."""
    _cache = {}

    def __init__(self, batch_split, device, pad_to=None):
        split = batch_split.detach().to('cpu', torch.int64)
        if split.numel() < 2:
            raise ValueError('batch_split needs at least two entries')
        sizes = split[1:] - split[:-1]
        if int(sizes.min()) < 0:
            raise ValueError('batch_split must be non-decreasing')
        self.B = int(split.numel() - 1)
        self.M = int(split[-1] - split[0])
        if int(split[0]) != 0:
            raise ValueError('batch_split must start at 0')
        self.n_max = int(sizes.max()) if self.B > 0 else 0
        self.slots = None
        self._slots_host, self._sizes_host = None, sizes
        if isinstance(pad_to, numbers.Integral):
            pad_to = int(pad_to)
        if pad_to is not None:
            if isinstance(pad_to, str):
                if pad_to != 'scene':
                    raise ValueError("pad_to must be None, an int, 'scene' or per-scene slot counts")
                slots = sizes.clone()
            elif isinstance(pad_to, int):
                slots = torch.full_like(sizes, pad_to)
            else:
                slots = torch.as_tensor(pad_to, dtype=torch.int64).reshape(-1)
                if slots.numel() != self.B:
                    raise ValueError('pad_to has %d entries for %d scenes' % (slots.numel(), self.B))
            if self.B > 0 and bool((slots < sizes).any()):
                raise ValueError('pad_to is smaller than a scene: padded slot counts must be >= the track counts')
            self.n_max = int(slots.max()) if self.B > 0 else 0
            if not isinstance(pad_to, int):
                self._slots_host = slots.to(torch.int64)
        # first row / size of every row's scene (training backward): built on the host, no device-side sync later
        # (the host copies stay: stacked_rows() builds the whole-sweep tables from them without reading the device back)
        # (numpy, not torch.repeat_interleave: torch's CPU kernel is an at::parallel_for -- on a 256-core host its thread
        # pool costs 0.7 ms per call warm and 7 ms cold for these ~300-element tables, every step of a trainer whose
        # batch_split changes; tools/diag/host_micro.py)
        sizes_np = sizes.numpy()
        self._row_base_host = torch.from_numpy(np.repeat(split[:-1].numpy(), sizes_np).astype(np.int32))
        self._row_count_host = torch.from_numpy(np.repeat(sizes_np, sizes_np).astype(np.int32))
        # ALL tables travel in ONE asynchronous copy from pinned memory: a pageable .to(device) is a blocking hipMemcpy that
        # first waits for everything queued on the stream -- with a new batch_split every step (the reference trainer) the host
        # could never run ahead of the GPU, which then idled between steps while the next batch was being prepared
        parts = [split.to(torch.int32), self._row_base_host, self._row_count_host]
        if self._slots_host is not None:
            parts.append(self._slots_host.to(torch.int32))
        flat = torch.cat(parts)
        flat = h2d_async(flat, device)
        o = 0
        self.starts = flat[o:o + self.B + 1]; o += self.B + 1
        self.row_base = flat[o:o + self.M]; o += self.M
        self.row_count = flat[o:o + self.M]; o += self.M
        if self._slots_host is not None:
            self.slots = flat[o:o + self.B]
        self.primary = torch.empty(max(self.M, 1), dtype=torch.uint8, device=device)
        check(lib().tnp_mark_primaries(ptr(self.starts), self.B, self.M, ptr(self.primary), stream_ptr()),
              'tnp_mark_primaries')
        self._mark = StreamMark() if torch.device(device).type == 'cuda' else None     # built on this stream (see get())

    def stacked_rows(self, S):
        """(row_base, row_count, row_padded) of S copies of the batch stacked along the rows (row r of step s is row s M + r):
        what the backward sweep's whole-sweep pair-cell launch indexes with; row_padded = slots the row's scene is padded to,
        None when every scene is padded to ``n_max``.  Built from the HOST copies of the tables once per (batch structure, S):
        no device read-back (a real trainer has a new batch_split every step, and a D2H copy here would drain the forward's
        queue in the middle of the backward pass), small pinned H2D copies."""
        got = self.__dict__.setdefault('_stacked', {}).get(S)
        if got is None:
            dev = self.row_base.device
            off = (torch.arange(S, dtype=torch.int32) * self.M)[:, None]
            tabs = [(off + self._row_base_host[None]).reshape(-1), self._row_count_host.repeat(S)]
            if self._slots_host is not None:
                tabs.append(torch.from_numpy(np.repeat(self._slots_host.numpy(), self._sizes_host.numpy()).astype(np.int32)).repeat(S))
            if dev.type == 'cuda':             # ONE pinned copy for the two / three tables
                flat = h2d_async(torch.cat(tabs), dev)
                n = tabs[0].numel()
                tabs = [flat[i * n:(i + 1) * n] for i in range(len(tabs))]
            else:
                tabs = [t.to(dev) for t in tabs]
            if len(tabs) == 2:
                tabs.append(None)
            if len(self._stacked) > 4:
                self._stacked.clear()
            got = self._stacked[S] = tuple(tabs) + (StreamMark() if dev.type == 'cuda' else None,)
        if got[3] is not None:
            got[3].join()
        return got[:3]

    @classmethod
    def get(cls, batch_split, device, pad_to=None):
        if isinstance(batch_split, (list, tuple)):
            batch_split = torch.tensor(batch_split, dtype=torch.int64)
        host = batch_split.detach().to('cpu', torch.int64)
        if isinstance(pad_to, numbers.Integral):
            pad_to = int(pad_to)
        if pad_to is not None and not isinstance(pad_to, (int, str)):
            pad_to = tuple(int(v) for v in torch.as_tensor(pad_to).reshape(-1).tolist())
        key = (str(device), host.numpy().tobytes(), pad_to)
        idx = cls._cache.get(key)
        if idx is None:
            if len(cls._cache) > 64:
                cls._cache.clear()
            idx = cls(host, device, pad_to)
            idx.key = key                 # content key: what hipGraph captures of a call shape are filed under
            cls._cache[key] = idx
        if idx._mark is not None:
            idx._mark.join()           # a cached index built on another stream: wait for its tables
        return idx


def linear_forward(x, weight, bias, relu=False, variant=0, out=None):
    """act(x @ weight.T + bias) on the matrix cores (tnp_linear_forward). x [M,K] fp32, weight [N,K]."""
    require_device(x, 'input')
    x = f32c(x)
    weight = f32c(weight, x.device)
    bias_t = f32c(bias, x.device) if bias is not None else None
    M, K = x.shape
    N = weight.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    check(lib().tnp_linear_forward(ptr(x), x.stride(0), ptr(weight), weight.stride(0), ptr(bias_t), ptr(out),
                                   out.stride(0), M, N, K, int(relu), int(variant), stream_ptr()), 'tnp_linear_forward')
    return out
