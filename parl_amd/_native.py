"""ctypes binding of libparl_hip.so (the C ABI declared in include/parl_hip.h).

The library is the product: if it is missing, or a call fails, this module raises — no entry of
the C ABI has a CPU twin in parl_amd, and a CPU tensor handed to an op raises (oracle/ is test
infrastructure and is never imported from here).  What the host framework keeps in PyTorch-ROCm by
design (north_star: "host code stays Python calling PyTorch-ROCm for the small policy/value
forward/backward") still runs ON THE DEVICE through the library's kernels: e.g. IMPALA's loss for a
shape the one-kernel loss has no instantiation for (A not in {2,3,4,6,9,18} or T > 256) is the
reference's formulas as torch ops around the fused V-trace kernel (algorithms/impala/impala.py,
`_vtrace_loss`) — a framework-level composition of device kernels, not a fallback off the device.

torch must be imported before the library is loaded so that libparl_hip.so binds to the same
libamdhip64.so.7 the torch allocator uses (device pointers are shared across the boundary).
"""
import ctypes
import os
import threading

import torch  # noqa: F401  (must precede the CDLL load, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# PARL_HIP_LIB: load another build of the same library (kernel experiments); default = in-tree
LIB_PATH = os.environ.get('PARL_HIP_LIB') or os.path.join(_HERE, 'libparl_hip.so')

c_f32p = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_p = ctypes.c_void_p
_u64 = ctypes.c_uint64
_i64 = ctypes.c_int64
_sz = ctypes.c_size_t
_d = ctypes.c_double

# name -> (restype, argtypes); kept in lock-step with include/parl_hip.h
# (tests/test_capi_symbols.py parses the header and checks every declared symbol is here
# and exported by the .so).
SIGNATURES = {
    'parlhip_version': (_i, []),
    'parlhip_source_hash': (ctypes.c_char_p, []),
    'parlhip_strerror': (ctypes.c_char_p, [_i]),
    'parlhip_last_hip_error': (_i, []),
    'parlhip_consume_device_errors': (_i, [_p]),
    'parlhip_vtrace_f32': (_i, [_p] * 8 + [_i, _i, _f, _f, _p]),
    'parlhip_vtrace_from_logits_f32':
    (_i, [_p] * 10 + [_i, _i, _i, _i, _f, _f, _f, _p]),
    'parlhip_impala_heads_loss_workspace_bytes': (_sz, [_i, _i]),
    'parlhip_impala_heads_loss_f32': (_i, [_p] * 15 + [_i, _i, _i, _i, _f, _f, _f, _f, _f, _p]),
    'parlhip_clip_adam_workspace_bytes': (_sz, [_i, _p]),
    'parlhip_clip_adam_f32': (_i, [_i] + [_p] * 7 + [_d, _d, _d, _d, _p, _p, _p]),
    'parlhip_impala_loss_f32': (_i, [_p] * 11 + [_i, _i, _i, _i, _f, _f, _f, _f, _f, _p]),
    'parlhip_gae_f32': (_i, [_p] * 7 + [_i, _i, _f, _f, _i, _i, _p]),
    'parlhip_gae_workspace_bytes': (_sz, [_i, _i]),
    'parlhip_gae_ws_f32': (_i, [_p] * 7 + [_i, _i, _f, _f, _i, _i, _p, _sz, _p]),
    'parlhip_discount_cumsum_f32': (_i, [_p, _p, _p, _i, _i, _f, _p]),
    'parlhip_adv_normalize_workspace_bytes': (_sz, [_i64]),
    'parlhip_adv_normalize_f32': (_i, [_p, _p, _p, _i64, _f, _p, _sz, _p, _p]),
    'parlhip_categorical_sample_f32': (_i, [_p, _p, _p, _i, _i, _p]),
    'parlhip_policy_head_sample_f32': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _u64, _u64, _u64, _p]),
    'parlhip_policy_head_sample_at_f32': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _u64, _p, _u64, _u64, _p]),
    'parlhip_policy_sample_f32':
    (_i, [_p, _i, _p, _p, _p, _i, _i, _u64, _u64, _u64, _p]),
    'parlhip_policy_sample_at_f32':
    (_i, [_p, _i, _p, _p, _p, _i, _i, _u64, _p, _u64, _u64, _p]),
    'parlhip_frame_post_tables_bytes': (_sz, [_i]),
    'parlhip_frame_post_tables_init': (_i, [_p, _i]),
    'parlhip_frame_post_u8': (_i, [_p, _p, _i64, _i, _p, _p, _i64, _i, _i, _p, _p]),
    'parlhip_frame_post_since_u8': (_i, [_p, _p, _i64, _i, _p, _p, _i64, _i, _i, _p, _p, _p, _p]),
    'parlhip_frame_post_step_u8': (_i, [_p, _p, _i64, _i, _p, _p, _i64, _i, _i, _p, _p, _p, _p, _p, _p, _p]),
    'parlhip_atari_state_bytes': (_sz, []),
    'parlhip_atari_frame_bytes': (_sz, []),
    'parlhip_atari_rom_table_bytes': (_sz, [ctypes.c_uint32]),
    'parlhip_atari_reset_cache_bytes': (_sz, []),
    'parlhip_atari_num_actions': (_i, [_i]),
    'parlhip_atari_native_cart': (ctypes.c_uint32, [_i]),
    'parlhip_atari_rom_table_build': (_i, [_p, ctypes.c_uint32, _p]),
    'parlhip_atari_reset_cache_build': (_i, [_p, ctypes.c_uint32, _i, _i64, _p, _p, _p]),
    'parlhip_atari_vec_reset':
    (_i, [_p, _p, ctypes.c_uint32, _i, _p, _p, _i, _u64, _u64, _i64, _p, _p]),
    'parlhip_atari_vec_step':
    (_i, [_p, _p, ctypes.c_uint32, _i, _p, _p, _p, _p, _p, _p, _p, _i, _u64, _u64, _i64, _p, _p, _p]),
    'parlhip_atari_vec_step_obs':
    (_i, [_p, _p, ctypes.c_uint32, _i, _p, _p, _p, _p, _p, _p, _p, _i, _u64, _u64, _i64, _p, _p, _p, _i, _p, _p, _p, _p,
          _p]),
    'parlhip_atari_vec_step_policy_obs':
    (_i, [_p, _p, ctypes.c_uint32, _i, _p, _p, _p, _p, _p, _p, _i, _u64, _u64, _i64, _p, _p, _p, _i, _p, _p, _p, _p,
          _p, _p, _p, _p, _p, _i, _i, _u64, _p, _u64, _u64, _p]),
    'parlhip_atari_vec_step_elastic':
    (_i, [_p, _p, ctypes.c_uint32, _i, _p, _p, _p, _p, _p, _p, _p, _i, _u64, _u64, _i64, _p, _p, _i, _i, _i, _i, _i] +
     [_p] * 7 + [_i] + [_p] * 4),
    'parlhip_atari_vec_step_elastic_obs':
    (_i, [_p, _p, ctypes.c_uint32, _i, _p, _p, _p, _p, _p, _p, _p, _i, _u64, _u64, _i64, _p, _p, _i, _i, _i, _i, _i] +
     [_p] * 7 + [_i] + [_p] * 4 + [_i, _p, _p]),
    'parlhip_stack_gather_ring_u8': (_i, [_p, _p, _p, _i, _i, _i, _p, _p, _i64, _p, _p]),
    'parlhip_stack_since_update_u8': (_i, [_p, _p, _p, _i, _p]),
    'parlhip_stack_gather_u8': (_i, [_p, _p, _i, _i, _p, _p, _i64, _p, _p]),
    'parlhip_atari42_conv12_u8_f32': (_i, [_p, _p, _p, _p, _p, _p, _i, _p]),
    'parlhip_atari42_conv12_ring_u8_f32': (_i, [_p, _p, _i, _i, _i] + [_p] * 6),
    'parlhip_atari42_conv12_weights_bytes': (_sz, []),
    'parlhip_atari42_conv12_weights_f32': (_i, [_p, _p, _p, _p]),
    'parlhip_atari42_conv12_packed_u8_f32': (_i, [_p, _p, _p, _p, _p, _i, _p]),
    'parlhip_atari42_conv12_ring_packed_u8_f32': (_i, [_p, _p, _i, _i, _i] + [_p] * 5),
    'parlhip_atari84_conv1_u8_f32': (_i, [_p, _p, _p, _p, _i, _p]),
    'parlhip_atari84_conv1_ring_u8_f32': (_i, [_p, _p, _i, _i, _i] + [_p] * 4),
    'parlhip_atari84_conv1_packed_u8_f32': (_i, [_p, _p, _p, _p, _i, _p]),
    'parlhip_atari84_conv1_ring_packed_u8_f32': (_i, [_p, _p, _i, _i, _i] + [_p] * 4),
    'parlhip_atari84_conv23_f32': (_i, [_p] * 7 + [_i, _p]),
    'parlhip_atari84_conv3_bwd_workspace_bytes': (_sz, [_i]),
    'parlhip_atari84_conv3_bwd_f32': (_i, [_p] * 4 + [_i] + [_p] * 4),
    'parlhip_atari84_conv2_bwd_workspace_bytes': (_sz, [_i]),
    'parlhip_atari84_conv2_bwd_f32': (_i, [_p] * 3 + [_i] + [_p] * 4),
    'parlhip_atari84_conv1_bwd_workspace_bytes': (_sz, [_i]),
    'parlhip_atari84_conv1_bwd_f32': (_i, [_p, _p, _i, _p, _p, _p]),
    'parlhip_atari42_conv12_bwd_workspace_bytes': (_sz, [_i]),
    'parlhip_atari42_conv12_bwd_f32': (_i, [_p] * 6 + [_i] + [_p] * 6),
    'parlhip_atari42_conv12_bwd_packed_f32': (_i, [_p] * 5 + [_i] + [_p] * 6),
    'parlhip_atari42_conv12_a1_bytes': (_sz, [_i]),
    'parlhip_atari42_conv12_packed_save_u8_f32': (_i, [_p] * 6 + [_i, _p]),
    'parlhip_atari42_conv12_bwd_saved_f32': (_i, [_p] * 6 + [_i] + [_p] * 6),
    'parlhip_vecnorm_obs_f64': (_i, [_p] * 7 + [_i, _i, _d, _d, _i, _p]),
    'parlhip_vecnorm_reward_f64': (_i, [_p] * 8 + [_i, _d, _d, _d, _p]),
    'parlhip_ppo_sample_batch_f32': (_i, [_p] * 13 + [_i64, _i64, _i, _i, _p]),
    'parlhip_episode_stats_accum_f64': (_i, [_p, _p, _i, _p, _p]),
}


class ParlHipError(RuntimeError):
    """A libparl_hip.so entry point returned a negative PARLHIP_E* code."""


_lib = None
_lock = threading.Lock()


def lib():
    """Load (once) and return the ctypes handle; raises if the extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise ImportError(
                    'parl_amd: %s not found. Build it with `python -c "import '
                    '__graft_entry__ as g; g.build()"` or `make -C parl_amd/csrc`. '
                    'There is no fallback path.' % LIB_PATH)
            handle = ctypes.CDLL(LIB_PATH)
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(handle, name)
                fn.restype = res
                fn.argtypes = args
            _lib = handle
    return _lib


def check(code, what):
    if code < 0:
        l = lib()
        msg = l.parlhip_strerror(code).decode()
        raise ParlHipError('%s failed: %s (code %d, hip error %d)' %
                           (what, msg, code, l.parlhip_last_hip_error()))
    return code


def ptr(t):
    """Device pointer of a contiguous CUDA/HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise ParlHipError('parl_amd ops need device (HIP) tensors; got a %s tensor. '
                           'There is no CPU fallback.' % t.device)
    if not t.is_contiguous():
        raise ParlHipError('parl_amd ops need contiguous tensors')
    return t.data_ptr()


def stream_ptr():
    """hipStream_t of torch's current stream as an integer handle."""
    return torch.cuda.current_stream().cuda_stream
