"""ctypes binding of libmvnerf_hip.so (include/mvnerf_hip.h).

The HIP library is the product; there is no CPU or PyTorch fallback.  If the shared object is
missing or fails to load, importing :func:`lib` raises ``RuntimeError`` telling the user to run
``python -c "import __graft_entry__ as g; g.build()"`` (or ``make -C thesis_clip_nerf_amd/csrc``).
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('MVNERF_LIB', os.path.join(_HERE, 'lib', 'libmvnerf_hip.so'))   # override: A/B builds
HEADER_PATH = os.path.join(os.path.dirname(_HERE), 'include', 'mvnerf_hip.h')

NET_PARAMS = 247300
Q7_ZERO, Q7_CLAMP = 0, 1


class TrainCall(ctypes.Structure):
    """mvnerf_train_call (include/mvnerf_hip.h): one training problem, field for field."""
    _fields_ = ([(n, c_void_p) for n in ('rays_o', 'rays_d', 'images', 'features', 'intrinsics', 'extrinsics_inv', 'u_coarse', 'u_fine', 'labels')] +
                [(n, c_int) for n in ('B', 'V', 'R', 'S', 'H', 'W')] +
                [('near_', c_double), ('far_', c_double), ('q7_mode', c_int), ('stop_fine_z', c_int), ('use_texel_tables', c_int)] +
                [(n, c_void_p) for n in ('net_coarse', 'net_fine', 'packed_coarse', 'packed_fine', 'split_coarse', 'split_fine',
                                         'bwd_streams_coarse', 'bwd_streams_fine', 'loss', 'grad', 'rgb', 'depth', 'fine_rgb', 'fine_depth',
                                         'd_features', 'workspace')] +
                [('workspace_bytes', c_size_t), ('fine_grad_event', c_void_p)])


class AdamState(ctypes.Structure):
    """mvnerf_adam_state (include/mvnerf_hip.h)."""
    _fields_ = [('m', c_void_p), ('v', c_void_p), ('lr_t', c_float), ('beta1', c_float), ('beta2', c_float), ('eps', c_float),
                ('clip', c_float), ('update_mask', c_void_p), ('repack', c_int)]


class GemmTnBatch(ctypes.Structure):
    """mvnerf_gemm_tn_batch (include/mvnerf_hip.h)."""
    _fields_ = ([(n, c_void_p) for n in ('g', 'a', 'g2', 'a2')] +
                [(n, ctypes.c_long) for n in ('g_batch_stride', 'a_batch_stride', 'g2_batch_stride', 'a2_batch_stride')] +
                [(n, c_int) for n in ('ldg', 'lda', 'ldg2', 'lda2', 'colsum_of')])


# name -> (restype, argtypes); must list every symbol include/mvnerf_hip.h declares
# (tests/test_abi.py cross-checks this table against the header and the built library).
SIGNATURES = {
    'mvnerf_abi_version': (c_int, []),
    'mvnerf_last_error': (c_char_p, []),
    'mvnerf_packed_net_floats': (c_size_t, []),
    'mvnerf_pack_net': (c_int, [c_void_p, c_void_p, c_void_p]),
    'mvnerf_get_rays': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p,
                                c_void_p, c_void_p]),
    'mvnerf_stratified_depths': (c_int, [c_void_p, c_int, c_int, c_double, c_double, c_void_p, c_void_p]),
    'mvnerf_field_eval': (c_int, [c_void_p] * 8 + [c_int] * 6 + [c_void_p] * 8),
    'mvnerf_query_workspace_bytes': (c_size_t, [c_int] * 3),
    'mvnerf_query_jvp': (c_int, [c_void_p] * 9 + [c_int] * 5 + [c_void_p] * 4),
    'mvnerf_query_vjp_scratch_bytes': (c_size_t, [c_int] * 3),
    'mvnerf_query_vjp': (c_int, [c_void_p] * 9 + [c_int] * 5 + [c_void_p] * 4),
    'mvnerf_stash_fused_acts': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'mvnerf_texel_table_bytes': (c_size_t, [c_int] * 4),
    'mvnerf_project_texels': (c_int, [c_void_p, c_void_p] + [c_int] * 4 + [c_void_p, c_void_p]),
    'mvnerf_project_texels2': (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p] * 3),
    'mvnerf_field_eval_table': (c_int, [c_void_p] * 9 + [c_int] * 6 + [c_void_p] * 8),
    'mvnerf_packed_net_bf16_bytes': (c_size_t, []),
    'mvnerf_project_texels_bf16': (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p] * 3),
    'mvnerf_pack_net_bf16': (c_int, [c_void_p, c_void_p, c_void_p]),
    'mvnerf_field_eval_bf16': (c_int, [c_void_p] * 10 + [c_int] * 6 + [c_void_p] * 6),
    'mvnerf_field_eval_bf16maps': (c_int, [c_void_p] * 10 + [c_int] * 6 + [c_void_p] * 6),
    'mvnerf_project_texels_bf16maps': (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p] * 3),
    'mvnerf_packed_net_split_bytes': (c_size_t, []),
    'mvnerf_pack_net_split': (c_int, [c_void_p, c_void_p, c_void_p]),
    'mvnerf_field_eval_split': (c_int, [c_void_p] * 10 + [c_int] * 6 + [c_void_p] * 8),
    'mvnerf_field_workspace_bytes': (c_size_t, [c_int, c_int, c_int]),
    'mvnerf_composite': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mvnerf_resample': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_void_p]),
    'mvnerf_resample_bwd': (c_int, [c_void_p] * 5 + [c_int, c_int, c_int, c_void_p, c_void_p]),
    'mvnerf_points_on_rays': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'mvnerf_project_points': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'mvnerf_camera_directions': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'mvnerf_position_encoding': (c_int, [c_void_p, c_long, c_int, c_float, c_void_p, c_void_p]),
    'mvnerf_bilinear_gather': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'mvnerf_sigma_to_alpha': (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p]),
    'mvnerf_sample_pdf': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mvnerf_readout': (c_int, [c_void_p, c_void_p, c_void_p, c_long, c_void_p, c_void_p]),
    'mvnerf_finish_view': (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p]),
    'mvnerf_set_deterministic': (c_int, [c_int]),
    'mvnerf_set_split_kernel': (c_int, [c_int]),
    'mvnerf_stash_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'mvnerf_field_backward_scratch_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'mvnerf_field_eval_stash': (c_int, [c_void_p] * 9 + [c_int] * 6 + [c_void_p] * 4),
    'mvnerf_field_eval_stash_split': (c_int, [c_void_p] * 10 + [c_int] * 6 + [c_void_p] * 4),
    'mvnerf_pack_bwd_streams': (c_int, [c_void_p, c_void_p, c_void_p]),
    'mvnerf_mse_grad': (c_int, [c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_void_p]),
    'mvnerf_composite_bwd': (c_int, [c_void_p] * 5 + [c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'mvnerf_field_backward': (c_int, [c_void_p] * 12 + [c_int] * 6 + [c_void_p] * 5),
    'mvnerf_gemm_nt_scratch_bytes': (c_size_t, [c_int] * 3),
    'mvnerf_gemm_tn_scratch_bytes': (c_size_t, [c_int] * 3),
    'mvnerf_gemm_tn': (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p] * 2),
    'mvnerf_gemm_tn_batched_scratch_bytes': (c_size_t, [c_int] * 5),
    'mvnerf_gemm_tn_batched': (c_int, [c_void_p] * 3 + [c_int] * 4 + [c_void_p] * 2),
    'mvnerf_gemm_nt': (c_int, [c_void_p] * 3 + [c_int] * 3 + [c_void_p] * 2),
    'mvnerf_gemm_nt_bias': (c_int, [c_void_p] * 4 + [c_int] * 3 + [c_void_p] * 2),
    'mvnerf_field_backward_table': (c_int, [c_void_p] * 14 + [c_int] * 6 + [c_void_p] * 5),
    'mvnerf_adam_clip': (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_float, c_float, c_float, c_float, c_float,
                                 c_void_p, c_void_p]),
    'mvnerf_grasp_head_packed_floats': (c_size_t, []),
    'mvnerf_grasp_head_pack': (c_int, [c_void_p] * 4),
    'mvnerf_grasp_head_fwd': (c_int, [c_void_p] * 4 + [c_long] + [c_void_p] * 3),
    'mvnerf_grasp_head_vjp': (c_int, [c_void_p] * 4 + [c_long] + [c_void_p] * 5),
    'mvnerf_grasp_head_vjp_bwd': (c_int, [c_void_p] * 6 + [c_long] + [c_void_p] * 5),
    'mvnerf_train_workspace_bytes': (c_size_t, [c_int] * 8),
    'mvnerf_loss_and_grads': (c_int, [ctypes.POINTER(TrainCall), c_void_p]),
    'mvnerf_apply_gradients': (c_int, [ctypes.POINTER(TrainCall), ctypes.POINTER(AdamState), c_void_p]),
    'mvnerf_train_step': (c_int, [ctypes.POINTER(TrainCall), ctypes.POINTER(AdamState), c_void_p]),
    'mvnerf_render_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int]),
    'mvnerf_render_fwd': (c_int, [c_void_p] * 10 + [c_int] * 6 + [c_double, c_double, c_int] + [c_void_p] * 6 +
                          [c_int, c_void_p]),
    'mvnerf_render_fwd_split': (c_int, [c_void_p] * 12 + [c_int] * 6 + [c_double, c_double, c_int] + [c_void_p] * 6 +
                                [c_int, c_void_p]),
}

_lib = None


def lib():
    """Load (once) and return the ctypes handle with argtypes set."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'{LIB_PATH} not found: the HIP extension is not built and there is no fallback path. '
            'Build it with `make -C thesis_clip_nerf_amd/csrc` (hipcc, gfx950) or `__graft_entry__.build()`.')
    try:
        handle = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # missing libamdhip64 etc.
        raise RuntimeError(f'cannot load {LIB_PATH}: {e}') from e
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name)
        fn.restype = res
        fn.argtypes = args
    _lib = handle
    return _lib


def check(rc, what='mvnerf'):
    """Map the C return code onto Python exceptions (<0: ValueError, >0: RuntimeError/HIP)."""
    if rc == 0:
        return
    msg = lib().mvnerf_last_error().decode('utf-8', 'replace')
    if rc < 0:
        raise ValueError(f'{what}: {msg} (code {rc})')
    raise RuntimeError(f'{what}: HIP error {rc}: {msg}')
