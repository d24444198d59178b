"""ctypes binding of libswem_hip.so (C ABI: include/swem_hip.h).

There is no CPU fallback: if the library is missing or a call fails this module raises.
Build the library with ``python -m swem_amd.build`` (or ``__graft_entry__.build()``).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (SWEM_HIP_LIB: another build of the same library -- the debug / tuning builds the tools make, e.g. tools/conv_stamps.py)
LIB_PATH = os.environ.get('SWEM_HIP_LIB') or os.path.join(_HERE, 'libswem_hip.so')

_p, _i, _ll, _f, _sz = C.c_void_p, C.c_int, C.c_longlong, C.c_float, C.c_size_t

# name -> (restype, argtypes).  Must list every symbol include/swem_hip.h declares
# (tests/test_abi.py cross-checks against the header).
SIGNATURES = {
    'swem_version': (_i, []),
    'swem_last_error': (C.c_char_p, []),
    'swem_device_cus': (_i, []),
    'swem_conv2d_workspace': (_sz, [_i] * 11),
    'swem_conv2d_nhwc_f32': (_i, [_p, _p, _i, _ll, _p, _i, _ll, _p, _i, _ll, _i, _i, _i, _p, _ll, _p, _p, _p, _ll, _p,
                                  _i, _i, _i, _i, _i, _i, _i, _p, _sz]),
    'swem_conv2d_nhwc_f32_planes': (_i, [_p, _p, _i, _ll, _p, _i, _ll, _p, _i, _ll, _i, _i, _i, _p, _ll, _p, _p, _p, _ll, _p,
                                         _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p, _i, _p, _i, _p]),
    'swem_conv2d_nhwc_bf16x3_planes': (_i, [_p, _p, _i, _ll, _ll, _p, _i, _ll, _ll, _p, _i, _ll, _ll, _i, _i, _i, _p, _p, _p, _p,
                                            _ll, _p, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p, _i, _p, _i]),
    'swem_conv2d_nhwc_bf16x3_planes_ctr': (_i, [_p, _p, _i, _ll, _ll, _p, _i, _ll, _ll, _p, _i, _ll, _ll, _i, _i, _i, _p, _p, _p, _p,
                                            _ll, _p, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p, _i, _p, _i, _p, _sz, _p]),
    'swem_conv2d_nhwc_bf16x3_planes_res': (_i, [_p, _p, _i, _ll, _ll, _p, _i, _ll, _ll, _p, _i, _ll, _ll, _i, _i, _i, _p, _p, _p, _p,
                                            _ll, _ll, _i, _ll, _p, _i, _i, _i, _i, _i, _i, _i, _p, _sz, _p, _i, _p, _i, _p, _sz, _p]),
    'swem_split_bf16x3_f32': (_i, [_p, _p, _p, _ll, _i, _i]),
    'swem_split_f16x2_f32': (_i, [_p, _p, _p, _ll, _i, _i, _p]),
    'swem_conv2d_nhwc_bf16x3': (_i, [_p, _p, _i, _ll, _ll, _p, _i, _ll, _ll, _p, _i, _ll, _ll, _i, _i, _i, _p, _p, _p, _p, _ll,
                                     _p, _i, _i, _i, _i, _i, _i, _i, _p, _sz]),
    'swem_bottleneck_f16x3': (_i, [_p, _p, _ll, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _ll, _p]),
    'swem_prep_key_input_f32': (_i, [_p, _p, _p, _p, _p, _i, _i, _i]),
    'swem_prep_value_input_f32': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i]),
    'swem_prep_input_s2d_f32': (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    'swem_maxpool3x3s2_nhwc_f32': (_i, [_p, _p, _p, _i, _i, _i, _i]),
    'swem_maxpool3x3s2_nhwc_f32_planes': (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _i, _p, _i, _p]),
    'swem_upsample_add_nhwc_f32': (_i, [_p, _p, _ll, _p, _p, _i, _i, _i, _i, _i, _i]),
    'swem_upsample_add_nhwc_f32_planes': (_i, [_p, _p, _ll, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p]),
    'swem_upsample_add_grouped_nhwc_f32': (_i, [_p, _p, _ll, _i, _p, _p, _i, _i, _i, _i, _i, _i]),
    'swem_upsample_add_grouped_nhwc_f32_planes': (_i, [_p, _p, _ll, _i, _p, _p, _i, _i, _i, _i, _i, _i, _p, _i, _p, _i, _p]),
    'swem_resize_planes_f32': (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i]),
    'swem_mask_prep_f32': (_i, [_p, _p, _i, _i, _i, _p, _i, _i, _p, _i, _i, _i, _i]),
    'swem_cbam_workspace': (_sz, [_i, _i, _i, _i]),
    'swem_cbam_f32': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _sz]),
    'swem_cbam_f32_planes': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _sz, _p, _i, _p, _i, _p]),
    'swem_pred_head_f32': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i]),
    'swem_decode_head_f32': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i]),
    'swem_argmax_onehot_i64': (_i, [_p, _p, _p, _p, _i, _i, _ll]),
    'swem_concat2_nhwc_f32': (_i, [_p, _p, _i, _ll, _p, _i, _ll, _p, _i, _ll]),
    'swem_lincomb_f32': (_i, [_p, _p, _f, _p, _f, _p, _ll]),
    'swem_inject_objects_f32': (_i, [_p, _p, _p, _p, _i, _i, _i, _ll]),
    'swem_pack_u8_i64': (_i, [_p, _p, _p, _ll]),
    'swem_transpose_f32': (_i, [_p, _p, _p, _i, _i, _i, _i]),
    'swem_em_pad': (_i, [_i]),
    'swem_em_norm_bases_f32': (_i, [_p, _p, _p, _i, _i, _i]),
    'swem_em_pack_bases_f32': (_i, [_p, _p, _p, _i, _i, _i]),
    'swem_em_ew_f32': (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _i, _i]),
    'swem_em_mstep_workspace': (_sz, [_i, _i, _i, _i]),
    'swem_em_mstep_f32': (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _sz]),
    'swem_memorize_workspace': (_sz, [_i, _i, _i, _i, _i]),
    'swem_memorize_f32': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _p, _sz]),
    'swem_memorize_packed_f32': (_i, [_p] * 13 + [_i] * 8 + [_f, _p, _sz, _p]),
    'swem_memorize_packed_clips_f32': (_i, [_p] * 13 + [_i] * 9 + [_f, _p, _sz, _p]),
    'swem_memorize_packed_keys_f32': (_i, [_p] * 9 + [_i] * 7 + [_f, _p, _sz]),
    'swem_memorize_packed_values_f32': (_i, [_p] * 8 + [_i] * 5 + [_p]),
    'swem_match_pad': (_i, [_i]),
    'swem_match_workspace': (_sz, [_i, _i, _i, _i, _i, _i, _i]),
    'swem_match_f32': (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _f, _i, _p, _sz]),
    'swem_match_pack_bank_f32': (_i, [_p] * 6 + [_i] * 6 + [_p]),
    'swem_match_packed_workspace': (_sz, [_i] * 6),
    'swem_match_packed_f32': (_i, [_p] * 7 + [_i] * 6 + [_f, _i, _p, _sz]),
    'swem_match_packed_f32_planes': (_i, [_p] * 7 + [_i] * 6 + [_f, _i, _p, _sz, _p, _i, _p, _i, _p]),
    'swem_match_packed_clips_f32': (_i, [_p] * 7 + [_i] * 7 + [_f, _i, _p, _sz, _p, _i, _p, _i, _p]),
    # ---- include/swem_hip_train.h
    'swem_vos_loss_workspace': (_sz, [_i, _i, _ll]),
    'swem_vos_loss_frame_fwd_f32': (_i, [_p, _p, _p, _ll, _p, _p, _p, _p, _p, _i, _i, _ll, _ll, _p, _p, _sz]),
    'swem_vos_loss_reduce_f32': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _ll, _ll, _p, _f]),
    'swem_vos_loss_frame_bwd_f32': (_i, [_p, _p, _p, _p, _ll, _p, _p, _p, _p, _i, _i, _i, _ll, _ll, _p, _f, _p]),
    'swem_adamw_f32': (_i, [_p, _p, _p, _p, _p, _ll, _f, _f, _f, _f, _f, _i]),
    'swem_adamw_gated_f32': (_i, [_p, _p, _p, _p, _p, _ll, _f, _f, _f, _f, _f, _i, _p, _i, _p]),
    'swem_fault_flags_f32': (_i, [_p, _p, _p]),
    'swem_memorize_train_f32': (_i, [_p] * 11 + [_i] * 6 + [_f, _p, _sz]),
    'swem_nu_update_bwd_workspace': (_sz, [_i, _i, _i, _i]),
    'swem_nu_update_bwd_f32': (_i, [_p] * 7 + [_i] * 4 + [_p, _sz]),
    'swem_match_bwd_workspace': (_sz, [_i] * 6),
    'swem_match_bwd_f32': (_i, [_p] * 11 + [_i] * 6 + [_f, _p, _sz]),
    'swem_conv2d_wgrad_workspace': (_sz, [_i] * 11),
    'swem_conv2d_wgrad_f32': (_i, [_p, _p, _p, _i, _ll, _p, _i, _ll, _p, _i, _ll, _i, _i, _i, _i, _i, _i, _i, _i, _i,
                                   _p, _i, _i, _p, _sz]),
    'swem_conv2d_wgrad_bf16x3_workspace': (_sz, [_i] * 12),
    'swem_conv2d_wgrad_bf16x3': (_i, [_p, _p, _ll] + [_p, _i, _ll, _ll] * 3 + [_i] * 9 + [_p, _i, _i, _i, _p, _sz]),
    'swem_conv2d_wgrad_f16x3': (_i, [_p, _p, _ll] + [_p, _i, _ll, _ll] * 3 + [_i] * 8 + [_p, _p, _i, _i, _i, _p, _sz]),
    'swem_split_f16x2_scaled_f32': (_i, [_p, _p, _p, _ll, _i, _p, _i, _p]),
    'swem_vec_scale_f32': (_i, [_p, _p, _p, _p, _i]),
    'swem_pack_filters_f16x2_f32': (_i, [_p, _p, _p, _i, _i, _p, _p]),
    'swem_colsum_workspace': (_sz, [_ll, _i]),
    'swem_colsum_f32': (_i, [_p, _p, _p, _p, _p, _ll, _i, _i, _p, _sz]),
    'swem_sum_batch_f32': (_i, [_p, _p, _p, _i, _ll, _i]),
    'swem_expand_groups_f32': (_i, [_p, _p, _p, _i, _i, _ll]),
    'swem_sum_groups_f32': (_i, [_p, _p, _p, _i, _i, _ll]),
    'swem_bn_act_f32': (_i, [_p, _p, _p, _p, _p, _p, _ll, _i, _i, _p]),
    'swem_bn_act_planes_f32': (_i, [_p, _p, _p, _p, _p, _p, _ll, _i, _i, _p, _i, _p]),
    'swem_bn_act_bwd_amax_parts': (_i, [_ll, _i]),
    'swem_bn_act_bwd_amax_f32': (_i, [_p] * 11 + [_ll, _i, _i, _p, _p, _sz]),
    'swem_bn_act_bwd_workspace': (_sz, [_ll, _i]),
    'swem_bn_act_bwd_f32': (_i, [_p] * 11 + [_ll, _i, _i, _p, _p, _sz]),
    'swem_cbam_bwd_workspace': (_sz, [_i, _i, _i, _i]),
    'swem_cbam_bwd_f32': (_i, [_p] * 16 + [_i] * 5 + [_p, _sz]),
    'swem_bn_fold_f32': (_i, [_p, _p, _p, _p, _p, _f, _p, _p, _p, _i]),
    'swem_glu_f32': (_i, [_p, _p, _p, _p, _ll]),
    'swem_glu_bwd_f32': (_i, [_p, _p, _p, _p, _p, _p, _ll]),
    'swem_add_f32': (_i, [_p, _p, _p, _p, _ll]),
    'swem_maxpool3x3s2_bwd_f32': (_i, [_p, _p, _p, _p, _i, _i, _i, _i]),
    'swem_maxpool3x3s2_bwd_y_f32': (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i]),
    'swem_upsample_bwd_nhwc_f32': (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i]),
    'swem_resize_bilinear_bwd_f32': (_i, [_p, _p, _p, _i, _i, _i, _i, _i]),
    'swem_decode_head_bwd_f32': (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p, _sz]),
    'swem_pred_head_bwd_workspace': (_sz, [_i, _i, _i, _i]),
    'swem_pred_head_bwd_f32': (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _p, _sz]),
    'swem_prep_value_input_bwd_f32': (_i, [_p, _p, _p, _i, _i, _i, _i, _i]),
}

_lib = None


class SwemHipError(RuntimeError):
    pass


class SwemRangeError(SwemHipError):
    """SWEM_FAULT_RANGE (include/swem_hip.h): a value beyond the fp16 range went into an fp16 operand pair of the f16x3
    arithmetic.  Nothing else is wrong with the launches: the caller re-runs the work in a full-range arithmetic
    (ops.PlanBook.to_full_range; the evaluator loops do)."""


def load():
    """Load the shared library once; raise (never fall back) if it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SwemHipError('%s not found: build it with `python -m swem_amd.build`; '
                               'swem_amd has no CPU fallback' % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def call(name, *args):
    """Call an int-returning entry point and raise on a negative status."""
    lib = load()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        raise SwemHipError('%s failed (%d): %s' % (name, rc, lib.swem_last_error().decode()))


def query(name, *args):
    """Call a size/int query (no status convention)."""
    return getattr(load(), name)(*args)
