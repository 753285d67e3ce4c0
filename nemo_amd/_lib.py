"""ctypes binding of libnemofilt.so (C ABI declared in include/nemofilt.h).

The library is the product: there is no CPU fallback.  If it is missing, or no HIP device is usable,
anything that needs to compute raises :class:`NemoFiltError` with the reason.
"""
import ctypes
import os

_here = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_here, "lib", "libnemofilt.so")

NF_F32 = 0
NF_F64 = 1
IN_COMPLEX, IN_REAL64, IN_REAL32 = 0, 1, 2


class NemoFiltError(RuntimeError):
    pass


_lib = None

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_ll = ctypes.c_longlong
c_double = ctypes.c_double
c_size_t = ctypes.c_size_t

NF_MF_MAXFREQ = 8
NF_MF_MAXN = 6          # most frequency maps nf_matched_filter_scales / nf_build_filter* take
_PD8 = c_void_p * NF_MF_MAXFREQ
_D8 = c_double * NF_MF_MAXFREQ


class MfTile(ctypes.Structure):
    """nf_mf_tile (include/nemofilt.h): what the filter scales of one tile share."""
    _fields_ = [('H', c_int), ('W', c_int), ('N', c_int), ('prec', c_int), ('covType', c_int), ('outputUnits', c_int),
                ('fMaps_d', c_void_p), ('noiseCov_d', c_void_p), ('w', _D8),
                ('A_d', c_void_p), ('B_d', c_void_p), ('C_d', c_void_p), ('psMask_d', c_void_p), ('surveyMask_d', c_void_p),
                ('medWeights_d', c_void_p), ('wtype', c_int), ('ny', c_int), ('nx', c_int), ('yEdges_d', c_void_p),
                ('xEdges_d', c_void_p), ('overlap', c_int), ('numBins', c_int), ('estimator', c_int),
                ('quantiles', ctypes.POINTER(c_double)), ('trimSizePix', c_int), ('apodPix', c_int), ('y0', c_double),
                ('centreYX_d', c_void_p), ('centreClass', c_int), ('centreRowLo', c_int), ('centreRowHi', c_int), ('halfWin', c_int), ('winY_d', c_void_p), ('winX_d', c_void_p),
                ('axisWinY_d', c_void_p), ('axisWinX_d', c_void_p), ('noiseTablesLanes', ctypes.c_uint), ('scaleBatch', c_int)]


class MfScale(ctypes.Structure):
    """nf_mf_scale (include/nemofilt.h): one scale's template tables and output buffers."""
    _fields_ = [('r_d', _PD8), ('prof_d', _PD8), ('slope_d', _PD8),
                ('ntab', c_int * NF_MF_MAXFREQ), ('r0', _D8), ('dlnr', _D8), ('amp', _D8), ('sign', _D8), ('templateScale', _D8),
                ('filt_d', c_void_p), ('signalNorm_d', c_void_p), ('relWeights_d', c_void_p), ('data_d', c_void_p),
                ('sn_d', c_void_p), ('rms_d', c_void_p), ('mask_d', c_void_p), ('snChunkRange_d', c_void_p)]


# name -> (restype, argtypes); every symbol include/nemofilt.h declares
SIGNATURES = {
    'nf_version': (c_int, []),
    'nf_last_error': (ctypes.c_char_p, []),
    'nf_device_count': (c_int, []),
    'nf_plan_create': (c_int, [c_int, c_int, ctypes.POINTER(c_void_p)]),
    'nf_plan_destroy': (c_int, [c_void_p]),
    'nf_fft2_forward': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    'nf_fft2_inverse': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'nf_fft2_inverse_fused': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_int, c_int, c_int, c_void_p]),
    'nf_cross_power': (c_int, [c_int, c_void_p, c_void_p, c_int, c_ll, c_void_p]),
    'nf_fft2_forward_half': (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p]),
    'nf_fft2_inverse_half': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, ctypes.POINTER(c_double),
                                     c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    'nf_fft2_inverse_half_epi': (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                         c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'nf_symmetrise_filter': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'nf_cross_power_half': (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'nf_build_filter_half': (c_int, [c_int, c_int, c_void_p, c_void_p, ctypes.POINTER(c_double), c_void_p, c_int, c_int, c_int,
                                     c_void_p]),
    'nf_build_filter_sym_half': (c_int, [c_int, c_int, c_void_p, c_void_p, ctypes.POINTER(c_double), c_void_p, c_void_p, c_int, c_int,
                                         c_int, c_void_p]),
    'nf_gaussian_filter': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p,
                                   c_int, c_void_p]),
    'nf_build_filter': (c_int, [c_int, c_int, c_void_p, c_void_p, ctypes.POINTER(c_double), c_void_p, c_int, c_ll, c_void_p]),
    'nf_apply_filter': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_ll, c_void_p]),
    'nf_sum_planes': (c_int, [c_void_p, c_void_p, c_int, c_ll, c_void_p]),
    'nf_window_mul': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'nf_paint_radial2': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                 c_void_p, c_int, c_double, c_double, c_double, c_double, c_double, c_double, c_void_p]),
    'nf_paint_radial': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_double,
                                c_double, c_double, c_double, c_void_p]),
    'nf_scale_mask': (c_int, [c_int, c_void_p, c_void_p, c_double, c_void_p, c_ll, c_void_p]),
    'nf_scale_mask_dev': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_ll, c_void_p]),
    'nf_noise_workspace_bytes': (c_size_t, [c_int, c_int, c_int, c_int, c_int]),
    'nf_noise_map': (c_int, [c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                             c_int, c_int, ctypes.POINTER(c_double), c_int, c_void_p, c_size_t, c_int, c_void_p]),
    'nf_noise_smart_workspace_bytes': (c_size_t, []),
    'nf_noise_map_smart': (c_int, [c_int, c_void_p, c_void_p, c_int, c_void_p, c_ll, c_int, c_void_p, c_size_t, c_void_p]),
    'nf_median_stack': (c_int, [c_int, c_void_p, c_void_p, c_int, c_ll, c_void_p]),
    'nf_edge_check': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'nf_finalize': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                            c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'nf_sn_chunks': (c_int, [c_int, c_int]),
    'nf_sn_cached': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_ll, c_void_p]),
    'nf_finalize_cached': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_int, c_void_p, c_void_p, c_void_p]),
    'nf_load_tile': (c_int, [c_int, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                             c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'nf_inject_model': (c_int, [c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, c_void_p, c_ll, c_void_p]),
    'nf_restore_zeros': (c_int, [c_int, c_void_p, c_void_p, c_ll, c_void_p]),
    'nf_subtract': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_ll, c_void_p]),
    'nf_convolve2d': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    'nf_convolve2d_f64': (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    'nf_median_filter': (c_int, [c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    'nf_disc_fill': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p, ctypes.POINTER(c_int), c_void_p]),
    'nf_disc_apply': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, ctypes.POINTER(c_int),
                              c_int, c_double, c_int, c_void_p, c_void_p]),
    'nf_paint_catalog': (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, ctypes.POINTER(c_int), c_void_p, c_ll, c_void_p,
                                 c_int, c_int, c_int, c_void_p]),
    'nf_add_model': (c_int, [c_int, c_void_p, c_void_p, c_void_p, c_ll, c_int, c_void_p]),
    'nf_label_components': (c_int, [c_int, c_void_p, c_double, c_int, c_void_p, c_int, c_int, c_void_p,
                                    ctypes.POINTER(c_int), c_void_p]),
    'nf_object_stats': (c_int, [c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'nf_label_chunk_flags_offset': (ctypes.c_longlong, [c_int, c_int]),
    'nf_label_components_kept': (c_int, [c_int, c_void_p, c_double, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'nf_object_stats_chunks': (c_int, [c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'nf_label_and_stats': (c_int, [c_int, c_void_p, c_double, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'nf_label_and_stats_batch': (c_int, [c_int, c_int, c_void_p, c_double, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    'nf_object_shapes': (c_int, [c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    'nf_resample_bilinear': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    'nf_rms_table': (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int,
                             c_void_p, c_void_p]),
    'nf_xmatch_workspace_bytes': (c_size_t, [c_int]),
    'nf_xmatch_self': (c_int, [c_void_p, c_void_p, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t,
                               c_void_p]),
    'nf_xmatch_static_max_rows': (c_int, []),
    'nf_xmatch_self_static': (c_int, [c_void_p, c_void_p, c_int, c_double, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'nf_map_max': (c_int, [c_int, c_void_p, c_ll, c_void_p, c_void_p]),
    'nf_ring_flags': (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    'nf_profile_enable': (c_int, [c_int]),
    'nf_profile_select': (c_int, [ctypes.c_ulonglong]),
    'nf_profile_num_tags': (c_int, []),
    'nf_profile_tag_name': (ctypes.c_char_p, [c_int]),
    'nf_profile_read': (c_int, [c_int, ctypes.POINTER(c_ll), ctypes.POINTER(c_double), ctypes.POINTER(c_double)]),
    'nf_plan_digit_reversal': (c_int, [c_void_p, c_int, c_int, c_void_p, c_void_p]),
    'nf_axis_filter': (c_int, [c_void_p, c_int, c_int, c_void_p, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    'nf_spline_eval_batch': (c_int, [c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                     c_void_p]),
    'nf_spline_eval': (c_int, [c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_void_p, c_void_p]),
    'nf_signal_norm': (c_int, [c_int, c_void_p, c_double, c_void_p, c_void_p]),
    'nf_matched_filter_workspace_bytes': (c_size_t, [ctypes.POINTER(MfTile)]),
    'nf_matched_filter_max_lanes': (c_int, []),
    'nf_matched_filter_scales': (c_int, [c_void_p, ctypes.POINTER(MfTile), ctypes.POINTER(MfScale), c_int, ctypes.POINTER(c_void_p),
                                         c_int, c_void_p]),
}


def load():
    """Load libnemofilt.so (once) and declare every entry point."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NemoFiltError("HIP library %s is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(make -C nemo_amd/csrc).  nemo_amd has no CPU fallback." % LIB_PATH)
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise NemoFiltError("could not load %s: %s" % (LIB_PATH, e))
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().nf_last_error()
        raise NemoFiltError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
