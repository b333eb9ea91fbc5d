"""ctypes binding of libcarta1_hip.so -- exactly the symbols include/carta1_hip.h declares.

There is no CPU path behind this module: if the library is missing or no HIP device is usable,
calls raise Carta1Error.  PyTorch is optional here (device tensors are passed as raw pointers).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# C1_LIB: another build of the same library (A/B timing of kernel variants inside one GPU session)
LIB_PATH = os.environ.get('C1_LIB') or os.path.join(HERE, 'lib', 'libcarta1_hip.so')

FRAME = 512
UNIT_BYTES = 212
SIGNAL_WHITE = 0
SIGNAL_PINK_BURSTS = 1
SIGNAL_MIXED = 2
SIGNAL_PARTIALS = 3


class Carta1Error(RuntimeError):
    def __init__(self, code, message):
        super().__init__('carta1_hip error %d: %s' % (code, message))
        self.code = code


class Tables(C.Structure):
    _fields_ = [('scale_factors', C.c_double * 64), ('window_short', C.c_double * 32),
                ('mdct_fwd64', C.c_double * 32), ('mdct_fwd256', C.c_double * 128),
                ('mdct_fwd512', C.c_double * 256), ('mdct_inv64', C.c_double * 32),
                ('mdct_inv256', C.c_double * 128), ('mdct_inv512', C.c_double * 256),
                ('fft_w', (C.c_double * 2) * 8), ('log1p_10', C.c_double)]


class EncodeOptions(C.Structure):
    _fields_ = [('biased_scale_factors', C.c_double * 64), ('transient_threshold', C.c_double),
                ('fixed_block_modes', C.c_int32 * 3), ('reserved', C.c_int32)]


# every exported symbol with its signature; tests/test_abi.py checks this list against the header
SIGNATURES = {
    'c1_abi_version': (C.c_int, []),
    'c1_last_error': (C.c_char_p, []),
    'c1_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'c1_get_default_tables': (C.c_int, [C.POINTER(Tables)]),
    'c1_set_tables': (C.c_int, [C.POINTER(Tables)]),
    'c1_default_encode_options': (C.c_int, [C.POINTER(EncodeOptions)]),
    'c1_ctx_create': (C.c_int, [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]),
    'c1_ctx_destroy': (C.c_int, [C.c_void_p]),
    'c1_ctx_synchronize': (C.c_int, [C.c_void_p]),
    'c1_ctx_set_profiling': (C.c_int, [C.c_void_p, C.c_int]),
    'c1_ctx_kernel_ms': (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    'c1_ctx_set_speculation': (C.c_int, [C.c_void_p, C.c_int]),
    'c1_ctx_set_decode_precision': (C.c_int, [C.c_void_p, C.c_int]),
    'c1_ctx_speculation_stats': (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.c_int]),
    'c1_ctx_speculation_deferred': (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    'c1_ctx_quantization_stats': (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    'c1_encode_device': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int,
                                   C.POINTER(EncodeOptions), C.c_void_p]),
    'c1_encode_batch': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int,
                                  C.POINTER(EncodeOptions), C.c_void_p]),
    'c1_encode_batch_multi': (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int,
                                        C.POINTER(EncodeOptions), C.c_void_p]),
    'c1_decode_batch_multi': (C.c_int, [C.POINTER(C.c_int), C.c_int, C.c_void_p, C.c_int, C.c_int64, C.c_int,
                                        C.POINTER(C.c_void_p)]),
    'c1_decode_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]),
    'c1_decode_batch': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]),
    'c1_enc_stream_create': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(EncodeOptions), C.POINTER(C.c_void_p)]),
    'c1_enc_stream_push': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int64, C.c_void_p]),
    'c1_enc_stream_destroy': (C.c_int, [C.c_void_p]),
    'c1_dec_stream_create': (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    'c1_dec_stream_push': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]),
    'c1_dec_stream_destroy': (C.c_int, [C.c_void_p]),
    'c1_generate_device': (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_int64, C.c_void_p]),
    'c1_pcm_from_int_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.POINTER(C.c_void_p)]),
    'c1_pcm_to_int16_device': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_void_p]),
    'c1_host_alloc': (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    'c1_host_free': (C.c_int, [C.c_void_p]),
    'c1_table_fast_paths': (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    'c1_encode_wav_batch': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.POINTER(EncodeOptions), C.c_void_p]),
    'c1_decode_wav16_batch': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p]),
    'c1_aea_header': (C.c_int, [C.c_char_p, C.c_uint32, C.c_int, C.c_void_p]),
    'c1_encode_stages_device': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int,
                                          C.POINTER(EncodeOptions), C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    'c1_detect_stages_device': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int,
                                          C.POINTER(EncodeOptions), C.c_void_p, C.c_void_p]),
    'c1_libm_device': (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int64]),
    'c1_detect_scores_device': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int,
                                          C.POINTER(EncodeOptions), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'c1_detect_spec_mags_device': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int, C.c_void_p, C.c_void_p]),
    'c1_log2f_error_device': (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_double)]),
    'c1_ctx_detection_stats': (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    'c1_alloc_bounds_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    'c1_spec_stages_device': (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_int,
                                        C.POINTER(EncodeOptions), C.c_void_p, C.c_void_p, C.c_void_p]),
    'c1_quantize': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'c1_dequantize': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    'c1_fft': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    'c1_qmf_analysis_batch': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    'c1_mdct_batch': (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    'c1_pack_spec_tap_device': (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                                          C.c_void_p, C.c_void_p]),
}

_lib = None


def load():
    """Load the shared library (no GPU needed for this step) and attach signatures."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Carta1Error(-1, 'libcarta1_hip.so is not built (%s); run `python -m carta1_amd.build` -- '
                                  'there is no CPU fallback' % LIB_PATH)
        # PyTorch-ROCm bundles its own libamdhip64 (same soname as /opt/rocm's).  Two HIP runtimes in one
        # process do not share the device, so when torch is installed let it load its runtime first; the
        # dynamic linker then binds this library to that same copy.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def check(rc):
    if rc != 0:
        raise Carta1Error(rc, load().c1_last_error().decode('utf-8', 'replace'))


def device_count():
    n = C.c_int(0)
    check(load().c1_device_count(C.byref(n)))
    return n.value


def default_tables():
    t = Tables()
    check(load().c1_get_default_tables(C.byref(t)))
    return t


def ptr_array(ptrs):
    return (C.c_void_p * len(ptrs))(*[C.c_void_p(int(p)) for p in ptrs])
