"""Host-side mirror (Python) of the reference interface around the hot path.

Mirrors, with the same names / argument meaning / error behaviour:
  EncoderOptions            codec/core/options.js:11-164
  encode_aea_pcm            encodeAeaPcm   codec/io/processor.js:597-617
  decode_aea_pcm            decodeAeaPcm   codec/io/processor.js:628-654
  EncoderStream / DecoderStream   one encode()/decode() closure + its BufferPool
                            (codec/pipeline/encoder.js:438-450, decoder.js:408-411)
  serialize_frame / deserialize_frame   codec/io/serialization.js:41-176 (host-side format code)
All arithmetic of the hot path runs on the GPU through libcarta1_hip.so (carta1_amd/capi.py); there
is no CPU fallback.  The JavaScript host with the reference's exact API is carta1_amd/js.
"""
import ctypes as C
import math
import struct

import numpy as np

from . import capi

AEA_HEADER_SIZE = 2048          # codec/core/constants.js:12-16
AEA_MAGIC = bytes([0x00, 0x08, 0x00, 0x00])
AEA_TITLE_OFFSET, AEA_TITLE_SIZE = 4, 256
AEA_FRAME_COUNT_OFFSET, AEA_CHANNEL_COUNT_OFFSET = 260, 264
BFU_AMOUNTS = (20, 28, 32, 36, 40, 44, 48, 52)
SPECS_PER_BFU = (8, 8, 8, 8, 4, 4, 4, 4, 8, 8, 8, 8, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 6, 7, 7, 7, 7, 9, 9, 9, 9,
                 10, 10, 10, 10, 12, 12, 12, 12, 12, 12, 12, 12, 20, 20, 20, 20, 20, 20, 20, 20)
WORD_LENGTH_BITS = (0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16)


_BIASED = None


def packaged_biased_table(bias):
    """pow(SCALE_FACTORS, bias) as V8 computed it, for the biases carta1_amd/biased_tables.json holds; else None"""
    global _BIASED
    if _BIASED is None:
        import json
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'biased_tables.json')
        raw = json.load(open(path))['biased_scale_factors_f64']
        _BIASED = {float(k): [struct.unpack('>d', bytes.fromhex(h))[0] for h in v] for k, v in raw.items()}
    return _BIASED.get(float(bias))


class EncoderOptions:
    """codec/core/options.js: same keys, defaults, ranges and error messages."""
    _RANGES = {'transientThresholdLow': (0.01, 2), 'transientThresholdMid': (0.01, 3),
               'transientThresholdHigh': (0.01, 4), 'allocationBias': (0.0, 5.0)}

    def __init__(self, options=None, biased_table=None):
        self.values = {'transientThresholdLow': 1.0, 'transientThresholdMid': 1.5, 'transientThresholdHigh': 2.0,
                       'allocationBias': 1.0, 'fixedBlockModes': None}
        self.biased_table = biased_table   # optional explicit pow(SCALE_FACTORS, bias) table (64 doubles)
        for k, v in (options or {}).items():
            if k in self.values:
                self.set_value(k, v)

    def set_value(self, key, value):
        if key not in self.values:
            raise ValueError('Unknown option: %s' % key)
        if key in self._RANGES:
            lo, hi = self._RANGES[key]
            if value < lo or value > hi:
                raise ValueError('Value for %s must be between %s and %s, got %s' % (key, lo, hi, value))
        self.values[key] = value

    def __getattr__(self, name):
        vals = self.__dict__.get('values', {})
        if name in vals:
            return vals[name]
        raise AttributeError(name)

    def to_c(self):
        """c1_encode_options.  The biased table is pow(SCALE_FACTORS[i], bias) (bitallocation.js:46-61) as the
        reference's engine computes it; Math.pow is not correctly rounded and differs between engines, so
        the package carries the tables V8 produced for the biases 0, 0.25, 0.5, 1, 1.5, 2, 3.3 and 5
        (carta1_amd/biased_tables.json, generated from the golden vectors).  Any other bias falls back to libm's pow,
        whose last bit may differ from V8's ("parity unpinned" for those) -- pass biased_table to pin it; the JavaScript
        host always uses its own engine's Math.pow."""
        o = capi.EncodeOptions()
        capi.check(capi.load().c1_default_encode_options(C.byref(o)))
        bias = float(self.values['allocationBias'])
        table = self.biased_table if self.biased_table is not None else packaged_biased_table(bias)
        if table is not None:
            for i in range(64):
                o.biased_scale_factors[i] = float(table[i])
        elif bias != 1.0:
            sf = [o.biased_scale_factors[i] for i in range(64)]
            for i in range(64):
                o.biased_scale_factors[i] = math.pow(sf[i], bias)
        o.transient_threshold = float(self.values['transientThresholdLow'])   # encoder.js:137-141
        fm = self.values['fixedBlockModes']
        if fm is not None and len(fm) != 3:
            raise ValueError('fixedBlockModes must have 3 entries, got %d' % len(fm))
        for b in range(3):
            o.fixed_block_modes[b] = int(fm[b]) if fm is not None else -1
        return o


class Context:
    """One c1_ctx: a device, a stream, tables and workspace on it."""

    def __init__(self, device=0, stream=None):
        self._h = C.c_void_p()
        capi.check(capi.load().c1_ctx_create(int(device), C.c_void_p(stream) if stream else None, C.byref(self._h)))
        self.device = device

    def close(self):
        if self._h:
            capi.load().c1_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        capi.check(capi.load().c1_ctx_synchronize(self._h))

    def set_profiling(self, on):
        capi.check(capi.load().c1_ctx_set_profiling(self._h, 1 if on else 0))

    def kernel_ms(self, name):
        ms, n = C.c_double(0), C.c_int(0)
        capi.check(capi.load().c1_ctx_kernel_ms(self._h, name.encode(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def set_speculation(self, mode):
        """0 = exact kernels only, 1 = material-local (default: runs the predictor rejects go to the exact kernels),
        2 = always speculate (include/carta1_hip.h)."""
        capi.check(capi.load().c1_ctx_set_speculation(self._h, int(mode)))

    def set_decode_precision(self, binary32):
        """False (default): bit-identical to the reference; True: binary32 arithmetic, PCM within rounding noise."""
        capi.check(capi.load().c1_ctx_set_decode_precision(self._h, 1 if binary32 else 0))

    def speculation_stats(self, reset=False):
        """(units encoded through the speculative pass, units among them redone by the exact kernels)."""
        u, r = C.c_uint64(0), C.c_uint64(0)
        capi.check(capi.load().c1_ctx_speculation_stats(self._h, C.byref(u), C.byref(r), 1 if reset else 0))
        return u.value, r.value

    def speculation_deferred(self):
        """units whose runs the speculative analysis handed to the exact kernels (material-local mode)"""
        u = C.c_uint64(0)
        capi.check(capi.load().c1_ctx_speculation_deferred(self._h, C.byref(u)))
        return int(u.value)

    # ---- host-resident batches ------------------------------------------------------------------
    def quantization_stats(self):
        """(units whose exact coefficients were quantized in binary32 with the guard band, units packed again in binary64)"""
        u, r = C.c_uint64(0), C.c_uint64(0)
        capi.check(capi.load().c1_ctx_quantization_stats(self._h, C.byref(u), C.byref(r)))
        return int(u.value), int(r.value)

    def detection_stats(self):
        """(units whose block modes the speculative transient detector decided, units among them rechecked exactly)"""
        u, r = C.c_uint64(0), C.c_uint64(0)
        capi.check(capi.load().c1_ctx_detection_stats(self._h, C.byref(u), C.byref(r)))
        return int(u.value), int(r.value)

    def encode(self, channels, options=None, halo_frames=0, out=None):
        """channels: list of 1 or 2 float32 arrays, each (halo_frames + frames) * 512 samples.
        Returns uint8 [frames * nch, 212], units interleaved L,R.  `out`: optional preallocated result (when it
        and the channels come from pinned_empty() the batch is streamed over PCIe in overlapping chunks)."""
        opts = (options or EncoderOptions()).to_c()
        chans = [np.ascontiguousarray(c, dtype=np.float32) for c in channels]
        n = len(chans[0])
        if any(len(c) != n for c in chans) or n % 512:
            raise ValueError('channels must have equal length, a multiple of 512')
        frames = n // 512 - halo_frames
        if out is None:
            units = np.zeros((max(frames, 0) * len(chans), 212), dtype=np.uint8)
        else:
            units = out
            if units.dtype != np.uint8 or not units.flags['C_CONTIGUOUS'] or units.size != max(frames, 0) * len(chans) * 212:
                raise ValueError('out must be a contiguous uint8 array of frames * channels * 212 bytes')
            units = units.reshape(-1, 212)
        ptrs = capi.ptr_array([c.ctypes.data + halo_frames * 512 * 4 for c in chans])
        capi.check(capi.load().c1_encode_batch(self._h, ptrs, len(chans), frames, halo_frames, C.byref(opts),
                                               units.ctypes.data))
        return units

    def decode(self, units, channels, halo_units=0, out=None):
        """units: uint8 [(halo_units + frames) * channels, 212].  Returns a list of float32 arrays (`out`: optional
        preallocated list of them, see encode())."""
        u = np.ascontiguousarray(units, dtype=np.uint8).reshape(-1, 212)
        frames = u.shape[0] // channels - halo_units
        if out is None:
            outs = [np.zeros(max(frames, 0) * 512, dtype=np.float32) for _ in range(channels)]
        else:
            outs = list(out)
            if len(outs) != channels or any(o.dtype != np.float32 or not o.flags['C_CONTIGUOUS'] or o.size != max(frames, 0) * 512 for o in outs):
                raise ValueError('out must be one contiguous float32 array of frames * 512 samples per channel')
        ptrs = capi.ptr_array([o.ctypes.data for o in outs])
        capi.check(capi.load().c1_decode_batch(self._h, u.ctypes.data + halo_units * channels * 212, channels,
                                               frames, halo_units, ptrs))
        return outs

    # ---- device-resident (raw device pointers, e.g. torch tensor .data_ptr()) ---------------------
    def encode_device(self, pcm_ptrs, frames, units_ptr, options=None, halo_frames=0, c_options=None):
        opts = c_options if c_options is not None else (options or EncoderOptions()).to_c()
        capi.check(capi.load().c1_encode_device(self._h, capi.ptr_array(pcm_ptrs), len(pcm_ptrs), frames, halo_frames,
                                                C.byref(opts), C.c_void_p(units_ptr)))

    def decode_device(self, units_ptr, channels, frames, pcm_ptrs, halo_units=0):
        capi.check(capi.load().c1_decode_device(self._h, C.c_void_p(units_ptr), channels, frames, halo_units,
                                                capi.ptr_array(pcm_ptrs)))

    def generate_device(self, signal, seed, frames, pcm_ptr):
        capi.check(capi.load().c1_generate_device(self._h, signal, seed, frames, C.c_void_p(pcm_ptr)))

    def encode_wav(self, raw, bits, channels, options=None, out=None):
        """raw: the body of a WAV file (interleaved little-endian integer PCM, bits = 16, 24 or 32) as bytes or a
        uint8 array.  Returns uint8 [ceil(samples / 512) * channels, 212].  Conversion happens on the device."""
        opts = (options or EncoderOptions()).to_c()
        buf = np.frombuffer(raw, dtype=np.uint8) if not isinstance(raw, np.ndarray) else np.ascontiguousarray(raw).view(np.uint8).reshape(-1)
        bps = bits // 8
        if buf.size % (bps * channels):
            raise ValueError('raw length is not a whole number of %d-bit %d-channel samples' % (bits, channels))
        samples = buf.size // (bps * channels)
        frames = (samples + 511) // 512
        units = out if out is not None else np.zeros((frames * channels, 212), dtype=np.uint8)
        if units.dtype != np.uint8 or not units.flags['C_CONTIGUOUS'] or units.size != frames * channels * 212:
            raise ValueError('out must be a contiguous uint8 array of frames * channels * 212 bytes')
        capi.check(capi.load().c1_encode_wav_batch(self._h, buf.ctypes.data, bits, channels, samples, C.byref(opts), units.ctypes.data))
        return units.reshape(-1, 212)

    def decode_wav16(self, units, channels, out=None):
        """units: uint8 [frames * channels, 212].  Returns int16 [frames * 512, channels] (a 16-bit WAV body)."""
        u = np.ascontiguousarray(units, dtype=np.uint8).reshape(-1, 212)
        frames = u.shape[0] // channels
        pcm = out if out is not None else np.zeros((frames * 512, channels), dtype=np.int16)
        if pcm.dtype != np.int16 or not pcm.flags['C_CONTIGUOUS'] or pcm.size != frames * 512 * channels:
            raise ValueError('out must be a contiguous int16 array of frames * 512 * channels samples')
        capi.check(capi.load().c1_decode_wav16_batch(self._h, u.ctypes.data, channels, frames, pcm.ctypes.data))
        return pcm

    def pcm_from_int_device(self, src_ptr, bits, channels, samples, pcm_ptrs):
        capi.check(capi.load().c1_pcm_from_int_device(self._h, C.c_void_p(src_ptr), bits, channels, samples, capi.ptr_array(pcm_ptrs)))

    def pcm_to_int16_device(self, pcm_ptrs, samples, dst_ptr):
        capi.check(capi.load().c1_pcm_to_int16_device(self._h, capi.ptr_array(pcm_ptrs), len(pcm_ptrs), samples, C.c_void_p(dst_ptr)))

    def encode_stages_device(self, pcm_ptrs, frames, bands_ptr, coefs_ptr, side_ptr, alloc_ptr, options=None,
                             halo_frames=0):
        opts = (options or EncoderOptions()).to_c()
        capi.check(capi.load().c1_encode_stages_device(
            self._h, capi.ptr_array(pcm_ptrs), len(pcm_ptrs), frames, halo_frames, C.byref(opts),
            C.c_void_p(bands_ptr), C.c_void_p(coefs_ptr), C.c_void_p(side_ptr), C.c_void_p(alloc_ptr)))

    def detect_stages_device(self, pcm_ptrs, frames, mags_ptr, modes_ptr, options=None, halo_frames=0):
        """The transient detector's magnitude spectra (256 floats per unit) and chosen block modes (1 byte per unit)."""
        opts = (options or EncoderOptions()).to_c()
        capi.check(capi.load().c1_detect_stages_device(
            self._h, capi.ptr_array(pcm_ptrs), len(pcm_ptrs), frames, halo_frames, C.byref(opts),
            C.c_void_p(mags_ptr), C.c_void_p(modes_ptr)))

    def detect_scores_device(self, pcm_ptrs, frames, scores_ptr, modes_ptr, open_ptr, options=None, halo_frames=0, speculative=True):
        """Per unit and band {lo, hi}: the speculative detector's interval for the transient score, or the reference's
        score twice; the block modes after the exact recheck; how many units the interval left open."""
        opts = (options or EncoderOptions()).to_c()
        capi.check(capi.load().c1_detect_scores_device(
            self._h, capi.ptr_array(pcm_ptrs), len(pcm_ptrs), frames, halo_frames, C.byref(opts), 1 if speculative else 0,
            C.c_void_p(scores_ptr), C.c_void_p(modes_ptr), C.c_void_p(open_ptr)))

    def detect_spec_mags_device(self, pcm_ptrs, frames, mags_ptr, bounds_ptr, halo_frames=0):
        """The speculative detector's binary32 magnitude spectra (256 floats per unit) and the bound per band (3 floats per unit)."""
        capi.check(capi.load().c1_detect_spec_mags_device(
            self._h, capi.ptr_array(pcm_ptrs), len(pcm_ptrs), frames, halo_frames, C.c_void_p(mags_ptr), C.c_void_p(bounds_ptr)))

    def log2f_error(self, first_bits, count):
        """(max relative error in units of 2^-24, max absolute error near 1) of the device's binary32 log2 over a range of bit patterns"""
        out = (C.c_double * 2)()
        capi.check(capi.load().c1_log2f_error_device(self._h, first_bits, count, out))
        return float(out[0]), float(out[1])

    def libm_device(self, fn, in_ptr, out_ptr, n):
        """Math.log / exp / log1p / log10 (fn 0..3) as the detector's kernels evaluate them, on n device doubles."""
        capi.check(capi.load().c1_libm_device(self._h, fn, C.c_void_p(in_ptr), C.c_void_p(out_ptr), n))

    def alloc_bounds_device(self, side_ptr, units, out_ptr, options=None):
        """Totals of the eight candidate BFU counts and the lower bounds the allocation prunes with (16 doubles per unit)."""
        opts = (options or EncoderOptions()).to_c()
        capi.check(capi.load().c1_alloc_bounds_device(self._h, C.c_void_p(side_ptr), units, C.byref(opts), C.c_void_p(out_ptr)))

    def spec_stages_device(self, pcm_ptrs, frames, coefs_ptr, eps_ptr, side_ptr, options=None, halo_frames=0):
        """The speculative binary32 analysis alone: coefficients, their proven error bounds, scale-factor indices."""
        opts = (options or EncoderOptions({'fixedBlockModes': [0, 0, 0]})).to_c()
        capi.check(capi.load().c1_spec_stages_device(
            self._h, capi.ptr_array(pcm_ptrs), len(pcm_ptrs), frames, halo_frames, C.byref(opts),
            C.c_void_p(coefs_ptr), C.c_void_p(eps_ptr), C.c_void_p(side_ptr)))

    # ---- the single-stage functions the reference exports (codec/index.js:30-35,42), host arrays ----------------
    def quantize(self, coefficients, scale_factor_index, bits_per_sample):
        """quantize, codec/coding/quantization.js:34-56 -> int32 array"""
        x = np.ascontiguousarray(coefficients, dtype=np.float32)
        out = np.zeros(x.size, dtype=np.int32)
        capi.check(capi.load().c1_quantize(self._h, x.ctypes.data, x.size, int(scale_factor_index), int(bits_per_sample), out.ctypes.data))
        return out

    def dequantize(self, quantized, scale_factor_index, bits_per_sample):
        """dequantize, quantization.js:65-78 -> float32 array"""
        q = np.ascontiguousarray(quantized, dtype=np.int32)
        out = np.zeros(q.size, dtype=np.float32)
        capi.check(capi.load().c1_dequantize(self._h, q.ctypes.data, q.size, int(scale_factor_index), int(bits_per_sample), out.ctypes.data))
        return out

    def fft(self, real, imag, w):
        """FFT.fft, codec/transforms/fft.js:14-68, in place on two contiguous float32 arrays; w: (cos, sin)(-2 pi / stride) for
        stride = 2, 4, .., n as the reference's engine computes them (float64, log2(n) pairs)"""
        if real.dtype != np.float32 or imag.dtype != np.float32 or not real.flags['C_CONTIGUOUS'] or not imag.flags['C_CONTIGUOUS'] or real.size != imag.size:
            raise ValueError('fft works in place on two contiguous float32 arrays of equal length')
        w = np.ascontiguousarray(w, dtype=np.float64)
        capi.check(capi.load().c1_fft(self._h, real.ctypes.data, imag.ctypes.data, real.size, w.ctypes.data))

    def qmf_analysis(self, pcm, halo_frames=0):
        """qmfAnalysisStage, codec/pipeline/encoder.js:57-96: pcm = (halo_frames + frames) * 512 samples of one channel
        -> float32 [frames, 512] (low128 | mid128 | high256)"""
        x = np.ascontiguousarray(pcm, dtype=np.float32)
        frames = x.size // 512 - halo_frames
        out = np.zeros((max(frames, 0), 512), dtype=np.float32)
        capi.check(capi.load().c1_qmf_analysis_batch(self._h, x.ctypes.data, frames, halo_frames, out.ctypes.data))
        return out

    def mdct(self, bands, block_modes, halo_frames=0):
        """mdctStage, encoder.js:170-349: bands float32 [(halo_frames + frames), 512], block_modes int [frames, 3]
        -> (coefficients [frames, 512], the bands as the reference leaves them [frames, 512])"""
        b = np.ascontiguousarray(bands, dtype=np.float32).reshape(-1, 512)
        frames = b.shape[0] - halo_frames
        m = np.ascontiguousarray(block_modes, dtype=np.int32).reshape(-1)
        if m.size != 3 * frames:
            raise ValueError('block_modes must hold three entries per frame')
        co = np.zeros((frames, 512), dtype=np.float32)
        bw = np.zeros((frames, 512), dtype=np.float32)
        capi.check(capi.load().c1_mdct_batch(self._h, b.ctypes.data, frames, halo_frames, m.ctypes.data, co.ctypes.data, bw.ctypes.data))
        return co, bw

    def pack_spec_tap_device(self, coefs_ptr, eps_ptr, side_ptr, alloc_ptr, units, units_out_ptr, lists_ptr, all_long=True):
        """Test tap: the speculative quantizer + packer on caller-supplied coefficients, bounds and records (device pointers)."""
        capi.check(capi.load().c1_pack_spec_tap_device(
            self._h, C.c_void_p(coefs_ptr), C.c_void_p(eps_ptr), C.c_void_p(side_ptr), C.c_void_p(alloc_ptr), units,
            1 if all_long else 0, C.c_void_p(units_out_ptr), C.c_void_p(lists_ptr)))


def encode_multi(channels, options=None, devices=(0,), out=None):
    """c1_encode_batch_multi: the batch sharded over `devices` (contiguous frame ranges, one host thread and context
    per entry, no collective).  channels: a stream from its start.  Same bytes as one device produces.
    out: a C-contiguous uint8 array of frames * channels * 212 bytes to write into (as Context.encode takes)."""
    opts = (options or EncoderOptions()).to_c()
    chans = [np.ascontiguousarray(c, dtype=np.float32) for c in channels]
    n = len(chans[0])
    if any(len(c) != n for c in chans) or n % 512:
        raise ValueError('channels must have equal length, a multiple of 512')
    frames = n // 512
    if out is None:
        units = np.zeros((frames * len(chans), 212), dtype=np.uint8)
    else:
        units = out
        if units.dtype != np.uint8 or not units.flags.c_contiguous or units.size != frames * len(chans) * 212:
            raise ValueError('out must be a C-contiguous uint8 array of frames * channels * 212 bytes')
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    capi.check(capi.load().c1_encode_batch_multi(devs, len(devices), capi.ptr_array([c.ctypes.data for c in chans]),
                                                 len(chans), frames, 0, C.byref(opts), units.ctypes.data))
    return units


def decode_multi(units, channels, devices=(0,), out=None):
    """c1_decode_batch_multi: units of a stream from its start -> list of float32 arrays (out: such a list to write into)."""
    u = np.ascontiguousarray(units, dtype=np.uint8).reshape(-1, 212)
    frames = u.shape[0] // channels
    if out is None:
        outs = [np.zeros(frames * 512, dtype=np.float32) for _ in range(channels)]
    else:
        outs = list(out)
        if len(outs) != channels or any(o.dtype != np.float32 or not o.flags.c_contiguous or o.size != frames * 512 for o in outs):
            raise ValueError('out must be one C-contiguous float32 array of frames * 512 samples per channel')
    devs = (C.c_int * len(devices))(*[int(d) for d in devices])
    capi.check(capi.load().c1_decode_batch_multi(devs, len(devices), u.ctypes.data, channels, frames, 0,
                                                 capi.ptr_array([o.ctypes.data for o in outs])))
    return outs


class EncoderStream:
    """What one encode() closure per channel + BufferPool is in the reference: push frames, get units,
    state carried on the device between calls."""

    def __init__(self, ctx, channels=1, options=None):
        self._ctx, self.channels = ctx, channels
        self._h = C.c_void_p()
        opts = (options or EncoderOptions()).to_c()
        capi.check(capi.load().c1_enc_stream_create(ctx._h, channels, C.byref(opts), C.byref(self._h)))

    def push(self, channels):
        chans = [np.ascontiguousarray(c, dtype=np.float32) for c in channels]
        if len(chans) != self.channels:
            raise ValueError('expected %d channels' % self.channels)
        frames = len(chans[0]) // 512
        units = np.zeros((frames * self.channels, 212), dtype=np.uint8)
        capi.check(capi.load().c1_enc_stream_push(self._h, capi.ptr_array([c.ctypes.data for c in chans]), frames,
                                                  units.ctypes.data))
        return units

    def close(self):
        if self._h:
            capi.load().c1_enc_stream_destroy(self._h)
            self._h = C.c_void_p()


class DecoderStream:
    def __init__(self, ctx, channels=1):
        self._ctx, self.channels = ctx, channels
        self._h = C.c_void_p()
        capi.check(capi.load().c1_dec_stream_create(ctx._h, channels, C.byref(self._h)))

    def push(self, units):
        u = np.ascontiguousarray(units, dtype=np.uint8).reshape(-1, 212)
        frames = u.shape[0] // self.channels
        outs = [np.zeros(frames * 512, dtype=np.float32) for _ in range(self.channels)]
        capi.check(capi.load().c1_dec_stream_push(self._h, u.ctypes.data, frames,
                                                  capi.ptr_array([o.ctypes.data for o in outs])))
        return outs

    def close(self):
        if self._h:
            capi.load().c1_dec_stream_destroy(self._h)
            self._h = C.c_void_p()


# ---- AEA container: codec/io/serialization.js:182-254 -------------------------------------------------
def aea_header(title='', frame_count=0, channel_count=1):
    h = bytearray(AEA_HEADER_SIZE)
    h[0:4] = AEA_MAGIC
    t = title.encode('utf-8')[:AEA_TITLE_SIZE - 1]
    h[AEA_TITLE_OFFSET:AEA_TITLE_OFFSET + len(t)] = t
    struct.pack_into('<I', h, AEA_FRAME_COUNT_OFFSET, frame_count)
    h[AEA_CHANNEL_COUNT_OFFSET] = channel_count
    return bytes(h)


def parse_aea_header(header):
    if len(header) != AEA_HEADER_SIZE:
        raise ValueError('Header must be %d bytes' % AEA_HEADER_SIZE)
    if bytes(header[0:4]) != AEA_MAGIC:
        raise ValueError('Invalid AEA file')
    end = bytes(header).find(b'\x00', AEA_TITLE_OFFSET)
    n = AEA_TITLE_SIZE if end < 0 else end - AEA_TITLE_OFFSET
    return {'title': bytes(header[AEA_TITLE_OFFSET:AEA_TITLE_OFFSET + n]).decode('utf-8', 'replace'),
            'frameCount': struct.unpack_from('<I', header, AEA_FRAME_COUNT_OFFSET)[0],
            'channelCount': header[AEA_CHANNEL_COUNT_OFFSET]}


def pinned_empty(shape, dtype=np.float32):
    """numpy array in page-locked host memory (c1_host_alloc).  Batch calls whose host buffers all live in such
    arrays stream the batch over PCIe in overlapping chunks (include/carta1_hip.h).  Freed with the array."""
    import weakref
    dtype = np.dtype(dtype)
    n = int(np.prod(shape))
    ptr = C.c_void_p()
    capi.check(capi.load().c1_host_alloc(max(1, n * dtype.itemsize), C.byref(ptr)))
    buf = (C.c_uint8 * (n * dtype.itemsize)).from_address(ptr.value)
    arr = np.frombuffer(buf, dtype=dtype, count=n).reshape(shape)
    weakref.finalize(buf, capi.load().c1_host_free, ptr)
    return arr


_default_ctx = None


def _ctx(ctx):
    global _default_ctx
    if ctx is not None:
        return ctx
    if _default_ctx is None:
        _default_ctx = Context(0)
    return _default_ctx


def _check_channels(channels):
    if (not isinstance(channels, (list, tuple)) or len(channels) not in (1, 2)
            or any(not (isinstance(c, np.ndarray) and c.dtype == np.float32) for c in channels)):
        raise TypeError('ATRAC1 encoding requires one or two Float32 channels')   # processor.js:603


def encode_pcm(channels, options=None, ctx=None):
    """Planar PCM of any length -> units; the final partial frame is zero padded and a shorter channel
    is padded to the longer one (frameBufferToFrames, processor.js:246-279)."""
    _check_channels(channels)
    n = max(len(c) for c in channels)
    frames = (n + 511) // 512
    padded = []
    for c in channels:
        p = np.zeros(frames * 512, dtype=np.float32)
        p[:len(c)] = c
        padded.append(p)
    return _ctx(ctx).encode(padded, options)


def decode_units(units, channels, ctx=None):
    return _ctx(ctx).decode(units, channels)


def encode_aea_pcm(channels, options=None, ctx=None):
    """encodeAeaPcm (processor.js:597-617): 2048-byte header + 212 bytes per unit, L/R interleaved;
    header frameCount counts sound units over both channels (processor.js:320-325)."""
    _check_channels(channels)
    options = dict(options or {})
    title = options.pop('title', 'encoded by carta1')
    units = encode_pcm(channels, EncoderOptions(options), ctx)
    return aea_header(title, units.shape[0], len(channels)) + units.tobytes()


def decode_aea_pcm(data, ctx=None):
    """decodeAeaPcm (processor.js:628-654): bytes / bytearray / ndarray(uint8) -> list of float32 arrays."""
    if isinstance(data, np.ndarray):
        data = data.tobytes()
    if not isinstance(data, (bytes, bytearray, memoryview)):
        raise TypeError('ATRAC1 decoding requires AEA bytes or a Blob')
    data = bytes(data)
    info = parse_aea_header(data[:AEA_HEADER_SIZE])
    body = data[AEA_HEADER_SIZE:]
    n_units = len(body) // 212            # a trailing partial unit is dropped (processor.js:516-521)
    nch = info['channelCount']
    units = np.frombuffer(body[:n_units * 212], dtype=np.uint8).reshape(-1, 212)
    if nch == 2 and n_units % 2:          # trailing lone L unit gets an all-zero partner (processor.js:222-232)
        dummy = np.zeros((1, 212), dtype=np.uint8)
        dummy[0, 0], dummy[0, 1] = 0xAC, 0x00   # modes 0,0,0; 20 BFUs; every word length 0 == _createDummyFrame
        units = np.concatenate([units, dummy])
    if nch not in (1, 2):
        raise ValueError('Unsupported channel count: %d' % nch)
    return _ctx(ctx).decode(units, nch)


# ---- sound unit <-> fields: codec/io/serialization.js:41-176 (format code, host side) ----------------
def deserialize_frame(unit):
    u = bytes(unit)
    if len(u) != 212:
        raise ValueError('Frame must be 212 bytes')
    bits = int.from_bytes(u, 'big')
    total = 212 * 8

    def get(pos, n):
        avail = max(0, min(n, total - pos))
        return (bits >> (total - pos - avail)) & ((1 << avail) - 1) if avail else 0
    header = get(0, 16)
    modes = [2 - ((header >> 14) & 3), 2 - ((header >> 12) & 3), 3 - ((header >> 10) & 3)]
    n = BFU_AMOUNTS[(header >> 5) & 7]
    wl = [get(16 + 4 * i, 4) for i in range(n)]
    sfi = [get(16 + 4 * n + 6 * i, 6) for i in range(n)]
    pos = 16 + 10 * n
    q = []
    for i in range(n):
        b = WORD_LENGTH_BITS[wl[i]]
        vals = []
        for _ in range(SPECS_PER_BFU[i]):
            v = 0
            if b:
                v = get(pos, b)
                pos += b
                if v >= 1 << (b - 1):
                    v -= 1 << b
            vals.append(v)
        q.append(vals)
    return {'nBfu': n, 'blockModes': modes, 'wordLengthIndices': wl, 'scaleFactorIndices': sfi,
            'quantizedCoefficients': q}


def serialize_frame(f):
    n = f['nBfu']
    header = ((2 - f['blockModes'][0]) << 14) | ((2 - f['blockModes'][1]) << 12) | ((3 - f['blockModes'][2]) << 10) \
        | (BFU_AMOUNTS.index(n) << 5)
    acc, nbits = header & 0xffff, 16
    for i in range(n):
        acc, nbits = (acc << 4) | (f['wordLengthIndices'][i] & 15), nbits + 4
    for i in range(n):
        acc, nbits = (acc << 6) | (f['scaleFactorIndices'][i] & 63), nbits + 6
    for i in range(n):
        b = WORD_LENGTH_BITS[f['wordLengthIndices'][i]]
        if b:
            for v in f['quantizedCoefficients'][i]:
                acc, nbits = (acc << b) | (v & ((1 << b) - 1)), nbits + b
    acc <<= 212 * 8 - nbits
    return acc.to_bytes(212, 'big')
