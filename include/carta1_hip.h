/*
 * carta1_hip.h -- C ABI of libcarta1_hip.so: the MI355X (gfx950) ATRAC1 hot path.
 *
 * This is the drop-in boundary for aynik/carta1's encode/decode hot path.  The
 * reference has no FFI of its own; the seam is the module boundary between
 * codec/pipeline/ and codec/io/processor.js (stay JavaScript) and
 * codec/transforms/{qmf,mdct,fft}.js, codec/analysis/transient.js,
 * codec/coding/{bitallocation,quantization}.js (replaced by HIP kernels).  Each
 * entry point below names the reference code it stands in for (file:line in
 * aynik/carta1 v1.1.10).  The N-API addon (carta1_amd/js/addon/) and the ctypes
 * binding (carta1_amd/capi.py) bind exactly these symbols; INTEGRATION.md shows
 * the reference-side patch.
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on error; c1_last_error()
 *    (thread-local) then holds the message.  There is NO CPU fallback: without a
 *    usable HIP device every compute entry point fails with C1_ERR_NO_DEVICE.
 *  - PCM is planar float32, one pointer per channel, 512 samples per frame.
 *  - sound units are 212 bytes each, interleaved L,R,L,R,... for stereo
 *    (codec/io/processor.js:125-130), unit index = frame * channels + channel.
 *  - "history": a frame's unit depends on at most the 650 PCM samples before it
 *    (266 with fixed block modes).  Batch calls take `halo_frames`
 *    = how many whole frames (0..2) of real PCM sit in memory directly BEFORE the
 *    pcm pointers; samples before that are taken as zero, which is exactly a
 *    stream start (codec/core/buffers.js:30-42 zero-initialised BufferPool).
 *  - decoded PCM of frame n depends on units n and n-1 only; decode calls take
 *    `halo_units` (0 or 1) = whether the unit(s) of frame -1 precede the pointer.
 */
#ifndef CARTA1_HIP_H
#define CARTA1_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C1_FRAME_SAMPLES 512
#define C1_UNIT_BYTES 212
#define C1_MAX_CHANNELS 2
#define C1_ABI_VERSION 3

enum {
  C1_OK = 0,
  C1_ERR_ARG = 1,        /* bad argument (message says which) */
  C1_ERR_NO_DEVICE = 2,  /* no HIP device / runtime: the product path never falls back to CPU */
  C1_ERR_HIP = 3,        /* a HIP call failed */
  C1_ERR_STATE = 4       /* handle used in the wrong state */
};

/* Numeric tables the reference builds at module load with V8's Math.sin/cos/pow/sqrt
 * (codec/core/constants.js:60-66,144-150; codec/transforms/mdct.js:27-36;
 * codec/transforms/fft.js:37-39).  libm differs from V8 in the last bit on some entries,
 * so the library never recomputes them: it carries the values the reference produced
 * (c1_get_default_tables) and a JavaScript host may install the ones its own V8 computes
 * (c1_set_tables) so results track the reference running in that same process. */
typedef struct c1_tables {
  double scale_factors[64];   /* SCALE_FACTORS            constants.js:144-150 */
  double window_short[32];    /* WINDOW_SHORT             constants.js:60-66   */
  double mdct_fwd64[32];      /* mdct64.sinCosTable       mdct.js:215          */
  double mdct_fwd256[128];    /* mdct256.sinCosTable      mdct.js:216          */
  double mdct_fwd512[256];    /* mdct512.sinCosTable      mdct.js:217          */
  double mdct_inv64[32];      /* imdct64.sinCosTable      mdct.js:219          */
  double mdct_inv256[128];    /* imdct256.sinCosTable     mdct.js:220          */
  double mdct_inv512[256];    /* imdct512.sinCosTable     mdct.js:221          */
  double fft_w[8][2];         /* (cos,sin)(-2*pi/stride), stride = 2..256; fft.js:37-39 */
  double log1p_10;            /* Math.log1p(10)           transient.js:211     */
} c1_tables;

/* EncoderOptions as the hot path consumes them (codec/core/options.js:17-23). */
typedef struct c1_encode_options {
  double biased_scale_factors[64]; /* pow(SCALE_FACTORS[i], allocationBias), bitallocation.js:46-61;
                                      computed by the HOST (its Math.pow), bias==1 -> SCALE_FACTORS */
  double transient_threshold;      /* options.transientThresholdLow: the pipeline uses the LOW
                                      threshold for all three bands (encoder.js:137-141) */
  int32_t fixed_block_modes[3];    /* options.fixedBlockModes, or {-1,-1,-1} for transient detection */
  int32_t reserved;
} c1_encode_options;

typedef struct c1_ctx c1_ctx; /* one per (device, stream): workspace + tables on that device */

/* ---- library ------------------------------------------------------------------------- */
int c1_abi_version(void);
const char *c1_last_error(void);
int c1_device_count(int *count);                 /* hipGetDeviceCount */
int c1_get_default_tables(c1_tables *out);       /* the V8-produced defaults compiled into the library */
int c1_set_tables(const c1_tables *tables);      /* NULL restores defaults; applies to contexts created afterwards */
/* fills biased_scale_factors for allocationBias == 1 (exact copy, bitallocation.js:51-52) and sets
 * threshold 1.0 / detection on: the EncoderOptions defaults (options.js:17-23) */
int c1_default_encode_options(c1_encode_options *out);
/* Diagnostics (host only, no device needed): which table-dependent shortcuts the kernels will take with the
 * tables currently installed.  Both are verified on the host against the plain formulation for the whole input
 * domain when tables are installed, and the kernels fall back to it when a check fails:
 *  scale_factor_bits  findScaleFactor (bitallocation.js:290-299) from the binary32 bit pattern
 *  dequant_reciprocal dequantize's (q * SF) / range (quantization.js:65-78): 1 = as multiply + two FMAs with RN(1 / range);
 *                     2 = moreover, after the store to the Float32 array, equal to q * RN(SF * RN(1 / range)) for every
 *                     (word length, scale factor, q): one product per BFU and one per coefficient */
int c1_table_fast_paths(int *scale_factor_bits, int *dequant_reciprocal);

/* ---- contexts ------------------------------------------------------------------------- */
int c1_ctx_create(int device, void *hip_stream /* hipStream_t or NULL = own stream */, c1_ctx **out);
int c1_ctx_destroy(c1_ctx *ctx);
int c1_ctx_synchronize(c1_ctx *ctx);
/* milliseconds the device spent in the named kernel during the most recent *_device call on this
 * context ("analysis", "allocate", "pack", "decode", "redo", or "total"), from HIP events on the context's
 * stream; c1_ctx_set_profiling(ctx, 1) must have been set before the call */
int c1_ctx_set_profiling(c1_ctx *ctx, int enabled);
int c1_ctx_kernel_ms(c1_ctx *ctx, const char *name, double *ms, int *launches);

/* Speculative encoding of fixed-block-mode streams.  The encoder's outputs are integers; they are
 * decisions taken on the MDCT coefficients (bitallocation.js:290-299, quantization.js:43-53).  For fixedBlockModes
 * [0,0,0] the library first computes the coefficients in binary32 together with a proven bound on their distance
 * from the reference's binary64-then-rounded values, accepts every sound unit whose decisions are the same for all
 * values within the bound, and re-encodes the others with the exact kernels: the result is bit-identical to the
 * exact path by construction.  mode: 0 = exact kernels only; 1 = material-local (default): every 16 frames of a
 * 64-frame run the speculative kernel predicts, from the scale-factor indices and its bound, how many decisions of a
 * unit the guards will leave open, and past a threshold hands the rest of that run to the exact kernels -- the choice
 * is taken per run inside the call, never carried from one call or stream to the next; 2 = always speculate.
 * In mode 1 a call of fewer than 64 sound units (a frame closure, a short streaming push) uses the exact kernels only:
 * it is bound by the number of launches behind it, and every shortcut adds some.
 * The environment variable C1_SPEC (0/1/2) sets the default of new contexts. */
int c1_ctx_set_speculation(c1_ctx *ctx, int mode);
/* units that stayed with the speculative analysis and units among them that were redone exactly, since the context
 * was created (or since the last call with reset != 0); synchronises the context's stream */
int c1_ctx_speculation_stats(c1_ctx *ctx, uint64_t *units, uint64_t *redone, int reset);
/* units of speculative calls whose runs the speculative analysis handed to the exact kernels (mode 1; cleared by
 * c1_ctx_speculation_stats(reset)); synchronises the context's stream */
int c1_ctx_speculation_deferred(c1_ctx *ctx, uint64_t *units);
/* The exact paths (transient detection, mixed fixed modes, runs the speculative analysis handed over)
 * quantize the reference's coefficients in binary32 with the same guard band and pack the few units it cannot certify
 * again in binary64 (0.07 % of noise-like units to 4 % of stationary partials): units packed that way so far and units
 * packed twice (cleared by c1_ctx_speculation_stats(reset)).  Like the detector below it is used whenever speculation is
 * not 0; no decision of the encode path depends on what a context encoded before. */
int c1_ctx_quantization_stats(c1_ctx *ctx, uint64_t *units, uint64_t *repacked);
/* Transient detection (blockSelectorStage, encoder.js:111-152) runs speculatively too unless speculation is 0: the
 * transient FFT (transient.js:17-35) in binary32, an interval that provably contains the reference's transient score
 * (transient.js:197-226), the decision `score > threshold` where the whole interval lies on one side, and the
 * reference's own arithmetic for the units left open.  Units decided that way so far and units among them that needed
 * the exact recheck (cleared by c1_ctx_speculation_stats(reset)). */
int c1_ctx_detection_stats(c1_ctx *ctx, uint64_t *units, uint64_t *rechecked);

/* Decoder arithmetic.  0 (default): the reference's -- binary64 operations, binary32 at every typed-array store --
 * decoded PCM bit-identical to the reference.  1: the same computation in binary32 throughout; the PCM then differs
 * from the reference's by rounding noise (measured RMS < 1e-7 on full-scale material; the task statement allows 1e-5
 * "otherwise").  Applies to every decode entry point of the context. */
int c1_ctx_set_decode_precision(c1_ctx *ctx, int binary32);

/* ---- encode: replaces the encode() frame closure body, encoder.js:438-450
 *      (qmfAnalysisStage :57-96, blockSelectorStage :111-152, mdctStage :170-349,
 *      quantizationStage :365-418) plus serializeFrame (serialization.js:41-98), batched ----- */

/* device-resident: pcm[c] and units are DEVICE pointers; asynchronous on the context's stream: the call only enqueues
 * work and never waits for the device (it blocks only to grow the workspace on a first, larger call, or when the
 * options change while earlier calls are still queued).
 * Device memory: the context keeps a workspace of 2.6 KB per sound unit (4.8 KB with transient detection) for the largest
 * batch it has seen, until it is destroyed.  A batch is kept in one piece when that fits (up to 2^24 frames per channel, or
 * C1_CHUNK_FRAMES from the environment) and is otherwise cut into chunks of what 90 % of the free device memory holds; the
 * output is the same bytes either way. */
int c1_encode_device(c1_ctx *ctx, const float *const *pcm, int channels, int64_t frames,
                     int halo_frames, const c1_encode_options *opts, uint8_t *units);
/* host-resident: copies in, runs c1_encode_device, copies out, synchronises.
 * This is what encodeAeaPcm's hot loop (processor.js:119-136) calls once per batch. */
int c1_encode_batch(c1_ctx *ctx, const float *const *pcm, int channels, int64_t frames,
                    int halo_frames, const c1_encode_options *opts, uint8_t *units);

/* The same batch sharded over several devices of this host (the hot loop of processor.js:119-136 has no
 * dependency between frames beyond a bounded PCM history): contiguous frame ranges, one per entry of `devices`, each
 * encoded by its own host thread on a context of that device from its 2 frames of real PCM history; no collective, the
 * ranges land in `units` by frame index.  A device may be listed more than once (each entry gets its own context).
 * Contexts come from a process-wide pool: created on first use, leased to one call at a time (concurrent calls get their
 * own), made anew after c1_set_tables().  Every shard streams its range in chunks as c1_encode_batch does.
 * Bit-identical to c1_encode_batch on a fresh context. */
int c1_encode_batch_multi(const int *devices, int n_devices, const float *const *pcm, int channels, int64_t frames,
                          int halo_frames, const c1_encode_options *opts, uint8_t *units);
/* decode twin: every range starts from the unit(s) of the frame before it */
int c1_decode_batch_multi(const int *devices, int n_devices, const uint8_t *units, int channels, int64_t frames,
                          int halo_units, float *const *pcm);

/* Page-locked host memory for the *_batch calls.  c1_encode_batch / c1_decode_batch stream a batch of more than
 * 65 536 frames per channel in chunks: upload of chunk i+1, kernels of chunk i and download of chunk i-1 overlap on
 * three streams.  From buffers in memory from c1_host_alloc (or otherwise registered with HIP) the PCIe link then runs
 * at its pinned-memory rate (12.9 M stereo frames/s on one MI355X host); from pageable buffers the runtime stages the
 * copies and the same calls reach 91 % of that (11.7 M; whole-batch copy, compute, copy managed 8.1 M). */
int c1_host_alloc(size_t bytes, void **out);
int c1_host_free(void *p);

/* ---- decode: replaces the decode() frame closure body, decoder.js:408-411
 *      (dequantizationStage :52-98, imdctStage :116-330, qmfSynthesisStage :349-389)
 *      plus deserializeFrame (serialization.js:111-176), batched ----------------------------- */
int c1_decode_device(c1_ctx *ctx, const uint8_t *units, int channels, int64_t frames,
                     int halo_units, float *const *pcm);
int c1_decode_batch(c1_ctx *ctx, const uint8_t *units, int channels, int64_t frames,
                    int halo_units, float *const *pcm);

/* ---- stateful streams: what one encode()/decode() closure + its BufferPool is
 *      (encoder.js:438-441, buffers.js:7-81).  The stream keeps the PCM / unit history on the
 *      device, so successive calls continue the same stream bit for bit. ---------------------- */
typedef struct c1_enc_stream c1_enc_stream;
typedef struct c1_dec_stream c1_dec_stream;
int c1_enc_stream_create(c1_ctx *ctx, int channels, const c1_encode_options *opts, c1_enc_stream **out);
int c1_enc_stream_push(c1_enc_stream *s, const float *const *pcm /* host */, int64_t frames,
                       uint8_t *units /* host, frames*channels*212 */);
int c1_enc_stream_destroy(c1_enc_stream *s);
int c1_dec_stream_create(c1_ctx *ctx, int channels, c1_dec_stream **out);
int c1_dec_stream_push(c1_dec_stream *s, const uint8_t *units /* host */, int64_t frames,
                       float *const *pcm /* host */);
int c1_dec_stream_destroy(c1_dec_stream *s);

/* ---- device-resident synthetic input for measurement ------------- */
enum { C1_SIGNAL_WHITE = 0, C1_SIGNAL_PINK_BURSTS = 1, C1_SIGNAL_MIXED = 2, C1_SIGNAL_PARTIALS = 3 };
/* Fills pcm (DEVICE pointer, frames*512 floats) with a signal of the given statistics.  Every 512-
 * frame segment restarts xorshift32 from a seed derived from (seed, segment), so segments are
 * generated in parallel; segment 0 with seed s reproduces the first 512 frames of the generators
 * in BASELINE.md section 4 exactly (the parity subset).  C1_SIGNAL_MIXED is the synthetic corpus of BASELINE configs[3]:
 * 512-frame segments cycling white noise, pink noise with bursts, stationary partials with slow amplitude modulation
 * and quiet white noise; C1_SIGNAL_PARTIALS is the tonal segment kind alone (parity subsets of both are checked by
 * copying the generated PCM back to the host). */
int c1_generate_device(c1_ctx *ctx, int signal, uint32_t seed, int64_t frames, float *pcm);

/* ---- the data formats either side of the path -------------- */
/* WAV PCM ingest, bin/cli.js:367-404 (WavReader._processFrameBuffer / _sampleToFloat): little-endian
 * interleaved integer PCM (bits = 16, 24 or 32) -> planar float32, value / 2^(bits-1).  Device pointers. */
int c1_pcm_from_int_device(c1_ctx *ctx, const void *interleaved, int bits, int channels,
                           int64_t samples_per_channel, float *const *pcm);
/* 16-bit WAV output, codec/io/processor.js:368-447: clamp to [-1,1], negative * 32768, positive * 32767,
 * DataView.setInt16 truncation; planar float32 -> little-endian interleaved int16.  Device pointers. */
int c1_pcm_to_int16_device(c1_ctx *ctx, const float *const *pcm, int channels,
                           int64_t samples_per_channel, int16_t *interleaved);
/* WAV body <-> sound units in one host call (what the reference's CLI does around the codec: WavReader ->
 * frameBufferToFrames -> encode, bin/cli.js:367-404, codec/io/processor.js:246-276; and decode -> createWavBlob,
 * processor.js:349-447).  The integer PCM crosses PCIe (half the bytes of float32 for 16 bit) and is converted on
 * the device; batches are streamed in chunks, at the pinned rate when the host buffers are page-locked.
 * samples_per_channel need not be a multiple of 512: the last frame is zero padded as frameBufferToFrames does.
 * units: ceil(samples_per_channel / 512) * channels * 212 bytes. */
int c1_encode_wav_batch(c1_ctx *ctx, const void *interleaved, int bits, int channels, int64_t samples_per_channel,
                        const c1_encode_options *opts, uint8_t *units);
/* decode `frames` frames to 16-bit interleaved PCM: frames * 512 * channels int16 */
int c1_decode_wav16_batch(c1_ctx *ctx, const uint8_t *units, int channels, int64_t frames, int16_t *interleaved);
/* AeaFile.createHeader, codec/io/serialization.js:190-211 (host side; the units a batch call returns are
 * already the AEA body: header + units is the whole file).  title is UTF-8, truncated to 255 bytes. */
int c1_aea_header(const char *title, uint32_t unit_count, int channels, uint8_t out[2048]);

/* ---- the single-stage functions the reference exports next to encode()/decode() (codec/index.js:30-35,42).  Host
 *      pointers, synchronous; the arithmetic runs on the device in the reference's own number model.  They serve
 *      applications that import these names; the hot path itself never calls them (it quantizes inside the packing kernel
 *      and transforms inside the analysis kernels). ------------------------------------------------------------------- */
/* quantize, codec/coding/quantization.js:34-56: out[i] = clamp(((x norm) +- 0.5) | 0), norm = ((1 << (bits - 1)) - 1) /
 * SCALE_FACTORS[sfi]; zeros when bits or sfi is 0 */
int c1_quantize(c1_ctx *ctx, const float *coefficients, int n, int scale_factor_index, int bits_per_sample, int32_t *out);
/* dequantize, quantization.js:65-78: Float32((q SCALE_FACTORS[sfi]) / ((1 << (bits - 1)) - 1)) */
int c1_dequantize(c1_ctx *ctx, const int32_t *quantized, int n, int scale_factor_index, int bits_per_sample, float *out);
/* FFT.fft, codec/transforms/fft.js:14-68: in place on real[n], imag[n], n a power of two; w = (cos, sin)(-2 pi / stride) for
 * stride = 2, 4, .., n as the HOST's Math.cos / Math.sin give them (log2(n) pairs; the reference computes them per call, :37-39) */
int c1_fft(c1_ctx *ctx, float *real, float *imag, int n, const double *w);
/* qmfAnalysisStage, codec/pipeline/encoder.js:57-96, for `frames` consecutive frames of one channel: pcm = (halo_frames + frames)
 * * 512 samples, the first halo_frames (0..2) being the stream's history (zero history = a fresh BufferPool); bands =
 * frames * 512 floats, low128 | mid128 | high256 (the high band behind its 39-sample delay) per frame */
int c1_qmf_analysis_batch(c1_ctx *ctx, const float *pcm, int64_t frames, int halo_frames, float *bands);
/* mdctStage, encoder.js:170-349, from band samples: bands = (halo_frames + frames) * 512 floats as above, the first frame (when
 * halo_frames = 1) being the previous frame of the stream, whose band tails make mdctOverlap (:309-316; none: a fresh pool's zero
 * overlap); block_modes = frames * 3 (0 long, else short); coefs = frames * 512 (as quantizationStage receives them);
 * bands_windowed (optional) = frames * 512: the band arrays as the reference leaves them, windowed in place (:244,292,314) */
int c1_mdct_batch(c1_ctx *ctx, const float *bands, int64_t frames, int halo_frames, const int32_t *block_modes, float *coefs,
                  float *bands_windowed);

/* ---- stage taps for bring-up and stage-level parity tests (device pointers) ---------------- */
/* bands: frames*channels*512 floats (low128|mid128|high256 per unit index, before windowing);
 * coefs: same shape (MDCT coefficients as quantizationStage receives them);
 * side:  frames*channels*64 bytes: sfi[52], modes byte (m0|m1<<2|m2<<4);  any of the three may be NULL */
int c1_encode_stages_device(c1_ctx *ctx, const float *const *pcm, int channels, int64_t frames,
                            int halo_frames, const c1_encode_options *opts, float *bands,
                            float *coefs, uint8_t *side, uint8_t *alloc /* frames*channels*32 or NULL */);

/* Stage taps of the transient detector (transient.js:17-226, blockSelectorStage encoder.js:111-152), device pointers:
 * mags:  frames*channels*256 floats, performFFT's magnitude spectra of the three bands (64 | 64 | 128 per unit index);
 * modes: frames*channels bytes, the block modes the detector chose (m0 | m1<<2 | m2<<4).  opts must ask for detection
 * (fixed_block_modes {-1,-1,-1}).  Either pointer may be NULL. */
int c1_detect_stages_device(c1_ctx *ctx, const float *const *pcm, int channels, int64_t frames, int halo_frames,
                            const c1_encode_options *opts, float *mags, uint8_t *modes);

/* Test tap of the detector's decisions.  scores: frames*channels*3*2 doubles, per unit and band {lo, hi}: with
 * speculative != 0 the interval the binary32 detector derives for calculateTransientScore (transient.js:197-226), else
 * the reference's score twice.  modes: frames*channels bytes (after the exact recheck of the open units);
 * open_units: one uint32 on the device, the number of units whose interval contained the threshold.  Device pointers,
 * any of them may be NULL. */
int c1_detect_scores_device(c1_ctx *ctx, const float *const *pcm, int channels, int64_t frames, int halo_frames,
                            const c1_encode_options *opts, int speculative, double *scores, uint8_t *modes,
                            uint32_t *open_units);
/* Test tap: the speculative detector's binary32 magnitude spectra (mags: frames*channels*256 floats, laid out as
 * c1_detect_stages_device's) and the bound it claims on the l2 distance of each band's spectrum from the reference's
 * (bounds: frames*channels*3 floats).  Device pointers. */
int c1_detect_spec_mags_device(c1_ctx *ctx, const float *const *pcm, int channels, int64_t frames, int halo_frames,
                               float *mags, float *bounds);
/* Test tap: the device's binary32 log2 (v_log_f32), which the speculative detector's flatness sums use, against
 * binary64 log2 over the bit patterns [first_bits, first_bits + count) of normal positive numbers.  out (host):
 * out[0] = max |r - log2 x| / |log2 x| in units of 2^-24 over the x with |log2 x| >= 2^-6, out[1] = max |r - log2 x|
 * over the others. */
int c1_log2f_error_device(c1_ctx *ctx, uint32_t first_bits, uint64_t count, double *out);

/* Test tap: Math.log (fn 0), Math.exp (1), Math.log1p (2), Math.log10 (3) as the reference's engine evaluates them and as
 * the detector's kernels use them (transient.js:129, :137, :185, :211; V8 src/base/ieee754.cc = fdlibm, not correctly
 * rounded, so the algorithm itself is part of the parity contract).  in, out: n doubles, device pointers. */
int c1_libm_device(c1_ctx *ctx, int fn, const double *in, double *out, int64_t n);

/* Test tap of the bit allocation (allocateBits, bitallocation.js:74-142).  The library runs the greedy heap only for the
 * candidate BFU counts that a lower bound on their total distortion does not exclude; this call shows both sides of
 * that comparison.  side: units*64 bytes as above (sfi[52] of every unit); out: units*16 doubles = the totals of the
 * eight candidates {20,28,32,36,40,44,48,52} as calculateTotalDistortion returns them (:157-190), then the seven lower
 * bounds, then (slot 15) the index of the candidate the pruned production path chose, -1 for the fallback.  Device
 * pointers.  tests/test_gpu_alloc_bound.py checks bound <= total everywhere and choice == brute-force minimum. */
int c1_alloc_bounds_device(c1_ctx *ctx, const uint8_t *side, int64_t units, const c1_encode_options *opts, double *out);

/* The speculative binary32 analysis on its own (diagnostics; tests/test_gpu_spec.py checks the bound with it):
 * coefs: frames*channels*512 floats = the binary32 coefficients; eps: frames*channels*4 floats = the proven bound on
 * |coefficient - reference coefficient| for bands 0, 1, 2 and a flag word (non-zero bit pattern: a scale-factor
 * index was not certain); side as above.  Fixed block modes [0,0,0] only (C1_ERR_ARG otherwise). */
int c1_spec_stages_device(c1_ctx *ctx, const float *const *pcm, int channels, int64_t frames, int halo_frames,
                          const c1_encode_options *opts, float *coefs, float *eps, uint8_t *side);

/* Test tap of the speculative quantizer on its own (k_pack<.., SPEC>: quantization.js:34-56 in binary32 behind the guard
 * band of DESIGN.md 3b, serializeFrame serialization.js:41-98).  The caller supplies what the analysis and allocation
 * kernels would: coefs units*512 floats (coefficient order of quantizationStage), eps units*4 floats (bounds of bands
 * 0..2; flag word: bit 0 = a scale-factor index is open), side units*64 bytes (sfi[52], modes byte), alloc units*32 bytes
 * (52 word-length nibbles, low nibble first; last dword: fallback flag bit 27, BFU-amount index bits 28..30).  Out:
 * units_out units*212 bytes, and lists = 8 + 3*units uint32: [0] [1] [2] the lengths of the redo, reallocation and
 * re-analysis lists, which start at 8, 8 + units and 8 + 2*units.  Device pointers.  tests/test_gpu_pack_guard.py
 * checks kernel == CPU model (tests/model/pack_model.c) on coefficients built to sit on the guard band's edge. */
int c1_pack_spec_tap_device(c1_ctx *ctx, const float *coefs, const float *eps, const uint8_t *side, const uint8_t *alloc,
                            int64_t units, int all_long, uint8_t *units_out, uint32_t *lists);

#ifdef __cplusplus
}
#endif
#endif /* CARTA1_HIP_H */
