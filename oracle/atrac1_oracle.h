/*
 * atrac1_oracle.h -- CPU restatement of the reference's ATRAC1 hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is the parity checker for the HIP path: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * Nothing under carta1_amd/ links, imports or calls it.
 *
 * Parity status: PINNED.  The reference (aynik/carta1 v1.1.10, JavaScript) was
 * run in the build container (tests/golden/gen/gen_golden.mjs) and this
 * restatement reproduces its outputs bit for bit on every committed fixture
 * (tests/test_oracle_golden.py): the config-1 known-answer unit, eleven 64-frame
 * stereo runs (212-byte units and decoded PCM), 2048-frame hashes, per-stage
 * intermediates, findScaleFactor at all 64 table boundaries, quantize vectors
 * and the ragged / silent / loud / denormal edge cases.
 *
 * Numeric model of the reference (ECMAScript): every arithmetic operation is an
 * IEEE-754 double operation with no fused multiply-add; every store into a
 * Float32Array rounds to binary32 (round to nearest even).  Build with
 * -ffp-contract=off and without -ffast-math (oracle/Makefile does).
 */
#ifndef ATRAC1_ORACLE_H
#define ATRAC1_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define C1O_FRAME 512
#define C1O_UNIT_BYTES 212
#define C1O_NUM_BFU 52

/* per-stream encoder state == the encoder half of BufferPool (codec/core/buffers.js:30-59) */
typedef struct c1o_enc_state {
  float qmf_low[46];     /* stage-1 delay line  (qmfDelays.lowBand)  */
  float qmf_mid[46];     /* stage-2 delay line  (qmfDelays.midBand)  */
  float qmf_high[39];    /* high-band delay     (qmfDelays.highBand) */
  float overlap[3][32];  /* mdctOverlap                                */
  float prev_mag[256];   /* transientDetection: 64 | 64 | 128          */
} c1o_enc_state;

/* per-stream decoder state == the decoder half of BufferPool (buffers.js:30-34, 68-72) */
typedef struct c1o_dec_state {
  float qmf_low[46];
  float qmf_mid[46];
  float qmf_high[39];
  float tail[3][16];     /* last 16 samples of imdctOverlap[band] */
} c1o_dec_state;

typedef struct c1o_options {
  int fixed_modes[3];    /* fixed_modes[0] < 0: transient detection on */
  double threshold;      /* options.transientThresholdLow (used for all three bands) */
  double biased_sf[64];  /* pow(SCALE_FACTORS[i], allocationBias), as the host computed it */
} c1o_options;

/* the frame-closure result of encode() (codec/pipeline/encoder.js:410-416) */
typedef struct c1o_fields {
  int nbfu;
  int modes[3];
  int wl[C1O_NUM_BFU];
  int sfi[C1O_NUM_BFU];
  int q[C1O_FRAME];      /* mantissas, BFU after BFU, SPECS_PER_BFU[b] each */
} c1o_fields;

void c1o_default_biased_sf(double bias, double out[64]); /* bias==1 exact copy; else libm pow (unpinned vs V8 for general bias) */
const double *c1o_scale_factors(void);

void c1o_enc_state_init(c1o_enc_state *s);
void c1o_dec_state_init(c1o_dec_state *s);

/* stage-level entry points (kernel bring-up and stage parity tests) */
void c1o_qmf_analysis_frame(c1o_enc_state *s, const float pcm[512], float bands[512]);
void c1o_block_modes(c1o_enc_state *s, const float bands[512], const c1o_options *o, int modes[3]);
void c1o_transient_mags(const float bands[512], float mags[256]);
/* Math.log (fn 0), exp (1), log1p (2), log10 (3) as V8 evaluates them (c1o_fdlibm.h), on an array */
void c1o_libm(int fn, const double *in, double *out, long n);
double c1o_transient_score(const float *cur, const float *prev, int n);   /* transient.js:63-226 */
int c1o_detect_transient(const float *cur, const float *prev, int n, double threshold);
void c1o_mdct_frame(c1o_enc_state *s, float bands[512], const int modes[3], float coefs[512]);
int c1o_find_scale_factor(const float *x, int n);
void c1o_allocate(const float coefs[512], const int modes[3], const double biased_sf[64],
                  int *nbfu, int wl[52], int sfi[52]);
void c1o_quantize_bfu(const float *x, int n, int sfi, int bits, int *out);
void c1o_dequantize_bfu(const int *q, int n, int sfi, int bits, float *out);

/* whole frame: encode() closure, serializeFrame, deserializeFrame, decode() closure */
void c1o_encode_frame(c1o_enc_state *s, const float pcm[512], const c1o_options *o, c1o_fields *out);
void c1o_pack_unit(const c1o_fields *f, uint8_t unit[212]);
void c1o_unpack_unit(const uint8_t unit[212], c1o_fields *f);
void c1o_decode_frame(c1o_dec_state *s, const c1o_fields *f, float pcm[512]);

/* streams: planar PCM per channel -> units interleaved L,R (processor.js:119-136), and back */
void c1o_encode_stream(const float *const *pcm, int channels, long frames, const c1o_options *o,
                       c1o_enc_state *states /* [channels], updated */, uint8_t *units);
void c1o_decode_stream(const uint8_t *units, int channels, long frames,
                       c1o_dec_state *states /* [channels], updated */, float *const *pcm);

/* synthetic signals of SURVEY.md section 8c / BASELINE.md section 4 (xorshift32) */
void c1o_gen_white(uint32_t seed, long n, float *out);
void c1o_gen_pinkT(uint32_t seed, long n, float *out);

/* formats either side of the path */
/* WavReader._sampleToFloat, bin/cli.js:394-404: interleaved LE int16/24/32 -> planar float32 */
void c1o_pcm_from_int(const uint8_t *interleaved, int bits, int channels, long samples, float *const *pcm);
/* _createMonoWavBlob/_createStereoWavBlob sample loops, codec/io/processor.js:379-392,429-447 */
void c1o_pcm_to_int16(const float *const *pcm, int channels, long samples, int16_t *interleaved);
/* AeaFile.createHeader, codec/io/serialization.js:190-211 */
void c1o_aea_header(const char *title, uint32_t frame_count, int channels, uint8_t out[2048]);

#ifdef __cplusplus
}
#endif
#endif
