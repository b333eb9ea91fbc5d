// Golden-vector generator: runs the JavaScript reference (aynik/carta1 v1.1.10,
// read in place from /root/reference through loader.mjs) and writes small
// fixtures into tests/golden/.  Only inputs/outputs and numeric tables are
// written — never reference source text.
//
//   cd tests/golden/gen && node --experimental-loader ./loader.mjs gen_golden.mjs
//
// The reference exists only in the build container; the fixtures travel.
import fs from 'fs'
import path from 'path'
import crypto from 'crypto'
import { fileURLToPath } from 'url'

import { encode } from '/root/reference/codec/pipeline/encoder.js'
import { decode } from '/root/reference/codec/pipeline/decoder.js'
import { serializeFrame, deserializeFrame, AeaFile } from '/root/reference/codec/io/serialization.js'
import { EncoderOptions } from '/root/reference/codec/core/options.js'
import { BufferPool } from '/root/reference/codec/core/buffers.js'
import * as K from '/root/reference/codec/core/constants.js'
import { qmfAnalysis, qmfSynthesis } from '/root/reference/codec/transforms/qmf.js'
import * as M from '/root/reference/codec/transforms/mdct.js'
import { FFT } from '/root/reference/codec/transforms/fft.js'
import { performFFT, detectTransient } from '/root/reference/codec/analysis/transient.js'
import { findScaleFactor, allocateBits } from '/root/reference/codec/coding/bitallocation.js'
import { quantize, dequantize, groupIntoBFUs } from '/root/reference/codec/coding/quantization.js'
import { qmfAnalysisStage, blockSelectorStage, mdctStage } from '/root/reference/codec/pipeline/encoder.js'

const OUT = path.resolve(path.dirname(fileURLToPath(import.meta.url)), '..')

// ---------- helpers ----------
const f64hex = (x) => { const b = Buffer.alloc(8); b.writeDoubleBE(x); return b.toString('hex') }
const f32hex = (x) => { const b = Buffer.alloc(4); b.writeFloatBE(x); return b.toString('hex') }
const bytesOf = (ta) => Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength)
const sha = (buf) => crypto.createHash('sha256').update(buf).digest('hex')
const writeJson = (name, obj) => fs.writeFileSync(path.join(OUT, name), JSON.stringify(obj, null, 1) + '\n')
const writeBin = (name, buf) => fs.writeFileSync(path.join(OUT, name), buf)

// xorshift32 PRNG of SURVEY.md §8c; u in [-1, 1)
function xorshift(seed) {
  let s = seed >>> 0
  return () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return (s / 4294967296) * 2 - 1 }
}
function white(seed, n) {
  const r = xorshift(seed); const x = new Float32Array(n)
  for (let i = 0; i < n; i++) x[i] = Math.fround(r() * 0.5)
  return x
}
function pinkT(seed, n) {
  const r = xorshift(seed); const x = new Float32Array(n); let p = 0
  for (let i = 0; i < n; i++) {
    const u = r(); p = 0.98 * p + 0.05 * u; let v = p
    if ((i >> 9) % 8 === 5 && (i % 512) >= 256) v += 0.8 * r()
    x[i] = v
  }
  return x
}
function sine(freq, n, amp = 1) {
  const x = new Float32Array(n)
  for (let i = 0; i < n; i++) x[i] = amp * Math.sin((2 * Math.PI * freq * i) / 44100)
  return x
}

function encodeChannels(chs, optValues) {
  // what encodeAeaPcm does (processor.js:597-617) minus the Blob: two independent
  // encoders, units interleaved L,R, last frame zero padded.
  const n = Math.max(...chs.map((c) => c.length))
  const frames = Math.ceil(n / 512)
  const encs = chs.map(() => encode(new EncoderOptions(optValues)))
  const units = []
  const fields = []
  for (let f = 0; f < frames; f++) {
    for (let c = 0; c < chs.length; c++) {
      const fr = new Float32Array(512)
      fr.set(chs[c].subarray(f * 512, Math.min((f + 1) * 512, chs[c].length)))
      const res = encs[c](fr)
      fields.push(res)
      units.push(serializeFrame(res))
    }
  }
  return { units, fields, frames }
}
function decodeUnits(units, nch) {
  const decs = []
  for (let c = 0; c < nch; c++) decs.push(decode())
  const frames = units.length / nch
  const out = []
  for (let c = 0; c < nch; c++) out.push(new Float32Array(frames * 512))
  for (let f = 0; f < frames; f++)
    for (let c = 0; c < nch; c++) out[c].set(decs[c](deserializeFrame(units[f * nch + c])), f * 512)
  return out
}
const concatUnits = (units) => Buffer.concat(units.map((u) => Buffer.from(u)))
// decoded PCM hashed as the survey did: per frame L then R
function pcmFrameInterleaved(pcm, frames) {
  const nch = pcm.length
  const out = new Float32Array(frames * 512 * nch)
  for (let f = 0; f < frames; f++)
    for (let c = 0; c < nch; c++) out.set(pcm[c].subarray(f * 512, (f + 1) * 512), (f * nch + c) * 512)
  return out
}
function modeHist(fields) {
  const h = {}
  for (const f of fields) { const k = f.blockModes.join(''); h[k] = (h[k] || 0) + 1 }
  return h
}

// ---------- 1. numeric tables ----------
{
  const mdctTab = (t) => Array.from(t.sinCosTable).map(f64hex)
  const fftW = {}
  for (let stride = 2; stride <= 256; stride <<= 1) {
    const a = (-2 * Math.PI) / stride
    fftW[stride] = [f64hex(Math.cos(a)), f64hex(Math.sin(a))]
  }
  const biased = {}
  for (const b of [0, 0.25, 0.5, 1, 1.5, 2, 3.3, 5]) {
    biased[String(b)] = Array.from(K.SCALE_FACTORS).map((s) => f64hex(b === 1 ? s : Math.pow(s, b)))
  }
  writeJson('tables.json', {
    note: 'hex = IEEE-754 big-endian bit pattern; values produced by V8 ' + process.versions.v8 + ' (node ' + process.version + ') running the reference constants',
    scale_factors_f64: Array.from(K.SCALE_FACTORS).map(f64hex),
    window_short_f64: Array.from(K.WINDOW_SHORT).map(f64hex),
    qmf_coeffs_f32: Array.from(K.QMF_COEFFS).map(f32hex),
    qmf_even_f32: Array.from(K.QMF_EVEN).map(f32hex),
    qmf_odd_f32: Array.from(K.QMF_ODD).map(f32hex),
    mdct_sincos_f64: {
      fwd64: mdctTab(M.mdct64), fwd256: mdctTab(M.mdct256), fwd512: mdctTab(M.mdct512),
      inv64: mdctTab(M.imdct64), inv256: mdctTab(M.imdct256), inv512: mdctTab(M.imdct512),
    },
    fft_w_f64: fftW,
    inv_power_of_two_f64: Array.from(K.INV_POWER_OF_TWO).map(f64hex),
    distortion_delta_factors_f64: Array.from(K.DISTORTION_DELTA_FACTORS).map(f64hex),
    word_length_delta_bits: Array.from(K.WORD_LENGTH_DELTA_BITS),
    word_length_bits: Array.from(K.WORD_LENGTH_BITS),
    specs_per_bfu: Array.from(K.SPECS_PER_BFU),
    bfu_start_long: Array.from(K.BFU_START_LONG),
    bfu_start_short: Array.from(K.BFU_START_SHORT),
    bfu_amounts: Array.from(K.BFU_AMOUNTS),
    log1p_10_f64: f64hex(Math.log1p(10)),
    biased_scale_factors_f64: biased,
  })
}

// ---------- 2. config 1: mono 1 kHz sine, one frame ----------
{
  const pcm = sine(1000, 512)
  const { units, fields } = encodeChannels([pcm], { fixedBlockModes: [0, 0, 0] })
  const f = fields[0]
  writeJson('config1_sine1k.json', {
    input: 'x[i] = fround(sin(2*pi*1000*i/44100)), i<512, mono; fixedBlockModes [0,0,0], bias 1',
    nBfu: f.nBfu, blockModes: f.blockModes,
    wordLengthIndices: Array.from(f.wordLengthIndices), scaleFactorIndices: Array.from(f.scaleFactorIndices),
    quantizedCoefficients: f.quantizedCoefficients.map((q) => Array.from(q)),
    unit_hex: Buffer.from(units[0]).toString('hex'),
    decoded_first_frame_sha256: sha(bytesOf(decodeUnits(units, 1)[0])),
  })
}

// ---------- 3. 64-frame stereo known answers (SURVEY.md §8c) + longer hashes ----------
const CASES = [
  ['white_m000_b0.5', 'white', { fixedBlockModes: [0, 0, 0], allocationBias: 0.5 }],
  ['white_m000_b1', 'white', { fixedBlockModes: [0, 0, 0], allocationBias: 1 }],
  ['white_m000_b2', 'white', { fixedBlockModes: [0, 0, 0], allocationBias: 2 }],
  ['white_m223_b0.5', 'white', { fixedBlockModes: [2, 2, 3], allocationBias: 0.5 }],
  ['white_m223_b1', 'white', { fixedBlockModes: [2, 2, 3], allocationBias: 1 }],
  ['white_m223_b2', 'white', { fixedBlockModes: [2, 2, 3], allocationBias: 2 }],
  ['white_detect', 'white', {}],
  ['pinkT_detect', 'pinkT', {}],
  ['pinkT_m000_b1', 'pinkT', { fixedBlockModes: [0, 0, 0] }],
  ['pinkT_m203_b1', 'pinkT', { fixedBlockModes: [2, 0, 3] }],
  ['pinkT_detect_thr0.3', 'pinkT', { transientThresholdLow: 0.3 }],
]
function makeInput(kind, frames) {
  const n = frames * 512
  return kind === 'white' ? [white(1, n), white(2, n)] : [pinkT(3, n), pinkT(4, n)]
}
{
  const index = {}
  for (const [name, kind, opts] of CASES) {
    const frames = 64
    const chs = makeInput(kind, frames)
    const { units, fields } = encodeChannels(chs, opts)
    const pcm = decodeUnits(units, 2)
    const ub = concatUnits(units)
    writeBin(`kat64_${name}.units.bin`, ub)
    // decoded PCM: the first 8 frames per channel in full (planar L then R), the rest by hash
    const head = new Float32Array(2 * 8 * 512)
    head.set(pcm[0].subarray(0, 8 * 512), 0); head.set(pcm[1].subarray(0, 8 * 512), 8 * 512)
    writeBin(`kat64_${name}.pcm8.bin`, bytesOf(head))
    index[name] = {
      signal: kind, seeds: kind === 'white' ? [1, 2] : [3, 4], frames, options: opts,
      input_L_sha256: sha(bytesOf(chs[0])), input_R_sha256: sha(bytesOf(chs[1])),
      units_sha256: sha(ub),
      decoded_LR_per_frame_sha256: sha(bytesOf(pcmFrameInterleaved(pcm, frames))),
      decoded_planar_L_sha256: sha(bytesOf(pcm[0])), decoded_planar_R_sha256: sha(bytesOf(pcm[1])),
      mode_hist: modeHist(fields),
    }
  }
  // longer runs, hashes only
  for (const [name, kind, opts] of CASES) {
    const frames = 2048
    const chs = makeInput(kind, frames)
    const { units, fields } = encodeChannels(chs, opts)
    const pcm = decodeUnits(units, 2)
    const ub = concatUnits(units)
    // modes per unit as one byte each: m0 | m1<<2 | m2<<4  (tiny, lets mode parity be pinned exactly)
    const modes = Buffer.from(fields.map((f) => f.blockModes[0] | (f.blockModes[1] << 2) | (f.blockModes[2] << 4)))
    writeBin(`long2048_${name}.modes.bin`, modes)
    index[name].long2048 = {
      frames, units_sha256: sha(ub),
      decoded_planar_L_sha256: sha(bytesOf(pcm[0])), decoded_planar_R_sha256: sha(bytesOf(pcm[1])),
      nbfu_hist: fields.reduce((h, f) => { h[f.nBfu] = (h[f.nBfu] || 0) + 1; return h }, {}),
      mode_hist: modeHist(fields),
    }
  }
  writeJson('kat_index.json', index)
}

// ---------- 4. per-stage intermediates for kernel bring-up (mono, white seed 1, 3 frames) ----------
{
  const pcm = white(1, 3 * 512)
  for (const [tag, modes] of [['m000', [0, 0, 0]], ['m223', [2, 2, 3]]]) {
    const pool = new BufferPool()
    const ctx = { bufferPool: pool, options: new EncoderOptions({ fixedBlockModes: modes }) }
    const s1 = qmfAnalysisStage(ctx), s2 = blockSelectorStage(ctx), s3 = mdctStage(ctx)
    const bandsOut = new Float32Array(3 * 512), coefOut = new Float32Array(3 * 512)
    const alloc = []
    for (let f = 0; f < 3; f++) {
      const a = s1(pcm.slice(f * 512, (f + 1) * 512))
      bandsOut.set(a.bands[0], f * 512); bandsOut.set(a.bands[1], f * 512 + 128); bandsOut.set(a.bands[2], f * 512 + 256)
      const c = s3(s2(a))
      coefOut.set(c.coefficients, f * 512)
      const g = groupIntoBFUs(c.coefficients, modes)
      const r = allocateBits(g.bfuData, g.bfuSizes, g.bfuCount, 1.0)
      alloc.push({ nBfu: r.bfuCount, wl: Array.from(r.allocation), sfi: Array.from(r.scaleFactorIndices) })
    }
    if (tag === 'm000') writeBin('stages_white1_bands.f32.bin', bytesOf(bandsOut)) // bands before windowing do not depend on modes
    writeBin(`stages_white1_${tag}_coefs.f32.bin`, bytesOf(coefOut))
    writeJson(`stages_white1_${tag}_alloc.json`, alloc)
  }
  // transient magnitudes + decisions, pinkT seed 3, 8 frames (frame 5 carries the burst)
  const p = pinkT(3, 8 * 512)
  const pool = new BufferPool()
  const ctx = { bufferPool: pool, options: new EncoderOptions({}) }
  const s1 = qmfAnalysisStage(ctx)
  const mags = new Float32Array(8 * 256)
  const dec = []
  let prev = [new Float32Array(64), new Float32Array(64), new Float32Array(128)]
  for (let f = 0; f < 8; f++) {
    const a = s1(p.slice(f * 512, (f + 1) * 512))
    const cur = [performFFT(a.bands[0], 128), performFFT(a.bands[1], 128), performFFT(a.bands[2], 256)]
    mags.set(cur[0], f * 256); mags.set(cur[1], f * 256 + 64); mags.set(cur[2], f * 256 + 128)
    const row = []
    for (const thr of [0.05, 0.1, 0.2, 0.3, 0.5, 1.0, 1.5])
      row.push([0, 1, 2].map((b) => (detectTransient(cur[b], prev[b], thr) ? 1 : 0)))
    dec.push(row)
    prev = cur
  }
  writeBin('stages_pinkT3_mags.f32.bin', bytesOf(mags))
  writeJson('stages_pinkT3_transient.json', { thresholds: [0.05, 0.1, 0.2, 0.3, 0.5, 1.0, 1.5], decisions_frame_thr_band: dec })
}

// ---------- 5. findScaleFactor at every table boundary ----------
{
  const rows = []
  const f32 = new Float32Array(1), u32 = new Uint32Array(f32.buffer)
  for (let i = 0; i < 64; i++) {
    f32[0] = K.SCALE_FACTORS[i]
    const centre = u32[0]
    for (let d = -4; d <= 4; d++) {
      u32[0] = centre + d
      rows.push([u32[0], findScaleFactor(new Float32Array([f32[0]]), 1)])
      rows.push([(u32[0] | 0x80000000) >>> 0, findScaleFactor(new Float32Array([-f32[0]]), 1)])
    }
  }
  for (const v of [0, 1e-45, 1e-40, 1e-38, 4.7e-7, 4.77e-7, 0.999, 1, 1.0000001, 2, 1000, 3.4e38]) {
    f32[0] = v; rows.push([u32[0], findScaleFactor(new Float32Array([f32[0]]), 1)])
  }
  writeJson('find_scale_factor.json', { note: '[f32 bit pattern, reference findScaleFactor([x],1)]', rows })
}

// ---------- 6. quantize / dequantize spot vectors ----------
{
  const r = xorshift(77)
  const rows = []
  for (let t = 0; t < 200; t++) {
    const sfi = 1 + Math.floor((r() * 0.5 + 0.5) * 63)
    const bits = K.WORD_LENGTH_BITS[1 + Math.floor((r() * 0.5 + 0.5) * 15)]
    const x = new Float32Array(8)
    for (let i = 0; i < 8; i++) x[i] = r() * K.SCALE_FACTORS[sfi] * (t % 10 === 0 ? 3 : 1.0)
    const q = quantize(x, sfi, bits)
    const d = dequantize(q, sfi, bits)
    rows.push({ sfi, bits, x: Array.from(x).map(f32hex), q: Array.from(q), d: Array.from(d).map(f32hex) })
  }
  // ToInt32 wrap corner: huge coefficients against a small scale factor
  for (const v of [3e9, -3e9, 1e12, -7.7e15, 3.4e38, 65536.5]) {
    const x = new Float32Array([v, -v, v / 3, 1])
    const q = quantize(x, 63, 16)
    rows.push({ sfi: 63, bits: 16, x: Array.from(x).map(f32hex), q: Array.from(q), d: Array.from(dequantize(q, 63, 16)).map(f32hex) })
  }
  writeJson('quantize.json', rows)
}

// ---------- 7. AEA header + ragged/edge inputs through the whole-file path ----------
{
  const hdr = AeaFile.createHeader('encoded by carta1', 4, 2)
  const cases = {}
  // 700-sample stereo (processor.test.js:95-108): 2 frames per channel, zero padded
  const a = white(11, 700), b = white(12, 700)
  let e = encodeChannels([a, b], {})
  cases.stereo700 = { units_hex: concatUnits(e.units).toString('hex'), decoded_sha256: decodeUnits(e.units, 2).map((c) => sha(bytesOf(c))) }
  // ragged stereo: L longer than R
  e = encodeChannels([white(13, 1300), white(14, 600)], { fixedBlockModes: [0, 0, 0] })
  cases.ragged_1300_600 = { units_sha256: sha(concatUnits(e.units)), units: e.units.length }
  // silence
  e = encodeChannels([new Float32Array(1024)], {})
  cases.silence_mono_2frames = { units_hex: concatUnits(e.units).toString('hex') }
  // mono sine sweep of amplitudes incl. > 1 (sfi clamp at 63)
  const loud = sine(440, 2048, 4.0)
  e = encodeChannels([loud], { fixedBlockModes: [0, 0, 0] })
  cases.loud_sine_mono = { units_sha256: sha(concatUnits(e.units)), decoded_sha256: sha(bytesOf(decodeUnits(e.units, 1)[0])) }
  // tiny amplitudes (denormal-range coefficients)
  const tiny = white(15, 1024); for (let i = 0; i < tiny.length; i++) tiny[i] *= 1e-36
  e = encodeChannels([tiny], { fixedBlockModes: [0, 0, 0] })
  cases.tiny_mono = { units_hex: concatUnits(e.units).toString('hex') }
  const tiny2 = white(15, 1024); for (let i = 0; i < tiny2.length; i++) tiny2[i] *= 1e-6
  e = encodeChannels([tiny2], {})
  cases.quiet_mono = { units_sha256: sha(concatUnits(e.units)), decoded_sha256: sha(bytesOf(decodeUnits(e.units, 1)[0])) }
  writeJson('aea_edge_cases.json', { header_hex_first_272: Buffer.from(hdr.subarray(0, 272)).toString('hex'), header_len: hdr.length, cases })
}
console.log('golden fixtures written to', OUT)
