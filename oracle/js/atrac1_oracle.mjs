// atrac1_oracle.mjs -- the encode half of oracle/atrac1_oracle.c in JavaScript, for the CPU baseline BASELINE.md section 5 asks
// for: "the reference Node.js CPU path timed on the same box's host cores".  The reference itself cannot travel to the GPU box;
// this restatement runs under the box's own Node, so it is the reference's number model executed by the reference's kind of
// engine (Float32Array stores round to binary32, everything else is a double operation, Math.* are the engine's).
//
// TEST INFRASTRUCTURE ONLY, like everything under oracle/: bench.py's cpu_baseline leg and tests/ run it; nothing under
// carta1_amd/ imports it.  Parity status: PINNED -- oracle/js/cpu_baseline.mjs refuses to time it unless it reproduces the
// reference's own golden vectors (tests/golden: the config-1 known answer and four 64-frame stereo runs, fixed modes and
// detection) byte for byte; tests/test_oracle_js.py runs that check on the CPU.
//
// Citations are file:line in aynik/carta1 v1.1.10, as in the C oracle it follows function by function.  Numeric tables are
// read from tests/golden/tables.json (the values the reference's initialisers produced), not recomputed.
import fs from 'fs'
import path from 'path'
import { fileURLToPath } from 'url'

const here = path.dirname(fileURLToPath(import.meta.url))
const T = JSON.parse(fs.readFileSync(path.resolve(here, '../../tests/golden/tables.json')))
const d64 = (h) => Buffer.from(h, 'hex').readDoubleBE(0)
const f32 = (h) => Buffer.from(h, 'hex').readFloatBE(0)

export const SPECS = T.specs_per_bfu, START_LONG = T.bfu_start_long, START_SHORT = T.bfu_start_short
export const BFU_AMOUNTS = T.bfu_amounts, WL_BITS = T.word_length_bits
const SCALE_FACTORS = Float64Array.from(T.scale_factors_f64.map(d64))
const WINDOW = Float64Array.from(T.window_short_f64.map(d64))
const QMF_EVEN = Float32Array.from(T.qmf_even_f32.map(f32)), QMF_ODD = Float32Array.from(T.qmf_odd_f32.map(f32))
const MDCT_FWD = { 64: Float64Array.from(T.mdct_sincos_f64.fwd64.map(d64)), 256: Float64Array.from(T.mdct_sincos_f64.fwd256.map(d64)),
  512: Float64Array.from(T.mdct_sincos_f64.fwd512.map(d64)) }
const FFT_W = {}
for (const s of [2, 4, 8, 16, 32, 64, 128, 256]) FFT_W[s] = T.fft_w_f64[s].map(d64)
const LOG1P_10 = d64(T.log1p_10_f64)
const DDF = Float64Array.from(T.distortion_delta_factors_f64.map(d64)), DBITS = T.word_length_delta_bits

export function biasedTable(bias) {   // buildBiasedScaleFactorTable, bitallocation.js:46-61
  const out = new Float64Array(64)
  for (let i = 0; i < 64; i++) out[i] = bias === 1 ? SCALE_FACTORS[i] : Math.pow(SCALE_FACTORS[i], bias)
  return out
}

export class EncState {   // the encoder half of BufferPool, buffers.js:30-59
  constructor() {
    this.qmfLow = new Float32Array(46); this.qmfMid = new Float32Array(46); this.qmfHigh = new Float32Array(39)
    this.overlap = [new Float32Array(32), new Float32Array(32), new Float32Array(32)]
    this.prevMag = new Float32Array(256)
    this.work = new Float32Array(46 + 512)
  }
}

// qmfAnalysis, qmf.js:19-50
function qmfAnalysis(input, n, delay, low, high, work) {
  work.set(delay, 0)
  work.set(input.subarray(0, n), 46)
  for (let i = 0; i < n / 2; i++) {
    let even = 0, odd = 0
    for (let j = 0; j < 24; j++) {
      even += work[2 * i + 47 - 2 * j] * QMF_EVEN[j]
      odd += work[2 * i + 46 - 2 * j] * QMF_ODD[j]
    }
    low[i] = even + odd
    high[i] = even - odd
  }
  delay.set(work.subarray(n, n + 46))
}

// FFT.fft, fft.js:14-68
function fftInPlace(re, im, size) {
  let bits = 0
  while ((1 << bits) < size) bits++
  for (let i = 0; i < size; i++) {
    let r = 0, t = i
    for (let b = 0; b < bits; b++) { r = (r << 1) | (t & 1); t >>= 1 }
    if (r > i) { let x = re[i]; re[i] = re[r]; re[r] = x; x = im[i]; im[i] = im[r]; im[r] = x }
  }
  for (let stride = 2; stride <= size; stride <<= 1) {
    const half = stride >> 1, wr = FFT_W[stride][0], wi = FFT_W[stride][1]
    for (let start = 0; start < size; start += stride) {
      let tr = 1, ti = 0
      for (let k = 0; k < half; k++) {
        const e = start + k, o = e + half
        const er = re[e], ei = im[e], or = re[o], oi = im[o]
        const xr = or * tr - oi * ti, xi = or * ti + oi * tr
        re[e] = er + xr; im[e] = ei + xi; re[o] = er - xr; im[o] = ei - xi
        const nr = tr * wr - ti * wi
        ti = tr * wi + ti * wr
        tr = nr
      }
    }
  }
}

// MDCT.transform, mdct.js:54-122
const fftRe = new Float32Array(256), fftIm = new Float32Array(256)
function mdctForward(input, size, out) {
  const tab = MDCT_FWD[size], n2 = size >> 1, n4 = size >> 2, n34 = 3 * n4, nfft = n2 >> 1
  const re = fftRe.subarray(0, nfft), im = fftIm.subarray(0, nfft)
  for (let i = 0; i < n4; i += 2) {
    const r = input[n34 - 1 - i] + input[n34 + i], m = input[n4 + i] - input[n4 - 1 - i]
    const c = tab[i], s = tab[i + 1]
    re[i >> 1] = r * c + m * s
    im[i >> 1] = m * c - r * s
  }
  for (let i = n4; i < n2; i += 2) {
    const r = input[n34 - 1 - i] - input[i - n4], m = input[n4 + i] + input[5 * n4 - 1 - i]
    const c = tab[i], s = tab[i + 1]
    re[i >> 1] = r * c + m * s
    im[i >> 1] = m * c - r * s
  }
  fftInPlace(re, im, nfft)
  for (let i = 0; i < nfft; i++) {
    const c = tab[2 * i], s = tab[2 * i + 1], r = re[i], m = im[i]
    out[2 * i] = -r * c - m * s
    out[n2 - 1 - 2 * i] = -r * s + m * c
  }
}

// performFFT, transient.js:17-35
function magnitudes(x, n, mag) {
  const re = fftRe.subarray(0, n), im = fftIm.subarray(0, n)
  re.set(x); im.fill(0)
  fftInPlace(re, im, n)
  for (let i = 0; i < n / 2; i++) mag[i] = Math.sqrt(re[i] * re[i] + im[i] * im[i])
}
function flatness(c) {   // transient.js:120-141
  let sumLog = 0, sumLin = 0, valid = 0
  for (let i = 0; i < c.length; i++) {
    const m = Math.abs(c[i])
    if (m > 1e-10) { sumLog += Math.log(m); sumLin += m; valid++ }
  }
  if (valid === 0) return 0
  const gm = Math.exp(sumLog / valid), am = sumLin / valid
  return am > 1e-10 ? gm / am : 0
}
function hfRatio(c) {   // transient.js:149-164
  let lo = 0, hi = 0
  const n = c.length
  for (let i = 0; i < n / 2; i++) lo += c[i] * c[i]
  for (let i = n / 2; i < n; i++) hi += c[i] * c[i]
  const tot = lo + hi
  return tot > 0 ? hi / tot : 0
}
function transientScore(cur, prev) {   // transient.js:63-226
  let flux = 0, curE = 0
  for (let i = 0; i < cur.length; i++) {
    const cm = Math.abs(cur[i]), pm = Math.abs(prev[i]), d = cm - pm
    if (d > 0) flux += d
    curE += cm * cm
  }
  flux = flux / (Math.sqrt(curE) || 1e-6)
  const flatChange = Math.abs(flatness(cur) - flatness(prev)), hfChange = Math.abs(hfRatio(cur) - hfRatio(prev))
  let ce = 0, pe = 0
  for (let i = 0; i < cur.length; i++) { ce += cur[i] * cur[i]; pe += prev[i] * prev[i] }
  const db = 10 * Math.log10(Math.max(ce, 1e-10) / Math.max(pe, 1e-10))
  const eChange = Math.max(0, db)
  return (flux + Math.sqrt(flatChange) + Math.log1p(hfChange * 10) / LOG1P_10 + Math.min(eChange / 30, 1)) / 4
}

const OFF = [0, 128, 256], LEN = [128, 128, 256], WSTART = [48, 48, 112], MOFF = [0, 64, 128]
const scratch = { low256: new Float32Array(256), high256: new Float32Array(256), bands: new Float32Array(512), coefs: new Float32Array(512),
  mags: new Float32Array(256), in512: new Float32Array(512), in64: new Float32Array(64), spec: new Float32Array(256) }

// applyTailWindowing, encoder.js:309-316
function tailWindow(samples, t0, overlap) {
  for (let i = 0; i < 32; i++) {
    const v = samples[t0 + i]
    overlap[i] = WINDOW[i] * v
    samples[t0 + i] = v * WINDOW[31 - i]
  }
}

// distributeBitsRDO + siftDown, bitallocation.js:203-281, :314-341
const hidx = new Int32Array(52), hpri = new Float32Array(52)
function siftDown(start, size) {
  let i = start
  const iv = hidx[i], pv = hpri[i]
  for (;;) {
    const l = 2 * i + 1, r = l + 1
    let mi = i, mp = pv
    if (l < size && hpri[l] > mp) { mi = l; mp = hpri[l] }
    if (r < size && hpri[r] > mp) mi = r
    if (mi === i) break
    hidx[i] = hidx[mi]; hpri[i] = hpri[mi]
    i = mi
  }
  hidx[i] = iv; hpri[i] = pv
}
function distributeBits(n, remaining, bsf, sfi, wl) {
  let hs = 0
  wl.fill(0)
  for (let b = 0; b < n; b++) {
    if (sfi[b] === 0) continue
    hidx[hs] = b
    hpri[hs] = (bsf[sfi[b]] * DDF[0]) / DBITS[0]
    hs++
  }
  if (hs === 0) return
  for (let i = (hs >> 1) - 1; i >= 0; i--) siftDown(i, hs)
  while (remaining > 0 && hs > 0) {
    const b = hidx[0], cur = wl[b], cost = DBITS[cur] * SPECS[b]
    if (cost > remaining || cost <= 0) {
      hidx[0] = hidx[hs - 1]; hpri[0] = hpri[hs - 1]; hs--
      if (hs > 0) siftDown(0, hs)
      continue
    }
    remaining -= cost
    const nxt = cur + 1
    wl[b] = nxt
    if (nxt < 15) {
      hpri[0] = (bsf[sfi[b]] * DDF[nxt]) / DBITS[nxt]
      siftDown(0, hs)
    } else {
      hidx[0] = hidx[hs - 1]; hpri[0] = hpri[hs - 1]; hs--
      if (hs > 0) siftDown(0, hs)
    }
  }
}

const zeroBit = new Float32Array(52), wlTry = new Int32Array(52)
const POW2_NEG = Float64Array.from({ length: 17 }, (_, b) => Math.pow(2, -b))   // INV_POWER_OF_TWO (exact)

// One frame of one channel through the encode() closure (encoder.js:438-450) and serializeFrame (serialization.js:41-98).
// opts: { fixedModes: [a, b, c] | null, threshold, biased: Float64Array(64) }.  Writes 212 bytes at unit[at ..].
export function encodeFrame(state, pcm, opts, unit, at) {
  const S = scratch, bands = S.bands, coefs = S.coefs
  // qmfAnalysisStage, encoder.js:57-96
  qmfAnalysis(pcm, 512, state.qmfLow, S.low256, S.high256, state.work)
  qmfAnalysis(S.low256, 256, state.qmfMid, bands.subarray(0, 128), bands.subarray(128, 256), state.work)
  bands.set(state.qmfHigh, 256)
  bands.set(S.high256.subarray(0, 217), 295)
  state.qmfHigh.set(S.high256.subarray(217, 256))
  // blockSelectorStage, encoder.js:111-152
  const modes = [0, 0, 0]
  if (opts.fixedModes) { modes[0] = opts.fixedModes[0]; modes[1] = opts.fixedModes[1]; modes[2] = opts.fixedModes[2] } else {
    for (let b = 0; b < 3; b++) magnitudes(bands.subarray(OFF[b], OFF[b] + LEN[b]), LEN[b], S.mags.subarray(MOFF[b], MOFF[b] + LEN[b] / 2))
    for (let b = 0; b < 3; b++) {
      const cur = S.mags.subarray(MOFF[b], MOFF[b] + LEN[b] / 2), prev = state.prevMag.subarray(MOFF[b], MOFF[b] + LEN[b] / 2)
      modes[b] = (transientScore(cur, prev) > opts.threshold ? 1 : 0) * Math.max(b + 1, 2)
    }
    state.prevMag.set(S.mags)
  }
  // mdctStage, encoder.js:170-349
  for (let b = 0; b < 3; b++) {
    const x = bands.subarray(OFF[b], OFF[b] + LEN[b]), ov = state.overlap[b], dst = coefs.subarray(OFF[b], OFF[b] + LEN[b])
    if (modes[b] === 0) {
      const size = b === 2 ? 512 : 256, inp = S.in512.subarray(0, size)
      inp.fill(0)
      inp.set(ov, WSTART[b])
      tailWindow(x, LEN[b] - 32, ov)
      inp.set(x, WSTART[b] + 32)
      mdctForward(inp, size, S.spec)
      if (b > 0) { for (let i = 0; i < LEN[b]; i++) dst[i] = S.spec[LEN[b] - 1 - i] } else dst.set(S.spec.subarray(0, LEN[b]))
    } else {
      for (let k = 0; k < LEN[b] / 32; k++) {
        S.in64.set(ov, 0)
        tailWindow(x, 32 * k, ov)
        S.in64.set(x.subarray(32 * k, 32 * k + 32), 32)
        mdctForward(S.in64, 64, S.spec)
        if (b > 0) { for (let i = 0; i < 32; i++) dst[32 * k + i] = S.spec[31 - i] } else dst.set(S.spec.subarray(0, 32), 32 * k)
      }
    }
  }
  // quantizationStage: groupIntoBFUs + allocateBits (bitallocation.js:74-142)
  const bsf = opts.biased
  const sfi = new Int32Array(52)
  const start = (b) => (modes[b >= 36 ? 2 : b >= 20 ? 1 : 0] === 0 ? START_LONG : START_SHORT)[b]
  for (let b = 0; b < 52; b++) {
    const s0 = start(b)
    let m = 0
    for (let i = 0; i < SPECS[b]; i++) { const a = Math.abs(coefs[s0 + i]); if (a > m) m = a }
    let s = 0
    if (m !== 0) { while (s < 63 && m > SCALE_FACTORS[s]) s++ }       // findScaleFactor (:290-299): smallest i with m <= SCALE_FACTORS[i]
    sfi[b] = s
    zeroBit[b] = s > 0 ? bsf[s] * 2 * SPECS[b] : 0
  }
  let best = Infinity, bestN = -1
  let wl = new Int32Array(52)
  for (let c = 0; c < 8; c++) {
    const n = BFU_AMOUNTS[c]
    distributeBits(n, 212 * 8 - 40 - n * 10, bsf, sfi, wlTry)
    let total = 0
    for (let b = 0; b < n; b++) {
      const bits = WL_BITS[wlTry[b]]
      if (bits === 0) { total += zeroBit[b]; continue }
      if (sfi[b] === 0) continue
      total += bsf[sfi[b]] * POW2_NEG[bits] * SPECS[b]
    }
    for (let b = n; b < 52; b++) total += zeroBit[b]
    if (total < best) { best = total; bestN = n; wl.set(wlTry) }
  }
  if (bestN < 0) { bestN = BFU_AMOUNTS[0]; wl.fill(0); sfi.fill(0) }
  // quantize (quantization.js:34-56) + serializeFrame (serialization.js:41-98), MSB first
  unit.fill(0, at, at + 212)
  let pos = at * 8
  const put = (v, n) => { for (let k = n - 1; k >= 0; k--, pos++) if ((v >>> k) & 1) unit[pos >> 3] |= 0x80 >> (pos & 7) }
  put((((2 - modes[0]) << 14) | ((2 - modes[1]) << 12) | ((3 - modes[2]) << 10) | (BFU_AMOUNTS.indexOf(bestN) << 5)) & 0xffff, 16)
  for (let b = 0; b < bestN; b++) put(wl[b], 4)
  for (let b = 0; b < bestN; b++) put(sfi[b], 6)
  for (let b = 0; b < bestN; b++) {
    const bits = WL_BITS[wl[b]]
    if (bits === 0) continue
    const s0 = start(b)
    if (sfi[b] === 0) { pos += bits * SPECS[b]; continue }
    const range = (1 << (bits - 1)) - 1, norm = range / SCALE_FACTORS[sfi[b]]
    for (let i = 0; i < SPECS[b]; i++) {
      const v = coefs[s0 + i] * norm
      const y = (v + (v >= 0 ? 0.5 : -0.5)) | 0
      put((y > range ? range : y < -range ? -range : y) & ((1 << bits) - 1), bits)
    }
  }
}

// encodeAeaPcm's hot loop (processor.js:119-136): frames of every channel in turn, units interleaved L, R
export function encodeStream(channels, frames, opts, states, units) {
  for (let f = 0; f < frames; f++)
    for (let c = 0; c < channels.length; c++)
      encodeFrame(states[c], channels[c].subarray(f * 512, (f + 1) * 512), opts, units, (f * channels.length + c) * 212)
}

// synthetic signals of SURVEY.md 8c (xorshift32)
export function xorshift(seed) {
  let s = seed >>> 0
  return () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return (s / 4294967296) * 2 - 1 }
}
export function white(seed, n, skip = 0) {
  const r = xorshift(seed), x = new Float32Array(n)
  for (let i = 0; i < skip; i++) r()
  for (let i = 0; i < n; i++) x[i] = r() * 0.5
  return x
}
export function pinkT(seed, n) {
  const r = xorshift(seed), x = new Float32Array(n)
  let p = 0
  for (let i = 0; i < n; i++) {
    const u = r(); p = 0.98 * p + 0.05 * u
    let v = p
    if ((i >> 9) % 8 === 5 && (i % 512) >= 256) v += 0.8 * r()
    x[i] = v
  }
  return x
}
