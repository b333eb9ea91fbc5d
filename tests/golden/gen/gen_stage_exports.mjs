// Golden vectors for the single-stage functions the reference exports next to encode()/decode() (codec/index.js:30-35,42):
// FFT.fft, qmfAnalysisStage, mdctStage (with the in-place windowing it leaves in the band arrays), quantize / dequantize at
// unusual word lengths.  Runs the JavaScript reference in place from /root/reference through loader.mjs and writes
// tests/golden/stage_exports.json -- inputs and outputs only, never reference source text.
//
//   cd tests/golden/gen && node --experimental-loader ./loader.mjs gen_stage_exports.mjs
import fs from 'fs'
import path from 'path'
import { fileURLToPath } from 'url'

import { FFT } from '/root/reference/codec/transforms/fft.js'
import { BufferPool } from '/root/reference/codec/core/buffers.js'
import { quantize, dequantize } from '/root/reference/codec/coding/quantization.js'
import { qmfAnalysisStage, mdctStage } from '/root/reference/codec/pipeline/encoder.js'

const OUT = path.resolve(path.dirname(fileURLToPath(import.meta.url)), '..')
const hex32 = (ta) => Buffer.from(ta.buffer, ta.byteOffset, ta.byteLength).toString('hex')     // little-endian element bytes

function xorshift(seed) {
  let s = seed >>> 0
  return () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return (s / 4294967296) * 2 - 1 }
}
function white(seed, n, amp = 0.5) {
  const r = xorshift(seed); const x = new Float32Array(n)
  for (let i = 0; i < n; i++) x[i] = Math.fround(r() * amp)
  return x
}

const out = { note: 'little-endian hex of Float32Array / Int32Array contents; inputs from xorshift32 (SURVEY.md 8c): white(seed, n, amp)' }

// FFT.fft in place, complex input
out.fft = []
for (const [n, seed] of [[2, 41], [8, 42], [64, 43], [256, 44], [1024, 45]]) {
  const real = white(seed, n, 1.0), imag = white(seed + 100, n, 1.0)
  const rec = { n, seed_real: seed, seed_imag: seed + 100, amp: 1.0 }
  FFT.fft(real, imag)
  rec.real = hex32(real); rec.imag = hex32(imag)
  out.fft.push(rec)
}

// qmfAnalysisStage -> mdctStage over consecutive frames of one stream, for several fixed block modes; what the stage
// functions return per frame (bands as mdctStage leaves them, coefficients)
out.stages = []
for (const modes of [[0, 0, 0], [2, 2, 3], [0, 2, 0], [2, 0, 3]]) {
  const pool = new BufferPool()
  const ctx = { bufferPool: pool }
  const qmf = qmfAnalysisStage(ctx), mdct = mdctStage(ctx)
  const pcm = white(51, 4 * 512)
  const frames = []
  for (let f = 0; f < 4; f++) {
    const a = qmf(pcm.subarray(f * 512, (f + 1) * 512))
    const raw = a.bands.map((b) => hex32(b))
    const r = mdct({ bands: a.bands, blockModes: modes, originalFrame: null })
    frames.push({ bands_raw: raw, bands_after: r.bands.map((b) => hex32(b)), coefficients: hex32(r.coefficients) })
  }
  out.stages.push({ seed: 51, modes, frames })
}

// quantize / dequantize at the ends of the word-length range
out.quantize = []
for (const [sfi, bits, seed, amp] of [[30, 2, 61, 0.004], [45, 16, 62, 0.05], [63, 12, 63, 1.5], [5, 3, 64, 1e-6], [20, 0, 65, 0.1], [0, 8, 66, 0.1]]) {
  const x = white(seed, 20, amp)
  const q = quantize(x, sfi, bits)
  out.quantize.push({ sfi, bits, seed, amp, n: 20, q: Array.from(q), d: hex32(dequantize(q, sfi, bits)) })
}

fs.writeFileSync(path.join(OUT, 'stage_exports.json'), JSON.stringify(out, null, 1) + '\n')
console.log('wrote stage_exports.json')
