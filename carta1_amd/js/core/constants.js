// Format constants of the ATRAC1 sound unit / AEA container and the numeric tables the device needs.
// The tables are computed HERE, by this process's own Math.sin/cos/pow/sqrt, with the same expressions
// the reference evaluates at module load (codec/core/constants.js:60-66,144-150;
// codec/transforms/mdct.js:27-36; codec/transforms/fft.js:37-39), and handed to the native library
// (c1_set_tables), so the GPU path tracks whatever V8 the host runs -- as the reference itself would.

export const SAMPLE_RATE = 44100
export const SAMPLES_PER_FRAME = 512
export const SOUND_UNIT_SIZE = 212
export const AEA_HEADER_SIZE = 2048
export const AEA_MAGIC = Uint8Array.of(0x00, 0x08, 0x00, 0x00)
export const AEA_TITLE_OFFSET = 4
export const AEA_TITLE_SIZE = 256
export const AEA_FRAME_COUNT_OFFSET = 260
export const AEA_CHANNEL_COUNT_OFFSET = 264
export const NUM_BFUS = 52

export const SPECS_PER_BFU = Int32Array.from(
  [].concat(rep(8, 4), rep(4, 4), rep(8, 4), rep(6, 12), rep(7, 4), rep(9, 4), rep(10, 4), rep(12, 8), rep(20, 8))
)
export const BFU_AMOUNTS = Int32Array.of(20, 28, 32, 36, 40, 44, 48, 52)
export const BFU_START_LONG = (() => {
  const out = new Int32Array(NUM_BFUS)
  for (let b = 1; b < NUM_BFUS; b++) out[b] = out[b - 1] + SPECS_PER_BFU[b - 1]
  return out
})()
export const WORD_LENGTH_BITS = Int32Array.from({ length: 16 }, (_, i) => (i === 0 ? 0 : i + 1))
export const SCALE_FACTORS = Float64Array.from({ length: 64 }, (_, i) => Math.pow(2.0, i / 3.0 - 21))

function rep(v, n) {
  return new Array(n).fill(v)
}

function mdctTable(size, scale) {
  const half = size >> 1
  const t = new Float64Array(half)
  const alpha = (2.0 * Math.PI) / (8.0 * size)
  const omega = (2.0 * Math.PI) / size
  const root = Math.sqrt(scale / size)
  for (let i = 0; i < size >> 2; i++) {
    const angle = omega * i + alpha
    t[2 * i] = root * Math.cos(angle)
    t[2 * i + 1] = root * Math.sin(angle)
  }
  return t
}

// Packed in the field order of `struct c1_tables` (include/carta1_hip.h).
export function buildNativeTables() {
  const parts = [
    SCALE_FACTORS,
    Float64Array.from({ length: 32 }, (_, i) => Math.sin(((i + 0.5) * Math.PI) / 64)),
    mdctTable(64, 0.5),
    mdctTable(256, 0.5),
    mdctTable(512, 1.0),
    mdctTable(64, 64 * 8),
    mdctTable(256, 256 * 8),
    mdctTable(512, 512 * 4),
  ]
  const w = new Float64Array(16)
  for (let k = 0, stride = 2; stride <= 256; stride <<= 1, k++) {
    const angle = (-2 * Math.PI) / stride
    w[2 * k] = Math.cos(angle)
    w[2 * k + 1] = Math.sin(angle)
  }
  parts.push(w, Float64Array.of(Math.log1p(10)))
  const out = new Float64Array(parts.reduce((n, p) => n + p.length, 0))
  let at = 0
  for (const p of parts) {
    out.set(p, at)
    at += p.length
  }
  return out
}
