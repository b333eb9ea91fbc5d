// Writes tests/golden/libm_v8_{log,exp,log1p,log10}.bin: (argument, result) pairs of float64, little endian, of the
// JavaScript engine's Math.log / Math.exp / Math.log1p / Math.log10 -- the functions the reference's transient score
// calls (codec/analysis/transient.js:129, :137, :185, :211).  Run with the Node of this image (12.22, V8 7.8):
//   node tests/golden/gen/gen_libm.mjs
// Arguments: edge cases of the fdlibm branch structure, then 10 000 draws of a fixed xorshift32 stream per function
// (magnitude-like values, values near 1, arbitrary bit patterns, Float32 values).
import fs from 'fs'
import path from 'path'
import { fileURLToPath } from 'url'

const out = path.join(path.dirname(fileURLToPath(import.meta.url)), '..')
let s = 0x9e3779b9 >>> 0
function rnd() { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return s / 4294967296 }
function bits(hi, lo) { const b = Buffer.alloc(8); b.writeUInt32LE(lo >>> 0, 0); b.writeUInt32LE(hi >>> 0, 4); return b.readDoubleLE(0) }
const N = 10000
const edge = [0, -0, 1, -1, 2, 0.5, 10, 1e-10, 1e-300, 5e-324, 1e300, Infinity, -Infinity, NaN, 0.7071067811865476, 1.4142135623730951,
  0.2928932188134524, -0.2928932188134525, -0.29289321881345254, 709.782712893384, 709.7827128933841, -745.1332191019411,
  -745.1332191019412, 0.34657359027997264, 1.0397207708399179, 1 + 2 ** -20, 1 - 2 ** -21, 1 + 2 ** -52, 1 - 2 ** -53, 2 ** -28, 2 ** -29, 2 ** -54]
function inputs(kind) {
  const v = edge.slice()
  for (let i = 0; i < N; i++) {
    const r = rnd(), t = rnd()
    let x
    if (kind === 'log' || kind === 'log10') {
      if (i % 4 === 0) x = Math.pow(10, -12 + 16 * r)
      else if (i % 4 === 1) x = 1 + (r - 0.5) * Math.pow(2, -Math.floor(t * 30))
      else if (i % 4 === 2) x = bits(Math.floor(r * 0x7ff00000) >>> 0, Math.floor(t * 4294967296))
      else x = Math.fround(Math.pow(10, -11 + 14 * r))
    } else if (kind === 'exp') {
      if (i % 4 === 0) x = -30 + 40 * r
      else if (i % 4 === 1) x = (r - 0.5) * Math.pow(2, -Math.floor(t * 40))
      else if (i % 4 === 2) x = -745 + 1455 * r
      else x = (r - 0.5) * 3
    } else {
      if (i % 4 === 0) x = 10 * r
      else if (i % 4 === 1) x = (r - 0.5) * Math.pow(2, -Math.floor(t * 60))
      else if (i % 4 === 2) x = -1 + 2.5 * r
      else x = Math.pow(10, -5 + 25 * r)
    }
    v.push(x)
  }
  return v
}
const fns = { log: Math.log, exp: Math.exp, log1p: Math.log1p, log10: Math.log10 }
for (const k of Object.keys(fns)) {
  const v = inputs(k)
  const b = Buffer.alloc(16 * v.length)
  v.forEach((x, i) => { b.writeDoubleLE(x, 16 * i); b.writeDoubleLE(fns[k](x), 16 * i + 8) })
  fs.writeFileSync(path.join(out, 'libm_v8_' + k + '.bin'), b)
}
console.log('node', process.version, 'v8', process.versions.v8)
