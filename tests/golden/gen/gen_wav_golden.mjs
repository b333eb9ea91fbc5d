// Golden vectors for the 16-bit WAV sample conversion (codec/io/processor.js:349-447), made by running the
// reference's AudioProcessor.createWavBlob under Node 12.  Node 12 has no global Blob; the stand-in below only
// keeps the ArrayBuffer the reference hands to `new Blob([...])` so it can be read back.
//   cd tests/golden/gen && node --experimental-loader ./loader.mjs gen_wav_golden.mjs
import fs from 'fs'
import path from 'path'
import { fileURLToPath } from 'url'
import { AudioProcessor } from '/root/reference/codec/io/processor.js'
globalThis.Blob = class { constructor(parts, opts) { this.parts = parts; this.type = opts && opts.type } }

const OUT = path.join(path.dirname(fileURLToPath(import.meta.url)), '..')
let s = 0x2545f491
const rnd = () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return s }
const special = [0, -0, 1, -1, 1.5, -1.5, NaN, Infinity, -Infinity, 1e-30, -1e-30, 0.5, -0.5, 0.99999994, -0.99999994,
  1 / 32767, -1 / 32768, 2 / 32767, 0.999969482421875, -0.999969482421875, 3.0517578125e-05, -3.0517578125e-05,
  0.25, -0.25, 0.333333343267, -0.333333343267, 1.00000012, -1.00000012, 32766.5 / 32767, -32767.5 / 32768]
const n = 2048
const mk = () => {
  const a = new Float32Array(n)
  for (let i = 0; i < n; i++) a[i] = i < special.length ? special[i] : (rnd() / 4294967296 * 2.4 - 1.2)
  return a
}
const L = mk(), R = mk().reverse()
const frames = (x) => [x.subarray(0, 512), x.subarray(512, 1024), x.subarray(1024, 1536), x.subarray(1536)]
const body = (blob) => Buffer.from(blob.parts[0]).subarray(44)
const mono = body(AudioProcessor.createWavBlob(frames(L), 1))
const fl = frames(L), fr = frames(R)
const stereo = body(AudioProcessor.createWavBlob(fl.map((l, i) => [l, fr[i]]), 2))
const header = Buffer.from(AudioProcessor.createWavBlob(frames(L), 1).parts[0]).subarray(0, 44)
const hex = (b) => Buffer.from(b.buffer, b.byteOffset, b.byteLength).toString('hex')
fs.writeFileSync(path.join(OUT, 'wav16.json'), JSON.stringify({
  note: 'reference AudioProcessor.createWavBlob: float32 LE input (hex), int16 LE body (hex, after the 44-byte header)',
  samples: n, left_f32: hex(L), right_f32: hex(R), mono_i16: mono.toString('hex'), stereo_i16: stereo.toString('hex'),
  mono_header: header.toString('hex') }, null, 1) + '\n')
console.log('wav16.json', mono.length, stereo.length)
