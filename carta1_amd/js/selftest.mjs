// node selftest.mjs [--gpu]   -- used by tests/test_js_host.py
// Without --gpu: host-side checks only (tables, serialization, options, addon loads and fails loudly).
// With --gpu: encode/decode through the addon against the reference's golden vectors.
import fs from 'fs'
import path from 'path'
import { fileURLToPath } from 'url'
import * as c1 from './index.js'
import { buildNativeTables } from './core/constants.js'
import { native } from './native.js'

const here = path.dirname(fileURLToPath(import.meta.url))
const G = path.resolve(here, '../../tests/golden')
const gpu = process.argv.includes('--gpu')
let failed = 0
const ok = (cond, what) => { if (!cond) { failed++; console.log('FAIL', what) } else console.log('ok  ', what) }
const hex = (u8) => Buffer.from(u8.buffer, u8.byteOffset, u8.byteLength).toString('hex')
const f64hex = (x) => { const b = Buffer.alloc(8); b.writeDoubleBE(x); return b.toString('hex') }

function xorshift(seed) { let s = seed >>> 0; return () => { s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0; return (s / 4294967296) * 2 - 1 } }
function white(seed, n) { const r = xorshift(seed); const x = new Float32Array(n); for (let i = 0; i < n; i++) x[i] = Math.fround(r() * 0.5); return x }
function pinkT(seed, n) { const r = xorshift(seed); const x = new Float32Array(n); let p = 0; for (let i = 0; i < n; i++) { const u = r(); p = 0.98 * p + 0.05 * u; let v = p; if ((i >> 9) % 8 === 5 && (i % 512) >= 256) v += 0.8 * r(); x[i] = v } return x }

async function main() {
  // tables computed by this V8 == the fixture (made by the reference under the generator's V8)
  const t = JSON.parse(fs.readFileSync(path.join(G, 'tables.json')))
  const mine = buildNativeTables()
  const want = [].concat(t.scale_factors_f64, t.window_short_f64, t.mdct_sincos_f64.fwd64, t.mdct_sincos_f64.fwd256,
    t.mdct_sincos_f64.fwd512, t.mdct_sincos_f64.inv64, t.mdct_sincos_f64.inv256, t.mdct_sincos_f64.inv512,
    [2, 4, 8, 16, 32, 64, 128, 256].flatMap((s) => t.fft_w_f64[s]), [t.log1p_10_f64])
  ok(mine.length === want.length && want.every((h, i) => f64hex(mine[i]) === h), 'host tables equal the reference tables bit for bit')
  ok(Array.from(c1.BFU_START_LONG).join() === t.bfu_start_long.join() && Array.from(c1.SPECS_PER_BFU).join() === t.specs_per_bfu.join(), 'BFU layout tables')
  for (const b of ['0.5', '2', '3.3']) {
    const o = new c1.EncoderOptions({ allocationBias: Number(b) }).toNative()
    ok(t.biased_scale_factors_f64[b].every((h, i) => f64hex(o[i]) === h), `biased scale factors, bias ${b}`)
  }
  // options behave like the reference's
  let msg = ''
  try { new c1.EncoderOptions({ allocationBias: 9 }) } catch (e) { msg = e.message }
  ok(msg === 'Value for allocationBias must be between 0 and 5, got 9', 'option range error message')
  const eo = new c1.EncoderOptions({ fixedBlockModes: [0, 2, 3], ignored: 1 })
  ok(eo.fixedBlockModes.join() === '0,2,3' && eo.transientThresholdMid === 1.5 && eo.getValue('allocationBias') === 1, 'option accessors')
  // sound unit <-> fields
  const k = JSON.parse(fs.readFileSync(path.join(G, 'config1_sine1k.json')))
  const unit = Uint8Array.from(Buffer.from(k.unit_hex, 'hex'))
  const f = c1.deserializeFrame(unit)
  ok(f.nBfu === 44 && Array.from(f.wordLengthIndices).join() === k.wordLengthIndices.join() &&
     Array.from(f.scaleFactorIndices).join() === k.scaleFactorIndices.join() &&
     f.quantizedCoefficients.every((q, i) => Array.from(q).join() === k.quantizedCoefficients[i].join()), 'deserializeFrame == reference fields')
  ok(hex(c1.serializeFrame(f)) === k.unit_hex, 'serializeFrame round trip')
  const kat = fs.readFileSync(path.join(G, 'kat64_pinkT_detect.units.bin'))
  let same = true
  for (let i = 0; i < 128; i++) { const u = Uint8Array.from(kat.subarray(i * 212, (i + 1) * 212)); if (hex(c1.serializeFrame(c1.deserializeFrame(u))) !== hex(u)) same = false }
  ok(same, 'serialize(deserialize(u)) == u on 128 reference units')
  const e = JSON.parse(fs.readFileSync(path.join(G, 'aea_edge_cases.json')))
  ok(hex(c1.AeaFile.createHeader('encoded by carta1', 4, 2).subarray(0, 272)) === e.header_hex_first_272, 'AEA header bytes')
  const info = c1.AeaFile.parseHeader(c1.AeaFile.createHeader('t', 7, 2))
  ok(info.title === 't' && info.frameCount === 7 && info.channelCount === 2, 'AEA header parse')
  msg = ''
  try { c1.deserializeFrame(new Uint8Array(5)) } catch (err) { msg = err.message }
  ok(msg === 'Frame must be 212 bytes', 'deserializeFrame length error')
  let rejected = null
  try { await c1.encodeAeaPcm([new Float64Array(4)]) } catch (err) { rejected = err }
  ok(rejected instanceof TypeError && rejected.message === 'ATRAC1 encoding requires one or two Float32 channels', 'encodeAeaPcm TypeError')
  rejected = null
  try { await c1.decodeAeaPcm('nope') } catch (err) { rejected = err }
  ok(rejected instanceof TypeError && rejected.message === 'ATRAC1 decoding requires AEA bytes or a Blob', 'decodeAeaPcm TypeError')
  ok(native().abiVersion() === 3, 'addon loads')
  // every name the reference exports (codec/index.js:26-47) is exported here, with the kind of value it is there
  {
    const fns = ['pipe', 'encode', 'decode', 'qmfAnalysisStage', 'mdctStage', 'serializeFrame', 'deserializeFrame', 'quantize', 'dequantize',
      'decodeAeaPcm', 'encodeAeaPcm']
    const classes = ['AeaFile', 'BufferPool', 'EncoderOptions', 'AudioProcessor', 'FFT']
    const tables = ['WORD_LENGTH_BITS', 'SPECS_PER_BFU', 'SCALE_FACTORS', 'BFU_START_LONG']
    ok(fns.every((n) => typeof c1[n] === 'function') && classes.every((n) => typeof c1[n] === 'function' && c1[n].prototype) &&
       tables.every((n) => c1[n] && c1[n].length > 0) && typeof c1.FFT.fft === 'function' && fns.length + classes.length + tables.length === 20,
       'all twenty exports of the reference\'s codec/index.js are present')
    // call shapes (no device needed): the stage factories take a context with a bufferPool and throw the reference's messages without one
    let m1 = '', m2 = ''
    try { c1.qmfAnalysisStage({}) } catch (x) { m1 = x.message }
    try { c1.mdctStage(null) } catch (x) { m2 = x.message }
    ok(m1 === 'qmfAnalysisStage: bufferPool is required' && m2 === 'mdctStage: bufferPool is required' &&
       typeof c1.qmfAnalysisStage({ bufferPool: new c1.BufferPool() }) === 'function' && typeof c1.mdctStage({ bufferPool: new c1.BufferPool() }) === 'function',
       'qmfAnalysisStage / mdctStage factories: context handling as in the reference')
    const one = new Float32Array([3]), zero = new Float32Array([0])
    c1.FFT.fft(one, zero)
    ok(one[0] === 3 && zero[0] === 0, 'FFT.fft of size 1 returns at once (fft.js:16)')
  }

  if (!gpu) {
    let err = null
    try { await c1.encodeAeaPcm([new Float32Array(512)]) } catch (x) { err = x }
    ok(err && /no HIP device|no ROCm/.test(err.message), 'no GPU -> the hot path throws (no CPU fallback): ' + (err && err.message))
  } else {
    // config 1 through the frame closure
    const pcm = new Float32Array(512)
    for (let i = 0; i < 512; i++) pcm[i] = Math.sin((2 * Math.PI * 1000 * i) / 44100)
    const enc = c1.encode(new c1.EncoderOptions({ fixedBlockModes: [0, 0, 0] }))
    const r = enc(pcm)
    ok(hex(c1.serializeFrame(r)) === k.unit_hex && r.quantizedCoefficients.length === 44, 'encode() closure == config-1 known answer')
    // 64-frame stereo known answers through encodeAeaPcm / decodeAeaPcm
    const idx = JSON.parse(fs.readFileSync(path.join(G, 'kat_index.json')))
    for (const name of Object.keys(idx)) {
      const c = idx[name]
      const n = 64 * 512
      const chs = c.signal === 'white' ? [white(1, n), white(2, n)] : [pinkT(3, n), pinkT(4, n)]
      const img = await c1.encodeAeaPcm(chs, Object.assign({ title: 'kat' }, c.options))
      const ref = fs.readFileSync(path.join(G, `kat64_${name}.units.bin`))
      const got = Buffer.from(img.buffer, img.byteOffset + 2048, img.length - 2048)
      ok(got.equals(ref) && img.length === 2048 + 128 * 212, `encodeAeaPcm ${name}: units == reference`)
      const pcmOut = await c1.decodeAeaPcm(img)
      const head = fs.readFileSync(path.join(G, `kat64_${name}.pcm8.bin`))
      const l8 = Buffer.from(pcmOut[0].buffer, pcmOut[0].byteOffset, 8 * 512 * 4)
      const r8 = Buffer.from(pcmOut[1].buffer, pcmOut[1].byteOffset, 8 * 512 * 4)
      ok(pcmOut.length === 2 && pcmOut[0].length === n && l8.equals(head.subarray(0, 16384)) && r8.equals(head.subarray(16384)), `decodeAeaPcm ${name}: PCM == reference`)
    }
    // the single-stage exports against the reference's own outputs (tests/golden/quantize.json, stage_exports.json)
    {
      const be32 = (h) => Buffer.from(h, 'hex').readFloatBE(0)
      const qv = JSON.parse(fs.readFileSync(path.join(G, 'quantize.json')))
      let okQ = true, okD = true
      for (const v of qv) {
        const x = Float32Array.from(v.x.map(be32))
        const q = c1.quantize(x, v.sfi, v.bits)
        if (!(q instanceof Int32Array) || Array.from(q).join() !== v.q.join()) okQ = false
        const d = c1.dequantize(Int32Array.from(v.q), v.sfi, v.bits)
        if (!(d instanceof Float32Array) || !v.d.every((h, i) => Object.is(be32(h), d[i]))) okD = false
      }
      ok(okQ, `quantize == reference on ${qv.length} vectors (incl. the | 0 wrap)`)
      ok(okD, 'dequantize == reference on the same vectors')
      const sx = JSON.parse(fs.readFileSync(path.join(G, 'stage_exports.json')))
      const wh = (seed, n, amp) => { const r = xorshift(seed); const x = new Float32Array(n); for (let i = 0; i < n; i++) x[i] = Math.fround(r() * amp); return x }
      let okE = true
      for (const v of sx.quantize) {
        const q = c1.quantize(wh(v.seed, v.n, v.amp), v.sfi, v.bits)
        if (Array.from(q).join() !== v.q.join() || hex(c1.dequantize(q, v.sfi, v.bits)) !== v.d) okE = false
      }
      ok(okE, 'quantize / dequantize at 2, 3, 12 and 16 bits, bits 0 and scale factor 0 == reference')
      let okF = true
      for (const v of sx.fft) {
        const re = wh(v.seed_real, v.n, v.amp), im = wh(v.seed_imag, v.n, v.amp)
        c1.FFT.fft(re, im)
        if (hex(re) !== v.real || hex(im) !== v.imag) okF = false
      }
      ok(okF, 'FFT.fft in place, sizes 2..1024 == reference bit for bit')
      let okB = true, okC = true, okW = true
      for (const run of sx.stages) {
        const pool = new c1.BufferPool()
        const qmf = c1.qmfAnalysisStage({ bufferPool: pool }), mdct = c1.mdctStage({ bufferPool: pool })
        const pcm = wh(run.seed, 4 * 512, 0.5)
        for (let fi = 0; fi < 4; fi++) {
          const a = qmf(pcm.subarray(fi * 512, (fi + 1) * 512).slice())
          if (a.bands.length !== 3 || !a.bands.every((b, i) => hex(b) === run.frames[fi].bands_raw[i])) okB = false
          const r = mdct({ bands: a.bands, blockModes: run.modes, originalFrame: 'tag' })
          if (hex(r.coefficients) !== run.frames[fi].coefficients || r.coefficients.length !== 512) okC = false
          if (r.bands !== a.bands || r.originalFrame !== 'tag' || !r.bands.every((b, i) => hex(b) === run.frames[fi].bands_after[i])) okW = false
        }
      }
      ok(okB, 'qmfAnalysisStage over 4 consecutive frames == reference bands')
      ok(okC, 'mdctStage, block modes [0,0,0] [2,2,3] [0,2,0] [2,0,3] == reference coefficients')
      ok(okW, 'mdctStage hands on the same band arrays, windowed in place as the reference leaves them')
      // pipe() of the two stages, as an application would compose them
      const p = c1.pipe({ bufferPool: new c1.BufferPool() }, c1.qmfAnalysisStage, () => (x) => Object.assign(x, { blockModes: [0, 0, 0] }), c1.mdctStage)
      const r0 = p(wh(51, 512, 0.5))
      ok(hex(r0.coefficients) === sx.stages[0].frames[0].coefficients, 'pipe(qmfAnalysisStage, ..., mdctStage) == reference')
    }
    // frame closures continue a stream; decode() closure on frame fields
    const x = pinkT(3, 8 * 512)
    const e2 = c1.encode(new c1.EncoderOptions())
    const d2 = c1.decode()
    const ref = fs.readFileSync(path.join(G, 'kat64_pinkT_detect.units.bin'))
    const refPcm = fs.readFileSync(path.join(G, 'kat64_pinkT_detect.pcm8.bin'))
    let okUnits = true, okPcm = true
    for (let fidx = 0; fidx < 8; fidx++) {
      const fields = e2(x.subarray(fidx * 512, (fidx + 1) * 512).slice())
      if (!Buffer.from(c1.serializeFrame(fields)).equals(ref.subarray(fidx * 2 * 212, fidx * 2 * 212 + 212))) okUnits = false
      const out = d2(fields)
      if (!Buffer.from(out.buffer, out.byteOffset, 2048).equals(refPcm.subarray(fidx * 2048, (fidx + 1) * 2048))) okPcm = false
    }
    ok(okUnits, 'encode() closure, 8 consecutive frames with detection == reference units')
    ok(okPcm, 'decode() closure, 8 consecutive frames == reference PCM')
    // 700-sample stereo (processor.test.js:95-108)
    const img = await c1.encodeAeaPcm([white(11, 700), white(12, 700)])
    ok(hex(img.subarray(2048)) === e.cases.stereo700.units_hex, 'ragged 700-sample stereo == reference')
    const out = await c1.decodeAeaPcm(img)
    ok(out.length === 2 && out[0].length === 1024 && out[1].length === 1024, 'decodeAeaPcm sizes')
    // streams API
    const frames = c1.AudioProcessor.frameBufferToFrames([white(11, 700), white(12, 700)])
    const fieldsList = await c1.AudioProcessor.collectFrames(c1.AudioProcessor.encodeStream(frames, { channelCount: 2 }))
    ok(fieldsList.length === 4 && hex(await c1.AudioProcessor.createAeaBytes(fieldsList, { title: 'encoded by carta1', channelCount: 2 })) === hex(img), 'AudioProcessor.encodeStream == encodeAeaPcm')
    // batched stream encoding: same frame fields as the per-frame closures
    {
      const src = [pinkT(3, 100 * 512), pinkT(4, 100 * 512)]
      const one = await c1.AudioProcessor.collectFrames(c1.AudioProcessor.encodeStream(c1.AudioProcessor.frameBufferToFrames(src), { channelCount: 2 }))
      const many = await c1.AudioProcessor.collectFrames(c1.AudioProcessor.encodeStream(c1.AudioProcessor.frameBufferToFrames(src), { channelCount: 2, batchFrames: 37 }))
      ok(one.length === 200 && many.length === 200 && one.every((f, i) => hex(c1.serializeFrame(f)) === hex(c1.serializeFrame(many[i]))), 'encodeStream with batchFrames 37 == per-frame closures')
    }
    // 16-bit WAV body in one native call == float conversion (value / 32768, bin/cli.js:394-404) + encodeAeaPcm
    {
      const ns = 700, wav = new Int16Array(ns * 2)
      const l = new Float32Array(ns), r = new Float32Array(ns)
      const a = white(31, ns), b = white(32, ns)
      for (let i = 0; i < ns; i++) {
        wav[2 * i] = Math.round(a[i] * 40000); wav[2 * i + 1] = Math.round(b[i] * 40000)
        l[i] = wav[2 * i] / 32768; r[i] = wav[2 * i + 1] / 32768
      }
      const viaFloat = await c1.encodeAeaPcm([l, r], {})
      const viaWav = c1.encodeWavPcm(wav, { channelCount: 2 })
      ok(hex(viaFloat) === hex(viaWav), 'encodeWavPcm(Int16Array) == encodeAeaPcm(float conversion)')
      const back = c1.decodeAeaToWav16(viaWav)
      const pcm = await c1.decodeAeaPcm(viaWav)
      let same = back.channelCount === 2 && back.samples.length === pcm[0].length * 2
      for (let i = 0; same && i < pcm[0].length; i++) {
        for (let c = 0; c < 2; c++) {
          const x = Math.max(-1, Math.min(1, pcm[c][i]))
          const want = (x < 0 ? x * 0x8000 : x * 0x7fff) | 0
          if (back.samples[2 * i + c] !== want) same = false
        }
      }
      ok(same, 'decodeAeaToWav16 == decodeAeaPcm + the 16-bit conversion of createWavBlob')
    }
    // two un-awaited encodes on the shared context (libuv pool threads): each must equal its sequential result
    {
      const a = [white(41, 300 * 512), white(42, 300 * 512)], b = [pinkT(43, 300 * 512), pinkT(44, 300 * 512)]
      const seqA = await c1.encodeAeaPcm(a, { fixedBlockModes: [0, 0, 0] })
      const seqB = await c1.encodeAeaPcm(b, {})
      const [parA, parB, parA2] = await Promise.all([c1.encodeAeaPcm(a, { fixedBlockModes: [0, 0, 0] }), c1.encodeAeaPcm(b, {}),
                                                     c1.encodeAeaPcm(a, { fixedBlockModes: [0, 0, 0] })])
      ok(hex(seqA) === hex(parA) && hex(seqB) === hex(parB) && hex(seqA) === hex(parA2), 'Promise.all of encodes on one context == sequential results')
      const [pa, pb] = await Promise.all([c1.decodeAeaPcm(seqA), c1.decodeAeaPcm(seqB)])
      const sa = await c1.decodeAeaPcm(seqA), sb = await c1.decodeAeaPcm(seqB)
      const same = (x, y) => Buffer.from(x.buffer, x.byteOffset, x.byteLength).equals(Buffer.from(y.buffer, y.byteOffset, y.byteLength))
      ok(same(pa[0], sa[0]) && same(pa[1], sa[1]) && same(pb[0], sb[0]) && same(pb[1], sb[1]), 'Promise.all of decodes on one context == sequential results')
    }
    // the batch sharded over device contexts (here the same device three times): same bytes, same PCM
    {
      const src = [pinkT(61, 500 * 512), white(62, 500 * 512)]
      const one = await c1.encodeAeaPcm(src, {})
      const three = await c1.encodeAeaPcm(src, { devices: [0, 0, 0] })
      ok(hex(one) === hex(three), 'encodeAeaPcm sharded over devices [0,0,0] == one device')
      const p1 = await c1.decodeAeaPcm(one), p3 = await c1.decodeAeaPcm(one, { devices: [0, 0, 0] })
      const same = (x, y) => Buffer.from(x.buffer, x.byteOffset, x.byteLength).equals(Buffer.from(y.buffer, y.byteOffset, y.byteLength))
      ok(same(p1[0], p3[0]) && same(p1[1], p3[1]), 'decodeAeaPcm sharded over devices [0,0,0] == one device')
    }
    // a stereo image that ends on a lone left unit: both decode entry points pair it with the dummy unit
    {
      const img = await c1.encodeAeaPcm([white(51, 3 * 512), white(52, 3 * 512)], {})
      const odd = img.slice(0, img.length - 212)
      const pcm = await c1.decodeAeaPcm(odd)
      const w16 = c1.decodeAeaToWav16(odd)
      let same = pcm[0].length === 3 * 512 && w16.samples.length === 3 * 512 * 2
      for (let i = 0; same && i < pcm[0].length; i++) {
        for (let c = 0; c < 2; c++) {
          const x = Math.max(-1, Math.min(1, pcm[c][i]))
          if (w16.samples[2 * i + c] !== ((x < 0 ? x * 0x8000 : x * 0x7fff) | 0)) same = false
        }
      }
      ok(same, 'odd unit count: decodeAeaToWav16 == decodeAeaPcm (dummy right unit)')
    }
    // page-locked PCM: a batch of more than one streaming chunk (32768 frames) must give the same bytes
    {
      const nf = 40000, plain = white(21, nf * 512)
      const pinned = c1.allocPinnedFloat32Array(nf * 512)
      pinned.set(plain)
      const a = await c1.encodeAeaPcm([plain], { fixedBlockModes: [0, 0, 0] })
      const b = await c1.encodeAeaPcm([pinned], { fixedBlockModes: [0, 0, 0] })
      ok(Buffer.from(a.buffer, a.byteOffset, a.length).equals(Buffer.from(b.buffer, b.byteOffset, b.length)), 'encodeAeaPcm from page-locked PCM (streamed) == plain')
      const back = await c1.decodeAeaPcm(b)
      ok(back.length === 1 && back[0].length === nf * 512, 'decodeAeaPcm of 40000 frames (page-locked output buffers)')
    }
  }
  console.log(failed ? `${failed} FAILED` : 'ALL OK')
  process.exit(failed ? 1 : 0)
}
main().catch((e) => { console.log('EXCEPTION', e); process.exit(2) })
