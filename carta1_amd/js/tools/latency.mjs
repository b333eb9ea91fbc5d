// node latency.mjs  -- round-trip time of the frame closures through the addon (one 512-sample frame per call, a GPU
// round trip each: js/pipeline/encoder.js) next to the reference's 0.233 ms of CPU work per mono frame (SURVEY.md 3.2,
// codec/pipeline/encoder.js:438-450), and of AudioProcessor.encodeStream with batchFrames.  Prints one JSON object.
import * as c1 from '../index.js'

function white(seed, n) {
  const out = new Float32Array(n)
  let s = seed >>> 0
  for (let i = 0; i < n; i++) {
    s ^= s << 13; s >>>= 0; s ^= s >>> 17; s ^= s << 5; s >>>= 0
    out[i] = Math.fround((s / 4294967296 * 2 - 1) * 0.5)
  }
  return out
}

async function main() {
  const frames = 400
  const x = white(1, (frames + 20) * 512)
  const enc = c1.encode(new c1.EncoderOptions())
  const dec = c1.decode()
  const fields = []
  for (let f = 0; f < 20; f++) fields.push(enc(x.subarray(f * 512, (f + 1) * 512).slice()))   // warm-up
  let t0 = process.hrtime.bigint()
  for (let f = 20; f < 20 + frames; f++) fields.push(enc(x.subarray(f * 512, (f + 1) * 512).slice()))
  const encMs = Number(process.hrtime.bigint() - t0) / 1e6 / frames
  for (let f = 0; f < 20; f++) dec(fields[f])
  t0 = process.hrtime.bigint()
  for (let f = 20; f < 20 + frames; f++) dec(fields[f])
  const decMs = Number(process.hrtime.bigint() - t0) / 1e6 / frames
  const out = { encode_closure_ms_per_frame: encMs, decode_closure_ms_per_frame: decMs,
                reference_cpu_ms_per_frame_measured_elsewhere: { encode: 0.233, decode: 0.095 } }
  for (const batch of [16, 64, 256]) {
    const n = 4096
    const src = [white(3, n * 512)]
    const t1 = process.hrtime.bigint()
    const got = await c1.AudioProcessor.collectFrames(c1.AudioProcessor.encodeStream(c1.AudioProcessor.frameBufferToFrames(src), { channelCount: 1, batchFrames: batch }))
    out['encodeStream_batchFrames_' + batch + '_ms_per_frame'] = Number(process.hrtime.bigint() - t1) / 1e6 / got.length
  }
  console.log(JSON.stringify(out))
}
main().catch((e) => { console.log(JSON.stringify({ error: String(e) })); process.exit(1) })
