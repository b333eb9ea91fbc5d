// node oracle/js/cpu_baseline.mjs [--check] [--frames N] [--threads T]
// The CPU baseline in the reference's own language (BASELINE.md section 5 items 1-2): encode of BASELINE configs[1]'s workload
// (stereo white noise, fixedBlockModes [0,0,0], bias 1) by oracle/js/atrac1_oracle.mjs under this host's Node -- one thread,
// then worker_threads x cores on disjoint frame ranges (each from its own 2-frame PCM history, as the product's shards are).
// Refuses to time anything unless the restatement first reproduces the reference's golden vectors byte for byte.
// Prints one JSON line.  --check: parity only.
import fs from 'fs'
import os from 'os'
import path from 'path'
import { fileURLToPath } from 'url'
import { Worker, isMainThread, parentPort, workerData } from 'worker_threads'
import { EncState, encodeStream, biasedTable, white, pinkT } from './atrac1_oracle.mjs'

const here = path.dirname(fileURLToPath(import.meta.url))
const G = path.resolve(here, '../../tests/golden')
const OPTS = { fixedModes: [0, 0, 0], threshold: 1.0, biased: biasedTable(1) }

function encodeRange(seedL, seedR, first, frames) {
  // the stream's frames [first, first + frames) from two frames of real history (SURVEY.md 5.1: 266 samples suffice)
  const halo = Math.min(2, first), n = (frames + halo) * 512
  const chans = [white(seedL, n, (first - halo) * 512), white(seedR, n, (first - halo) * 512)]
  const units = new Uint8Array((frames + halo) * 2 * 212)
  encodeStream(chans, frames + halo, OPTS, [new EncState(), new EncState()], units)
  return units.subarray(halo * 2 * 212)
}

if (!isMainThread) {
  const { first, frames } = workerData
  const t0 = process.hrtime.bigint()
  encodeRange(1, 2, first, frames)
  parentPort.postMessage(Number(process.hrtime.bigint() - t0) / 1e9)
} else {
  const args = process.argv.slice(2)
  const arg = (name, dflt) => { const i = args.indexOf(name); return i >= 0 ? Number(args[i + 1]) : dflt }
  // ---- parity first ----
  const idx = JSON.parse(fs.readFileSync(path.join(G, 'kat_index.json')))
  const checks = {}
  const k1 = JSON.parse(fs.readFileSync(path.join(G, 'config1_sine1k.json')))
  {
    const pcm = new Float32Array(512)
    for (let i = 0; i < 512; i++) pcm[i] = Math.sin((2 * Math.PI * 1000 * i) / 44100)
    const u = new Uint8Array(212)
    encodeStream([pcm], 1, OPTS, [new EncState()], u)
    checks.config1 = Buffer.from(u).toString('hex') === k1.unit_hex
  }
  for (const name of ['white_m000_b1', 'white_m223_b1', 'white_detect', 'pinkT_detect']) {
    const c = idx[name], n = 64 * 512
    const chans = c.signal === 'white' ? [white(1, n), white(2, n)] : [pinkT(3, n), pinkT(4, n)]
    const o = { fixedModes: c.options.fixedBlockModes || null, threshold: c.options.transientThresholdLow || 1.0, biased: biasedTable(c.options.allocationBias || 1) }
    const u = new Uint8Array(128 * 212)
    encodeStream(chans, 64, o, [new EncState(), new EncState()], u)
    checks[name] = Buffer.from(u).equals(fs.readFileSync(path.join(G, `kat64_${name}.units.bin`)))
  }
  // shard == stream: a range encoded from its 2-frame history equals the same frames of the whole stream
  {
    const whole = encodeRange(1, 2, 0, 40), part = encodeRange(1, 2, 17, 23)
    checks.range_from_history = Buffer.from(part).equals(Buffer.from(whole.subarray(17 * 424)))
  }
  const parity = Object.values(checks).every(Boolean)
  if (!parity || args.includes('--check')) {
    console.log(JSON.stringify({ parity, checks }))
    process.exit(parity ? 0 : 1)
  }
  // ---- timing ----
  const frames = arg('--frames', 8192), threads = arg('--threads', Math.max(1, os.cpus().length))
  const t0 = process.hrtime.bigint()
  encodeRange(1, 2, 0, frames)
  const one = frames / (Number(process.hrtime.bigint() - t0) / 1e9)
  const each = frames
  const w0 = process.hrtime.bigint()
  const jobs = []
  for (let t = 0; t < threads; t++) {
    jobs.push(new Promise((resolve, reject) => {
      const w = new Worker(fileURLToPath(import.meta.url), { workerData: { first: t * each, frames: each } })
      w.on('message', resolve); w.on('error', reject)
    }))
  }
  Promise.all(jobs).then((times) => {
    const wall = Number(process.hrtime.bigint() - w0) / 1e9
    // all_threads: the workers encode side by side; the slowest one's encode time is the job's (module load and JIT warm-up of
    // a fresh worker, ~0.3 s each, are left out of it and shown by wall_s)
    console.log(JSON.stringify({ parity, checks, node: process.version, one_thread: one, one_thread_frames: frames, threads,
      all_threads: (each * threads) / Math.max(...times), frames_per_thread: each, wall_s: wall, unit: 'stereo frames/s' }))
  }).catch((e) => { console.error(e); process.exit(2) })
}
