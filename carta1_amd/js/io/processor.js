// AudioProcessor + encodeAeaPcm / decodeAeaPcm: the stream/file level API of the reference
// (codec/io/processor.js:37-671), with the per-frame hot loop (processor.js:119-136, :193-237)
// replaced by ONE batched native call per buffer.  WAV blob helpers are outside the hot-path scope
// (SURVEY.md section 8f) and are not provided.
import { EncoderOptions } from '../core/options.js'
import { BufferPool } from '../core/buffers.js'
import { SAMPLES_PER_FRAME, AEA_HEADER_SIZE, SOUND_UNIT_SIZE } from '../core/constants.js'
import { encode } from '../pipeline/encoder.js'
import { decode } from '../pipeline/decoder.js'
import { serializeFrame, deserializeFrame, AeaFile } from './serialization.js'
import { native, context } from '../native.js'

function padChannels(channels) {
  const longest = Math.max(...channels.map((c) => c.length))
  const frames = Math.ceil(longest / SAMPLES_PER_FRAME)
  return {
    frames,
    padded: channels.map((c) => {
      if (c.length === frames * SAMPLES_PER_FRAME) return c
      const p = new Float32Array(frames * SAMPLES_PER_FRAME) // zero padding: processor.js:246-279
      p.set(c)
      return p
    }),
  }
}

export async function encodeAeaPcm(channels, options = {}) {
  if (!Array.isArray(channels) || (channels.length !== 1 && channels.length !== 2) ||
      channels.some((channel) => !(channel instanceof Float32Array))) {
    throw new TypeError('ATRAC1 encoding requires one or two Float32 channels')
  }
  // options.devices (not in the reference): device indices to shard the frame batch over, e.g. [0, 1, 2, 3]; contiguous
  // frame ranges, one context and host thread per entry, no collective -- same bytes as one device
  const { title = 'encoded by carta1', devices, ...encoderValues } = options
  const encoderOptions = new EncoderOptions(encoderValues)
  const { frames, padded } = padChannels(channels)
  const unitCount = frames * channels.length
  const image = new Uint8Array(AEA_HEADER_SIZE + unitCount * SOUND_UNIT_SIZE)
  image.set(AeaFile.createHeader(title, unitCount, channels.length), 0) // frameCount counts units: processor.js:320-325
  if (frames > 0) {
    const where = Array.isArray(devices) && devices.length ? devices : context()
    const units = await native().encodeBatchAsync(where, padded, 0, encoderOptions.toNative())
    image.set(units, AEA_HEADER_SIZE)
  }
  return image
}

export async function decodeAeaPcm(input, options = {}) {
  let bytes
  if (input instanceof Uint8Array) bytes = input
  else if (input instanceof ArrayBuffer) bytes = new Uint8Array(input)
  else if (typeof Blob !== 'undefined' && input instanceof Blob) bytes = new Uint8Array(await input.arrayBuffer())
  else throw new TypeError('ATRAC1 decoding requires AEA bytes or a Blob')
  const { info, units } = AudioProcessor.parseAea(bytes)
  const nch = info.channelCount
  if (nch !== 1 && nch !== 2) throw new Error(`Unsupported channel count: ${nch}`)
  let body = units
  const count = units.length / SOUND_UNIT_SIZE
  if (nch === 2 && count % 2 === 1) {
    // trailing lone left unit is paired with the reference's dummy frame (processor.js:222-232)
    body = new Uint8Array(units.length + SOUND_UNIT_SIZE)
    body.set(units)
    body[units.length] = 0xac
  }
  if (body.length === 0) return nch === 1 ? [new Float32Array(0)] : [new Float32Array(0), new Float32Array(0)]
  const where = Array.isArray(options.devices) && options.devices.length ? options.devices : context()
  return native().decodeBatchAsync(where, body, nch, 0)
}

// WAV body (interleaved little-endian integer PCM: Int16Array, or a Uint8Array with bits = 16, 24 or 32) -> AEA image.
// What the reference's CLI does with WavReader + encodeStream (bin/cli.js:367-404, processor.js:246-276) as one
// native call: the integer samples cross PCIe and are converted on the device.
export function encodeWavPcm(wavBody, options = {}) {
  const { title = 'encoded by carta1', channelCount = 1, bits = 16, ...encoderValues } = options
  if (!(wavBody instanceof Int16Array) && !(wavBody instanceof Uint8Array)) {
    throw new TypeError('ATRAC1 WAV encoding requires an Int16Array or a Uint8Array of sample bytes')
  }
  if (channelCount !== 1 && channelCount !== 2) throw new TypeError('ATRAC1 encoding requires one or two channels')
  const encoderOptions = new EncoderOptions(encoderValues)
  const units = native().encodeWavBatch(context(), wavBody, wavBody instanceof Int16Array ? 16 : bits, channelCount, encoderOptions.toNative())
  const image = new Uint8Array(AEA_HEADER_SIZE + units.length)
  image.set(AeaFile.createHeader(title, units.length / SOUND_UNIT_SIZE, channelCount), 0)
  image.set(units, AEA_HEADER_SIZE)
  return image
}

// AEA image -> { channelCount, samples: Int16Array (interleaved) }: decode + the 16-bit conversion of createWavBlob
// (processor.js:349-447) on the device
export function decodeAeaToWav16(bytes) {
  if (!(bytes instanceof Uint8Array)) throw new TypeError('ATRAC1 decoding requires AEA bytes')
  const { info, units } = AudioProcessor.parseAea(bytes)
  const nch = info.channelCount
  if (nch !== 1 && nch !== 2) throw new Error(`Unsupported channel count: ${nch}`)
  // a stereo image that ends on a lone left unit: the reference pairs it with a dummy right unit (processor.js:222-232)
  let whole = units
  const rest = units.length % (nch * SOUND_UNIT_SIZE)
  if (rest >= SOUND_UNIT_SIZE) {
    whole = new Uint8Array(units.length - rest + nch * SOUND_UNIT_SIZE)
    whole.set(units.subarray(0, units.length - rest + SOUND_UNIT_SIZE))
    whole[units.length - rest + SOUND_UNIT_SIZE] = 0xac            // the dummy unit's two header bytes, zeros after
  } else if (rest) {
    whole = units.subarray(0, units.length - rest)
  }
  return { channelCount: nch, samples: whole.length ? native().decodeWav16Batch(context(), whole, nch) : new Int16Array(0) }
}

export class AudioProcessor {
  static encodeAeaPcm(channels, options = {}) { return encodeAeaPcm(channels, options) }
  static decodeAeaPcm(input) { return decodeAeaPcm(input) }

  // Streams of frames in, frame fields out: one closure per channel, as processor.js:69-136.
  // options.batchFrames (default 1 = a result after every frame, like the reference): with N > 1 the frames are
  // collected and handed to the device N at a time (one native stream for all channels); same fields, N times fewer
  // device round trips.
  static async *encodeStream(audioFrames, options = {}) {
    const { channelCount = 1, onProgress, encoderOptions, batchFrames = 1 } = options
    if (channelCount !== 1 && channelCount !== 2) throw new Error(`Unsupported channel count: ${channelCount}`)
    const opts = encoderOptions || new EncoderOptions()
    let frameIndex = 0
    if (batchFrames > 1) {
      const addon = native()
      const stream = addon.encStreamCreate(context(), channelCount, opts.toNative())
      let pending = 0
      let buffers = null
      const flush = function* () {
        const units = addon.encStreamPush(stream, buffers.map((b) => b.subarray(0, pending * SAMPLES_PER_FRAME)))
        for (let f = 0; f < pending; f++) {
          for (let c = 0; c < channelCount; c++) {
            const at = (f * channelCount + c) * SOUND_UNIT_SIZE
            const fields = deserializeFrame(units.subarray(at, at + SOUND_UNIT_SIZE))
            if (opts.fixedBlockModes) fields.blockModes = opts.fixedBlockModes
            yield fields
          }
          if (onProgress) onProgress(frameIndex++)
        }
        pending = 0
      }
      for await (const frame of audioFrames) {
        const parts = channelCount === 1 ? [frame] : frame
        if (!buffers) buffers = parts.map(() => new Float32Array(batchFrames * SAMPLES_PER_FRAME))
        for (let c = 0; c < channelCount; c++) {
          if (!(parts[c] instanceof Float32Array) || parts[c].length !== SAMPLES_PER_FRAME) {
            throw new Error(`encode: expected a Float32Array of ${SAMPLES_PER_FRAME} samples`)
          }
          buffers[c].set(parts[c], pending * SAMPLES_PER_FRAME)
        }
        if (++pending === batchFrames) yield* flush()
      }
      if (pending) yield* flush()
      return
    }
    const encoders = []
    for (let c = 0; c < channelCount; c++) encoders.push(encode(opts, new BufferPool()))
    for await (const frame of audioFrames) {
      const parts = channelCount === 1 ? [frame] : frame
      for (let c = 0; c < channelCount; c++) yield encoders[c](parts[c])
      if (onProgress) onProgress(frameIndex++)
    }
  }

  static async *decodeStream(encodedFrames, options = {}) {
    const { channelCount = 1, onProgress } = options
    if (channelCount !== 1 && channelCount !== 2) throw new Error(`Unsupported channel count: ${channelCount}`)
    const decoders = []
    for (let c = 0; c < channelCount; c++) decoders.push(decode(new BufferPool()))
    let frameIndex = 0
    let pending = []
    for await (const frame of encodedFrames) {
      pending.push(frame)
      if (pending.length < channelCount) continue
      const out = pending.map((f, c) => decoders[c](f))
      pending = []
      yield channelCount === 1 ? out[0] : out
      if (onProgress) onProgress(frameIndex++)
    }
    if (pending.length === 1 && channelCount === 2) {
      yield [decoders[0](pending[0]), decoders[1](AudioProcessor._createDummyFrame())]
      if (onProgress) onProgress(frameIndex++)
    }
  }

  static *frameBufferToFrames(buffers, frameSize = SAMPLES_PER_FRAME) {
    if (buffers.length !== 1 && buffers.length !== 2) throw new Error(`Unsupported channel count: ${buffers.length}`)
    const longest = Math.max(...buffers.map((b) => b.length))
    for (let at = 0; at < longest; at += frameSize) {
      const frames = buffers.map((b) => {
        const f = new Float32Array(frameSize)
        if (at < b.length) f.set(b.subarray(at, Math.min(at + frameSize, b.length)))
        return f
      })
      yield buffers.length === 1 ? frames[0] : frames
    }
  }

  static async collectFrames(frameStream) {
    const frames = []
    for await (const frame of frameStream) frames.push(frame)
    return frames
  }

  static _createDummyFrame() {
    return { nBfu: 0, blockModes: [0, 0, 0], scaleFactorIndices: new Int32Array(0), wordLengthIndices: new Int32Array(0), quantizedCoefficients: [] }
  }

  // AEA image from a stream of frame fields (createAeaBlob of the reference returns a Blob; a
  // Uint8Array is returned here so this also runs where Blob does not exist).
  static async createAeaBytes(encodedFrames, options = {}) {
    const { title = 'encoded by atrac1.js', channelCount = 1 } = options
    const units = []
    for await (const frame of encodedFrames) units.push(serializeFrame(frame))
    const image = new Uint8Array(AEA_HEADER_SIZE + units.length * SOUND_UNIT_SIZE)
    image.set(AeaFile.createHeader(title, units.length, channelCount), 0)
    units.forEach((u, i) => image.set(u, AEA_HEADER_SIZE + i * SOUND_UNIT_SIZE))
    return image
  }

  static async createAeaBlob(encodedFrames, options = {}) {
    const bytes = await AudioProcessor.createAeaBytes(encodedFrames, options)
    if (typeof Blob === 'undefined') throw new Error('Blob is not available in this runtime; use createAeaBytes')
    return new Blob([bytes], { type: 'application/octet-stream' })
  }

  static parseAea(bytes) {
    const info = AeaFile.parseHeader(bytes.subarray(0, AEA_HEADER_SIZE))
    const whole = Math.floor((bytes.length - AEA_HEADER_SIZE) / SOUND_UNIT_SIZE) // a partial trailing unit is dropped
    return { info, units: bytes.subarray(AEA_HEADER_SIZE, AEA_HEADER_SIZE + whole * SOUND_UNIT_SIZE) }
  }

  static async parseAeaBlob(blob) {
    const { info, units } = AudioProcessor.parseAea(new Uint8Array(await blob.arrayBuffer()))
    const frameData = []
    for (let at = 0; at < units.length; at += SOUND_UNIT_SIZE) frameData.push(units.slice(at, at + SOUND_UNIT_SIZE))
    return { info, frameData }
  }

  static *deserializedFrameStream(frameData) {
    for (const frame of frameData) yield deserializeFrame(frame)
  }
}
