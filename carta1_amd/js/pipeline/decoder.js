// decode(bufferPool) -> frame closure (codec/pipeline/decoder.js:408-411): dequantization, IMDCT with
// overlap-add and QMF synthesis run as one HIP kernel; decoder state lives in the pool's native stream.
import { BufferPool } from '../core/buffers.js'
import { SOUND_UNIT_SIZE } from '../core/constants.js'
import { serializeFrame } from '../io/serialization.js'
import { native, context } from '../native.js'
import { throwError } from '../utils.js'

export function decode(bufferPool = new BufferPool()) {
  if (!bufferPool) throwError('imdctStage: bufferPool is required')
  return (frameData) => {
    const addon = native()
    if (!bufferPool.decoderStream) bufferPool.decoderStream = addon.decStreamCreate(context(), 1)
    let unit
    if (frameData instanceof Uint8Array) unit = frameData
    else if (!frameData.nBfu) {
      // the reference's padding frame (processor.js:300-308): no BFUs -> all-zero spectrum
      unit = new Uint8Array(SOUND_UNIT_SIZE)
      unit[0] = 0xac
    } else unit = serializeFrame(frameData)
    return addon.decStreamPush(bufferPool.decoderStream, unit, 1)[0]
  }
}
