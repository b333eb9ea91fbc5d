// encode(options, bufferPool) -> frame closure, the reference's public encoder entry point
// (codec/pipeline/encoder.js:438-450).  The four stages the reference composes there -- QMF analysis,
// block selection, MDCT, RDO quantization -- run as HIP kernels; the per-stream state the reference
// keeps in the BufferPool lives in a native stream handle owned by the pool.  The closure returns the
// same fields object: { nBfu, scaleFactorIndices, wordLengthIndices, quantizedCoefficients, blockModes }.
import { EncoderOptions } from '../core/options.js'
import { BufferPool } from '../core/buffers.js'
import { SAMPLES_PER_FRAME } from '../core/constants.js'
import { deserializeFrame } from '../io/serialization.js'
import { native, context } from '../native.js'
import { throwError } from '../utils.js'

export function encode(options = new EncoderOptions(), bufferPool = new BufferPool()) {
  if (!bufferPool) throwError('qmfAnalysisStage: bufferPool is required')
  if (!options) throwError('blockSelectorStage: options is required')
  return (pcmSamples) => {
    if (!(pcmSamples instanceof Float32Array) || pcmSamples.length !== SAMPLES_PER_FRAME) {
      throwError(`encode: expected a Float32Array of ${SAMPLES_PER_FRAME} samples`)
    }
    const addon = native()
    // options are read per call, as the reference's stages do (encoder.js:131,381)
    const packed = options.toNative()
    const key = packed.join(',')
    if (!bufferPool.encoderStream || bufferPool.encoderOptionsKey !== key) {
      if (bufferPool.encoderStream && bufferPool.encoderOptionsKey !== key) {
        throwError('encode: options changed on a live stream; use a new BufferPool')
      }
      bufferPool.encoderStream = addon.encStreamCreate(context(), 1, packed)
      bufferPool.encoderOptionsKey = key
    }
    const unit = addon.encStreamPush(bufferPool.encoderStream, [pcmSamples])
    const fields = deserializeFrame(unit)
    if (options.fixedBlockModes) fields.blockModes = options.fixedBlockModes // same array, as encoder.js:131-132
    return fields
  }
}
