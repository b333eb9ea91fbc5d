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

// qmfAnalysisStage(context) and mdctStage(context): the two pipeline stages the reference also exports on their own
// (codec/pipeline/encoder.js:57-96, :170-349; codec/index.js:30-31), with the reference's call shapes.  Their state -- the QMF
// delay lines, the MDCT overlap -- is a bounded function of the previous frame (SURVEY.md 5.1), so the pool keeps that frame's
// input and the device rebuilds the state from it: bit-identical to carrying the reference's buffers along.
export function qmfAnalysisStage(stageContext) {
  const bufferPool = (stageContext && stageContext.bufferPool) || throwError('qmfAnalysisStage: bufferPool is required')
  return (pcmSamples) => {
    if (!(pcmSamples instanceof Float32Array) || pcmSamples.length !== SAMPLES_PER_FRAME) {
      throwError(`qmfAnalysisStage: expected a Float32Array of ${SAMPLES_PER_FRAME} samples`)
    }
    const prev = bufferPool.qmfHistory
    const pcm = new Float32Array((prev ? 2 : 1) * SAMPLES_PER_FRAME)
    if (prev) pcm.set(prev, 0)
    pcm.set(pcmSamples, prev ? SAMPLES_PER_FRAME : 0)
    const bands = native().qmfAnalysis(context(), pcm, prev ? 1 : 0)
    bufferPool.qmfHistory = pcmSamples.slice()
    return { bands: [bands.slice(0, 128), bands.slice(128, 256), bands.slice(256, 512)] }
  }
}

export function mdctStage(stageContext) {
  const bufferPool = (stageContext && stageContext.bufferPool) || throwError('mdctStage: bufferPool is required')
  return (input) => {
    const { bands, blockModes, originalFrame } = input
    const prev = bufferPool.mdctPreviousBands
    const all = new Float32Array((prev ? 2 : 1) * 512)
    const at = prev ? 512 : 0
    if (prev) all.set(prev, 0)
    all.set(bands[0], at); all.set(bands[1], at + 128); all.set(bands[2], at + 256)
    bufferPool.mdctPreviousBands = all.slice(at, at + 512)   // the samples as they came in: the next frame's overlap is made of them
    const [coefficients, windowed] = native().mdctFromBands(context(), all, prev ? 1 : 0, Int32Array.from(blockModes))
    // the reference windows the band arrays it was given in place and hands the same arrays on (encoder.js:244,292,314)
    bands[0].set(windowed.subarray(0, 128)); bands[1].set(windowed.subarray(128, 256)); bands[2].set(windowed.subarray(256, 512))
    return { bands, coefficients, blockModes, originalFrame }
  }
}
