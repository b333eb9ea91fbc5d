// carta1-amd: the carta1 API surface for the ATRAC1 hot path, computed on an MI355X.
// Same names and call shapes as the reference's codec/index.js:26-47 for everything on or next to the
// hot path.  Not re-exported: quantize, dequantize, FFT, qmfAnalysisStage, mdctStage -- in the
// reference those are single-frame CPU functions of the very stages that are HIP kernels here; their
// results are reachable through encode()/decode() and through the stage taps of the C ABI.
export { encode } from './pipeline/encoder.js'
export { decode } from './pipeline/decoder.js'
export { serializeFrame, deserializeFrame, AeaFile } from './io/serialization.js'
export { AudioProcessor, encodeAeaPcm, decodeAeaPcm, encodeWavPcm, decodeAeaToWav16 } from './io/processor.js'
export { BufferPool } from './core/buffers.js'
export { EncoderOptions } from './core/options.js'
export { pipe } from './utils.js'
export { WORD_LENGTH_BITS, SPECS_PER_BFU, SCALE_FACTORS, BFU_START_LONG } from './core/constants.js'
export { deviceCount, allocPinnedFloat32Array } from './native.js'
