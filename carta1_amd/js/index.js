// carta1-amd: the carta1 API surface for the ATRAC1 hot path, computed on an MI355X.
// Same names and call shapes as the reference's codec/index.js:26-47 -- every one of its twenty exports, so that an
// application's `import { ... } from 'carta1'` can be pointed here unchanged (INTEGRATION.md 1).
export { encode, qmfAnalysisStage, mdctStage } from './pipeline/encoder.js'
export { decode } from './pipeline/decoder.js'
export { serializeFrame, deserializeFrame, AeaFile } from './io/serialization.js'
export { AudioProcessor, encodeAeaPcm, decodeAeaPcm, encodeWavPcm, decodeAeaToWav16 } from './io/processor.js'
export { quantize, dequantize } from './coding/quantization.js'
export { FFT } from './transforms/fft.js'
export { BufferPool } from './core/buffers.js'
export { EncoderOptions } from './core/options.js'
export { pipe } from './utils.js'
export { WORD_LENGTH_BITS, SPECS_PER_BFU, SCALE_FACTORS, BFU_START_LONG } from './core/constants.js'
export { deviceCount, allocPinnedFloat32Array } from './native.js'
