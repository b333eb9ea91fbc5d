// quantize / dequantize: the reference's single-BFU functions (codec/coding/quantization.js:34-78), same signatures, computed
// on the device in the reference's arithmetic (c1_quantize / c1_dequantize).  The encoder itself quantizes inside its packing
// kernel and never comes through here; these exist for code that imports the names.
import { native, context } from '../native.js'

export function quantize(coefficients, scaleFactorIndex, bitsPerSample) {
  const x = coefficients instanceof Float32Array ? coefficients : Float32Array.from(coefficients)
  return native().quantize(context(), x, scaleFactorIndex | 0, bitsPerSample | 0)
}

export function dequantize(quantized, scaleFactorIndex, bitsPerSample) {
  const q = quantized instanceof Int32Array ? quantized : Int32Array.from(quantized)
  return native().dequantize(context(), q, scaleFactorIndex | 0, bitsPerSample | 0)
}
