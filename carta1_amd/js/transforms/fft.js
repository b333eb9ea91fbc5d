// FFT.fft(real, imag): the reference's in-place radix-2 transform (codec/transforms/fft.js:14-68) on the device (c1_fft).
// The per-stage twiddle steps (cos, sin)(-2 pi / stride) come from THIS engine's Math.cos / Math.sin, as they would if the
// reference ran in this process (fft.js:37-39); the twiddle recurrence and the Float32Array roundings are the kernel's.
import { native, context } from '../native.js'

export class FFT {
  static fft(real, imag) {
    const size = real.length
    if (size === 1) return
    if (!(real instanceof Float32Array) || !(imag instanceof Float32Array)) throw new TypeError('FFT.fft works in place on Float32Arrays')
    const stages = Math.round(Math.log2(size))
    const w = new Float64Array(2 * stages)
    for (let s = 0, stride = 2; s < stages; s++, stride <<= 1) {
      const angle = (-2 * Math.PI) / stride
      w[2 * s] = Math.cos(angle)
      w[2 * s + 1] = Math.sin(angle)
    }
    native().fft(context(), real, imag, w)
  }
}
