// BufferPool: in the reference (codec/core/buffers.js:7-81) this object IS the per-stream codec state
// (QMF delay lines, MDCT overlap, transient history, IMDCT tails).  Here that state lives on the GPU
// inside a native stream handle; the pool owns the handle so that, exactly as in the reference, passing
// the same pool to encode()/decode() continues the same stream and a fresh pool starts a new one.
export class BufferPool {
  constructor() {
    this.encoderStream = null // c1_enc_stream, created by the first encode() closure call
    this.decoderStream = null // c1_dec_stream
    this.encoderOptionsKey = null
    this.qmfHistory = null // qmfAnalysisStage on its own: the previous frame's PCM (the QMF delay lines are made of it)
    this.mdctPreviousBands = null // mdctStage on its own: the previous frame's band samples (mdctOverlap is made of their tails)
  }
}
