// 212-byte sound unit <-> frame fields, and the 2048-byte AEA header.  Host-side format code (the
// reference keeps it in JavaScript too: codec/io/serialization.js:41-254).  Layout: 16-bit header
// (2-mode, 2-mode, 3-mode, BFU-amount index), nBfu x 4-bit word-length index, nBfu x 6-bit scale-factor
// index, mantissas in two's complement, three zero bytes at the end.
import {
  SOUND_UNIT_SIZE, BFU_AMOUNTS, SPECS_PER_BFU, WORD_LENGTH_BITS, AEA_HEADER_SIZE, AEA_MAGIC,
  AEA_TITLE_OFFSET, AEA_TITLE_SIZE, AEA_FRAME_COUNT_OFFSET, AEA_CHANNEL_COUNT_OFFSET,
} from '../core/constants.js'
import { BitWriter, BitReader } from './bitstream.js'

export function serializeFrame(frameData) {
  const unit = new Uint8Array(SOUND_UNIT_SIZE)
  const w = new BitWriter(unit)
  const n = frameData.nBfu
  const modes = frameData.blockModes
  w.write(2 - modes[0], 2)
  w.write(2 - modes[1], 2)
  w.write(3 - modes[2], 2)
  w.write(BFU_AMOUNTS.indexOf(n) & 7, 5) // 3-bit amount index followed by two zero bits
  w.write(0, 5)
  for (let b = 0; b < n; b++) w.write(frameData.wordLengthIndices[b], 4)
  for (let b = 0; b < n; b++) w.write(frameData.scaleFactorIndices[b], 6)
  for (let b = 0; b < n; b++) {
    const bits = WORD_LENGTH_BITS[frameData.wordLengthIndices[b]]
    if (bits === 0) continue
    const q = frameData.quantizedCoefficients[b]
    for (let i = 0; i < q.length; i++) w.write(q[i] & ((1 << bits) - 1), bits)
  }
  unit[SOUND_UNIT_SIZE - 3] = unit[SOUND_UNIT_SIZE - 2] = unit[SOUND_UNIT_SIZE - 1] = 0
  return unit
}

export function deserializeFrame(buffer) {
  if (buffer.length !== SOUND_UNIT_SIZE) throw new Error(`Frame must be ${SOUND_UNIT_SIZE} bytes`)
  const r = new BitReader(buffer)
  const blockModes = [2 - r.read(2), 2 - r.read(2), 3 - r.read(2)]
  r.read(2)
  const nBfu = BFU_AMOUNTS[r.read(3)]
  r.read(5)
  const wordLengthIndices = new Int32Array(nBfu)
  const scaleFactorIndices = new Int32Array(nBfu)
  for (let b = 0; b < nBfu; b++) wordLengthIndices[b] = r.read(4)
  for (let b = 0; b < nBfu; b++) scaleFactorIndices[b] = r.read(6)
  const quantizedCoefficients = []
  for (let b = 0; b < nBfu; b++) {
    const bits = WORD_LENGTH_BITS[wordLengthIndices[b]]
    const q = new Int32Array(SPECS_PER_BFU[b])
    if (bits > 0) for (let i = 0; i < q.length; i++) q[i] = r.readSigned(bits)
    quantizedCoefficients.push(q)
  }
  return { nBfu, scaleFactorIndices, wordLengthIndices, quantizedCoefficients, blockModes }
}

export class AeaFile {
  static createHeader(title = '', frameCount = 0, channelCount = 1) {
    const header = new Uint8Array(AEA_HEADER_SIZE)
    header.set(AEA_MAGIC, 0)
    const text = Buffer.from(String(title), 'utf8')
    header.set(text.subarray(0, Math.min(text.length, AEA_TITLE_SIZE - 1)), AEA_TITLE_OFFSET)
    new DataView(header.buffer).setUint32(AEA_FRAME_COUNT_OFFSET, frameCount, true)
    header[AEA_CHANNEL_COUNT_OFFSET] = channelCount
    return header
  }

  static parseHeader(header) {
    if (header.length !== AEA_HEADER_SIZE) throw new Error(`Header must be ${AEA_HEADER_SIZE} bytes`)
    for (let i = 0; i < AEA_MAGIC.length; i++) if (header[i] !== AEA_MAGIC[i]) throw new Error('Invalid AEA file')
    const end = header.indexOf(0, AEA_TITLE_OFFSET)
    const len = end === -1 ? AEA_TITLE_SIZE : end - AEA_TITLE_OFFSET
    const title = Buffer.from(header.buffer, header.byteOffset + AEA_TITLE_OFFSET, len).toString('utf8')
    const view = new DataView(header.buffer, header.byteOffset, header.byteLength)
    return { title, frameCount: view.getUint32(AEA_FRAME_COUNT_OFFSET, true), channelCount: header[AEA_CHANNEL_COUNT_OFFSET] }
  }
}
