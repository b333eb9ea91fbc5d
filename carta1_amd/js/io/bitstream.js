// MSB-first bit cursor over a byte buffer: the sound unit's bit order (reference behaviour:
// codec/io/bitstream.js:15-82, including "reads past the end return what was read").
export class BitWriter {
  constructor(bytes) {
    this.bytes = bytes
    this.pos = 0
  }

  write(value, count) {
    for (let k = count - 1; k >= 0; k--, this.pos++) {
      const byte = this.pos >> 3
      if (byte >= this.bytes.length) return
      if ((value >>> k) & 1) this.bytes[byte] |= 0x80 >> (this.pos & 7)
    }
  }
}

export class BitReader {
  constructor(bytes) {
    this.bytes = bytes
    this.pos = 0
  }

  read(count) {
    let value = 0
    for (let k = 0; k < count; k++) {
      const p = this.pos + k
      if (p >> 3 >= this.bytes.length) break
      value = (value << 1) | ((this.bytes[p >> 3] >> (7 - (p & 7))) & 1)
    }
    this.pos += count
    return value
  }

  readSigned(count) {
    const v = this.read(count)
    return v >= 1 << (count - 1) ? v - (1 << count) : v
  }
}

export function packBits(buffer, bitPosition, value, bitCount) {
  const w = new BitWriter(buffer)
  w.pos = bitPosition
  const mask = bitCount >= 32 ? 0xffffffff : (1 << bitCount) - 1
  // clear then set, so packing over non-zero bytes behaves like the reference's masked write
  for (let k = 0; k < bitCount; k++) {
    const p = bitPosition + k
    if (p >> 3 < buffer.length) buffer[p >> 3] &= ~(0x80 >> (p & 7))
  }
  w.write(value & mask, bitCount)
}

export function unpackBits(buffer, bitPosition, bitCount) {
  const r = new BitReader(buffer)
  r.pos = bitPosition
  return r.read(bitCount)
}

export function unpackSignedBits(buffer, bitPosition, bitCount) {
  const r = new BitReader(buffer)
  r.pos = bitPosition
  return r.readSigned(bitCount)
}
