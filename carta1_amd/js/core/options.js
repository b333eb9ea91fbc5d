// EncoderOptions: same keys, defaults, ranges, accessors and error messages as the reference's
// codec/core/options.js:11-164, plus toNative() which packs what the hot path consumes.
import { SCALE_FACTORS } from './constants.js'

const SPEC = {
  transientThresholdLow: { default: 1.0, name: 'Low Band Transient Threshold', range: [0.01, 2], step: 0.01,
    description: 'Transient detection threshold of the 0-5.5 kHz band; lower is more sensitive.' },
  transientThresholdMid: { default: 1.5, name: 'Mid Band Transient Threshold', range: [0.01, 3], step: 0.01,
    description: 'Transient detection threshold of the 5.5-11 kHz band; lower is more sensitive.' },
  transientThresholdHigh: { default: 2.0, name: 'High Band Transient Threshold', range: [0.01, 4], step: 0.01,
    description: 'Transient detection threshold of the 11-22 kHz band; lower is more sensitive.' },
  allocationBias: { default: 1.0, name: 'Bit allocation bias', range: [0.0, 5.0], step: 0.01,
    description: 'Higher values spend more bits on loud spectral components, lower values spread them.' },
  fixedBlockModes: { default: null, name: 'Fixed block modes', type: 'array',
    description: 'Skip transient detection and use [low, mid, high] block modes: low/mid 0 or 2, high 0 or 3.' },
}

export class EncoderOptions {
  constructor(options = {}) {
    this.values = {}
    this.metadata = {}
    for (const key of Object.keys(SPEC)) {
      this.values[key] = SPEC[key].default
      this.metadata[key] = Object.assign({}, SPEC[key])
    }
    if (options) this.setOptions(options)
  }

  setOptions(options) {
    for (const key of Object.keys(options)) {
      if (key in this.values) this.setValue(key, options[key])
    }
  }

  setValue(key, value) {
    const meta = this.metadata[key]
    if (!meta) throw new Error(`Unknown option: ${key}`)
    if (meta.type !== 'array') {
      const [min, max] = meta.range
      if (value < min || value > max) {
        throw new Error(`Value for ${key} must be between ${min} and ${max}, got ${value}`)
      }
    }
    this.values[key] = value
  }

  getValue(key) {
    if (!(key in this.values)) throw new Error(`Unknown option: ${key}`)
    return this.values[key]
  }

  get transientThresholdLow() { return this.values.transientThresholdLow }
  get transientThresholdMid() { return this.values.transientThresholdMid }
  get transientThresholdHigh() { return this.values.transientThresholdHigh }
  get allocationBias() { return this.values.allocationBias }
  get fixedBlockModes() { return this.values.fixedBlockModes }

  getMetadata(key) { return this.metadata[key] }
  getAllMetadata() { return Object.assign({}, this.metadata) }
  reset() { for (const key of Object.keys(this.metadata)) this.values[key] = this.metadata[key].default }
  toObject() { return { values: Object.assign({}, this.values), metadata: Object.assign({}, this.metadata) } }

  // Float64Array(68) for the addon: biased scale factors (reference: buildBiasedScaleFactorTable,
  // codec/coding/bitallocation.js:46-61 -- Math.pow here, in the host's V8), the one threshold the
  // pipeline reads (encoder.js:137-141), and the fixed modes or -1.
  toNative() {
    const out = new Float64Array(68)
    const bias = this.allocationBias
    for (let i = 0; i < 64; i++) out[i] = bias === 1 ? SCALE_FACTORS[i] : Math.pow(SCALE_FACTORS[i], bias)
    out[64] = this.transientThresholdLow
    const modes = this.fixedBlockModes
    for (let b = 0; b < 3; b++) out[65 + b] = modes ? modes[b] : -1
    return out
  }
}
