// Loads the N-API addon (addon/carta1_napi.node -> lib/libcarta1_hip.so) and owns the default device
// context.  There is no JavaScript fallback for the hot path: if the addon or a HIP device is missing,
// the first call throws the library's error.
import { createRequire } from 'module'
import { buildNativeTables } from './core/constants.js'

const require = createRequire(import.meta.url)
let addon = null
let defaultCtx = null

export function native() {
  if (!addon) {
    try {
      addon = require('./addon/carta1_napi.node')
    } catch (e) {
      throw new Error(`carta1-amd: native addon not built (${e.message}); run \`make -C carta1_amd/js/addon\` -- there is no CPU fallback`)
    }
    addon.setTables(buildNativeTables())
  }
  return addon
}

export function context(device = 0) {
  if (device !== 0) return native().ctxCreate(device)
  if (!defaultCtx) defaultCtx = native().ctxCreate(0)
  return defaultCtx
}

export function deviceCount() {
  return native().deviceCount()
}

// Float32Array in page-locked host memory: PCM placed there is uploaded at the pinned PCIe rate and large batches
// are streamed (upload, kernels and download of consecutive chunks overlap).  Garbage collected like any array.
export function allocPinnedFloat32Array(length) {
  return new Float32Array(native().allocPinned(length * 4))
}
