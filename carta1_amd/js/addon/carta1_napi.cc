// carta1_napi.cc -- thin N-API addon over the C ABI of include/carta1_hip.h.
// One JavaScript function per C entry point; no codec logic lives here.  Batch calls come in a
// synchronous form and an asynchronous one (napi_async_work on the libuv pool, resolving a Promise)
// so encodeAeaPcm / decodeAeaPcm stay async like the reference's (codec/io/processor.js:597,628).
#include <node_api.h>

#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/carta1_hip.h"

#define NAPI_OK(call)                                                     \
  do {                                                                    \
    if ((call) != napi_ok) {                                              \
      napi_throw_error(env, nullptr, "N-API call failed: " #call);        \
      return nullptr;                                                     \
    }                                                                     \
  } while (0)

namespace {

napi_value throw_c1(napi_env env, int rc) {
  std::string msg = std::string("carta1_hip: ") + c1_last_error();
  napi_throw_error(env, rc == C1_ERR_NO_DEVICE ? "C1_ERR_NO_DEVICE" : (rc == C1_ERR_ARG ? "C1_ERR_ARG" : "C1_ERR_HIP"), msg.c_str());
  return nullptr;
}

bool get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv) {
  size_t argc = want;
  if (napi_get_cb_info(env, info, &argc, argv, nullptr, nullptr) != napi_ok || argc < want) {
    napi_throw_type_error(env, nullptr, "wrong number of arguments");
    return false;
  }
  return true;
}

template <typename T>
bool get_external(napi_env env, napi_value v, T **out) {
  void *p = nullptr;
  if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
    napi_throw_type_error(env, nullptr, "expected a native handle");
    return false;
  }
  *out = static_cast<T *>(p);
  return true;
}

bool get_typed(napi_env env, napi_value v, napi_typedarray_type want, void **data, size_t *len) {
  napi_typedarray_type t;
  napi_value ab;
  size_t off;
  bool is = false;
  if (napi_is_typedarray(env, v, &is) != napi_ok || !is ||
      napi_get_typedarray_info(env, v, &t, len, data, &ab, &off) != napi_ok || t != want) {
    napi_throw_type_error(env, nullptr, "typed array of the wrong kind");
    return false;
  }
  return true;
}

// options arrive as one Float64Array(68): biased[64], threshold, mode0, mode1, mode2
bool get_options(napi_env env, napi_value v, c1_encode_options *o) {
  void *d;
  size_t n;
  if (!get_typed(env, v, napi_float64_array, &d, &n)) return false;
  if (n != 68) { napi_throw_type_error(env, nullptr, "options must be a Float64Array(68)"); return false; }
  const double *p = static_cast<const double *>(d);
  memset(o, 0, sizeof *o);
  memcpy(o->biased_scale_factors, p, 64 * sizeof(double));
  o->transient_threshold = p[64];
  for (int b = 0; b < 3; b++) o->fixed_block_modes[b] = (int32_t)p[65 + b];
  return true;
}

bool get_channels(napi_env env, napi_value arr, std::vector<float *> *ptrs, size_t *samples) {
  uint32_t n = 0;
  bool is = false;
  if (napi_is_array(env, arr, &is) != napi_ok || !is || napi_get_array_length(env, arr, &n) != napi_ok || n < 1 || n > 2) {
    napi_throw_type_error(env, nullptr, "channels must be an array of one or two Float32Array");
    return false;
  }
  for (uint32_t c = 0; c < n; c++) {
    napi_value e;
    void *d;
    size_t len;
    if (napi_get_element(env, arr, c, &e) != napi_ok || !get_typed(env, e, napi_float32_array, &d, &len)) return false;
    if (c == 0) *samples = len;
    else if (len != *samples) { napi_throw_type_error(env, nullptr, "channels must have equal length"); return false; }
    ptrs->push_back(static_cast<float *>(d));
  }
  return true;
}

// ArrayBuffer of `bytes` bytes.  Large PCM buffers come from page-locked memory (c1_host_alloc) so that the batch calls
// stream them over PCIe in overlapping chunks; they are released by the garbage collector (c1_host_free).
constexpr size_t kPinnedThreshold = (size_t)32768 * 512 * 4;   // one streaming chunk of one channel
void free_pinned(napi_env, void *data, void *) { c1_host_free(data); }
bool make_buffer(napi_env env, size_t bytes, bool pinned, void **data, napi_value *ab) {
  if (pinned && bytes > 0) {
    void *p = nullptr;
    if (c1_host_alloc(bytes, &p) == C1_OK && p) {
      if (napi_create_external_arraybuffer(env, p, bytes, free_pinned, nullptr, ab) == napi_ok) { *data = p; return true; }
      c1_host_free(p);
    }
  }
  return napi_create_arraybuffer(env, bytes, data, ab) == napi_ok;
}
napi_value make_u8(napi_env env, size_t n, uint8_t **data) {
  napi_value ab, ta;
  void *p;
  if (!make_buffer(env, n, false, &p, &ab) || napi_create_typedarray(env, napi_uint8_array, n, ab, 0, &ta) != napi_ok) return nullptr;
  *data = static_cast<uint8_t *>(p);
  return ta;
}
napi_value make_f32(napi_env env, size_t n, float **data) {
  napi_value ab, ta;
  void *p;
  if (!make_buffer(env, n * 4, n * 4 > kPinnedThreshold, &p, &ab) || napi_create_typedarray(env, napi_float32_array, n, ab, 0, &ta) != napi_ok) return nullptr;
  *data = static_cast<float *>(p);
  return ta;
}

// allocPinned(bytes) -> ArrayBuffer in page-locked memory: PCM placed there is uploaded at the pinned PCIe rate
napi_value AllocPinned(napi_env env, napi_callback_info info) {
  napi_value argv[1];
  if (!get_args(env, info, 1, argv)) return nullptr;
  int64_t bytes = 0;
  NAPI_OK(napi_get_value_int64(env, argv[0], &bytes));
  if (bytes < 0) { napi_throw_range_error(env, nullptr, "bytes must be >= 0"); return nullptr; }
  void *p = nullptr;
  const int rc = c1_host_alloc((size_t)bytes, &p);
  if (rc) return throw_c1(env, rc);
  napi_value ab;
  if (bytes == 0) { NAPI_OK(napi_create_arraybuffer(env, 0, &p, &ab)); return ab; }
  if (napi_create_external_arraybuffer(env, p, (size_t)bytes, free_pinned, nullptr, &ab) != napi_ok) {
    c1_host_free(p);
    napi_throw_error(env, nullptr, "allocation failed");
    return nullptr;
  }
  return ab;
}

// ---- library ---------------------------------------------------------------------------------------
napi_value DeviceCount(napi_env env, napi_callback_info) {
  int n = 0;
  const int rc = c1_device_count(&n);
  if (rc) return throw_c1(env, rc);
  napi_value v;
  NAPI_OK(napi_create_int32(env, n, &v));
  return v;
}
napi_value AbiVersion(napi_env env, napi_callback_info) {
  napi_value v;
  NAPI_OK(napi_create_int32(env, c1_abi_version(), &v));
  return v;
}
// tables travel as one Float64Array in the field order of c1_tables
napi_value GetDefaultTables(napi_env env, napi_callback_info) {
  c1_tables t;
  const int rc = c1_get_default_tables(&t);
  if (rc) return throw_c1(env, rc);
  napi_value ab, ta;
  void *p;
  const size_t n = sizeof(c1_tables) / sizeof(double);
  NAPI_OK(napi_create_arraybuffer(env, sizeof t, &p, &ab));
  memcpy(p, &t, sizeof t);
  NAPI_OK(napi_create_typedarray(env, napi_float64_array, n, ab, 0, &ta));
  return ta;
}
napi_value SetTables(napi_env env, napi_callback_info info) {
  napi_value argv[1];
  if (!get_args(env, info, 1, argv)) return nullptr;
  napi_valuetype vt;
  NAPI_OK(napi_typeof(env, argv[0], &vt));
  int rc;
  if (vt == napi_null || vt == napi_undefined) rc = c1_set_tables(nullptr);
  else {
    void *d;
    size_t n;
    if (!get_typed(env, argv[0], napi_float64_array, &d, &n)) return nullptr;
    if (n != sizeof(c1_tables) / sizeof(double)) { napi_throw_type_error(env, nullptr, "tables: wrong length"); return nullptr; }
    c1_tables t;
    memcpy(&t, d, sizeof t);
    rc = c1_set_tables(&t);
  }
  if (rc) return throw_c1(env, rc);
  return nullptr;
}

// ---- contexts ---------------------------------------------------------------------------------------
void FinalizeCtx(napi_env, void *data, void *) { c1_ctx_destroy(static_cast<c1_ctx *>(data)); }
napi_value CtxCreate(napi_env env, napi_callback_info info) {
  napi_value argv[1];
  if (!get_args(env, info, 1, argv)) return nullptr;
  int32_t dev = 0;
  NAPI_OK(napi_get_value_int32(env, argv[0], &dev));
  c1_ctx *ctx = nullptr;
  const int rc = c1_ctx_create(dev, nullptr, &ctx);
  if (rc) return throw_c1(env, rc);
  napi_value ext;
  NAPI_OK(napi_create_external(env, ctx, FinalizeCtx, nullptr, &ext));
  return ext;
}

// ---- batches ------------------------------------------------------------------------------------------
// encodeBatch(ctx, [Float32Array...], haloFrames, options) -> Uint8Array(frames*channels*212)
napi_value EncodeBatch(napi_env env, napi_callback_info info) {
  napi_value argv[4];
  if (!get_args(env, info, 4, argv)) return nullptr;
  c1_ctx *ctx;
  std::vector<float *> ch;
  size_t samples = 0;
  int32_t halo = 0;
  c1_encode_options o;
  if (!get_external(env, argv[0], &ctx) || !get_channels(env, argv[1], &ch, &samples)) return nullptr;
  NAPI_OK(napi_get_value_int32(env, argv[2], &halo));
  if (!get_options(env, argv[3], &o)) return nullptr;
  if (samples % 512 || (int64_t)(samples / 512) < halo) { napi_throw_type_error(env, nullptr, "PCM length must be a multiple of 512"); return nullptr; }
  const int64_t frames = (int64_t)(samples / 512) - halo;
  uint8_t *units;
  napi_value out = make_u8(env, (size_t)frames * ch.size() * C1_UNIT_BYTES, &units);
  if (!out) { napi_throw_error(env, nullptr, "allocation failed"); return nullptr; }
  const float *p[2] = {ch[0] + (size_t)halo * 512, ch.size() > 1 ? ch[1] + (size_t)halo * 512 : nullptr};
  const int rc = c1_encode_batch(ctx, p, (int)ch.size(), frames, halo, &o, units);
  if (rc) return throw_c1(env, rc);
  return out;
}

// decodeBatch(ctx, Uint8Array units, channels, haloUnits) -> [Float32Array...]
napi_value DecodeBatch(napi_env env, napi_callback_info info) {
  napi_value argv[4];
  if (!get_args(env, info, 4, argv)) return nullptr;
  c1_ctx *ctx;
  void *d;
  size_t n;
  int32_t channels = 1, halo = 0;
  if (!get_external(env, argv[0], &ctx) || !get_typed(env, argv[1], napi_uint8_array, &d, &n)) return nullptr;
  NAPI_OK(napi_get_value_int32(env, argv[2], &channels));
  NAPI_OK(napi_get_value_int32(env, argv[3], &halo));
  if (channels < 1 || channels > 2 || n % ((size_t)channels * C1_UNIT_BYTES)) { napi_throw_type_error(env, nullptr, "units: wrong length"); return nullptr; }
  const int64_t frames = (int64_t)(n / ((size_t)channels * C1_UNIT_BYTES)) - halo;
  if (frames < 0) { napi_throw_type_error(env, nullptr, "units shorter than the halo"); return nullptr; }
  napi_value arr;
  NAPI_OK(napi_create_array_with_length(env, channels, &arr));
  float *p[2] = {nullptr, nullptr};
  for (int c = 0; c < channels; c++) {
    napi_value ta = make_f32(env, (size_t)frames * 512, &p[c]);
    if (!ta) { napi_throw_error(env, nullptr, "allocation failed"); return nullptr; }
    NAPI_OK(napi_set_element(env, arr, c, ta));
  }
  const int rc = c1_decode_batch(ctx, static_cast<const uint8_t *>(d) + (size_t)halo * channels * C1_UNIT_BYTES, channels, frames, halo, p);
  if (rc) return throw_c1(env, rc);
  return arr;
}

// encodeWavBatch(ctx, Int16Array | Uint8Array wavBody, bits, channels, options) -> Uint8Array units
// (c1_encode_wav_batch: interleaved little-endian integer PCM converted on the device, ragged tail zero padded)
napi_value EncodeWavBatch(napi_env env, napi_callback_info info) {
  napi_value argv[5];
  if (!get_args(env, info, 5, argv)) return nullptr;
  c1_ctx *ctx;
  if (!get_external(env, argv[0], &ctx)) return nullptr;
  bool is_ta = false;
  NAPI_OK(napi_is_typedarray(env, argv[1], &is_ta));
  if (!is_ta) { napi_throw_type_error(env, nullptr, "wav body must be an Int16Array or a Uint8Array"); return nullptr; }
  napi_typedarray_type tt;
  size_t len = 0, off = 0;
  void *data = nullptr;
  napi_value ab;
  NAPI_OK(napi_get_typedarray_info(env, argv[1], &tt, &len, &data, &ab, &off));
  if (tt != napi_int16_array && tt != napi_uint8_array) { napi_throw_type_error(env, nullptr, "wav body must be an Int16Array or a Uint8Array"); return nullptr; }
  const size_t bytes = tt == napi_int16_array ? len * 2 : len;
  int32_t bits = 16, channels = 1;
  NAPI_OK(napi_get_value_int32(env, argv[2], &bits));
  NAPI_OK(napi_get_value_int32(env, argv[3], &channels));
  c1_encode_options o;
  if (!get_options(env, argv[4], &o)) return nullptr;
  if ((bits != 16 && bits != 24 && bits != 32) || channels < 1 || channels > 2 || bytes % ((size_t)channels * (bits / 8))) {
    napi_throw_type_error(env, nullptr, "wav body: bits must be 16, 24 or 32, channels 1 or 2, and the length a whole number of samples");
    return nullptr;
  }
  const int64_t samples = (int64_t)(bytes / ((size_t)channels * (bits / 8)));
  const int64_t frames = (samples + 511) / 512;
  uint8_t *units;
  napi_value out = make_u8(env, (size_t)frames * channels * C1_UNIT_BYTES, &units);
  if (!out) { napi_throw_error(env, nullptr, "allocation failed"); return nullptr; }
  const int rc = c1_encode_wav_batch(ctx, data, bits, channels, samples, &o, units);
  if (rc) return throw_c1(env, rc);
  return out;
}

// decodeWav16Batch(ctx, Uint8Array units, channels) -> Int16Array (interleaved, frames * 512 * channels)
napi_value DecodeWav16Batch(napi_env env, napi_callback_info info) {
  napi_value argv[3];
  if (!get_args(env, info, 3, argv)) return nullptr;
  c1_ctx *ctx;
  void *d;
  size_t n;
  int32_t channels = 1;
  if (!get_external(env, argv[0], &ctx) || !get_typed(env, argv[1], napi_uint8_array, &d, &n)) return nullptr;
  NAPI_OK(napi_get_value_int32(env, argv[2], &channels));
  if (channels < 1 || channels > 2 || n % ((size_t)channels * C1_UNIT_BYTES)) { napi_throw_type_error(env, nullptr, "units: wrong length"); return nullptr; }
  const int64_t frames = (int64_t)(n / ((size_t)channels * C1_UNIT_BYTES));
  const size_t count = (size_t)frames * 512 * channels;
  napi_value ab, ta;
  void *p;
  if (!make_buffer(env, count * 2, count * 2 > kPinnedThreshold, &p, &ab) || napi_create_typedarray(env, napi_int16_array, count, ab, 0, &ta) != napi_ok) {
    napi_throw_error(env, nullptr, "allocation failed");
    return nullptr;
  }
  const int rc = c1_decode_wav16_batch(ctx, static_cast<const uint8_t *>(d), channels, frames, static_cast<int16_t *>(p));
  if (rc) return throw_c1(env, rc);
  return ta;
}

// ---- asynchronous batches: the typed arrays are kept alive by references while the work runs ----------
struct AsyncJob {
  napi_async_work work = nullptr;
  napi_deferred deferred = nullptr;
  napi_ref keep[4] = {nullptr, nullptr, nullptr, nullptr};
  int nkeep = 0;
  c1_ctx *ctx = nullptr;
  std::vector<int> devices;          // non-empty: the *_multi entry points (one context per entry, owned by the library)
  bool encode = true;
  int channels = 1, halo = 0;
  int64_t frames = 0;
  c1_encode_options opts;
  const float *in[2] = {nullptr, nullptr};
  float *outp[2] = {nullptr, nullptr};
  const uint8_t *units_in = nullptr;
  uint8_t *units_out = nullptr;
  napi_ref result = nullptr;
  int rc = 0;
  std::string err;
};
void AsyncExecute(napi_env, void *data) {
  AsyncJob *j = static_cast<AsyncJob *>(data);
  if (!j->devices.empty())
    j->rc = j->encode ? c1_encode_batch_multi(j->devices.data(), (int)j->devices.size(), j->in, j->channels, j->frames, j->halo, &j->opts, j->units_out)
                      : c1_decode_batch_multi(j->devices.data(), (int)j->devices.size(), j->units_in, j->channels, j->frames, j->halo, j->outp);
  else
    j->rc = j->encode ? c1_encode_batch(j->ctx, j->in, j->channels, j->frames, j->halo, &j->opts, j->units_out)
                      : c1_decode_batch(j->ctx, j->units_in, j->channels, j->frames, j->halo, j->outp);
  if (j->rc) j->err = c1_last_error();   // thread-local: read it on the worker thread
}
void AsyncComplete(napi_env env, napi_status, void *data) {
  AsyncJob *j = static_cast<AsyncJob *>(data);
  if (j->rc) {
    napi_value msg, e;
    napi_create_string_utf8(env, ("carta1_hip: " + j->err).c_str(), NAPI_AUTO_LENGTH, &msg);
    napi_create_error(env, nullptr, msg, &e);
    napi_reject_deferred(env, j->deferred, e);
  } else {
    napi_value v;
    napi_get_reference_value(env, j->result, &v);
    napi_resolve_deferred(env, j->deferred, v);
  }
  for (int i = 0; i < j->nkeep; i++) napi_delete_reference(env, j->keep[i]);
  napi_delete_reference(env, j->result);
  napi_delete_async_work(env, j->work);
  delete j;
}
napi_value start_job(napi_env env, AsyncJob *j, const char *name) {
  napi_value promise, rname;
  if (napi_create_promise(env, &j->deferred, &promise) != napi_ok ||
      napi_create_string_utf8(env, name, NAPI_AUTO_LENGTH, &rname) != napi_ok ||
      napi_create_async_work(env, nullptr, rname, AsyncExecute, AsyncComplete, j, &j->work) != napi_ok ||
      napi_queue_async_work(env, j->work) != napi_ok) {
    delete j;
    napi_throw_error(env, nullptr, "could not queue async work");
    return nullptr;
  }
  return promise;
}
// argv[0]: a context, or (multi) an array of device indices
bool get_ctx_or_devices(napi_env env, napi_value v, AsyncJob *j) {
  bool is_array = false;
  if (napi_is_array(env, v, &is_array) != napi_ok) return false;
  if (!is_array) return get_external(env, v, &j->ctx);
  uint32_t n = 0;
  if (napi_get_array_length(env, v, &n) != napi_ok || n < 1 || n > 64) return false;
  for (uint32_t i = 0; i < n; i++) {
    napi_value e;
    int32_t d = 0;
    if (napi_get_element(env, v, i, &e) != napi_ok || napi_get_value_int32(env, e, &d) != napi_ok) return false;
    j->devices.push_back(d);
  }
  return true;
}
// keep `v` alive until the job is done; false (and an exception) when the engine cannot reference it -- the job must
// then not start: its buffers could be collected under it
bool keep_alive(napi_env env, AsyncJob *j, napi_value v) {
  napi_ref r = nullptr;
  if (napi_create_reference(env, v, 1, &r) != napi_ok || !r) {
    napi_throw_error(env, nullptr, "carta1: cannot reference an argument of the asynchronous call");
    return false;
  }
  j->keep[j->nkeep++] = r;
  return true;
}
void drop_job(napi_env env, AsyncJob *j) {
  for (int i = 0; i < j->nkeep; i++) if (j->keep[i]) napi_delete_reference(env, j->keep[i]);
  if (j->result) napi_delete_reference(env, j->result);
  delete j;
}

napi_value EncodeBatchAsync(napi_env env, napi_callback_info info) {
  napi_value argv[4];
  if (!get_args(env, info, 4, argv)) return nullptr;
  AsyncJob *j = new AsyncJob();
  std::vector<float *> ch;
  size_t samples = 0;
  int32_t halo = 0;
  if (!get_ctx_or_devices(env, argv[0], j) || !get_channels(env, argv[1], &ch, &samples) ||
      napi_get_value_int32(env, argv[2], &halo) != napi_ok || !get_options(env, argv[3], &j->opts) ||
      samples % 512 || (int64_t)(samples / 512) < halo) {
    delete j;
    bool pending = false;
    napi_is_exception_pending(env, &pending);
    if (!pending) napi_throw_type_error(env, nullptr, "bad arguments");
    return nullptr;
  }
  j->encode = true;
  j->channels = (int)ch.size();
  j->halo = halo;
  j->frames = (int64_t)(samples / 512) - halo;
  for (size_t c = 0; c < ch.size(); c++) j->in[c] = ch[c] + (size_t)halo * 512;
  napi_value out = make_u8(env, (size_t)j->frames * ch.size() * C1_UNIT_BYTES, &j->units_out);
  if (!out || (j->frames > 0 && !j->units_out)) {
    delete j;
    napi_throw_error(env, nullptr, "could not allocate the result");
    return nullptr;
  }
  // the context (or device list) and the input arrays must outlive the job
  if (!keep_alive(env, j, argv[0]) || !keep_alive(env, j, argv[1])) { drop_job(env, j); return nullptr; }
  if (napi_create_reference(env, out, 1, &j->result) != napi_ok) { j->result = nullptr; drop_job(env, j); napi_throw_error(env, nullptr, "carta1: cannot reference the result"); return nullptr; }
  return start_job(env, j, "carta1.encodeBatch");
}
napi_value DecodeBatchAsync(napi_env env, napi_callback_info info) {
  napi_value argv[4];
  if (!get_args(env, info, 4, argv)) return nullptr;
  AsyncJob *j = new AsyncJob();
  void *d;
  size_t n;
  int32_t channels = 1, halo = 0;
  if (!get_ctx_or_devices(env, argv[0], j) || !get_typed(env, argv[1], napi_uint8_array, &d, &n) ||
      napi_get_value_int32(env, argv[2], &channels) != napi_ok || napi_get_value_int32(env, argv[3], &halo) != napi_ok ||
      channels < 1 || channels > 2 || n % ((size_t)channels * C1_UNIT_BYTES) ||
      (int64_t)(n / ((size_t)channels * C1_UNIT_BYTES)) < halo) {
    delete j;
    bool pending = false;
    napi_is_exception_pending(env, &pending);
    if (!pending) napi_throw_type_error(env, nullptr, "bad arguments");
    return nullptr;
  }
  j->encode = false;
  j->channels = channels;
  j->halo = halo;
  j->frames = (int64_t)(n / ((size_t)channels * C1_UNIT_BYTES)) - halo;
  j->units_in = static_cast<const uint8_t *>(d) + (size_t)halo * channels * C1_UNIT_BYTES;
  napi_value arr;
  napi_create_array_with_length(env, channels, &arr);
  for (int c = 0; c < channels; c++) {
    napi_value ta = make_f32(env, (size_t)j->frames * 512, &j->outp[c]);
    if (!ta || (j->frames > 0 && !j->outp[c])) {
      delete j;
      napi_throw_error(env, nullptr, "could not allocate the result");
      return nullptr;
    }
    napi_set_element(env, arr, c, ta);
  }
  if (!keep_alive(env, j, argv[0]) || !keep_alive(env, j, argv[1])) { drop_job(env, j); return nullptr; }   // context and units outlive the job
  if (napi_create_reference(env, arr, 1, &j->result) != napi_ok) { j->result = nullptr; drop_job(env, j); napi_throw_error(env, nullptr, "carta1: cannot reference the result"); return nullptr; }
  return start_job(env, j, "carta1.decodeBatch");
}

// ---- stateful streams: the native half of one encode()/decode() closure --------------------------------
void FinalizeEnc(napi_env, void *data, void *) { c1_enc_stream_destroy(static_cast<c1_enc_stream *>(data)); }
void FinalizeDec(napi_env, void *data, void *) { c1_dec_stream_destroy(static_cast<c1_dec_stream *>(data)); }
napi_value EncStreamCreate(napi_env env, napi_callback_info info) {
  napi_value argv[3];
  if (!get_args(env, info, 3, argv)) return nullptr;
  c1_ctx *ctx;
  int32_t channels = 1;
  c1_encode_options o;
  if (!get_external(env, argv[0], &ctx)) return nullptr;
  NAPI_OK(napi_get_value_int32(env, argv[1], &channels));
  if (!get_options(env, argv[2], &o)) return nullptr;
  c1_enc_stream *s = nullptr;
  const int rc = c1_enc_stream_create(ctx, channels, &o, &s);
  if (rc) return throw_c1(env, rc);
  napi_value ext;
  NAPI_OK(napi_create_external(env, s, FinalizeEnc, nullptr, &ext));
  // keep the context alive as long as the stream is
  napi_ref ref;
  napi_create_reference(env, argv[0], 1, &ref);
  return ext;
}
napi_value EncStreamPush(napi_env env, napi_callback_info info) {
  napi_value argv[2];
  if (!get_args(env, info, 2, argv)) return nullptr;
  c1_enc_stream *s;
  std::vector<float *> ch;
  size_t samples = 0;
  if (!get_external(env, argv[0], &s) || !get_channels(env, argv[1], &ch, &samples)) return nullptr;
  if (samples % 512) { napi_throw_type_error(env, nullptr, "PCM length must be a multiple of 512"); return nullptr; }
  const int64_t frames = (int64_t)(samples / 512);
  uint8_t *units;
  napi_value out = make_u8(env, (size_t)frames * ch.size() * C1_UNIT_BYTES, &units);
  const float *p[2] = {ch[0], ch.size() > 1 ? ch[1] : nullptr};
  const int rc = c1_enc_stream_push(s, p, frames, units);
  if (rc) return throw_c1(env, rc);
  return out;
}
napi_value DecStreamCreate(napi_env env, napi_callback_info info) {
  napi_value argv[2];
  if (!get_args(env, info, 2, argv)) return nullptr;
  c1_ctx *ctx;
  int32_t channels = 1;
  if (!get_external(env, argv[0], &ctx)) return nullptr;
  NAPI_OK(napi_get_value_int32(env, argv[1], &channels));
  c1_dec_stream *s = nullptr;
  const int rc = c1_dec_stream_create(ctx, channels, &s);
  if (rc) return throw_c1(env, rc);
  napi_value ext;
  NAPI_OK(napi_create_external(env, s, FinalizeDec, nullptr, &ext));
  napi_ref ref;
  napi_create_reference(env, argv[0], 1, &ref);
  return ext;
}
napi_value DecStreamPush(napi_env env, napi_callback_info info) {
  napi_value argv[3];
  if (!get_args(env, info, 3, argv)) return nullptr;
  c1_dec_stream *s;
  void *d;
  size_t n;
  int32_t channels = 1;
  if (!get_external(env, argv[0], &s) || !get_typed(env, argv[1], napi_uint8_array, &d, &n)) return nullptr;
  NAPI_OK(napi_get_value_int32(env, argv[2], &channels));
  if (channels < 1 || channels > 2 || n % ((size_t)channels * C1_UNIT_BYTES)) { napi_throw_type_error(env, nullptr, "units: wrong length"); return nullptr; }
  const int64_t frames = (int64_t)(n / ((size_t)channels * C1_UNIT_BYTES));
  napi_value arr;
  NAPI_OK(napi_create_array_with_length(env, channels, &arr));
  float *p[2] = {nullptr, nullptr};
  for (int c = 0; c < channels; c++) {
    napi_value ta = make_f32(env, (size_t)frames * 512, &p[c]);
    NAPI_OK(napi_set_element(env, arr, c, ta));
  }
  const int rc = c1_dec_stream_push(s, static_cast<const uint8_t *>(d), frames, p);
  if (rc) return throw_c1(env, rc);
  return arr;
}

// ---- the single-stage functions of the reference's export surface (codec/index.js:30-35,42) ---------------------------
napi_value Quantize(napi_env env, napi_callback_info info) {      // (ctx, Float32Array, sfi, bits) -> Int32Array
  napi_value argv[4];
  if (!get_args(env, info, 4, argv)) return nullptr;
  c1_ctx *ctx; void *d; size_t n; int32_t sfi = 0, bits = 0;
  if (!get_external(env, argv[0], &ctx) || !get_typed(env, argv[1], napi_float32_array, &d, &n)) return nullptr;
  NAPI_OK(napi_get_value_int32(env, argv[2], &sfi));
  NAPI_OK(napi_get_value_int32(env, argv[3], &bits));
  napi_value ab, ta; void *q;
  NAPI_OK(napi_create_arraybuffer(env, n * 4, &q, &ab));
  NAPI_OK(napi_create_typedarray(env, napi_int32_array, n, ab, 0, &ta));
  const int rc = c1_quantize(ctx, static_cast<const float *>(d), (int)n, sfi, bits, static_cast<int32_t *>(q));
  if (rc) return throw_c1(env, rc);
  return ta;
}
napi_value Dequantize(napi_env env, napi_callback_info info) {    // (ctx, Int32Array, sfi, bits) -> Float32Array
  napi_value argv[4];
  if (!get_args(env, info, 4, argv)) return nullptr;
  c1_ctx *ctx; void *d; size_t n; int32_t sfi = 0, bits = 0;
  if (!get_external(env, argv[0], &ctx) || !get_typed(env, argv[1], napi_int32_array, &d, &n)) return nullptr;
  NAPI_OK(napi_get_value_int32(env, argv[2], &sfi));
  NAPI_OK(napi_get_value_int32(env, argv[3], &bits));
  float *x;
  napi_value out = make_f32(env, n, &x);
  const int rc = c1_dequantize(ctx, static_cast<const int32_t *>(d), (int)n, sfi, bits, x);
  if (rc) return throw_c1(env, rc);
  return out;
}
napi_value Fft(napi_env env, napi_callback_info info) {           // (ctx, real, imag, Float64Array w) in place
  napi_value argv[4];
  if (!get_args(env, info, 4, argv)) return nullptr;
  c1_ctx *ctx; void *re, *im, *w; size_t n, n2, nw;
  if (!get_external(env, argv[0], &ctx) || !get_typed(env, argv[1], napi_float32_array, &re, &n) ||
      !get_typed(env, argv[2], napi_float32_array, &im, &n2) || !get_typed(env, argv[3], napi_float64_array, &w, &nw)) return nullptr;
  size_t stages = 0;
  while (((size_t)1 << stages) < n) stages++;
  if (n2 != n || nw < 2 * stages) { napi_throw_type_error(env, nullptr, "fft: real, imag of equal length and log2(n) twiddle pairs"); return nullptr; }
  const int rc = c1_fft(ctx, static_cast<float *>(re), static_cast<float *>(im), (int)n, static_cast<const double *>(w));
  if (rc) return throw_c1(env, rc);
  napi_value undef;
  napi_get_undefined(env, &undef);
  return undef;
}
napi_value QmfAnalysis(napi_env env, napi_callback_info info) {   // (ctx, Float32Array pcm incl. halo, haloFrames) -> Float32Array bands
  napi_value argv[3];
  if (!get_args(env, info, 3, argv)) return nullptr;
  c1_ctx *ctx; void *d; size_t n; int32_t halo = 0;
  if (!get_external(env, argv[0], &ctx) || !get_typed(env, argv[1], napi_float32_array, &d, &n)) return nullptr;
  NAPI_OK(napi_get_value_int32(env, argv[2], &halo));
  if (n % 512 || halo < 0 || (size_t)halo > n / 512) { napi_throw_type_error(env, nullptr, "qmfAnalysis: whole frames of 512 samples"); return nullptr; }
  const int64_t frames = (int64_t)(n / 512) - halo;
  float *b;
  napi_value out = make_f32(env, (size_t)frames * 512, &b);
  const int rc = c1_qmf_analysis_batch(ctx, static_cast<const float *>(d), frames, halo, b);
  if (rc) return throw_c1(env, rc);
  return out;
}
napi_value MdctFromBands(napi_env env, napi_callback_info info) { // (ctx, Float32Array bands incl. halo, haloFrames, Int32Array modes) -> [coefs, bandsWindowed]
  napi_value argv[4];
  if (!get_args(env, info, 4, argv)) return nullptr;
  c1_ctx *ctx; void *d, *m; size_t n, nm; int32_t halo = 0;
  if (!get_external(env, argv[0], &ctx) || !get_typed(env, argv[1], napi_float32_array, &d, &n)) return nullptr;
  NAPI_OK(napi_get_value_int32(env, argv[2], &halo));
  if (!get_typed(env, argv[3], napi_int32_array, &m, &nm)) return nullptr;
  if (n % 512 || halo < 0 || halo > 1 || (size_t)halo > n / 512 || nm != 3 * (n / 512 - (size_t)halo)) { napi_throw_type_error(env, nullptr, "mdctFromBands: whole frames and three block modes per frame"); return nullptr; }
  const int64_t frames = (int64_t)(n / 512) - halo;
  float *c, *bw;
  napi_value coefs = make_f32(env, (size_t)frames * 512, &c), windowed = make_f32(env, (size_t)frames * 512, &bw), arr;
  const int rc = c1_mdct_batch(ctx, static_cast<const float *>(d), frames, halo, static_cast<const int32_t *>(m), c, bw);
  if (rc) return throw_c1(env, rc);
  NAPI_OK(napi_create_array_with_length(env, 2, &arr));
  NAPI_OK(napi_set_element(env, arr, 0, coefs));
  NAPI_OK(napi_set_element(env, arr, 1, windowed));
  return arr;
}

napi_value Init(napi_env env, napi_value exports) {
  const napi_property_descriptor props[] = {
      {"abiVersion", nullptr, AbiVersion, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"deviceCount", nullptr, DeviceCount, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"getDefaultTables", nullptr, GetDefaultTables, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"setTables", nullptr, SetTables, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"ctxCreate", nullptr, CtxCreate, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"allocPinned", nullptr, AllocPinned, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"encodeBatch", nullptr, EncodeBatch, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"decodeBatch", nullptr, DecodeBatch, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"encodeWavBatch", nullptr, EncodeWavBatch, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"decodeWav16Batch", nullptr, DecodeWav16Batch, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"encodeBatchAsync", nullptr, EncodeBatchAsync, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"decodeBatchAsync", nullptr, DecodeBatchAsync, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"encStreamCreate", nullptr, EncStreamCreate, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"encStreamPush", nullptr, EncStreamPush, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"decStreamCreate", nullptr, DecStreamCreate, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"decStreamPush", nullptr, DecStreamPush, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"quantize", nullptr, Quantize, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"dequantize", nullptr, Dequantize, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"fft", nullptr, Fft, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"qmfAnalysis", nullptr, QmfAnalysis, nullptr, nullptr, nullptr, napi_default, nullptr},
      {"mdctFromBands", nullptr, MdctFromBands, nullptr, nullptr, nullptr, napi_default, nullptr},
  };
  napi_define_properties(env, exports, sizeof props / sizeof props[0], props);
  return exports;
}

}  // namespace

NAPI_MODULE(NODE_GYP_MODULE_NAME, Init)
