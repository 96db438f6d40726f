// libmodszmq: the MODS descriptor-daemon protocol (see include/mods_zmq.h for the reference citations).
#include "../../include/mods_zmq.h"
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <png.h>
#include <zmq.h>

static thread_local std::string g_err;
static void set_err(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}

extern "C" {

const char *mods_zmq_last_error(void) { return g_err.c_str(); }
void mods_zmq_free(void *p) { free(p); }

// cv::Mat::convertTo(CV_8U) of a float: saturate_cast<uchar>(cvRound(v)) - round half to even, clamp to 0..255
static inline unsigned char to_u8(float v) {
  const double r = nearbyint((double)v);      // default rounding mode: ties to even
  if (!(r > 0)) return 0;
  return r >= 255 ? 255 : (unsigned char)r;
}

int mods_zmq_encode_request(const float *patches, int n, int ps, unsigned char **png, size_t *len) {
  if (!patches || n <= 0 || ps <= 0 || !png || !len) { set_err("encode: bad arguments"); return MODS_ZMQ_E_ARG; }
  std::vector<unsigned char> pix((size_t)n * ps * ps);
  for (size_t i = 0; i < pix.size(); i++) pix[i] = to_u8(patches[i]);
  png_image img;
  memset(&img, 0, sizeof(img));
  img.version = PNG_IMAGE_VERSION;
  img.width = (png_uint_32)ps;
  img.height = (png_uint_32)((size_t)n * ps);
  img.format = PNG_FORMAT_GRAY;
  png_alloc_size_t bytes = 0;
  if (!png_image_write_to_memory(&img, nullptr, &bytes, 0, pix.data(), 0, nullptr)) { set_err("png: %s", img.message); return MODS_ZMQ_E_PNG; }
  unsigned char *buf = (unsigned char *)malloc(bytes);
  if (!buf) { set_err("out of memory"); return MODS_ZMQ_E_PNG; }
  if (!png_image_write_to_memory(&img, buf, &bytes, 0, pix.data(), 0, nullptr)) { free(buf); set_err("png: %s", img.message); return MODS_ZMQ_E_PNG; }
  *png = buf; *len = (size_t)bytes;
  return MODS_ZMQ_OK;
}

int mods_zmq_decode_request(const unsigned char *png, size_t len, unsigned char **pixels, int *n, int *ps) {
  if (!png || !len || !pixels || !n || !ps) { set_err("decode: bad arguments"); return MODS_ZMQ_E_ARG; }
  png_image img;
  memset(&img, 0, sizeof(img));
  img.version = PNG_IMAGE_VERSION;
  if (!png_image_begin_read_from_memory(&img, png, len)) { set_err("png: %s", img.message); return MODS_ZMQ_E_PNG; }
  img.format = PNG_FORMAT_GRAY;          // cv2.imdecode(buf, 0): 8-bit grey whatever the file holds
  if (img.width == 0 || img.height % img.width != 0) { png_image_free(&img); set_err("request image %ux%u is not a column of square patches", img.width, img.height); return MODS_ZMQ_E_PNG; }
  // the protocol sends at most MODS_ZMQ_MAX_PATCHES patches per request (imagerepresentation.cpp:27-61); anything larger is
  // refused before a byte of it is decompressed (decompression bombs, absurd patch sizes)
  if (img.width > MODS_ZMQ_MAX_PATCH_SIZE || img.height / img.width > MODS_ZMQ_MAX_PATCHES) {
    set_err("request image %ux%u exceeds the protocol limits (%d patches of at most %d px)", img.width, img.height, MODS_ZMQ_MAX_PATCHES, MODS_ZMQ_MAX_PATCH_SIZE);
    png_image_free(&img);
    return MODS_ZMQ_E_ARG;
  }
  unsigned char *buf = (unsigned char *)malloc(PNG_IMAGE_SIZE(img));
  if (!buf || !png_image_finish_read(&img, nullptr, buf, 0, nullptr)) { free(buf); set_err("png: %s", img.message); png_image_free(&img); return MODS_ZMQ_E_PNG; }
  *pixels = buf; *ps = (int)img.width; *n = (int)(img.height / img.width);
  return MODS_ZMQ_OK;
}

// one request / reply: at most MODS_ZMQ_MAX_PATCHES patches
static int request_once(const char *endpoint, const float *patches, int n, int ps, float *out, size_t cap, int *dim, int timeout_ms) {
  unsigned char *png = nullptr;
  size_t len = 0;
  int rc = mods_zmq_encode_request(patches, n, ps, &png, &len);
  if (rc) return rc;
  void *ctx = zmq_ctx_new();
  void *sock = ctx ? zmq_socket(ctx, ZMQ_REQ) : nullptr;
  rc = MODS_ZMQ_E_SOCKET;
  zmq_msg_t reply;
  bool have_reply = false;
  do {
    if (!sock) { set_err("zmq: %s", zmq_strerror(zmq_errno())); break; }
    const int linger = 0;
    zmq_setsockopt(sock, ZMQ_LINGER, &linger, sizeof(linger));
    if (timeout_ms > 0) { zmq_setsockopt(sock, ZMQ_RCVTIMEO, &timeout_ms, sizeof(timeout_ms)); zmq_setsockopt(sock, ZMQ_SNDTIMEO, &timeout_ms, sizeof(timeout_ms)); }
    if (zmq_connect(sock, endpoint) != 0) { set_err("zmq_connect(%s): %s", endpoint, zmq_strerror(zmq_errno())); break; }
    if (zmq_send(sock, png, len, 0) < 0) { set_err("zmq_send: %s", zmq_strerror(zmq_errno())); break; }
    zmq_msg_init(&reply);
    have_reply = true;
    if (zmq_msg_recv(&reply, sock, 0) < 0) { set_err("zmq_recv from %s: %s", endpoint, zmq_strerror(zmq_errno())); break; }
    const size_t bytes = zmq_msg_size(&reply);
    const size_t floats = bytes / sizeof(float);
    if (bytes % sizeof(float) != 0 || floats % (size_t)n != 0 || floats == 0) { set_err("reply of %zu bytes for %d patches", bytes, n); rc = MODS_ZMQ_E_REPLY; break; }
    if (floats > cap) { set_err("reply of %zu floats exceeds the output buffer (%zu)", floats, cap); rc = MODS_ZMQ_E_REPLY; break; }
    memcpy(out, zmq_msg_data(&reply), bytes);
    *dim = (int)(floats / (size_t)n);       // desc_size = inMsg.size() / kps.size()
    rc = MODS_ZMQ_OK;
  } while (0);
  if (have_reply) zmq_msg_close(&reply);
  if (sock) zmq_close(sock);
  if (ctx) zmq_ctx_term(ctx);
  free(png);
  return rc;
}

int mods_zmq_describe(const char *endpoint, const float *patches, int n, int ps, float *out, size_t out_cap_floats, int *dim, int timeout_ms) {
  if (!endpoint || !dim || n < 0 || ps <= 0 || (n > 0 && (!patches || !out))) { set_err("describe: bad arguments"); return MODS_ZMQ_E_ARG; }
  *dim = 0;
  size_t written = 0;
  for (int done = 0; done < n; done += MODS_ZMQ_MAX_PATCHES) {       // consecutive requests of at most 2000 patches
    const int m = n - done < MODS_ZMQ_MAX_PATCHES ? n - done : MODS_ZMQ_MAX_PATCHES;
    int d = 0;
    const int rc = request_once(endpoint, patches + (size_t)done * ps * ps, m, ps, out + written, out_cap_floats - written, &d, timeout_ms);
    if (rc) return rc;
    if (*dim && d != *dim) { set_err("descriptor size changed between requests (%d, %d)", *dim, d); return MODS_ZMQ_E_REPLY; }
    *dim = d;
    written += (size_t)m * d;
  }
  return MODS_ZMQ_OK;
}

int mods_zmq_descriptor_hook(void *user, const float *patches, int n, int ps, float *out, size_t out_cap_floats, int *dim) {
  return mods_zmq_describe((const char *)user, patches, n, ps, out, out_cap_floats, dim, 0);
}

int mods_zmq_serve(const char *bind_endpoint, mods_zmq_model_fn model, void *user, int max_requests) {
  if (!bind_endpoint || !model) { set_err("serve: bad arguments"); return MODS_ZMQ_E_ARG; }
  void *ctx = zmq_ctx_new();
  void *sock = ctx ? zmq_socket(ctx, ZMQ_REP) : nullptr;
  if (!sock || zmq_bind(sock, bind_endpoint) != 0) {
    set_err("zmq_bind(%s): %s", bind_endpoint, zmq_strerror(zmq_errno()));
    if (sock) zmq_close(sock);
    if (ctx) zmq_ctx_term(ctx);
    return MODS_ZMQ_E_SOCKET;
  }
  int rc = MODS_ZMQ_OK;
  std::vector<float> out;
  for (int served = 0; max_requests <= 0 || served < max_requests; served++) {
    zmq_msg_t req;
    zmq_msg_init(&req);
    if (zmq_msg_recv(&req, sock, 0) < 0) { set_err("zmq_recv: %s", zmq_strerror(zmq_errno())); zmq_msg_close(&req); rc = MODS_ZMQ_E_SOCKET; break; }
    const size_t len = zmq_msg_size(&req);
    if (len == 0) { zmq_msg_close(&req); zmq_send(sock, "", 0, 0); break; }      // shutdown request
    unsigned char *pix = nullptr;
    int n = 0, ps = 0, dim = 0;
    int e = mods_zmq_decode_request((const unsigned char *)zmq_msg_data(&req), len, &pix, &n, &ps);
    zmq_msg_close(&req);
    if (!e) {
      out.resize((size_t)n * 512);
      e = model(user, pix, n, ps, out.data(), out.size(), &dim);
      if (!e && (dim <= 0 || (size_t)dim * n > out.size())) e = MODS_ZMQ_E_REPLY;
    }
    free(pix);
    // a REP socket must answer: an empty reply tells the client that the request was not understood
    if (e) zmq_send(sock, "", 0, 0);
    else zmq_send(sock, out.data(), sizeof(float) * (size_t)n * dim, 0);
  }
  zmq_close(sock);
  zmq_ctx_term(ctx);
  return rc;
}

}  // extern "C"
