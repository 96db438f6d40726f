/*
 * mods_zmq.h — the ZMQ descriptor-daemon wire protocol of MODS, both ends, as a small C library
 * (libmodszmq.so; links libzmq and libpng, neither of which libmodsgpu.so depends on).
 *
 * Replaces / serves (reference root relative):
 *   client  DescribeWithZmq                  imagerepresentation.cpp:21-103
 *           request  = ONE message: the PNG encoding of an 8-bit single-channel image of size (ps*n) x ps,
 *                      the n patches stacked in a column (ExtractPatchesColumn, synth-detection.cpp:38-132,
 *                      fp32 -> 8 bit as cv::imencode does: convertTo(CV_8U) = round half to even, saturate);
 *                      at most 2000 patches per request, longer lists go out in consecutive requests
 *           reply    = ONE message: n * dim little-endian float32, row-major; dim = bytes / 4 / n
 *           a fresh REQ socket is connected for every request and closed afterwards
 *   server  build/desc_server.py:107-127, affnet_server.py, orinet_server.py: REP socket bound to tcp://\*:port,
 *           one reply per request
 */
#ifndef MODS_ZMQ_H
#define MODS_ZMQ_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MODS_ZMQ_OK 0
#define MODS_ZMQ_E_ARG (-1)
#define MODS_ZMQ_E_PNG (-2)
#define MODS_ZMQ_E_SOCKET (-3)
#define MODS_ZMQ_E_REPLY (-4)      /* reply size is not a multiple of 4 * n, or does not fit the output buffer */
#define MODS_ZMQ_MAX_PATCHES 2000  /* per request (imagerepresentation.cpp:27) */
#define MODS_ZMQ_MAX_PATCH_SIZE 256  /* px; the shipped configurations use 32 (a request wider than this is refused) */

const char *mods_zmq_last_error(void);

/* client: n fp32 patches [n][ps][ps] -> out[n][*dim].  timeout_ms <= 0: wait for ever (the reference does). */
int mods_zmq_describe(const char *endpoint, const float *patches, int n, int ps, float *out, size_t out_cap_floats, int *dim,
                      int timeout_ms);

/* mods_descriptor_fn of include/mods_hip.h: user = endpoint string (e.g. "tcp://localhost:5555"); waits for ever */
int mods_zmq_descriptor_hook(void *user, const float *patches, int n, int ps, float *out, size_t out_cap_floats, int *dim);

/* the two message bodies */
int mods_zmq_encode_request(const float *patches, int n, int ps, unsigned char **png, size_t *len);      /* free with mods_zmq_free */
int mods_zmq_decode_request(const unsigned char *png, size_t len, unsigned char **pixels, int *n, int *ps);   /* pixels: (n*ps) x ps bytes */
void mods_zmq_free(void *p);

/* server: binds a REP socket and answers requests with `model`.  model(user, patches_u8 [n][ps][ps], n, ps, out, out_cap_floats,
 * &dim) fills out[n][dim] and returns 0.  Returns after max_requests requests (<= 0: never), or when a request of zero
 * bytes arrives (shutdown message; answered with an empty reply). */
typedef int (*mods_zmq_model_fn)(void *user, const unsigned char *patches, int n, int ps, float *out, size_t out_cap_floats, int *dim);
int mods_zmq_serve(const char *bind_endpoint, mods_zmq_model_fn model, void *user, int max_requests);

#ifdef __cplusplus
}
#endif
#endif
