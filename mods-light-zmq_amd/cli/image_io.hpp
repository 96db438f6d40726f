// Image input of the mods CLI: PNG (libpng), JPEG (libjpeg) and binary PGM/PPM, to a grey float image.
//
// Reference behaviour: cv::imread (mods.cpp:116-118).  With [Computing] LoadColor=1 a colour file stays
// colour and GenerateSynthImageCorr averages it, (B + G + R) / 3.0 on float planes (synth-detection.cpp:343-354): a cv::MatExpr
// that OpenCV evaluates as addWeighted(B + G, 1/3, R, 1/3, 0) with float weights, fused: fma(B + G, a, R * a), a = (float)(1/3.)
// (the reading that reproduces the reference's README counts, tools/readme_count_hunt.py);
// with LoadColor=0 imread itself converts, i.e. OpenCV's fixed-point BT.601 luma
// (R*4899 + G*9617 + B*1868 + 8192) >> 14.  PNG decoding is exact; JPEG is decoded with the image's libjpeg (ISLOW DCT, the
// library default, fancy upsampling on): a JPEG decoder's output is only defined up to +-1 per sample, so pixel values can
// differ from those of the libjpeg-turbo that an OpenCV build carries.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <png.h>

namespace modscli {

struct GreyImage { int w = 0, h = 0; std::vector<float> px; };

static inline float grey_of(const unsigned char *rgb, bool average) {
  if (average) {
    const float a = (float)(1.0 / 3.0);
    return std::fmaf((float)rgb[2] + (float)rgb[1], a, (float)rgb[0] * a);         // fma(B + G, a, R * a)
  }
  return (float)((rgb[0] * 4899 + rgb[1] * 9617 + rgb[2] * 1868 + 8192) >> 14);
}

// decoded images are limited to 2^28 pixels (16 k x 16 k): a crafted header must not become a multi-gigabyte allocation
constexpr double kMaxImagePixels = 268435456.0;

static bool read_pnm(const std::string &fn, bool average, GreyImage *out, std::string *err) {
  FILE *f = fopen(fn.c_str(), "rb");
  if (!f) { *err = "cannot open " + fn; return false; }
  char magic[3] = {0, 0, 0};
  int w = 0, h = 0, maxv = 0;
  auto next_int = [&](int *v) {
    int c;
    for (;;) {
      c = fgetc(f);
      if (c == '#') { while (c != '\n' && c != EOF) c = fgetc(f); continue; }
      if (c == EOF) return false;
      if (!isspace(c)) break;
    }
    int n = 0;
    while (c != EOF && isdigit(c)) { n = n * 10 + (c - '0'); c = fgetc(f); }
    *v = n;
    return true;
  };
  if (fread(magic, 1, 2, f) != 2 || magic[0] != 'P' || (magic[1] != '5' && magic[1] != '6')) { fclose(f); *err = fn + ": not a binary PGM/PPM"; return false; }
  if (!next_int(&w) || !next_int(&h) || !next_int(&maxv) || w <= 0 || h <= 0 || maxv <= 0 || maxv > 255) { fclose(f); *err = fn + ": bad PNM header"; return false; }
  if ((double)w * h > kMaxImagePixels) { fclose(f); *err = fn + ": image exceeds the limit of 2^28 pixels"; return false; }
  const int ch = magic[1] == '6' ? 3 : 1;
  std::vector<unsigned char> buf((size_t)w * h * ch);
  if (fread(buf.data(), 1, buf.size(), f) != buf.size()) { fclose(f); *err = fn + ": truncated"; return false; }
  fclose(f);
  out->w = w; out->h = h; out->px.resize((size_t)w * h);
  for (size_t i = 0; i < (size_t)w * h; i++) {
    if (ch == 1) out->px[i] = (float)buf[i];
    else out->px[i] = grey_of(&buf[3 * i], average);
  }
  return true;
}

static bool read_png(const std::string &fn, bool average, GreyImage *out, std::string *err) {
  png_image img;
  memset(&img, 0, sizeof(img));
  img.version = PNG_IMAGE_VERSION;
  if (!png_image_begin_read_from_file(&img, fn.c_str())) { *err = fn + ": " + img.message; return false; }
  if ((double)img.width * img.height > kMaxImagePixels) { png_image_free(&img); *err = fn + ": image exceeds the limit of 2^28 pixels"; return false; }
  const bool colour = (img.format & PNG_FORMAT_FLAG_COLOR) != 0;
  img.format = colour ? PNG_FORMAT_RGB : PNG_FORMAT_GRAY;      // 8-bit, alpha dropped as cv::imread(IMREAD_COLOR) does
  std::vector<unsigned char> buf(PNG_IMAGE_SIZE(img));
  if (!png_image_finish_read(&img, nullptr, buf.data(), 0, nullptr)) { *err = fn + ": " + img.message; png_image_free(&img); return false; }
  out->w = (int)img.width; out->h = (int)img.height;
  out->px.resize((size_t)out->w * out->h);
  for (size_t i = 0; i < out->px.size(); i++) {
    if (!colour) out->px[i] = (float)buf[i];
    else out->px[i] = grey_of(&buf[3 * i], average);
  }
  return true;
}

// libmodsjpeg.so (cli/jpeg_read.c): the decoder sits behind a C function so that only that library carries the run path of
// the tree libjpeg lives in
extern "C" int mods_jpeg_read(const char *fn, int colour, unsigned char **out, int *w, int *h, int *ch, char *err);
extern "C" void mods_jpeg_free(unsigned char *p);
static bool read_jpeg(const std::string &fn, bool average, GreyImage *out, std::string *err) {
  unsigned char *buf = nullptr;
  int w = 0, h = 0, ch = 0;
  char msg[256] = {0};
  if (mods_jpeg_read(fn.c_str(), average ? 1 : 0, &buf, &w, &h, &ch, msg)) { *err = msg; return false; }
  out->w = w; out->h = h; out->px.resize((size_t)w * h);
  for (size_t i = 0; i < (size_t)w * h; i++) out->px[i] = ch == 1 ? (float)buf[i] : grey_of(&buf[3 * i], average);
  mods_jpeg_free(buf);
  return true;
}

// average = true: LoadColor=1 semantics ((B+G+R)/3); false: imread-as-grey semantics
static bool read_image(const std::string &fn, bool average, GreyImage *out, std::string *err) {
  FILE *f = fopen(fn.c_str(), "rb");
  if (!f) { *err = "cannot open " + fn; return false; }
  unsigned char sig[8] = {0};
  const size_t n = fread(sig, 1, 8, f);
  fclose(f);
  if (n >= 8 && !png_sig_cmp(sig, 0, 8)) return read_png(fn, average, out, err);
  if (n >= 2 && sig[0] == 'P' && (sig[1] == '5' || sig[1] == '6')) return read_pnm(fn, average, out, err);
  if (n >= 2 && sig[0] == 0xFF && sig[1] == 0xD8) return read_jpeg(fn, average, out, err);
  *err = fn + ": unknown image format (PNG, JPEG, binary PGM/PPM supported)";
  return false;
}

}  // namespace modscli
