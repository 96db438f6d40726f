/* JPEG decoding for the mods CLI, in its own small shared library (libmodsjpeg.so): the image's libjpeg lives in the conda tree,
 * and a run path to that tree on the executable itself would also redirect its libstdc++.  Plain C, no other dependency.
 * cv::imread semantics (mods.cpp:116-118): colour != 0 -> 3 interleaved RGB bytes per pixel (grey files replicated by libjpeg),
 * colour == 0 -> libjpeg's own grey conversion.  ISLOW DCT (the library default). */
#include <setjmp.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <jpeglib.h>

struct mods_jpeg_err { struct jpeg_error_mgr pub; jmp_buf jb; char msg[JMSG_LENGTH_MAX]; };

static void mods_jpeg_fail(j_common_ptr c) {
  struct mods_jpeg_err *e = (struct mods_jpeg_err *)c->err;
  (*c->err->format_message)(c, e->msg);
  longjmp(e->jb, 1);
}

/* Returns 0 and a malloc'ed buffer of w * h * ch bytes (release with mods_jpeg_free), or -1 and a message in err[256]. */
int mods_jpeg_read(const char *fn, int colour, unsigned char **out, int *w, int *h, int *ch, char *err) {
  FILE *f = fopen(fn, "rb");
  struct jpeg_decompress_struct d;
  struct mods_jpeg_err je;
  unsigned char *volatile buf = NULL;
  volatile int created = 0;       /* jpeg_create_decompress itself can fail (library / header mismatch): nothing to destroy then */
  if (!f) { snprintf(err, 256, "cannot open %s", fn); return -1; }
  memset(&d, 0, sizeof(d));
  d.err = jpeg_std_error(&je.pub);
  je.pub.error_exit = mods_jpeg_fail;
  if (setjmp(je.jb)) {
    if (created) jpeg_destroy_decompress(&d);
    fclose(f); free(buf);
    snprintf(err, 256, "%s: %s", fn, je.msg);
    return -1;
  }
  jpeg_create_decompress(&d);
  created = 1;
  jpeg_stdio_src(&d, f);
  jpeg_read_header(&d, TRUE);
  d.out_color_space = colour ? JCS_RGB : JCS_GRAYSCALE;
  jpeg_start_decompress(&d);
  *w = (int)d.output_width; *h = (int)d.output_height; *ch = (int)d.output_components;
  /* the same pixel limit as the other decoders of the command line (MODS_MAX_IMAGE_PIXELS, image_io.hpp): a crafted header
   * must not turn into a multi-gigabyte allocation */
  if ((double)*w * (double)*h > 268435456.0) {
    jpeg_destroy_decompress(&d); fclose(f);
    snprintf(err, 256, "%s: %d x %d pixels exceed the limit of 2^28", fn, *w, *h);
    return -1;
  }
  if (*w <= 0 || *h <= 0 || (*ch != 1 && *ch != 3)) {
    jpeg_destroy_decompress(&d); fclose(f);
    snprintf(err, 256, "%s: unsupported JPEG layout", fn);
    return -1;
  }
  buf = (unsigned char *)malloc((size_t)*w * *h * *ch);
  if (!buf) { jpeg_destroy_decompress(&d); fclose(f); snprintf(err, 256, "%s: out of memory", fn); return -1; }
  while (d.output_scanline < d.output_height) {
    unsigned char *rp = buf + (size_t)d.output_scanline * *w * *ch;
    jpeg_read_scanlines(&d, &rp, 1);
  }
  jpeg_finish_decompress(&d);
  jpeg_destroy_decompress(&d);
  fclose(f);
  *out = buf;
  return 0;
}

void mods_jpeg_free(unsigned char *p) { free(p); }
