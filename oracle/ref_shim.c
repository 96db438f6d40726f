/* ORACLE — TEST INFRASTRUCTURE ONLY.
 * The reference seeds its RANSAC with srand(time(NULL)) (degensac/exp_ranH.c:823,
 * exp_ranF.c:832), so two runs never agree.  The reference library is linked with
 * -Wl,--wrap=time: its calls to time() land here and return a value the tests pin.  Nothing of
 * the reference's arithmetic is replaced.
 *
 * LAPACK integer width: degensac/lapwrap.h:12 declares `typedef ptrdiff_t lapack_int` (64-bit) and
 * passes such integers (and reads `info`) by address.  The LAPACK of this image is MKL's single
 * dynamic library, whose default interface takes 32-bit integers: it would write only the low half
 * of the uninitialised 64-bit `info`, and lap_SVD/lap_eig would report failure depending on stack
 * garbage (singulF then replaces F by the identity, Ftools.c:284-287).  MKL is therefore switched
 * to its 64-bit-integer interface before its first use, which is what the reference's declaration
 * asks for. */
#include <stdlib.h>
#include <time.h>

__attribute__((constructor)) static void oracle_ref_lapack_ilp64(void) { setenv("MKL_INTERFACE_LAYER", "ILP64", 1); }

time_t __real_time(time_t *t);

static long g_pinned = -1;

void oracle_ref_pin_time(long v) { g_pinned = v; }   /* v < 0: unpin */

time_t __wrap_time(time_t *t) {
  if (g_pinned >= 0) {
    if (t) *t = (time_t)g_pinned;
    return (time_t)g_pinned;
  }
  return __real_time(t);
}
