/* ORACLE — TEST INFRASTRUCTURE ONLY.
 * The reference seeds its RANSAC with srand(time(NULL)) (degensac/exp_ranH.c:823,
 * exp_ranF.c:832), so two runs never agree.  The reference library is linked with
 * -Wl,--wrap=time: its calls to time() land here and return a value the tests pin.  Nothing of
 * the reference's arithmetic is replaced. */
#include <time.h>

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
