/* normals_shim.c -- TEST INFRASTRUCTURE (ours; calls only the reference's public mtrand API, src/util/mtrand/mtrand.h).
 * LD_PRELOADed into a reference deck executable, it records every normal the run draws through mt_drandn -- the value
 * and how many 32-bit words of the Mersenne twister the draw consumed (the reference's ziggurat takes a variable number:
 * mtrand.c:395-440) -- into $VPIC_NORMALS_OUT.<rank> as {double value; uint8 words} arrays.  The HIP deck host replays
 * such a file (VPIC_HIP_NORMALS) so that a deck that calls maxwellian_rand loads the reference's particles bit for
 * bit without the reference's ziggurat tables entering this repository.  Built and used by oracle/trecon.py only. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef struct mt_rng mt_rng_t;
static double (*real_drandn)(mt_rng_t *);
static size_t (*get_size)(mt_rng_t *);
static void (*get_state)(mt_rng_t *, void *);
static void (*set_state)(mt_rng_t *, const void *, size_t);
static mt_rng_t *(*new_rng)(unsigned);
static unsigned (*urand)(mt_rng_t *);
static mt_rng_t *shadow;
static void *s0, *s1, *s2;
static size_t sz;
static double *val; static unsigned char *wrd; static long n, cap;

static void boot(mt_rng_t *rng) {
  real_drandn = (double (*)(mt_rng_t *))dlsym(RTLD_NEXT, "mt_drandn");
  get_size = (size_t (*)(mt_rng_t *))dlsym(RTLD_NEXT, "get_mt_rng_size");
  get_state = (void (*)(mt_rng_t *, void *))dlsym(RTLD_NEXT, "get_mt_rng_state");
  set_state = (void (*)(mt_rng_t *, const void *, size_t))dlsym(RTLD_NEXT, "set_mt_rng_state");
  new_rng = (mt_rng_t *(*)(unsigned))dlsym(RTLD_NEXT, "new_mt_rng");
  urand = (unsigned (*)(mt_rng_t *))dlsym(RTLD_NEXT, "mt_urand");
  sz = get_size(rng);
  s0 = malloc(sz); s1 = malloc(sz); s2 = malloc(sz);
  shadow = new_rng(0);
}

double mt_drandn(mt_rng_t *rng) {
  if (!real_drandn) boot(rng);
  get_state(rng, s0);
  const double v = real_drandn(rng);
  get_state(rng, s1);
  /* step a copy of the generator word by word until it is where the real one is */
  set_state(shadow, s0, sz);
  int words = 0;
  for (;;) {
    get_state(shadow, s2);
    if (!memcmp(s1, s2, sz)) break;
    if (++words > 200) { fprintf(stderr, "normals_shim: cannot follow the generator\n"); abort(); }
    (void)urand(shadow);
  }
  if (n == cap) { cap = cap ? 2 * cap : 1 << 16; val = realloc(val, cap * sizeof(double)); wrd = realloc(wrd, cap); }
  val[n] = v; wrd[n] = (unsigned char)words; n++;
  return v;
}

__attribute__((destructor)) static void done(void) {
  const char *base = getenv("VPIC_NORMALS_OUT");
  if (!base || !n) return;
  const char *rank = getenv("PMI_RANK");
  char path[4096];
  snprintf(path, sizeof(path), "%s.%s", base, rank ? rank : "0");
  FILE *f = fopen(path, "wb");
  if (!f) return;
  long long count = n;
  fwrite(&count, 8, 1, f); fwrite(val, 8, n, f); fwrite(wrd, 1, n, f);
  fclose(f);
}
