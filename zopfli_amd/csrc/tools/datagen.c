/*
 * Deterministic synthetic corpora for parity tests and bench.py.
 *
 * enwik8 / Calgary / Silesia are not available offline, so every workload in this
 * repo is one of six seeded classes.  All randomness comes from one 64-bit LCG
 * (x = x*6364136223846793005 + 1442695040888963407, output x>>33), so the bytes
 * are identical on every toolchain and on the GPU box.
 *
 *   T  text-like   : 4096-word vocabulary, Zipf(1) word choice, sentences
 *   X  markup-like : <rec id=".." ts=".." kind="..">..</rec> records
 *   R  uniform random bytes
 *   Z  long runs of 4 symbols (run lengths 1..5000)
 *   B  two-symbol i.i.d., p = 0.85
 *   P  PNG-like    : per-row filter byte + filtered smooth RGBA gradient + noise
 *   M  mixed       : concatenation of the above in runs (Silesia stand-in)
 */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint64_t x; } Lcg;

static uint32_t lcg_next(Lcg* g) {
  g->x = g->x * 6364136223846793005ULL + 1442695040888963407ULL;
  return (uint32_t)(g->x >> 33);
}
static uint32_t lcg_below(Lcg* g, uint32_t n) { return lcg_next(g) % n; }

/* ---------------------------------------------------------------- text --- */
#define VOCAB 4096
typedef struct {
  char words[VOCAB][12];
  uint8_t len[VOCAB];
  uint32_t cum[VOCAB]; /* cumulative Zipf weights, scaled */
} Vocab;

static void vocab_init(Vocab* v, Lcg* g) {
  /* rough English letter frequencies (per 1000) */
  static const char letters[] = "etaoinshrdlcumwfgypbvkjxqz";
  static const int freq[] = {127, 91, 82, 75, 70, 67, 63, 61, 60, 43, 40, 28, 28,
                             24, 24, 22, 20, 20, 19, 15, 10, 8, 2, 2, 1, 1};
  int cumf[26], tot = 0, i, j;
  uint64_t acc = 0;
  for (i = 0; i < 26; i++) { tot += freq[i]; cumf[i] = tot; }
  for (i = 0; i < VOCAB; i++) {
    int L = 2 + (int)lcg_below(g, 9); /* 2..10 */
    v->len[i] = (uint8_t)L;
    for (j = 0; j < L; j++) {
      int r = (int)lcg_below(g, (uint32_t)tot), k = 0;
      while (cumf[k] <= r) k++;
      v->words[i][j] = letters[k];
    }
    v->words[i][L] = 0;
    acc += (uint64_t)(1u << 24) / (uint64_t)(i + 1); /* Zipf(1) */
    v->cum[i] = (uint32_t)acc;
  }
}

static int vocab_pick(const Vocab* v, Lcg* g) {
  uint32_t r = lcg_next(g) % v->cum[VOCAB - 1];
  int lo = 0, hi = VOCAB - 1;
  while (lo < hi) {
    int mid = (lo + hi) >> 1;
    if (v->cum[mid] > r) hi = mid; else lo = mid + 1;
  }
  return lo;
}

static void gen_text(uint8_t* out, size_t n, uint64_t seed) {
  Lcg g; Vocab* v = (Vocab*)malloc(sizeof(Vocab));
  size_t p = 0; int sentence_left, cap = 1;
  g.x = seed;
  vocab_init(v, &g);
  sentence_left = 4 + (int)lcg_below(&g, 14);
  while (p < n) {
    int w = vocab_pick(v, &g), j;
    for (j = 0; j < v->len[w] && p < n; j++) {
      char c = v->words[w][j];
      if (cap && j == 0) c = (char)(c - 32);
      out[p++] = (uint8_t)c;
    }
    cap = 0;
    if (--sentence_left == 0) {
      if (p < n) out[p++] = '.';
      if (lcg_below(&g, 8) == 0) { if (p < n) out[p++] = '\n'; }
      else if (p < n) out[p++] = ' ';
      sentence_left = 4 + (int)lcg_below(&g, 14);
      cap = 1;
    } else {
      if (lcg_below(&g, 12) == 0 && p < n) out[p++] = ',';
      if (p < n) out[p++] = ' ';
    }
  }
  free(v);
}

/* -------------------------------------------------------------- markup --- */
static size_t put_str(uint8_t* out, size_t p, size_t n, const char* s) {
  while (*s && p < n) out[p++] = (uint8_t)*s++;
  return p;
}
static size_t put_num(uint8_t* out, size_t p, size_t n, uint32_t v) {
  char buf[16]; int k = 0;
  do { buf[k++] = (char)('0' + v % 10); v /= 10; } while (v);
  while (k && p < n) out[p++] = (uint8_t)buf[--k];
  return p;
}

static void gen_markup(uint8_t* out, size_t n, uint64_t seed) {
  static const char* kinds[] = {"create", "update", "delete", "touch", "merge", "split"};
  Lcg g; Vocab* v = (Vocab*)malloc(sizeof(Vocab));
  size_t p = 0; uint32_t id = 1000, ts = 1700000000u;
  g.x = seed;
  vocab_init(v, &g);
  while (p < n) {
    int nw = 1 + (int)lcg_below(&g, 9), i;
    p = put_str(out, p, n, "<rec id=\"");
    p = put_num(out, p, n, id); id += 1 + lcg_below(&g, 3);
    p = put_str(out, p, n, "\" ts=\"");
    p = put_num(out, p, n, ts); ts += lcg_below(&g, 600);
    p = put_str(out, p, n, "\" kind=\"");
    p = put_str(out, p, n, kinds[lcg_below(&g, 6)]);
    p = put_str(out, p, n, "\">");
    for (i = 0; i < nw; i++) {
      int w = vocab_pick(v, &g);
      if (i) p = put_str(out, p, n, " ");
      p = put_str(out, p, n, v->words[w]);
    }
    p = put_str(out, p, n, "</rec>\n");
  }
  free(v);
}

/* --------------------------------------------------------------- others --- */
static void gen_random(uint8_t* out, size_t n, uint64_t seed) {
  Lcg g; size_t p; g.x = seed;
  for (p = 0; p < n; p++) out[p] = (uint8_t)(lcg_next(&g) >> 11);
}

static void gen_runs(uint8_t* out, size_t n, uint64_t seed) {
  static const int runlen[8] = {1, 2, 3, 5, 40, 300, 700, 5000};
  static const uint8_t sym[4] = {0x00, 0x20, 0x41, 0xff};
  Lcg g; size_t p = 0; g.x = seed;
  while (p < n) {
    int L = runlen[lcg_below(&g, 8)];
    uint8_t c = sym[lcg_below(&g, 4)];
    while (L-- && p < n) out[p++] = c;
  }
}

static void gen_twosym(uint8_t* out, size_t n, uint64_t seed) {
  Lcg g; size_t p; g.x = seed;
  for (p = 0; p < n; p++) out[p] = (lcg_below(&g, 100) < 85) ? 'a' : 'b';
}

static uint8_t paeth(int a, int b, int c) {
  int pp = a + b - c, pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c);
  if (pa <= pb && pa <= pc) return (uint8_t)a;
  return (uint8_t)(pb <= pc ? b : c);
}

static void gen_pnglike(uint8_t* out, size_t n, uint64_t seed) {
  /* rows of 1 + 4*W bytes, Paeth-filtered RGBA gradient with low-amplitude noise */
  const int W = 1024;
  const size_t stride = (size_t)4 * W;
  uint8_t* prev = (uint8_t*)calloc(stride, 1);
  uint8_t* cur = (uint8_t*)malloc(stride);
  Lcg g; size_t p = 0; int y = 0;
  g.x = seed;
  while (p < n) {
    int x, ch;
    for (x = 0; x < W; x++) {
      for (ch = 0; ch < 4; ch++) {
        int base = (ch == 3) ? 255 : ((x * (ch + 1)) / 8 + y * (3 - ch) / 4);
        int noise = (int)lcg_below(&g, 4) == 0 ? (int)lcg_below(&g, 3) - 1 : 0;
        cur[4 * x + ch] = (uint8_t)(base + noise);
      }
    }
    out[p++] = 4; /* filter type: Paeth */
    for (x = 0; x < (int)stride && p < n; x++) {
      int a = x >= 4 ? cur[x - 4] : 0, b = prev[x], c = x >= 4 ? prev[x - 4] : 0;
      out[p++] = (uint8_t)(cur[x] - paeth(a, b, c));
    }
    memcpy(prev, cur, stride);
    y++;
  }
  free(prev); free(cur);
}

static void gen_class(char cls, uint8_t* out, size_t n, uint64_t seed);

static void gen_mixed(uint8_t* out, size_t n, uint64_t seed) {
  static const char order[] = "TXPRZTBX";
  Lcg g; size_t p = 0; int k = 0; g.x = seed;
  while (p < n) {
    /* runs of ~1/16 of the total, min 4 KiB; B kept short (pathological tail) */
    size_t run = n / 16 + 4096 + lcg_below(&g, 4096);
    char cls = order[k++ % 8];
    if (cls == 'B' && run > 65536) run = 65536;
    if (run > n - p) run = n - p;
    gen_class(cls, out + p, run, seed * 31 + (uint64_t)k);
    p += run;
  }
}

static void gen_class(char cls, uint8_t* out, size_t n, uint64_t seed) {
  switch (cls) {
    case 'T': gen_text(out, n, seed); break;
    case 'X': gen_markup(out, n, seed); break;
    case 'R': gen_random(out, n, seed); break;
    case 'Z': gen_runs(out, n, seed); break;
    case 'B': gen_twosym(out, n, seed); break;
    case 'P': gen_pnglike(out, n, seed); break;
    case 'M': gen_mixed(out, n, seed); break;
    default: memset(out, 0, n); break;
  }
}

/* C ABI used through ctypes: fill out[0..n) with class `cls`, given seed. */
int zopfli_amd_datagen(char cls, uint64_t seed, uint8_t* out, size_t n) {
  if (!out && n) return -1;
  gen_class(cls, out, n, seed);
  return 0;
}
