/* ORACLE — test infrastructure only. MD5 (RFC 1321) for the HEVC decoded-picture-hash SEI the CTC streams carry
 * (cfg/hm/ctc-hm-geometry-ai.cfg:65). The reference vendors dependencies/libmd5/libmd5.c for the same purpose;
 * this is an independent restatement, pinned against python hashlib in tests/test_oracle_codec.py::test_md5_against_hashlib. */
#include <math.h>
#include "hevc_common.h"

static uint32_t K[64];
static int k_init = 0;
static const uint8_t S[64] = {7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 7, 12, 17, 22, 5, 9,  14, 20, 5, 9,
                              14, 20, 5, 9,  14, 20, 5, 9,  14, 20, 4, 11, 16, 23, 4, 11, 16, 23, 4, 11, 16, 23,
                              4, 11, 16, 23, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21, 6, 10, 15, 21};

typedef struct { uint32_t a, b, c, d; uint64_t len; uint8_t buf[64]; int nbuf; } md5_ctx;

static void md5_block(md5_ctx* c, const uint8_t* p) {
  uint32_t M[16];
  for (int i = 0; i < 16; i++) M[i] = p[4 * i] | (p[4 * i + 1] << 8) | (p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
  uint32_t A = c->a, B = c->b, C = c->c, D = c->d;
  for (int i = 0; i < 64; i++) {
    uint32_t F; int g;
    if (i < 16) { F = (B & C) | (~B & D); g = i; }
    else if (i < 32) { F = (D & B) | (~D & C); g = (5 * i + 1) & 15; }
    else if (i < 48) { F = B ^ C ^ D; g = (3 * i + 5) & 15; }
    else { F = C ^ (B | ~D); g = (7 * i) & 15; }
    F = F + A + K[i] + M[g];
    A = D; D = C; C = B;
    B = B + ((F << S[i]) | (F >> (32 - S[i])));
  }
  c->a += A; c->b += B; c->c += C; c->d += D;
}
static void md5_begin(md5_ctx* c) {
  if (!k_init) { for (int i = 0; i < 64; i++) K[i] = (uint32_t)floor(fabs(sin((double)(i + 1))) * 4294967296.0); k_init = 1; }
  c->a = 0x67452301u; c->b = 0xefcdab89u; c->c = 0x98badcfeu; c->d = 0x10325476u; c->len = 0; c->nbuf = 0;
}
static void md5_update(md5_ctx* c, const uint8_t* p, size_t n) {
  c->len += n;
  while (n) {
    if (c->nbuf == 0 && n >= 64) { md5_block(c, p); p += 64; n -= 64; continue; }
    size_t k = 64 - c->nbuf; if (k > n) k = n;
    memcpy(c->buf + c->nbuf, p, k); c->nbuf += (int)k; p += k; n -= k;
    if (c->nbuf == 64) { md5_block(c, c->buf); c->nbuf = 0; }
  }
}
static void md5_end(md5_ctx* c, uint8_t out[16]) {
  uint64_t bits = c->len * 8; uint8_t pad = 0x80; md5_update(c, &pad, 1);
  pad = 0; while (c->nbuf != 56) md5_update(c, &pad, 1);
  uint8_t l[8]; for (int i = 0; i < 8; i++) l[i] = (uint8_t)(bits >> (8 * i));
  md5_update(c, l, 8);
  uint32_t v[4] = {c->a, c->b, c->c, c->d};
  for (int i = 0; i < 16; i++) out[i] = (uint8_t)(v[i >> 2] >> (8 * (i & 3)));
}

void oracle_md5(const uint8_t* data, size_t n, uint8_t out[16]) { md5_ctx c; md5_begin(&c); md5_update(&c, data, n); md5_end(&c, out); }

/* Picture hash per H.265 D.3.19: samples as 1 byte (bit depth 8) or 2 bytes little-endian, raster order. */
void oracle_md5_plane(const uint16_t* p, int w, int h, int bit_depth, uint8_t out[16]) {
  md5_ctx c; md5_begin(&c);
  uint8_t row[2 * HEVC_MAX_W];
  for (int y = 0; y < h; y++) {
    const uint16_t* s = p + (size_t)y * w;
    if (bit_depth <= 8) { for (int x = 0; x < w; x++) row[x] = (uint8_t)s[x]; md5_update(&c, row, (size_t)w); }
    else { for (int x = 0; x < w; x++) { row[2 * x] = (uint8_t)s[x]; row[2 * x + 1] = (uint8_t)(s[x] >> 8); } md5_update(&c, row, (size_t)w * 2); }
  }
  md5_end(&c, out);
}
