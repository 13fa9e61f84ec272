// Host-side text I/O of the DOTA devkit mirrors (no device code): the Task1_<class>.txt reader and writer of the tile ->
// full-image merge, restating what DOTA_devkit/ResultMerge_multi_process.py:186-233 does per line with str.split, two
// regular expressions, float() and round()/str() -- in one pass over the file instead of ~20 Python calls per line
// (the merge of a 25k-line class file spent 130 of its 134 ms there).  strtod is correctly rounded like Python's float(),
// printf's %.1f / %.2f are the correctly rounded decimals Python's round() picks; both have an exact fast path for the numbers
// such files hold (fast_decimal, fast_fixed), with strtod / snprintf behind it.  Anything that is not the plain layout
// (10 single-space separated fields, tile name `<orig>__<rate>__<x>___<y>`, plain decimal numbers) makes the reader return
// OBB_ERR_BAD_ARG and the Python layer takes its line-by-line path, which behaves like the reference on such input.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <string_view>
#include <unordered_map>
#include "obb_hip.h"

namespace {

inline bool is_space(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }
inline bool is_digit(char c) { return c >= '0' && c <= '9'; }

// Exact fast path of the decimal -> double conversion (Clinger): a literal without exponent whose digits, read as one
// integer, stay below 2^53 and whose fraction has at most 22 digits is integer / 10^k with BOTH operands exactly
// representable, so the one IEEE division is the correctly rounded result -- the value strtod and Python's float() return.
// (Every number of a Task1 file is of this kind; strtod took 60 % of the reader's time.)
const double kPow10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19,
                           1e20, 1e21, 1e22};
bool fast_decimal(const char* b, const char* e, double* out) {
  const char* p = b;
  bool neg = false;
  if (*p == '+' || *p == '-') { neg = *p == '-'; p++; }
  unsigned long long m = 0;
  int sig = 0, frac = 0;
  bool dot = false;
  for (; p < e; p++) {
    const char c = *p;
    if (c == '.') { if (dot) return false; dot = true; continue; }
    if (!is_digit(c)) return false;                       // an exponent: the general path
    if (m != 0 || c != '0') { if (++sig > 15) return false; }
    m = m * 10 + (unsigned long long)(c - '0');           // < 10^15 < 2^53
    if (dot && ++frac > 22) return false;
  }
  const double v = (double)m / kPow10[frac];
  *out = neg ? -v : v;
  return true;
}

// Exact fast path of printf("%.<D>f") for D = 1, 2 and finite |v| < 1e15: v = m * 2^e exactly, so round-half-even of
// v * 10^D is integer arithmetic on m * 10^D (below 2^60) -- the digits glibc's correctly rounded printf and Python's
// round() produce.  Returns the number of characters, 0 when the value is outside the fast path.
int fast_fixed(double v, int D, char* out) {
  unsigned long long bits;
  memcpy(&bits, &v, 8);
  const bool neg = (bits >> 63) != 0;
  const int ex = (int)((bits >> 52) & 0x7ff);
  unsigned long long m = bits & ((1ull << 52) - 1);
  if (ex == 0x7ff) return 0;
  int e2;
  if (ex == 0) e2 = -1074; else { m |= 1ull << 52; e2 = ex - 1075; }
  if (e2 > -3) return 0;                                 // |v| >= 2^50 ~ 1.1e15: the general path
  const unsigned long long scale = D == 1 ? 10ull : 100ull;
  const unsigned long long P = m * scale;                // < 2^53 * 100 < 2^60
  const int s = -e2;
  unsigned long long q = 0;
  if (s < 61) {
    q = P >> s;
    const unsigned long long rem = P & ((1ull << s) - 1), half = 1ull << (s - 1);
    if (rem > half || (rem == half && (q & 1ull))) q++;
  }                                                      // (s >= 61: P < 2^60 <= half, the value rounds to zero)
  unsigned long long ip = q / scale, fp = q % scale;
  char tmp[24];
  int n = 0;
  do { tmp[n++] = (char)('0' + ip % 10); ip /= 10; } while (ip);
  int w = 0;
  if (neg) out[w++] = '-';
  while (n) out[w++] = tmp[--n];
  out[w++] = '.';
  if (D == 2) out[w++] = (char)('0' + fp / 10);
  out[w++] = (char)('0' + fp % 10);
  return w;
}

// plain decimal literal (what float() and strtod agree on without exception): [+-] digits [. digits] [e[+-]digits]
bool parse_double(const char* b, const char* e, double* out) {
  if (b >= e || e - b > 60) return false;
  const char* p = b;
  if (*p == '+' || *p == '-') p++;
  int nd = 0;
  while (p < e && is_digit(*p)) { p++; nd++; }
  if (p < e && *p == '.') { p++; while (p < e && is_digit(*p)) { p++; nd++; } }
  if (nd == 0) return false;
  if (p < e && (*p == 'e' || *p == 'E')) {
    p++;
    if (p < e && (*p == '+' || *p == '-')) p++;
    int ne = 0;
    while (p < e && is_digit(*p)) { p++; ne++; }
    if (ne == 0) return false;
  }
  if (p != e) return false;
  if (fast_decimal(b, e, out)) return true;
  char buf[64];
  memcpy(buf, b, (size_t)(e - b));
  buf[e - b] = 0;
  char* end = nullptr;
  *out = strtod(buf, &end);
  return end == buf + (e - b);
}

// Python reads these files in text mode: "\r\n" and a lone "\r" end a line too, and str.strip() also removes the separators
// 0x1c-0x1f and a few non-ASCII spaces.  The native readers split at "\n" and strip ASCII blanks only, so they decline
// a buffer with a lone "\r", and a line whose ends could be such a character.
bool plain_newlines(const char* text, int64_t len) {
  const char* p = text;
  const char* end = text + len;
  while ((p = (const char*)memchr(p, '\r', (size_t)(end - p))) != nullptr) {
    if (p + 1 >= end) return true;                       // a "\r" that ends the buffer only ends the last line
    if (p[1] != '\n') return false;
    p++;
  }
  return true;
}
inline bool odd_edge(const char* b, const char* e) {     // [b, e) non-empty, already stripped of ASCII blanks
  const unsigned char f = (unsigned char)b[0], l = (unsigned char)e[-1];
  if ((f >= 0x1c && f <= 0x1f) || (l >= 0x1c && l <= 0x1f)) return true;
  // UTF-8 of U+0085, U+00A0, U+1680, U+2000-200A, U+2028, U+2029, U+202F, U+205F, U+3000 starts with C2 / E1 / E2 / E3; at the
  // end of a line their last byte is 0x80-0xBF after such a lead byte: names in other scripts are common, so look closer
  auto ws_at = [](const unsigned char* q, const unsigned char* lim) -> int {   // length of a Unicode space starting at q, or 0
    if (q + 1 < lim && q[0] == 0xC2 && (q[1] == 0x85 || q[1] == 0xA0)) return 2;
    if (q + 2 < lim && q[0] == 0xE1 && q[1] == 0x9A && q[2] == 0x80) return 3;
    if (q + 2 < lim && q[0] == 0xE2 && q[1] == 0x80 && ((q[2] >= 0x80 && q[2] <= 0x8A) || q[2] == 0xA8 || q[2] == 0xA9 || q[2] == 0xAF)) return 3;
    if (q + 2 < lim && q[0] == 0xE2 && q[1] == 0x81 && q[2] == 0x9F) return 3;
    if (q + 2 < lim && q[0] == 0xE3 && q[1] == 0x80 && q[2] == 0x80) return 3;
    return 0;
  };
  const unsigned char* ub = (const unsigned char*)b;
  const unsigned char* ue = (const unsigned char*)e;
  if (ws_at(ub, ue)) return true;
  if (e - b >= 2 && ws_at(ue - 2, ue) == 2) return true;
  if (e - b >= 3 && ws_at(ue - 3, ue) == 3) return true;
  return false;
}

bool parse_int(const char* b, const char* e, long long* out) {
  if (b >= e || e - b > 18) return false;
  long long v = 0;
  for (const char* p = b; p < e; p++) { if (!is_digit(*p)) return false; v = v * 10 + (*p - '0'); }
  *out = v;
  return true;
}

}  // namespace

extern "C" {

int64_t obb_task1_parse_tiles(const char* text, int64_t len, int64_t max_lines, double* dets9, int32_t* name_off, int32_t* name_len,
                              int32_t* group, int32_t* group_first, int64_t* n_groups) {
  if (!text || len < 0 || max_lines < 0 || !dets9 || !name_off || !name_len || !group || !group_first || !n_groups) return OBB_ERR_BAD_ARG;
  if (!plain_newlines(text, len)) return OBB_ERR_BAD_ARG;
  std::unordered_map<std::string_view, int32_t> ids;          // keys point into `text`: no copies
  std::string_view prev_name;
  int32_t prev_id = -1;
  int64_t n = 0;
  const char* p = text;
  const char* end = text + len;
  while (p < end) {
    const char* le = (const char*)memchr(p, '\n', (size_t)(end - p));
    const char* next = le ? le + 1 : end;
    if (!le) le = end;
    const char* b = p;
    const char* e = le;
    while (b < e && is_space(*b)) b++;                    // str.strip()
    while (e > b && is_space(e[-1])) e--;
    p = next;
    if (b == e) return OBB_ERR_BAD_ARG;                   // an empty line: the reference raises on it
    if (n >= max_lines || odd_edge(b, e)) return OBB_ERR_BAD_ARG;
    // ---- 10 fields separated by single spaces
    const char* fb[10];
    const char* fe[10];
    int nf = 0;
    const char* q = b;
    while (true) {
      const char* sp = (const char*)memchr(q, ' ', (size_t)(e - q));
      if (nf == 10) return OBB_ERR_BAD_ARG;
      fb[nf] = q; fe[nf] = sp ? sp : e; nf++;
      if (!sp) break;
      q = sp + 1;
    }
    if (nf != 10) return OBB_ERR_BAD_ARG;
    for (int k = 0; k < 10; k++) if (fb[k] == fe[k]) return OBB_ERR_BAD_ARG;
    const char* s = fb[0];
    const char* se = fe[0];
    // ---- oriname = subname.split('__')[0]
    const char* us = nullptr;
    for (const char* t = s; t + 1 < se; t++) if (t[0] == '_' && t[1] == '_') { us = t; break; }
    if (!us) return OBB_ERR_BAD_ARG;
    // ---- re.findall(r'__\d+___\d+', subname)[0]: leftmost "__" digits "___" digits
    long long x = 0, y = 0;
    bool have_xy = false;
    for (const char* t = s; t + 1 < se && !have_xy; t++) {
      if (t[0] != '_' || t[1] != '_') continue;
      const char* d0 = t + 2;
      const char* d1 = d0;
      while (d1 < se && is_digit(*d1)) d1++;
      if (d1 == d0 || d1 + 3 > se || d1[0] != '_' || d1[1] != '_' || d1[2] != '_') continue;
      const char* g0 = d1 + 3;
      const char* g1 = g0;
      while (g1 < se && is_digit(*g1)) g1++;
      if (g1 == g0) continue;
      if (!parse_int(d0, d1, &x) || !parse_int(g0, g1, &y)) return OBB_ERR_BAD_ARG;
      have_xy = true;
    }
    if (!have_xy) return OBB_ERR_BAD_ARG;
    // ---- re.findall(r'__([\d+\.]+)__\d+___', subname)[0]
    double rate = 0.0;
    bool have_rate = false;
    for (const char* t = s; t + 1 < se && !have_rate; t++) {
      if (t[0] != '_' || t[1] != '_') continue;
      const char* r0 = t + 2;
      const char* r1 = r0;
      while (r1 < se && (is_digit(*r1) || *r1 == '+' || *r1 == '.')) r1++;
      if (r1 == r0 || r1 + 2 > se || r1[0] != '_' || r1[1] != '_') continue;
      const char* d0 = r1 + 2;
      const char* d1 = d0;
      while (d1 < se && is_digit(*d1)) d1++;
      if (d1 == d0 || d1 + 3 > se || d1[0] != '_' || d1[1] != '_' || d1[2] != '_') continue;
      if (!parse_double(r0, r1, &rate)) return OBB_ERR_BAD_ARG;   // float(rate) would raise
      have_rate = true;
    }
    if (!have_rate) return OBB_ERR_BAD_ARG;
    // ---- numbers: float(poly + x) / float(rate), confidence last (:205-209)
    double* d = dets9 + n * 9;
    if (!parse_double(fb[1], fe[1], &d[8])) return OBB_ERR_BAD_ARG;
    for (int k = 0; k < 8; k++) {
      double v;
      if (!parse_double(fb[2 + k], fe[2 + k], &v)) return OBB_ERR_BAD_ARG;
      d[k] = (v + (double)((k & 1) ? y : x)) / rate;
    }
    name_off[n] = (int32_t)(s - text);
    name_len[n] = (int32_t)(us - s);
    const std::string_view nm(s, (size_t)(us - s));
    if (prev_id < 0 || nm != prev_name) {                  // (the tiles of one source image usually follow each other)
      auto ins = ids.emplace(nm, (int32_t)ids.size());
      if (ins.second) group_first[ins.first->second] = (int32_t)n;
      prev_name = nm; prev_id = ins.first->second;
    }
    group[n] = prev_id;
    n++;
  }
  *n_groups = (int64_t)ids.size();
  return n;
}

// `image score x1 y1 .. x4 y4` per line (dota_evaluation_task1.py:152-160: strip, split(' '), float()): the plain layout only
int64_t obb_task1_parse_dets(const char* text, int64_t len, int64_t max_lines, double* conf, double* bb8, int32_t* name_off,
                             int32_t* name_len) {
  if (!text || len < 0 || max_lines < 0 || !conf || !bb8 || !name_off || !name_len) return OBB_ERR_BAD_ARG;
  if (!plain_newlines(text, len)) return OBB_ERR_BAD_ARG;
  int64_t n = 0;
  const char* p = text;
  const char* end = text + len;
  while (p < end) {
    const char* le = (const char*)memchr(p, '\n', (size_t)(end - p));
    const char* next = le ? le + 1 : end;
    if (!le) le = end;
    const char* b = p;
    const char* e = le;
    while (b < e && is_space(*b)) b++;
    while (e > b && is_space(e[-1])) e--;
    p = next;
    if (b == e || n >= max_lines || odd_edge(b, e)) return OBB_ERR_BAD_ARG;   // (an empty line makes the reference raise)
    const char* q = b;
    for (int k = 0; k < 10; k++) {
      const char* sp = (const char*)memchr(q, ' ', (size_t)(e - q));
      const char* fe = sp ? sp : e;
      if (fe == q || (k < 9) != (sp != nullptr)) return OBB_ERR_BAD_ARG;      // an empty field, too few or too many fields
      if (k == 0) { name_off[n] = (int32_t)(q - text); name_len[n] = (int32_t)(fe - q); }
      else if (!parse_double(q, fe, k == 1 ? &conf[n] : &bb8[n * 8 + (k - 2)])) return OBB_ERR_BAD_ARG;
      q = fe + 1;
    }
    n++;
  }
  return n;
}

// `x1 y1 x2 y2 x3 y3 x4 y4 name [difficult]` per line (dota_evaluation_task1.py:21-53): lines with fewer than 9 fields are
// skipped like there; difficult = 0 without the tenth field.  More than 10 fields, an empty field, a difficult flag that is
// not plain digits: declined.
int64_t obb_task1_parse_gt(const char* text, int64_t len, int64_t max_lines, double* bbox8, int32_t* name_off, int32_t* name_len,
                           int32_t* difficult) {
  if (!text || len < 0 || max_lines < 0 || !bbox8 || !name_off || !name_len || !difficult) return OBB_ERR_BAD_ARG;
  if (!plain_newlines(text, len)) return OBB_ERR_BAD_ARG;
  int64_t n = 0;
  const char* p = text;
  const char* end = text + len;
  while (p < end) {
    const char* le = (const char*)memchr(p, '\n', (size_t)(end - p));
    const char* next = le ? le + 1 : end;
    if (!le) le = end;
    const char* b = p;
    const char* e = le;
    while (b < e && is_space(*b)) b++;
    while (e > b && is_space(e[-1])) e--;
    p = next;
    if (b == e) continue;                                 // '' -> [''] : fewer than 9 fields, skipped
    if (odd_edge(b, e)) return OBB_ERR_BAD_ARG;
    const char* fb[10];
    const char* fe[10];
    int nf = 0;
    const char* q = b;
    while (true) {
      const char* sp = (const char*)memchr(q, ' ', (size_t)(e - q));
      if (nf == 10) return OBB_ERR_BAD_ARG;               // an eleventh field: the reference's record has no 'difficult' key
      fb[nf] = q; fe[nf] = sp ? sp : e; nf++;
      if (!sp) break;
      q = sp + 1;
    }
    if (nf < 9) continue;                                 // (:26-27; empty fields count as fields there too)
    for (int k = 0; k < nf; k++) if (fb[k] == fe[k]) return OBB_ERR_BAD_ARG;
    if (n >= max_lines) return OBB_ERR_BAD_ARG;
    for (int k = 0; k < 8; k++) if (!parse_double(fb[k], fe[k], &bbox8[n * 8 + k])) return OBB_ERR_BAD_ARG;
    name_off[n] = (int32_t)(fb[8] - text);
    name_len[n] = (int32_t)(fe[8] - fb[8]);
    long long dv = 0;
    if (nf == 10 && (!parse_int(fb[9], fe[9], &dv) || dv > 0x7fffffffLL)) return OBB_ERR_BAD_ARG;   // int('1_0'), int('+1'), int(' 1'): the general path
    difficult[n] = (int32_t)dv;
    n++;
  }
  return n;
}

// "<name> <conf> <c1> .. <c8>\n" per row: str(round(conf, 2)) and str(round(c, 1)) of :218-233 for |conf| < 1e13, |c| < 1e14
// (while the fixed-point string has at most 15 significant digits it IS the shortest repr of the rounded double)
int64_t obb_task1_format_rows(const char* text, const int32_t* name_off, const int32_t* name_len, const double* dets9, const int64_t* rows,
                              int64_t n_rows, char* out, int64_t out_cap) {
  if (!text || !name_off || !name_len || !dets9 || (n_rows > 0 && !rows) || !out || out_cap < 0) return OBB_ERR_BAD_ARG;
  int64_t w = 0;
  char num[400];
  for (int64_t i = 0; i < n_rows; i++) {
    const int64_t r = rows[i];
    const double* d = dets9 + r * 9;
    const int nl = name_len[r];
    if (w + nl + 2 > out_cap) return OBB_ERR_WORKSPACE;
    memcpy(out + w, text + name_off[r], (size_t)nl);
    w += nl;
    out[w++] = ' ';
    int k = fast_fixed(d[8], 2, num);
    if (k == 0) k = snprintf(num, sizeof num, "%.2f", d[8]);
    if (k <= 0 || k >= (int)sizeof num) return OBB_ERR_BAD_ARG;
    if (num[k - 1] == '0') k--;                          // '0.50' -> '0.5', '1.00' -> '1.0' (one decimal always stays)
    if (w + k + 1 > out_cap) return OBB_ERR_WORKSPACE;
    memcpy(out + w, num, (size_t)k);
    w += k;
    for (int c = 0; c < 8; c++) {
      k = fast_fixed(d[c], 1, num);
      if (k == 0) k = snprintf(num, sizeof num, "%.1f", d[c]);
      if (k <= 0 || k >= (int)sizeof num) return OBB_ERR_BAD_ARG;
      if (w + k + 2 > out_cap) return OBB_ERR_WORKSPACE;
      out[w++] = ' ';
      memcpy(out + w, num, (size_t)k);
      w += k;
    }
    out[w++] = '\n';
  }
  return w;
}

}  // extern "C"
