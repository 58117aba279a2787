// ABI version + thread-local error string for libcsam_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

static thread_local char g_err[512] = "";

extern "C" {
void csam_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
int csam_abi_version(void) { return 1; }

// Host helper: COCO compressed-RLE string of run lengths (the arithmetic of pycocotools' rleToString, which
// the reference reaches through segment_anything_cs/utils/amg.py:294-300): 5 data bits + continuation bit
// per character, offset 48, runs after the third delta-coded against counts[i-2].  HOST pointers.
// Returns the string length, or -1 if `cap` is too small (13 characters per run always suffice).
long csam_coco_rle_string(const long long* counts, long n, char* out, long cap) {
  long p = 0;
  for (long i = 0; i < n; ++i) {
    long long x = counts[i];
    if (i > 2) x -= counts[i - 2];
    bool more = true;
    while (more) {
      char c = (char)(x & 0x1f);
      x >>= 5;
      more = (c & 0x10) ? (x != -1) : (x != 0);
      if (more) c |= 0x20;
      if (p >= cap) return -1;
      out[p++] = (char)(c + 48);
    }
  }
  return p;
}
// n strings in one call: counts = the run lengths of all masks back to back, offs[n + 1] their boundaries; the strings
// go to `out` back to back with their boundaries in out_offs[n + 1].  Returns the total length or -1 (cap too small).
long csam_coco_rle_strings(const long long* counts, const long long* offs, long n, char* out, long cap,
                           long long* out_offs) {
  long p = 0;
  for (long m = 0; m < n; ++m) {
    out_offs[m] = p;
    const long r = csam_coco_rle_string(counts + offs[m], (long)(offs[m + 1] - offs[m]), out + p, cap - p);
    if (r < 0) return -1;
    p += r;
  }
  out_offs[n] = p;
  return p;
}
const char* csam_last_error(void) { return g_err; }
}
