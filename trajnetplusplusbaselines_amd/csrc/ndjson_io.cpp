// Host-side reader / writer of the evaluator feed, plain C++ (no HIP): the TrajNet++ .ndjson test file
// in, the prediction file out.  The reference reads a test file through trajnetplusplustools.Reader (one json.loads per line) and
// writes predictions through trajnetplusplustools.writers.trajnet (one dict + json.dumps per row), evaluator/
// trajnet_evaluator.py:29-65, evaluator/write_utils.py:42-81; at 40 agents per scene a test file holds ~840 rows per scene and
// the prediction file ~440, so with the forward pass at 0.12 ms per scene the Python JSON work (1.3 ms + 1.0 ms per scene) was
// 95 % of data.predict_dataset (profiles/round5_g_predict_dataset_throughput.txt).  These two functions do the same work on
// columnar arrays: ~0.1 ms per scene together.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/trajnet_hip.h"

namespace {

struct Cur { const char *p, *end; };

inline void ws(Cur &c) { while (c.p < c.end && (*c.p == ' ' || *c.p == '\t' || *c.p == '\r' || *c.p == '\n')) ++c.p; }

// a JSON string starting at the opening quote; the key text is returned only when it has no escapes (all keys we look for)
inline bool str(Cur &c, const char **s, size_t *n, bool *plain) {
    if (c.p >= c.end || *c.p != '"') return false;
    const char *b = ++c.p;
    *plain = true;
    while (c.p < c.end && *c.p != '"') {
        if (*c.p == '\\') { *plain = false; ++c.p; if (c.p >= c.end) return false; }
        ++c.p;
    }
    if (c.p >= c.end) return false;
    *s = b; *n = (size_t)(c.p - b);
    ++c.p;
    return true;
}

bool skip_value(Cur &c, int depth);

inline bool skip_number(Cur &c) {
    const char *b = c.p;
    while (c.p < c.end && (*c.p == '-' || *c.p == '+' || *c.p == '.' || *c.p == 'e' || *c.p == 'E' || (*c.p >= '0' && *c.p <= '9'))) ++c.p;
    return c.p > b;
}

bool skip_value(Cur &c, int depth) {
    if (depth > 32) return false;
    ws(c);
    if (c.p >= c.end) return false;
    const char ch = *c.p;
    if (ch == '"') { const char *s; size_t n; bool pl; return str(c, &s, &n, &pl); }
    if (ch == '{' || ch == '[') {
        const char close = ch == '{' ? '}' : ']';
        ++c.p; ws(c);
        if (c.p < c.end && *c.p == close) { ++c.p; return true; }
        for (;;) {
            if (ch == '{') {
                const char *s; size_t n; bool pl;
                ws(c);
                if (!str(c, &s, &n, &pl)) return false;
                ws(c);
                if (c.p >= c.end || *c.p != ':') return false;
                ++c.p;
            }
            if (!skip_value(c, depth + 1)) return false;
            ws(c);
            if (c.p >= c.end) return false;
            if (*c.p == ',') { ++c.p; continue; }
            if (*c.p == close) { ++c.p; return true; }
            return false;
        }
    }
    if (c.end - c.p >= 4 && !strncmp(c.p, "null", 4)) { c.p += 4; return true; }
    if (c.end - c.p >= 4 && !strncmp(c.p, "true", 4)) { c.p += 4; return true; }
    if (c.end - c.p >= 5 && !strncmp(c.p, "false", 5)) { c.p += 5; return true; }
    if (c.end - c.p >= 3 && !strncmp(c.p, "NaN", 3)) { c.p += 3; return true; }
    if (c.end - c.p >= 8 && !strncmp(c.p, "Infinity", 8)) { c.p += 8; return true; }
    return skip_number(c);
}

// integer token (no fraction / exponent); false = not an integer -> the caller gives the file back to the Python reader
inline bool int_value(Cur &c, int64_t *v) {
    ws(c);
    const char *b = c.p;
    if (c.p < c.end && *c.p == '-') ++c.p;
    const char *d = c.p;
    while (c.p < c.end && *c.p >= '0' && *c.p <= '9') ++c.p;
    if (c.p == d || c.p - d > 18) return false;
    if (c.p < c.end && (*c.p == '.' || *c.p == 'e' || *c.p == 'E')) return false;
    char tmp[24];
    memcpy(tmp, b, (size_t)(c.p - b)); tmp[c.p - b] = 0;
    *v = strtoll(tmp, nullptr, 10);
    return true;
}

inline bool float_value(Cur &c, double *v) {
    ws(c);
    const char *b = c.p;
    // plain decimals with at most 15 significant digits (every coordinate of a data file): mantissa / 10^k with both operands
    // exact doubles is ONE correctly rounded division = what strtod / Python's float() give
    {
        static const double p10[] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
        const char *q = b;
        bool neg = false;
        if (q < c.end && *q == '-') { neg = true; ++q; }
        uint64_t mant = 0;
        int digits = 0, frac = 0;
        const char *d0 = q;
        while (q < c.end && *q >= '0' && *q <= '9') { mant = mant * 10 + (uint64_t)(*q - '0'); ++digits; ++q; }
        if (q > d0) {
            if (q < c.end && *q == '.') {
                ++q;
                const char *f0 = q;
                while (q < c.end && *q >= '0' && *q <= '9') { mant = mant * 10 + (uint64_t)(*q - '0'); ++digits; ++frac; ++q; }
                if (q == f0) digits = 99;
            }
            if (digits <= 15 && !(q < c.end && (*q == 'e' || *q == 'E'))) {
                const double m = (double)mant / p10[frac];
                *v = neg ? -m : m;
                c.p = q;
                return true;
            }
        }
    }
    if (!skip_number(c)) return false;
    char tmp[64];
    const size_t n = (size_t)(c.p - b);
    if (n >= sizeof(tmp)) return false;
    memcpy(tmp, b, n); tmp[n] = 0;
    char *e = nullptr;
    *v = strtod(tmp, &e);               // correctly rounded, what Python's float() of the same token gives
    return e == tmp + n;
}

inline bool key_is(const char *s, size_t n, const char *k) { return strlen(k) == n && !memcmp(s, k, n); }

inline char *put_uint(char *o, uint64_t v) {
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *o++ = tmp[--n];
    return o;
}

inline char *put_int(char *o, int64_t v) {
    if (v < 0) { *o++ = '-'; return put_uint(o, (uint64_t)(-(v + 1)) + 1u); }
    return put_uint(o, (uint64_t)v);
}

// repr(round(v, 2)) of Python for a finite double, json.dumps' spelling otherwise
inline char *put_coord(char *o, double v) {
    if (v != v) { memcpy(o, "NaN", 3); return o + 3; }
    if (isinf(v)) { if (v < 0) { memcpy(o, "-Infinity", 9); return o + 9; } memcpy(o, "Infinity", 8); return o + 8; }
    const double a = fabs(v);
    if (a >= 1e15) {
        // (never a coordinate.)  Python's own recipe: round = print with two decimals and parse back, repr = the shortest
        // digit string that parses back to the same double, with ".0" behind an integer mantissa
        char tmp[400];
        snprintf(tmp, sizeof(tmp), "%.2f", v);
        const double r = strtod(tmp, nullptr);
        int prec = 1, ex = 0;
        for (; prec <= 17; ++prec) {                                  // shortest round-trip digits, read in exponent form
            sprintf(tmp, "%.*e", prec - 1, r);
            if (strtod(tmp, nullptr) == r) break;
        }
        ex = atoi(strchr(tmp, 'e') + 1);
        int n;
        if (ex >= 16) {                                               // repr switches to exponent notation at 1e16
            n = sprintf(o, "%.*e", prec - 1, r);
            char *e = strchr(o, 'e');                                 // C prints e+16, Python too; mantissa zeros are already minimal
            (void)e;
        } else {
            const int dec = prec - 1 - ex > 0 ? prec - 1 - ex : 0;
            n = sprintf(o, "%.*f", dec, r);
            if (dec == 0) { o[n++] = '.'; o[n++] = '0'; }
        }
        return o + n;
    }
    // round(v, 2) is the correctly rounded two-decimal value of the BINARY number (half to even on exact ties), its repr that
    // decimal string with a trailing zero dropped.  a * 100 is within one rounding of the exact product, so away from a tie
    // (|fraction - 0.5| > 1e-6 >> the rounding) the nearest integer of the product is the exact answer ...
    if (a < 1e9) {
        const double t = a * 100.0, fl = floor(t), fr = t - fl;
        if (fabs(fr - 0.5) > 1e-6) {
            const uint64_t k = (uint64_t)fl + (fr > 0.5 ? 1u : 0u);
            if (signbit(v)) *o++ = '-';
            o = put_uint(o, k / 100);
            *o++ = '.';
            const unsigned f2 = (unsigned)(k % 100);
            *o++ = (char)('0' + f2 / 10);
            if (f2 % 10) *o++ = (char)('0' + f2 % 10);
            return o;
        }
    }
    // ... and at (or next to) a tie the C library decides on the exact value
    int n = sprintf(o, "%.2f", v);
    if (o[n - 1] == '0') --n;
    return o + n;
}

}  // namespace

// Parse a TrajNet++ .ndjson buffer into columns.  Track records that carry a non-null "prediction_number" are skipped (the
// reader of the test files ignores stored predictions).  Returns 0, or -(line number) of the first line it does not understand
// (non-integer frame / pedestrian / scene fields, malformed JSON): the caller then falls back to the general Python reader.
extern "C" TNP_API int64_t tnp_ndjson_parse(const char *buf, size_t n, int64_t cap, int64_t *t_frame, int64_t *t_ped, double *t_x,
                                            double *t_y, int64_t *n_tracks, int64_t *s_id, int64_t *s_ped, int64_t *s_start,
                                            int64_t *s_end, int64_t *n_scenes) {
    int64_t nt = 0, nsc = 0, line = 0;
    const char *p = buf, *end = buf + n;
    while (p < end) {
        const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
        const char *le = nl ? nl : end;
        ++line;
        Cur c = {p, le};
        p = nl ? nl + 1 : end;
        ws(c);
        if (c.p >= c.end) continue;                       // blank line
        if (*c.p != '{') return -line;
        ++c.p; ws(c);
        bool done_line = false;
        while (!done_line) {                              // top-level keys of the record
            const char *ks; size_t kn; bool pl;
            if (!str(c, &ks, &kn, &pl)) return -line;
            ws(c);
            if (c.p >= c.end || *c.p != ':') return -line;
            ++c.p; ws(c);
            const bool is_track = pl && key_is(ks, kn, "track"), is_scene = pl && key_is(ks, kn, "scene");
            if ((is_track || is_scene) && c.p < c.end && *c.p == '{') {
                ++c.p; ws(c);
                int64_t f = 0, pd = 0, sid = 0, ss = 0, se = 0;
                double x = 0.0, y = 0.0;
                unsigned have = 0;
                bool stored_prediction = false;
                if (c.p < c.end && *c.p == '}') { ++c.p; }
                else for (;;) {
                    ws(c);
                    if (!str(c, &ks, &kn, &pl)) return -line;
                    ws(c);
                    if (c.p >= c.end || *c.p != ':') return -line;
                    ++c.p; ws(c);
                    bool ok = true;
                    if (is_track && pl && key_is(ks, kn, "f")) { ok = int_value(c, &f); have |= 1; }
                    else if (pl && key_is(ks, kn, "p")) { ok = int_value(c, &pd); have |= 2; }
                    else if (is_track && pl && key_is(ks, kn, "x")) { ok = float_value(c, &x); have |= 4; }
                    else if (is_track && pl && key_is(ks, kn, "y")) { ok = float_value(c, &y); have |= 8; }
                    else if (is_track && pl && key_is(ks, kn, "prediction_number")) {
                        if (c.end - c.p >= 4 && !strncmp(c.p, "null", 4)) c.p += 4;
                        else { stored_prediction = true; ok = skip_value(c, 0); }
                    }
                    else if (is_scene && pl && key_is(ks, kn, "id")) { ok = int_value(c, &sid); have |= 16; }
                    else if (is_scene && pl && key_is(ks, kn, "s")) { ok = int_value(c, &ss); have |= 32; }
                    else if (is_scene && pl && key_is(ks, kn, "e")) { ok = int_value(c, &se); have |= 64; }
                    else ok = skip_value(c, 0);
                    if (!ok) return -line;
                    ws(c);
                    if (c.p >= c.end) return -line;
                    if (*c.p == ',') { ++c.p; continue; }
                    if (*c.p == '}') { ++c.p; break; }
                    return -line;
                }
                if (is_track) {
                    if (!stored_prediction) {
                        if ((have & 15) != 15 || nt >= cap) return -line;
                        t_frame[nt] = f; t_ped[nt] = pd; t_x[nt] = x; t_y[nt] = y; ++nt;
                    }
                } else {
                    if ((have & (2 | 16 | 32 | 64)) != (2 | 16 | 32 | 64) || nsc >= cap) return -line;
                    s_id[nsc] = sid; s_ped[nsc] = pd; s_start[nsc] = ss; s_end[nsc] = se; ++nsc;
                }
            } else if (!skip_value(c, 0)) return -line;
            ws(c);
            if (c.p >= c.end) return -line;
            if (*c.p == ',') { ++c.p; ws(c); continue; }
            if (*c.p == '}') { ++c.p; done_line = true; }
            else return -line;
        }
        ws(c);
        if (c.p < c.end) return -line;                    // trailing text behind the record
    }
    *n_tracks = nt; *n_scenes = nsc;
    return 0;
}

// The prediction-file lines of a batch of scenes, as evaluator/write_utils.py:42-81 lays them out through
// trajnetplusplustools.writers.trajnet: per scene one scene record (fps 2.5, tag 0), then per mode the primary's pred_length
// rows followed by every neighbour's, coordinates as repr(round(float(x), 2)).
//   pred        [n_modes][pred_length][M][2] float64, M = split[n_scenes] tracks (scene s = columns split[s] .. split[s + 1])
//   ped         [M] pedestrian ids in column order (column split[s] = the primary)
//   scene_id / first_frame / frame_diff / scene_start / scene_end  [n_scenes]: first_frame = frame of the first predicted row
// Returns the number of bytes written, or -1 when `cap` is too small (the caller sizes it with tnp_format_predictions_bound).
extern "C" TNP_API int64_t tnp_format_predictions(const double *pred, int n_modes, int pred_length, int64_t M, int64_t n_scenes,
                                                  const int64_t *split, const int64_t *ped, const int64_t *scene_id,
                                                  const int64_t *first_frame, const int64_t *frame_diff, const int64_t *scene_start,
                                                  const int64_t *scene_end, char *out, size_t cap) {
    char *o = out;
    const size_t line_bound = 200;
    for (int64_t s = 0; s < n_scenes; ++s) {
        if ((size_t)(o - out) + line_bound > cap) return -1;
        const int64_t lo = split[s], hi = split[s + 1];
        memcpy(o, "{\"scene\": {\"id\": ", 17); o += 17; o = put_int(o, scene_id[s]);
        memcpy(o, ", \"p\": ", 7); o += 7; o = put_int(o, ped[lo]);
        memcpy(o, ", \"s\": ", 7); o += 7; o = put_int(o, scene_start[s]);
        memcpy(o, ", \"e\": ", 7); o += 7; o = put_int(o, scene_end[s]);
        memcpy(o, ", \"fps\": 2.5, \"tag\": 0}}\n", 25); o += 25;
        for (int m = 0; m < n_modes; ++m)
            for (int64_t col = lo; col < hi; ++col)
                for (int t = 0; t < pred_length; ++t) {
                    if ((size_t)(o - out) + line_bound > cap) return -1;
                    const double *v = pred + (((size_t)m * pred_length + t) * M + col) * 2;
                    memcpy(o, "{\"track\": {\"f\": ", 16); o += 16; o = put_int(o, first_frame[s] + t * frame_diff[s]);
                    memcpy(o, ", \"p\": ", 7); o += 7; o = put_int(o, ped[col]);
                    memcpy(o, ", \"x\": ", 7); o += 7; o = put_coord(o, v[0]);
                    memcpy(o, ", \"y\": ", 7); o += 7; o = put_coord(o, v[1]);
                    memcpy(o, ", \"prediction_number\": ", 23); o += 23; o = put_int(o, m);
                    memcpy(o, ", \"scene_id\": ", 14); o += 14; o = put_int(o, scene_id[s]);
                    memcpy(o, "}}\n", 3); o += 3;
                }
    }
    return (int64_t)(o - out);
}

extern "C" TNP_API size_t tnp_format_predictions_bound(int n_modes, int pred_length, int64_t M, int64_t n_scenes) {
    return (size_t)200 * ((size_t)n_scenes + (size_t)n_modes * (size_t)pred_length * (size_t)M + 2);
}
