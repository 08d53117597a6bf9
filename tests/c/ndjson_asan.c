/* The evaluator feed's host functions (csrc/ndjson_io.cpp) under AddressSanitizer / UBSan on the CPU build: well-formed and
 * hostile buffers (truncated records, runaway strings, deep nesting, huge numbers, no trailing newline, buffers that end exactly at
 * a token), every call on a heap copy WITHOUT a terminating NUL so that any read past the end is caught; the formatter with its
 * exact bound and with bounds that are too small.  tests/test_evaluator_feed.py compiles ndjson_io.cpp + this file with
 * g++ -fsanitize=address,undefined and runs it. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "trajnet_hip.h"

static int64_t parse(const char *text, size_t n, int64_t *nt, int64_t *ns) {
    char *buf = (char *)malloc(n ? n : 1);              /* exact size, no NUL */
    memcpy(buf, text, n);
    int64_t cap = 1;
    for (size_t i = 0; i < n; ++i) cap += text[i] == '\n';
    int64_t *tf = (int64_t *)malloc(cap * 8), *tp = (int64_t *)malloc(cap * 8), *s0 = (int64_t *)malloc(cap * 8),
            *s1 = (int64_t *)malloc(cap * 8), *s2 = (int64_t *)malloc(cap * 8), *s3 = (int64_t *)malloc(cap * 8);
    double *tx = (double *)malloc(cap * 8), *ty = (double *)malloc(cap * 8);
    *nt = *ns = -1;
    const int64_t rc = tnp_ndjson_parse(buf, n, cap, tf, tp, tx, ty, nt, s0, s1, s2, s3, ns);
    free(buf); free(tf); free(tp); free(s0); free(s1); free(s2); free(s3); free(tx); free(ty);
    return rc;
}

int main(void) {
    int64_t nt, ns;
    const char *good = "{\"scene\": {\"id\": 1, \"p\": 7, \"s\": 0, \"e\": 200, \"fps\": 2.5, \"tag\": [1, [2, 3]]}}\n"
                       "{\"track\": {\"f\": 10, \"p\": 7, \"x\": -1.25, \"y\": 3.0e1, \"extra\": {\"a\": \"}\\\"\"}}}\n"
                       "\n  \n{ \"track\" : { \"y\" : 2 , \"x\" : 1 , \"p\" : 8 , \"f\" : 20 , \"prediction_number\": null } }";
    if (parse(good, strlen(good), &nt, &ns) != 0 || nt != 2 || ns != 1) { printf("good buffer: nt %lld ns %lld\n", (long long)nt, (long long)ns); return 1; }
    /* every prefix of the good buffer: either parsed or refused, never a read past the end */
    for (size_t n = 0; n <= strlen(good); ++n) parse(good, n, &nt, &ns);
    const char *bad[] = {
        "{\"track\": {\"f\": 1, \"p\": 1, \"x\": 0.0, \"y\": 0.0}",             /* missing brace */
        "{\"track\": {\"f\": 1.5, \"p\": 1, \"x\": 0.0, \"y\": 0.0}}",          /* non-integer frame */
        "{\"track\": {\"f\": 1, \"p\": 1, \"x\": \"unterminated",                 /* runaway string */
        "{\"track\": {\"f\": 1, \"p\": 1, \"x\": 0.0}}",                          /* missing y */
        "{\"scene\": {\"id\": 1, \"p\": 7, \"s\": 0}}",                           /* missing e */
        "{\"track\": {\"f\": 99999999999999999999999, \"p\": 1, \"x\": 0, \"y\": 0}}",
        "{\"track\": {\"f\": 1, \"p\": 1, \"x\": 1e999999999999999999999999999999999999999999999999999999999999999999999, \"y\": 0}}",
        "{\"track\": {\"f\": 1, \"p\": 1, \"x\": 0, \"y\": 0}} trailing",
        "[1, 2, 3]", "{", "{\"", "{\"track\"", "{\"track\":", "{\"track\": {", "{\"track\": {\"f\"", "{\"track\": {\"f\": -", "nul",
        "{\"a\": [[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[[1]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]]}"};
    for (size_t i = 0; i < sizeof(bad) / sizeof(bad[0]); ++i)
        if (parse(bad[i], strlen(bad[i]), &nt, &ns) == 0) { printf("hostile buffer %zu accepted\n", i); return 2; }

    /* formatter: exact bound, then bounds that are too small (must return -1, never write past `cap`) */
    enum { M = 5, T = 12, MODES = 2, S = 2 };
    double pred[MODES * T * M * 2];
    for (int i = 0; i < MODES * T * M * 2; ++i) pred[i] = (i % 7 == 0) ? NAN : (i % 11 == 0 ? -INFINITY : (i - 60) * 123456.789 / 7.0);
    pred[3] = 0.125; pred[5] = -0.001; pred[9] = 5e15; pred[13] = 1e16; pred[15] = 99999999.995;
    const int64_t split[S + 1] = {0, 2, 5}, ped[M] = {7, 8, -3, 123456789012LL, 0}, sid[S] = {1, -2}, ff[S] = {100, 9000000000LL}, fd[S] = {10, 1},
                  st[S] = {0, 5}, en[S] = {200, 25};
    const size_t bound = tnp_format_predictions_bound(MODES, T, M, S);
    char *out = (char *)malloc(bound);
    const int64_t n = tnp_format_predictions(pred, MODES, T, M, S, split, ped, sid, ff, fd, st, en, out, bound);
    if (n <= 0 || (size_t)n > bound) { printf("formatter: %lld of %zu\n", (long long)n, bound); return 3; }
    int lines = 0;
    for (int64_t i = 0; i < n; ++i) lines += out[i] == '\n';
    if (lines != S + MODES * T * M) { printf("formatter: %d lines\n", lines); return 4; }
    free(out);
    for (size_t cap = 0; cap < 600; cap += 37) {
        char *small = (char *)malloc(cap ? cap : 1);
        if (tnp_format_predictions(pred, MODES, T, M, S, split, ped, sid, ff, fd, st, en, small, cap) != -1) { printf("small cap %zu accepted\n", cap); return 5; }
        free(small);
    }
    printf("ok: %lld bytes, %d lines\n", (long long)n, lines);
    return 0;
}
