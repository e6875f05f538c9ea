/*
 * oracle/pigo_oracle.c -- CPU ORACLE.  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A plain-C restatement of the reference's cascade-scan hot path, statement by statement:
 *
 *     (*Pigo).Unpack                 /root/reference/core/pigo.go:51-110
 *     (*Pigo).classifyRegion         /root/reference/core/pigo.go:113-147
 *     (*Pigo).classifyRotatedRegion  /root/reference/core/pigo.go:150-191
 *     (*Pigo).RunCascade             /root/reference/core/pigo.go:212-258
 *     (*Pigo).ClusterDetections      /root/reference/core/pigo.go:262-308
 *     abs / min / max / pow          /root/reference/core/utils.go:8-29,42-52
 *
 * plus a restatement of the one third-party algorithm the path depends on that is NOT under
 * /root/reference: Go's standard-library `sort.Slice` (pattern-defeating quicksort,
 * `sort/zsortfunc.go`, Go 1.19 .. 1.22; the reference pins `go 1.22` in go.mod:3 and its CI runs
 * 1.21/1.22).  It is unstable, so the order of *tied* Q values inside ClusterDetections is a
 * property of that exact algorithm.
 *
 * PARITY UNPINNED.  The reference is pure Go and there is no Go toolchain in this image or on
 * the GPU box, so the reference itself cannot be executed to mint vectors, and its own tests
 * (core/pigo_test.go:68-84, core/flploc_test.go:102-153) pin only "at least one cluster" and
 * "exactly one cluster with Scale>50" on testdata/sample.jpg.  This file is therefore pinned by
 *   (1) those two weak invariants (tests/test_oracle.py),
 *   (2) an independent vectorised NumPy restatement (oracle/np_restatement.py) that must agree
 *       bit-for-bit on every committed fixture, and
 *   (3) the survey-time probe values recorded in SURVEY.md Appendix C.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (pigo_amd/) never links, imports or falls back to it.
 *
 * Build: see oracle/Makefile  (gcc -O2 -ffp-contract=off, no -ffast-math: float32 adds must stay
 * sequential and unfused, exactly like the Go compiler on amd64).
 *
 * Go's `int` is 64-bit on amd64: every integer below is `long long` so no intermediate can wrap
 * differently from the reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef long long gint; /* Go `int` on amd64 */

/* Detection, core/pigo.go:195-200.  Same field order; 32 bytes like Go/amd64. */
typedef struct {
    gint row, col, scale;
    float q;
} oracle_det;

/* Pigo struct, core/pigo.go:37-43 */
typedef struct {
    int8_t *tree_codes;
    float *tree_pred;
    float *tree_threshold;
    uint32_t tree_depth;
    uint32_t tree_num;
    size_t n_codes, n_pred;
} oracle_pigo;

#define ORACLE_OK 0
#define ORACLE_ERR_PANIC (-1) /* the Go code would panic (slice index out of range) */
#define ORACLE_ERR_ALLOC (-2)

/* ---- core/utils.go ------------------------------------------------------------------------- */

/* utils.go:8-13 */
static gint go_abs(gint x) { return x < 0 ? -x : x; }
/* utils.go:16-21 */
static gint go_min(gint a, gint b) { return a < b ? a : b; }
/* utils.go:24-29 */
static gint go_max(gint a, gint b) { return a > b ? a : b; }
/* utils.go:42-52: square-and-multiply on float64 */
static double go_pow(double base, gint exp)
{
    double result = 1.0;
    while (exp > 0) {
        if (exp % 2 == 1)
            result *= base;
        exp >>= 1;
        base *= base;
    }
    return result;
}

static uint32_t le32(const uint8_t *p)
{
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

/* ---- Unpack, pigo.go:51-110 ------------------------------------------------------------------ */

void oracle_free(oracle_pigo *pg)
{
    if (!pg)
        return;
    free(pg->tree_codes);
    free(pg->tree_pred);
    free(pg->tree_threshold);
    free(pg);
}

int oracle_unpack(const uint8_t *packet, size_t len, oracle_pigo **out)
{
    *out = NULL;
    size_t pos = 8; /* pigo.go:61 "We skip the first 8 bytes" */
    if (len < pos + 4)
        return ORACLE_ERR_PANIC; /* binary.LittleEndian.Uint32 on a short slice panics */
    uint32_t tree_depth = le32(packet + pos); /* :64 */
    pos += 4;
    if (len < pos + 4)
        return ORACLE_ERR_PANIC;
    uint32_t tree_num = le32(packet + pos); /* :68 */
    pos += 4;                               /* :75 */
    if (tree_depth > 20)
        return ORACLE_ERR_ALLOC; /* 4*2^d bytes per tree: refuse absurd depths instead of OOM */

    gint ncode = (gint)(4 * go_pow(2, (gint)tree_depth) - 4); /* :81 */
    gint nleaf = (gint)go_pow(2, (gint)tree_depth);           /* :89 */

    oracle_pigo *pg = (oracle_pigo *)calloc(1, sizeof(*pg));
    if (!pg)
        return ORACLE_ERR_ALLOC;
    pg->tree_depth = tree_depth;
    pg->tree_num = tree_num;
    /* check the whole packet length first (Go would panic at the first short slice) */
    unsigned long long need = (unsigned long long)pos + (unsigned long long)tree_num * ((unsigned long long)ncode + 4ull * nleaf + 4ull);
    if ((unsigned long long)len < need) {
        free(pg);
        return ORACLE_ERR_PANIC;
    }
    pg->n_codes = (size_t)tree_num * (size_t)(ncode + 4);
    pg->n_pred = (size_t)tree_num * (size_t)nleaf;
    pg->tree_codes = (int8_t *)malloc(pg->n_codes ? pg->n_codes : 1);
    pg->tree_pred = (float *)malloc((pg->n_pred ? pg->n_pred : 1) * sizeof(float));
    pg->tree_threshold = (float *)malloc((tree_num ? tree_num : 1) * sizeof(float));
    if (!pg->tree_codes || !pg->tree_pred || !pg->tree_threshold) {
        oracle_free(pg);
        return ORACLE_ERR_ALLOC;
    }
    size_t nc = 0, np = 0;
    for (gint t = 0; t < (gint)tree_num; t++) { /* :77 */
        /* :79 four zero bytes in front of each tree so that node k lives at byte 4k */
        pg->tree_codes[nc++] = 0;
        pg->tree_codes[nc++] = 0;
        pg->tree_codes[nc++] = 0;
        pg->tree_codes[nc++] = 0;
        /* :81-84 unsigned -> signed reinterpretation */
        memcpy(pg->tree_codes + nc, packet + pos, (size_t)ncode);
        nc += (size_t)ncode;
        pos += (size_t)ncode; /* :86 */
        for (gint i = 0; i < nleaf; i++) { /* :89-95 */
            uint32_t u = le32(packet + pos);
            float f;
            memcpy(&f, &u, 4);
            pg->tree_pred[np++] = f;
            pos += 4;
        }
        uint32_t u = le32(packet + pos); /* :96-100 */
        float f;
        memcpy(&f, &u, 4);
        pg->tree_threshold[t] = f;
        pos += 4;
    }
    *out = pg;
    return ORACLE_OK; /* :103-109: error is always nil */
}

uint32_t oracle_tree_depth(const oracle_pigo *pg) { return pg->tree_depth; }
uint32_t oracle_tree_num(const oracle_pigo *pg) { return pg->tree_num; }
const int8_t *oracle_tree_codes(const oracle_pigo *pg) { return pg->tree_codes; }
const float *oracle_tree_pred(const oracle_pigo *pg) { return pg->tree_pred; }
const float *oracle_tree_threshold(const oracle_pigo *pg) { return pg->tree_threshold; }

/* ---- classifyRegion, pigo.go:113-147 ---------------------------------------------------------- */

/* `*panic` is set when an index would be out of range in Go (slice bounds check). `last_tree`
 * (optional) receives the index of the tree at which the window was rejected (tree_num if it
 * survived) -- instrumentation for the reject-profile statistics, not part of the reference. */
static float classify_region(const oracle_pigo *pg, gint r, gint c, gint s, gint tree_depth, const uint8_t *pixels,
                             gint npixels, gint dim, int *panic, gint *last_tree)
{
    gint root = 0;
    float out = 0.0f;

    r = r * 256; /* :119 */
    c = c * 256; /* :120 */

    if (pg->tree_num > 0) {                           /* :122 */
        for (gint i = 0; i < (gint)pg->tree_num; i++) { /* :123 */
            gint idx = 1;
            for (gint j = 0; j < (gint)pg->tree_depth; j++) { /* :125 */
                gint x1 = ((r + (gint)pg->tree_codes[root + 4 * idx + 0] * s) >> 8) * dim +
                          ((c + (gint)pg->tree_codes[root + 4 * idx + 1] * s) >> 8); /* :126 */
                gint x2 = ((r + (gint)pg->tree_codes[root + 4 * idx + 2] * s) >> 8) * dim +
                          ((c + (gint)pg->tree_codes[root + 4 * idx + 3] * s) >> 8); /* :127 */
                if (x1 < 0 || x1 >= npixels || x2 < 0 || x2 >= npixels) {
                    *panic = 1;
                    return -1.0f;
                }
                idx = 2 * idx + (pixels[x1] <= pixels[x2] ? 1 : 0); /* :129-135 */
            }
            out += pg->tree_pred[tree_depth * i + idx - tree_depth]; /* :137 (float32 add) */

            if (out <= pg->tree_threshold[i]) { /* :139 */
                if (last_tree)
                    *last_tree = i;
                return -1.0f;
            }
            root += 4 * tree_depth; /* :142 */
        }
        if (last_tree)
            *last_tree = (gint)pg->tree_num;
        return out - pg->tree_threshold[pg->tree_num - 1]; /* :144 */
    }
    if (last_tree)
        *last_tree = 0;
    return 0.0f; /* :146 */
}

/* ---- classifyRotatedRegion, pigo.go:150-191 --------------------------------------------------- */

static const gint q_cos_table[33] = {256, 251, 236, 212, 181, 142, 97, 49, 0, -49, -97, -142, -181, -212, -236, -251, -256,
                                     -251, -236, -212, -181, -142, -97, -49, 0, 49, 97, 142, 181, 212, 236, 251, 256}; /* :156 */
static const gint q_sin_table[33] = {0, 49, 97, 142, 181, 212, 236, 251, 256, 251, 236, 212, 181, 142, 97, 49, 0,
                                     -49, -97, -142, -181, -212, -236, -251, -256, -251, -236, -212, -181, -142, -97, -49, 0}; /* :157 */

static float classify_rotated_region(const oracle_pigo *pg, gint r, gint c, gint s, gint tree_depth, double a, gint nrows,
                                     gint ncols, const uint8_t *pixels, gint npixels, gint dim, int *panic, gint *last_tree)
{
    gint root = 0;
    float out = 0.0f;
    (void)ncols; /* unused in the reference too (quirk Q1: columns are clamped with nrows-1) */

    gint ai = (gint)(32.0 * a); /* :159 int(32.0*a) truncates */
    if (ai < 0 || ai > 32) {
        *panic = 1;
        return -1.0f;
    }
    gint qsin = s * q_sin_table[ai]; /* :159 */
    gint qcos = s * q_cos_table[ai]; /* :160 */

    if (pg->tree_num > 0) {
        for (gint i = 0; i < (gint)pg->tree_num; i++) {
            gint idx = 1;
            for (gint j = 0; j < (gint)pg->tree_depth; j++) {
                const int8_t *tc = pg->tree_codes + root + 4 * idx;
                gint r1 = go_abs(go_min(nrows - 1, go_max(0, 65536 * r + qcos * (gint)tc[0] - qsin * (gint)tc[1]) >> 16)); /* :167 */
                gint c1 = go_abs(go_min(nrows - 1, go_max(0, 65536 * c + qsin * (gint)tc[0] + qcos * (gint)tc[1]) >> 16)); /* :168 */
                gint r2 = go_abs(go_min(nrows - 1, go_max(0, 65536 * r + qcos * (gint)tc[2] - qsin * (gint)tc[3]) >> 16)); /* :170 */
                gint c2 = go_abs(go_min(nrows - 1, go_max(0, 65536 * c + qsin * (gint)tc[2] + qcos * (gint)tc[3]) >> 16)); /* :171 */
                gint x1 = r1 * dim + c1, x2 = r2 * dim + c2;
                if (x1 < 0 || x1 >= npixels || x2 < 0 || x2 >= npixels) {
                    *panic = 1;
                    return -1.0f;
                }
                idx = 2 * idx + (pixels[x1] <= pixels[x2] ? 1 : 0); /* :173-179 */
            }
            out += pg->tree_pred[tree_depth * i + idx - tree_depth]; /* :181 */

            if (out <= pg->tree_threshold[i]) { /* :183 */
                if (last_tree)
                    *last_tree = i;
                return -1.0f;
            }
            root += 4 * tree_depth; /* :186 */
        }
        if (last_tree)
            *last_tree = (gint)pg->tree_num;
        return out - pg->tree_threshold[pg->tree_num - 1]; /* :188 */
    }
    if (last_tree)
        *last_tree = 0;
    return 0.0f;
}

/* single-window entry points for the tests */
float oracle_classify_region(const oracle_pigo *pg, gint r, gint c, gint s, const uint8_t *pixels, gint npixels, gint dim,
                             int *panic)
{
    *panic = 0;
    return classify_region(pg, r, c, s, (gint)go_pow(2, (gint)pg->tree_depth), pixels, npixels, dim, panic, NULL);
}

float oracle_classify_rotated_region(const oracle_pigo *pg, gint r, gint c, gint s, double a, gint nrows, gint ncols,
                                     const uint8_t *pixels, gint npixels, gint dim, int *panic)
{
    *panic = 0;
    return classify_rotated_region(pg, r, c, s, (gint)go_pow(2, (gint)pg->tree_depth), a, nrows, ncols, pixels, npixels, dim,
                                   panic, NULL);
}

/* ---- RunCascade, pigo.go:212-258 --------------------------------------------------------------- */

/*
 * Returns the number of detections the reference would return (may exceed `cap`; only the first
 * `cap` are stored), or ORACLE_ERR_PANIC.  `n_windows` (optional) receives the number of
 * classified windows; `hist` (optional, tree_num+1 entries) receives, per tree index, how many
 * windows were rejected AT that tree (entry tree_num = survivors).
 */
gint oracle_run_cascade(const oracle_pigo *pg, const uint8_t *pixels, gint npixels, gint rows, gint cols, gint dim,
                        gint min_size, gint max_size, double shift_factor, double scale_factor, double angle,
                        oracle_det *out, gint cap, gint *n_windows, uint64_t *hist)
{
    gint ndet = 0, nwin = 0;
    gint tree_depth = (gint)go_pow(2, (gint)pg->tree_depth); /* :216 (quirk Q3: this is 2^depth) */
    float q;
    gint scale = min_size; /* :219 */
    int panic = 0;

    while (scale <= max_size) { /* :226 */
        double m = shift_factor * (double)scale;
        gint step = (gint)(m > 1.0 ? m : 1.0); /* :227 int(math.Max(shift*scale, 1)) */
        gint offset = scale / 2 + 1;           /* :228 */

        for (gint row = offset; row <= rows - offset; row += step) {     /* :230 */
            for (gint col = offset; col <= cols - offset; col += step) { /* :231 */
                gint last = 0;
                if (angle > 0.0) {   /* :232 */
                    if (angle > 1.0) /* :233 */
                        angle = 1.0;
                    q = classify_rotated_region(pg, row, col, scale, tree_depth, angle, rows, cols, pixels, npixels, dim, &panic,
                                                &last); /* :236 */
                } else {
                    q = classify_region(pg, row, col, scale, tree_depth, pixels, npixels, dim, &panic, &last); /* :238 */
                }
                if (panic)
                    return ORACLE_ERR_PANIC;
                nwin++;
                if (hist)
                    hist[last]++;
                if (q > 0.0f) { /* :246 */
                    if (ndet < cap) {
                        out[ndet].row = row;
                        out[ndet].col = col;
                        out[ndet].scale = scale;
                        out[ndet].q = q;
                    }
                    ndet++;
                }
            }
        }
        /* :255 scale = int(float64(scale) + math.Max(2, float64(scale)*ScaleFactor - float64(scale))) */
        double grow = (double)scale * scale_factor - (double)scale;
        scale = (gint)((double)scale + (grow > 2.0 ? grow : 2.0));
    }
    if (n_windows)
        *n_windows = nwin;
    return ndet;
}

/* ---- Go sort.Slice restated (sort/zsortfunc.go, Go 1.19-1.22), specialised to Detection.Q ------ */

typedef struct {
    oracle_det *d;
} less_swap;

static int ls_less(less_swap *s, gint i, gint j) { return s->d[i].q < s->d[j].q; } /* pigo.go:264-266 */
static void ls_swap(less_swap *s, gint i, gint j)
{
    oracle_det t = s->d[i];
    s->d[i] = s->d[j];
    s->d[j] = t;
}

static void insertion_sort_func(less_swap *data, gint a, gint b)
{
    for (gint i = a + 1; i < b; i++)
        for (gint j = i; j > a && ls_less(data, j, j - 1); j--)
            ls_swap(data, j, j - 1);
}

static void sift_down_func(less_swap *data, gint lo, gint hi, gint first)
{
    gint root = lo;
    for (;;) {
        gint child = 2 * root + 1;
        if (child >= hi)
            return;
        if (child + 1 < hi && ls_less(data, first + child, first + child + 1))
            child++;
        if (!ls_less(data, first + root, first + child))
            return;
        ls_swap(data, first + root, first + child);
        root = child;
    }
}

static void heap_sort_func(less_swap *data, gint a, gint b)
{
    gint first = a, lo = 0, hi = b - a;
    for (gint i = (hi - 1) / 2; i >= 0; i--)
        sift_down_func(data, i, hi, first);
    for (gint i = hi - 1; i >= 0; i--) {
        ls_swap(data, first, first + i);
        sift_down_func(data, lo, i, first);
    }
}

static gint partition_func(less_swap *data, gint a, gint b, gint pivot, int *already_partitioned)
{
    ls_swap(data, a, pivot);
    gint i = a + 1, j = b - 1;
    while (i <= j && ls_less(data, i, a))
        i++;
    while (i <= j && !ls_less(data, j, a))
        j--;
    if (i > j) {
        ls_swap(data, j, a);
        *already_partitioned = 1;
        return j;
    }
    ls_swap(data, i, j);
    i++;
    j--;
    for (;;) {
        while (i <= j && ls_less(data, i, a))
            i++;
        while (i <= j && !ls_less(data, j, a))
            j--;
        if (i > j)
            break;
        ls_swap(data, i, j);
        i++;
        j--;
    }
    ls_swap(data, j, a);
    *already_partitioned = 0;
    return j;
}

static gint partition_equal_func(less_swap *data, gint a, gint b, gint pivot)
{
    ls_swap(data, a, pivot);
    gint i = a + 1, j = b - 1;
    for (;;) {
        while (i <= j && !ls_less(data, a, i))
            i++;
        while (i <= j && ls_less(data, a, j))
            j--;
        if (i > j)
            break;
        ls_swap(data, i, j);
        i++;
        j--;
    }
    return i;
}

static int partial_insertion_sort_func(less_swap *data, gint a, gint b)
{
    const gint max_steps = 5, shortest_shifting = 50;
    gint i = a + 1;
    for (gint j = 0; j < max_steps; j++) {
        while (i < b && !ls_less(data, i, i - 1))
            i++;
        if (i == b)
            return 1;
        if (b - a < shortest_shifting)
            return 0;
        ls_swap(data, i, i - 1);
        if (i - a >= 2) {
            for (gint k = i - 1; k >= 1; k--) {
                if (!ls_less(data, k, k - 1))
                    break;
                ls_swap(data, k, k - 1);
            }
        }
        if (b - i >= 2) {
            for (gint k = i + 1; k < b; k++) {
                if (!ls_less(data, k, k - 1))
                    break;
                ls_swap(data, k, k - 1);
            }
        }
    }
    return 0;
}

static int bits_len(uint64_t x)
{
    int n = 0;
    while (x) {
        n++;
        x >>= 1;
    }
    return n;
}

static uint64_t xorshift_next(uint64_t *r)
{
    *r ^= *r << 13;
    *r ^= *r >> 7;
    *r ^= *r << 17;
    return *r;
}

static void break_patterns_func(less_swap *data, gint a, gint b)
{
    gint length = b - a;
    if (length >= 8) {
        uint64_t random = (uint64_t)length;
        uint64_t modulus = 1ull << bits_len((uint64_t)length);
        gint idx = a + (length / 4) * 2 - 1;
        for (gint i = 0; i < 3; i++) {
            gint other = (gint)(xorshift_next(&random) & (modulus - 1));
            if (other >= length)
                other -= length;
            ls_swap(data, idx - 1 + i, a + other);
        }
    }
}

static void order2_func(less_swap *data, gint *a, gint *b, gint *swaps)
{
    if (ls_less(data, *b, *a)) {
        (*swaps)++;
        gint t = *a;
        *a = *b;
        *b = t;
    }
}

static gint median_func(less_swap *data, gint a, gint b, gint c, gint *swaps)
{
    order2_func(data, &a, &b, swaps);
    order2_func(data, &b, &c, swaps);
    order2_func(data, &a, &b, swaps);
    return b;
}

static gint median_adjacent_func(less_swap *data, gint a, gint *swaps) { return median_func(data, a - 1, a, a + 1, swaps); }

enum { HINT_UNKNOWN = 0, HINT_INCREASING, HINT_DECREASING };

static gint choose_pivot_func(less_swap *data, gint a, gint b, int *hint)
{
    const gint shortest_ninther = 50, max_swaps = 4 * 3;
    gint l = b - a, swaps = 0;
    gint i = a + l / 4 * 1, j = a + l / 4 * 2, k = a + l / 4 * 3;
    if (l >= 8) {
        if (l >= shortest_ninther) {
            i = median_adjacent_func(data, i, &swaps);
            j = median_adjacent_func(data, j, &swaps);
            k = median_adjacent_func(data, k, &swaps);
        }
        j = median_func(data, i, j, k, &swaps);
    }
    if (swaps == 0)
        *hint = HINT_INCREASING;
    else if (swaps == max_swaps)
        *hint = HINT_DECREASING;
    else
        *hint = HINT_UNKNOWN;
    return j;
}

static void reverse_range_func(less_swap *data, gint a, gint b)
{
    gint i = a, j = b - 1;
    while (i < j) {
        ls_swap(data, i, j);
        i++;
        j--;
    }
}

static void pdqsort_func(less_swap *data, gint a, gint b, gint limit)
{
    const gint max_insertion = 12;
    int was_balanced = 1, was_partitioned = 1;
    for (;;) {
        gint length = b - a;
        if (length <= max_insertion) {
            insertion_sort_func(data, a, b);
            return;
        }
        if (limit == 0) {
            heap_sort_func(data, a, b);
            return;
        }
        if (!was_balanced) {
            break_patterns_func(data, a, b);
            limit--;
        }
        int hint;
        gint pivot = choose_pivot_func(data, a, b, &hint);
        if (hint == HINT_DECREASING) {
            reverse_range_func(data, a, b);
            pivot = (b - 1) - (pivot - a);
            hint = HINT_INCREASING;
        }
        if (was_balanced && was_partitioned && hint == HINT_INCREASING) {
            if (partial_insertion_sort_func(data, a, b))
                return;
        }
        if (a > 0 && !ls_less(data, a - 1, pivot)) {
            gint mid = partition_equal_func(data, a, b, pivot);
            a = mid;
            continue;
        }
        int already;
        gint mid = partition_func(data, a, b, pivot, &already);
        was_partitioned = already;
        gint left_len = mid - a, right_len = b - mid;
        gint balance_threshold = length / 8;
        if (left_len < right_len) {
            was_balanced = left_len >= balance_threshold;
            pdqsort_func(data, a, mid, limit);
            a = mid + 1;
        } else {
            was_balanced = right_len >= balance_threshold;
            pdqsort_func(data, mid + 1, b, limit);
            b = mid;
        }
    }
}

/* sort.Slice(detections, func(i, j) bool { return detections[i].Q < detections[j].Q })  pigo.go:264-266 */
void oracle_sort_by_q(oracle_det *d, gint n)
{
    less_swap ls = {d};
    pdqsort_func(&ls, 0, n, (gint)bits_len((uint64_t)n));
}

/* ---- ClusterDetections, pigo.go:262-308 --------------------------------------------------------- */

static double dmax(double a, double b) { return a > b ? a : b; } /* math.Max on finite inputs */
static double dmin(double a, double b) { return a < b ? a : b; } /* math.Min on finite inputs */

/* calcIoU, pigo.go:268-278 */
static double calc_iou(const oracle_det *det1, const oracle_det *det2)
{
    double r1 = (double)det1->row, c1 = (double)det1->col, s1 = (double)det1->scale;
    double r2 = (double)det2->row, c2 = (double)det2->col, s2 = (double)det2->scale;
    double over_row = dmax(0, dmin(r1 + s1 / 2, r2 + s2 / 2) - dmax(r1 - s1 / 2, r2 - s2 / 2));
    double over_col = dmax(0, dmin(c1 + s1 / 2, c2 + s2 / 2) - dmax(c1 - s1 / 2, c2 - s2 / 2));
    return over_row * over_col / (s1 * s1 + s2 * s2 - over_row * over_col);
}

double oracle_calc_iou(const oracle_det *a, const oracle_det *b) { return calc_iou(a, b); }

/*
 * Sorts `detections` in place (as the reference does) and writes the clusters to `out`
 * (capacity >= n is always enough: at most one cluster per detection).  Returns the cluster count.
 * `n_ties` (optional) receives the number of adjacent equal-Q pairs after sorting: when it is
 * non-zero the result depends on Go's unstable sort and parity on that input is UNPINNED.
 */
gint oracle_cluster_detections(oracle_det *detections, gint n, double iou_threshold, oracle_det *out, gint *n_ties)
{
    oracle_sort_by_q(detections, n); /* :264 */
    if (n_ties) {
        gint t = 0;
        for (gint i = 1; i < n; i++)
            if (detections[i].q == detections[i - 1].q)
                t++;
        *n_ties = t;
    }
    unsigned char *assignments = (unsigned char *)calloc((size_t)(n > 0 ? n : 1), 1); /* :279 */
    gint nclusters = 0;                                                               /* :280 */
    for (gint i = 0; i < n; i++) {                                                    /* :282 */
        if (!assignments[i]) {                                                        /* :285 */
            gint r = 0, c = 0, s = 0, cnt = 0;
            float q = 0.0f;
            for (gint j = 0; j < n; j++) {                                            /* :290 */
                if (calc_iou(&detections[i], &detections[j]) > iou_threshold) {       /* :293 */
                    assignments[j] = 1;
                    r += detections[j].row;
                    c += detections[j].col;
                    s += detections[j].scale;
                    q += detections[j].q; /* float32, in j order */
                    cnt++;
                }
            }
            if (cnt > 0) { /* :302 */
                out[nclusters].row = r / cnt;
                out[nclusters].col = c / cnt;
                out[nclusters].scale = s / cnt;
                out[nclusters].q = q;
                nclusters++;
            }
        }
    }
    free(assignments);
    return nclusters;
}

/* ======================================================================================================
 * RgbToGrayscale  (the step immediately before RunCascade)
 *
 *     RgbToGrayscale              /root/reference/core/grayscale.go:8-23
 *     (*Canvas).rgbaToGrayscale   /root/reference/wasm/canvas/canvas.go:179-191   (kind 2)
 *
 * The reference calls src.At(x, y).RGBA() (grayscale.go:14), which is Go standard-library code that is
 * NOT under /root/reference: package image/color (go 1.22, go.mod:3).  Its published behaviour, restated:
 *     color.NRGBA.RGBA():  r = R; r |= r << 8; r *= A; r /= 0xff   (uint32; same for g, b)   -> kind 0
 *     color.RGBA.RGBA():   r = R; r |= r << 8                      (already premultiplied)   -> kind 1
 * GetImage / DecodeImage hand the CLI an *image.NRGBA (core/image.go:12-90), the reference's own test builds
 * an *image.RGBA (core/grayscale_test.go:15).  The mix is evaluated in float64, left to right, unfused
 * (Go/amd64, GOAMD64=v1), and uint8() of a float64 truncates toward zero (grayscale.go:15-19).
 *
 * PARITY UNPINNED like the rest of this file (no Go toolchain); pinned by the one known answer the
 * reference's test implies (R=G=B=v, A=255 gives v; core/grayscale_test.go:14-34) and the NumPy restatement.
 * ====================================================================================================== */
#define ORACLE_PIX_NRGBA 0
#define ORACLE_PIX_RGBA 1
#define ORACLE_PIX_CANVAS 2

static uint32_t go_rgba16(uint32_t c8, uint32_t a8, int kind)
{
    uint32_t r = c8;
    r |= r << 8;
    if (kind == ORACLE_PIX_NRGBA) {
        r *= a8;
        r /= 0xff;
    }
    return r;
}

/* pix: rows of `stride` bytes, 4 bytes {R,G,B,A} per pixel; out: width*height bytes, out[y*width + x]
 * (grayscale.go:10,15).  Returns 0, or -1 for an unknown kind. */
int oracle_rgb_to_grayscale(const uint8_t *pix, gint width, gint height, gint stride, int kind, uint8_t *out)
{
    if (kind < 0 || kind > 2) return -1;
    for (gint y = 0; y < height; y++) {          /* grayscale.go:12 */
        for (gint x = 0; x < width; x++) {       /* grayscale.go:13 */
            const uint8_t *p = pix + y * stride + 4 * x;
            if (kind == ORACLE_PIX_CANVAS) {
                /* canvas.go:184-187: uint8(math.Round(0.2126*R + 0.7152*G + 0.0722*B)); alpha ignored */
                double v = 0.2126 * (double)p[0] + 0.7152 * (double)p[1];
                v = v + 0.0722 * (double)p[2];
                out[y * width + x] = (uint8_t)round(v); /* math.Round: half away from zero, like C round() */
                continue;
            }
            const uint32_t r = go_rgba16(p[0], p[3], kind), g = go_rgba16(p[1], p[3], kind), b = go_rgba16(p[2], p[3], kind);
            double v = 0.299 * (double)r + 0.587 * (double)g; /* grayscale.go:16-17 */
            v = v + 0.114 * (double)b;                         /* grayscale.go:18 */
            out[y * width + x] = (uint8_t)(v / 256);           /* grayscale.go:15,19: truncation */
        }
    }
    return 0;
}

/* ======================================================================================================
 * Pupil / facial-landmark localisation  (SURVEY.md section 8, rows f2 + f3: the step right after the scan)
 *
 *     PuplocCascade, UnpackCascade      /root/reference/core/puploc.go:22-103   (wire format, row f3)
 *     (*PuplocCascade).classifyRegion   /root/reference/core/puploc.go:106-154
 *     ...classifyRotatedRegion          /root/reference/core/puploc.go:157-217
 *     (*PuplocCascade).RunDetector      /root/reference/core/puploc.go:239-277
 *     (*PuplocCascade).GetLandmarkPoint /root/reference/core/flploc.go:36-57
 *
 * Two things in RunDetector are NOT a function of its arguments, so the restatement takes them as inputs:
 *   - the perturbations come from the global math/rand source (puploc.go:248-250): `rnd` holds the 3*Perturbs
 *     values rand.Float32() would return, in draw order (row, col, scale of perturbation 0, then 1, ...);
 *   - the three 63-entry result arrays live in a sync.Pool object that is never cleared (puploc.go:228-237,
 *     242-243) and are sorted WHOLE (puploc.go:267-269), so with Perturbs < 63 the entries [Perturbs, 63) are
 *     whatever the previous user of that pool object left behind: `pool` (3*63 floats, in/out) is that object;
 *     NULL means a brand-new one (all zeros, puploc.go:232-234).
 * float32 arithmetic is evaluated operation by operation (Go/amd64 does not fuse; -ffp-contract=off here).
 * PARITY UNPINNED like the rest of this file: the reference's tests only ask for "eyes found"
 * (core/puploc_test.go:34-80) and "15 landmark points" (core/flploc_test.go).
 * ====================================================================================================== */
typedef struct {
    int8_t *tree_codes; /* stages*trees*(4*2^depth - 4) bytes, no pad in front of a tree (puploc.go:76-79) */
    float *tree_preds;  /* stages*trees*2^depth*2 */
    float scales;
    uint32_t stages, trees, tree_depth;
    size_t n_codes, n_preds;
} oracle_puploc;

typedef struct { /* Puploc, puploc.go:14-19 */
    gint row, col;
    float scale;
    gint perturbs;
} oracle_puploc_det;

void oracle_puploc_free(oracle_puploc *plc)
{
    if (!plc) return;
    free(plc->tree_codes);
    free(plc->tree_preds);
    free(plc);
}

/* UnpackCascade, puploc.go:38-103 */
int oracle_puploc_unpack(const uint8_t *packet, size_t len, oracle_puploc **out)
{
    *out = NULL;
    if (len < 16) return ORACLE_ERR_PANIC; /* binary.LittleEndian.Uint32 on a short slice panics (:51-66) */
    uint32_t stages = le32(packet + 0);    /* :51 */
    uint32_t u32scales = le32(packet + 4); /* :55 */
    float scales;
    memcpy(&scales, &u32scales, 4);        /* :57 */
    uint32_t trees = le32(packet + 8);     /* :61 */
    uint32_t tree_depth = le32(packet + 12); /* :65 */
    size_t pos = 16;
    if (tree_depth > 20) return ORACLE_ERR_ALLOC;
    gint depth = (gint)go_pow(2, (gint)tree_depth); /* :73 */
    unsigned long long per_tree = (unsigned long long)(4 * depth - 4) + 8ull * (unsigned long long)depth;
    unsigned long long need = 16ull + (unsigned long long)stages * trees * per_tree;
    if ((unsigned long long)len < need) return ORACLE_ERR_PANIC; /* packet[pos : pos+4*depth-4] past the end (:75) */
    oracle_puploc *plc = (oracle_puploc *)calloc(1, sizeof(*plc));
    if (!plc) return ORACLE_ERR_ALLOC;
    plc->stages = stages;
    plc->scales = scales;
    plc->trees = trees;
    plc->tree_depth = tree_depth;
    plc->n_codes = (size_t)stages * trees * (size_t)(4 * depth - 4);
    plc->n_preds = (size_t)stages * trees * (size_t)depth * 2;
    plc->tree_codes = (int8_t *)malloc(plc->n_codes ? plc->n_codes : 1);
    plc->tree_preds = (float *)malloc((plc->n_preds ? plc->n_preds : 1) * sizeof(float));
    if (!plc->tree_codes || !plc->tree_preds) {
        oracle_puploc_free(plc);
        return ORACLE_ERR_ALLOC;
    }
    size_t nc = 0, np = 0;
    for (gint s = 0; s < (gint)stages; s++) {     /* :69 */
        for (gint t = 0; t < (gint)trees; t++) {  /* :71 */
            memcpy(plc->tree_codes + nc, packet + pos, (size_t)(4 * depth - 4)); /* :75-78 */
            nc += (size_t)(4 * depth - 4);
            pos += (size_t)(4 * depth - 4);       /* :80 */
            for (gint i = 0; i < depth; i++)      /* :83 */
                for (gint l = 0; l < 2; l++) {    /* :84 */
                    uint32_t u = le32(packet + pos);
                    float f;
                    memcpy(&f, &u, 4);
                    plc->tree_preds[np++] = f;    /* :85-88 */
                    pos += 4;
                }
        }
    }
    *out = plc;
    return ORACLE_OK;
}

uint32_t oracle_puploc_stages(const oracle_puploc *p) { return p->stages; }
uint32_t oracle_puploc_trees(const oracle_puploc *p) { return p->trees; }
uint32_t oracle_puploc_depth(const oracle_puploc *p) { return p->tree_depth; }
float oracle_puploc_scales(const oracle_puploc *p) { return p->scales; }
const int8_t *oracle_puploc_codes(const oracle_puploc *p) { return p->tree_codes; }
const float *oracle_puploc_preds(const oracle_puploc *p) { return p->tree_preds; }

/* int8 negation as Go does it: -plc.treeCodes[..] is evaluated in int8 and wraps (-(-128) == -128), puploc.go:123-124 */
static gint neg_i8(int8_t v) { return (gint)(int8_t)(uint8_t)(0u - (uint8_t)v); }

/* classifyRegion, puploc.go:106-154.  res = {r, c, s}; *panic set when a pixel index would be out of range. */
static void puploc_classify_region(const oracle_puploc *plc, float r, float c, float s, gint tree_depth, gint nrows, gint ncols,
                                   const uint8_t *pixels, gint npixels, gint dim, int flip_v, float *res, int *panic)
{
    gint c1, c2, root = 0;
    for (gint i = 0; i < (gint)plc->stages; i++) {     /* :112 */
        float dr = 0.0f, dc = 0.0f;                     /* :113 */
        for (gint j = 0; j < (gint)plc->trees; j++) {  /* :115 */
            gint idx = 0;
            for (gint k = 0; k < (gint)plc->tree_depth; k++) { /* :117 */
                const gint sr = (gint)llround((double)s);       /* int(math.Round(float64(s))) */
                gint r1 = go_min(nrows - 1, go_max(0, (256 * (gint)r + (gint)plc->tree_codes[root + 4 * idx + 0] * sr) >> 8)); /* :118 */
                gint r2 = go_min(nrows - 1, go_max(0, (256 * (gint)r + (gint)plc->tree_codes[root + 4 * idx + 2] * sr) >> 8)); /* :119 */
                if (flip_v) { /* :123-125 */
                    c1 = go_min(ncols - 1, go_max(0, (256 * (gint)c + neg_i8(plc->tree_codes[root + 4 * idx + 1]) * sr) >> 8));
                    c2 = go_min(ncols - 1, go_max(0, (256 * (gint)c + neg_i8(plc->tree_codes[root + 4 * idx + 3]) * sr) >> 8));
                } else {      /* :126-128 */
                    c1 = go_min(ncols - 1, go_max(0, (256 * (gint)c + (gint)plc->tree_codes[root + 4 * idx + 1] * sr) >> 8));
                    c2 = go_min(ncols - 1, go_max(0, (256 * (gint)c + (gint)plc->tree_codes[root + 4 * idx + 3] * sr) >> 8));
                }
                const gint i1 = r1 * dim + c1, i2 = r2 * dim + c2;
                if (i1 < 0 || i1 >= npixels || i2 < 0 || i2 >= npixels) {
                    *panic = 1;
                    return;
                }
                idx = 2 * idx + 1 + (pixels[i1] > pixels[i2] ? 1 : 0); /* :130-136: bintest is p1 > p2 here */
            }
            const gint lut = 2 * ((gint)plc->trees * tree_depth * i + tree_depth * j + idx - (tree_depth - 1)); /* :138 */
            dr += plc->tree_preds[lut + 0];                            /* :140 */
            if (flip_v) dc += -plc->tree_preds[lut + 1];               /* :141-145 */
            else dc += plc->tree_preds[lut + 1];
            root += 4 * tree_depth - 4;                                /* :146 */
        }
        r += dr * s;          /* :149 */
        c += dc * s;          /* :150 */
        s *= plc->scales;     /* :151 */
    }
    res[0] = r;
    res[1] = c;
    res[2] = s;
}

/* classifyRotatedRegion, puploc.go:157-217 */
static void puploc_classify_rotated_region(const oracle_puploc *plc, float r, float c, float s, double a, gint tree_depth, gint nrows,
                                           gint ncols, const uint8_t *pixels, gint npixels, gint dim, int flip_v, float *res, int *panic)
{
    static const float q_cos_table[33] = {256, 251, 236, 212, 181, 142, 97, 49, 0, -49, -97, -142, -181, -212, -236, -251, -256,
                                          -251, -236, -212, -181, -142, -97, -49, 0, 49, 97, 142, 181, 212, 236, 251, 256}; /* :163 */
    static const float q_sin_table[33] = {0, 49, 97, 142, 181, 212, 236, 251, 256, 251, 236, 212, 181, 142, 97, 49, 0,
                                          -49, -97, -142, -181, -212, -236, -251, -256, -251, -236, -212, -181, -142, -97, -49, 0}; /* :164 */
    gint row1, col1, row2, col2, root = 0;
    const gint ai = (gint)(32.0 * a);
    if (ai < 0 || ai > 32) {
        *panic = 1;
        return;
    }
    const float qsin = s * q_sin_table[ai]; /* :166 */
    const float qcos = s * q_cos_table[ai]; /* :167 */
    for (gint i = 0; i < (gint)plc->stages; i++) {
        float dr = 0.0f, dc = 0.0f;
        for (gint j = 0; j < (gint)plc->trees; j++) {
            gint idx = 0;
            for (gint k = 0; k < (gint)plc->tree_depth; k++) {
                row1 = (gint)plc->tree_codes[root + 4 * idx + 0]; /* :175 */
                row2 = (gint)plc->tree_codes[root + 4 * idx + 2]; /* :176 */
                if (flip_v) { /* :180-182 */
                    col1 = neg_i8(plc->tree_codes[root + 4 * idx + 1]);
                    col2 = neg_i8(plc->tree_codes[root + 4 * idx + 3]);
                } else {      /* :183-185 */
                    col1 = (gint)plc->tree_codes[root + 4 * idx + 1];
                    col2 = (gint)plc->tree_codes[root + 4 * idx + 3];
                }
                const gint r1 = go_min(nrows - 1, go_max(0, 65536 * (gint)r + (gint)qcos * row1 - (gint)qsin * col1) >> 16); /* :188 */
                const gint cc1 = go_min(ncols - 1, go_max(0, 65536 * (gint)c + (gint)qsin * row1 + (gint)qcos * col1) >> 16); /* :189 */
                const gint r2 = go_min(nrows - 1, go_max(0, 65536 * (gint)r + (gint)qcos * row2 - (gint)qsin * col2) >> 16); /* :190 */
                const gint cc2 = go_min(ncols - 1, go_max(0, 65536 * (gint)c + (gint)qsin * row2 + (gint)qcos * col2) >> 16); /* :191 */
                const gint i1 = r1 * dim + cc1, i2 = r2 * dim + cc2;
                if (i1 < 0 || i1 >= npixels || i2 < 0 || i2 >= npixels) {
                    *panic = 1;
                    return;
                }
                idx = 2 * idx + 1 + (pixels[i1] <= pixels[i2] ? 1 : 0); /* :193-199: bintest is px1 <= px2 here */
            }
            const gint lut = 2 * ((gint)plc->trees * tree_depth * i + tree_depth * j + idx - (tree_depth - 1)); /* :201 */
            dr += plc->tree_preds[lut + 0];
            if (flip_v) dc += -plc->tree_preds[lut + 1];
            else dc += plc->tree_preds[lut + 1];
            root += 4 * tree_depth - 4;
        }
        r += dr * s; /* :212 */
        c += dc * s;
        s *= plc->scales;
    }
    res[0] = r;
    res[1] = c;
    res[2] = s;
}

/* one perturbation, exported for the tests: which==0 upright, 1 rotated */
int oracle_puploc_classify(const oracle_puploc *plc, float r, float c, float s, double a, int rotated, gint nrows, gint ncols,
                           const uint8_t *pixels, gint npixels, gint dim, int flip_v, float *res)
{
    int panic = 0;
    const gint td = (gint)go_pow(2, (gint)plc->tree_depth);
    if (rotated) puploc_classify_rotated_region(plc, r, c, s, a, td, nrows, ncols, pixels, npixels, dim, flip_v, res, &panic);
    else puploc_classify_region(plc, r, c, s, td, nrows, ncols, pixels, npixels, dim, flip_v, res, &panic);
    return panic ? ORACLE_ERR_PANIC : ORACLE_OK;
}

static int f32_less(const void *a, const void *b) /* plocSort.Less: q[i] < q[j]; values are finite, so any sort gives one result */
{
    const float x = *(const float *)a, y = *(const float *)b;
    return x < y ? -1 : (y < x ? 1 : 0);
}

/* RunDetector, puploc.go:239-277.  rnd: 3*Perturbs uniform [0,1) float32 in draw order; pool: 3*63 floats in/out or NULL. */
int oracle_puploc_run_detector(const oracle_puploc *plc, const oracle_puploc_det *pl, const uint8_t *pixels, gint npixels, gint rows,
                               gint cols, gint dim, double angle, int flip_v, const float *rnd, float *pool, oracle_puploc_det *out)
{
    float fresh[3 * 63];
    memset(fresh, 0, sizeof fresh);          /* plcPool.New: make([]float32, 63) x 3  (:231-235) */
    float *det_rows = pool ? pool : fresh, *det_cols = det_rows + 63, *det_scale = det_rows + 126;
    const gint tree_depth = (gint)go_pow(2, (gint)plc->tree_depth); /* :245 */
    float res[3];
    int panic = 0;
    if (pl->perturbs > 63) return ORACLE_ERR_PANIC; /* det.rows[63] = ...: index out of range (:262) -- after 63 iterations of work */
    for (gint i = 0; i < pl->perturbs; i++) {        /* :247 */
        const float u0 = rnd[3 * i + 0], u1 = rnd[3 * i + 1], u2 = rnd[3 * i + 2];
        float row = (float)pl->row + ((float)pl->scale * 0.15f) * (0.5f - u0); /* :248 */
        float col = (float)pl->col + ((float)pl->scale * 0.15f) * (0.5f - u1); /* :249 */
        float sc = (float)pl->scale * (0.925f + 0.15f * u2);                   /* :250 */
        if (angle > 0.0) {            /* :252 */
            if (angle > 1.0) angle = 1.0; /* :253-255 */
            puploc_classify_rotated_region(plc, row, col, sc, angle, tree_depth, rows, cols, pixels, npixels, dim, flip_v, res, &panic);
        } else {
            puploc_classify_region(plc, row, col, sc, tree_depth, rows, cols, pixels, npixels, dim, flip_v, res, &panic);
        }
        if (panic) return ORACLE_ERR_PANIC;
        det_rows[i] = res[0];  /* :262-264 */
        det_cols[i] = res[1];
        det_scale[i] = res[2];
    }
    qsort(det_rows, 63, sizeof(float), f32_less);  /* sort.Sort(plocSort(det.rows)): all 63 entries (:267-269) */
    qsort(det_cols, 63, sizeof(float), f32_less);
    qsort(det_scale, 63, sizeof(float), f32_less);
    const gint mid = (gint)llround((double)pl->perturbs / 2); /* int(math.Round(float64(pl.Perturbs)/2)) (:273) */
    if (mid < 0 || mid >= 63) return ORACLE_ERR_PANIC;
    out->row = (gint)det_rows[mid];   /* :273 */
    out->col = (gint)det_cols[mid];   /* :274 */
    out->scale = det_scale[mid];      /* :275 */
    out->perturbs = 0;                /* the returned &Puploc{} leaves Perturbs at its zero value */
    return ORACLE_OK;
}

/* GetLandmarkPoint, flploc.go:36-57 */
int oracle_get_landmark_point(const oracle_puploc *plc, const oracle_puploc_det *left_eye, const oracle_puploc_det *right_eye,
                              const uint8_t *pixels, gint npixels, gint rows, gint cols, gint dim, gint perturb, int flip_v,
                              const float *rnd, float *pool, oracle_puploc_det *out)
{
    const gint dx = (left_eye->row - right_eye->row) * (left_eye->row - right_eye->row); /* :37 */
    const gint dy = (left_eye->col - right_eye->col) * (left_eye->col - right_eye->col); /* :38 */
    const double dist = sqrt((double)(dx + dy));                                         /* :39 */
    const double row = (double)(left_eye->row + right_eye->row) / 2.0 + 0.25 * dist;     /* :41 */
    const double col = (double)(left_eye->col + right_eye->col) / 2.0 + 0.15 * dist;     /* :42 */
    const double scale = 3.0 * dist;                                                     /* :43 */
    oracle_puploc_det flploc;
    flploc.row = (gint)row;          /* :48 */
    flploc.col = (gint)col;          /* :49 */
    flploc.scale = (float)scale;     /* :50 */
    flploc.perturbs = perturb;       /* :51 */
    return oracle_puploc_run_detector(plc, &flploc, pixels, npixels, rows, cols, dim, 0.0, flip_v, rnd, pool, out); /* :53-56 */
}
