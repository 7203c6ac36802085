/* CPU oracle for the bipartite matcher.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's shared object; the product path never does.
 *
 * What it restates
 * ----------------
 * The reference's matcher is a third-party dependency that is not under
 * /root/reference: scipy.optimize.linear_sum_assignment (version unpinned in the
 * reference; scipy 1.15.3 is installed here), called from
 * ModelComponents/losses_and_metrics.py:240-243 on the fp32 slice
 * cost[i, :n_i, :].  scipy's solver is the rectangular shortest-augmenting-path
 * algorithm of D. F. Crouse, "On implementing 2D rectangular assignment
 * algorithms", IEEE TAES 52(4), 2016, run in fp64.  This file restates that
 * published algorithm in plain C with the behaviours that decide ties:
 *   - a tall matrix (nr > nc) is transposed first and the result re-sorted by row;
 *   - the candidate columns of one augmentation are scanned in a list that is
 *     initialised in REVERSE order (nc-1 .. 0) and shrunk by swap-with-last;
 *   - the next column is the first strict minimum in scan order, except that a
 *     later column with an equal value replaces it when that column is unassigned;
 *   - the reduced cost is evaluated as ((minVal + c) - u[i]) - v[j] in fp64.
 *
 * PARITY STATUS: pinned.  tests/test_oracle_lsap.py checks this restatement
 * against the real scipy on random, tied, constant, tall, wide and +inf matrices.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* returns 0 ok, -1 infeasible, -2 invalid (NaN / -inf) */
static int solve_wide(int nr, int nc, const double *cost, int64_t *col4row_out)
{
    double *u = calloc(nr, sizeof(double));
    double *v = calloc(nc, sizeof(double));
    double *spc = malloc(nc * sizeof(double));
    int *path = malloc(nc * sizeof(int));
    int *col4row = malloc(nr * sizeof(int));
    int *row4col = malloc(nc * sizeof(int));
    char *SR = malloc(nr), *SC = malloc(nc);
    int *remaining = malloc(nc * sizeof(int));
    int rc = 0;
    for (int i = 0; i < nr; i++) col4row[i] = -1;
    for (int j = 0; j < nc; j++) { row4col[j] = -1; path[j] = -1; }

    for (int cur = 0; cur < nr && rc == 0; cur++) {
        double minVal = 0;
        int num_remaining = nc;
        for (int it = 0; it < nc; it++) remaining[it] = nc - it - 1;
        memset(SR, 0, nr);
        memset(SC, 0, nc);
        for (int j = 0; j < nc; j++) spc[j] = INFINITY;
        int sink = -1, i = cur;
        while (sink == -1) {
            int index = -1;
            double lowest = INFINITY;
            SR[i] = 1;
            for (int it = 0; it < num_remaining; it++) {
                int j = remaining[it];
                double r = minVal + cost[(size_t)i * nc + j] - u[i] - v[j];
                if (r < spc[j]) { path[j] = i; spc[j] = r; }
                if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) {
                    lowest = spc[j];
                    index = it;
                }
            }
            minVal = lowest;
            if (minVal == INFINITY) { rc = -1; break; }
            int j = remaining[index];
            if (row4col[j] == -1) sink = j; else i = row4col[j];
            SC[j] = 1;
            remaining[index] = remaining[--num_remaining];
        }
        if (rc) break;
        u[cur] += minVal;
        for (int r = 0; r < nr; r++)
            if (SR[r] && r != cur) u[r] += minVal - spc[col4row[r]];
        for (int j = 0; j < nc; j++)
            if (SC[j]) v[j] -= minVal - spc[j];
        int j = sink;
        for (;;) {
            int r = path[j];
            row4col[j] = r;
            int t = col4row[r]; col4row[r] = j; j = t;
            if (r == cur) break;
        }
    }
    if (rc == 0) for (int i = 0; i < nr; i++) col4row_out[i] = col4row[i];
    free(u); free(v); free(spc); free(path); free(col4row); free(row4col);
    free(SR); free(SC); free(remaining);
    return rc;
}

/* cost: nr x nc row-major fp32 (as the reference passes it); rows/cols: min(nr,nc)
 * int64 each, sorted by row, exactly like scipy's return value. */
int lsap_oracle_f32(int nr, int nc, const float *cost, int64_t *rows, int64_t *cols)
{
    if (nr == 0 || nc == 0) return 0;
    int transpose = nc < nr;
    int R = transpose ? nc : nr, C = transpose ? nr : nc;
    double *d = malloc((size_t)R * C * sizeof(double));
    for (int i = 0; i < nr; i++)
        for (int j = 0; j < nc; j++) {
            double x = (double)cost[(size_t)i * nc + j];
            if (x != x || x == -INFINITY) { free(d); return -2; }
            if (transpose) d[(size_t)j * nr + i] = x; else d[(size_t)i * nc + j] = x;
        }
    int64_t *c4r = malloc(R * sizeof(int64_t));
    int rc = solve_wide(R, C, d, c4r);
    if (rc == 0) {
        if (!transpose) {
            for (int i = 0; i < R; i++) { rows[i] = i; cols[i] = c4r[i]; }
        } else {
            /* c4r[j] = original row matched to original column j; emit sorted by row */
            int k = 0;
            for (int r = 0; r < nr; r++)
                for (int j = 0; j < R; j++)
                    if (c4r[j] == r) { rows[k] = r; cols[k] = j; k++; }
        }
    }
    free(c4r); free(d);
    return rc;
}

/* batched form used by the cpu_baseline leg: mask[b, rows, cols] = 1 as in
 * losses_and_metrics.py:234-245 */
int lsap_oracle_mask_f32(int B, int M, int N, const float *cost, const int32_t *num_objects, float *mask)
{
    int64_t *rows = malloc((M > N ? M : N) * sizeof(int64_t));
    int64_t *cols = malloc((M > N ? M : N) * sizeof(int64_t));
    memset(mask, 0, (size_t)B * M * N * sizeof(float));
    int rc = 0;
    for (int b = 0; b < B && rc == 0; b++) {
        int n = num_objects[b];
        rc = lsap_oracle_f32(n, N, cost + (size_t)b * M * N, rows, cols);
        int k = n < N ? n : N;
        for (int t = 0; t < k && rc == 0; t++)
            mask[(size_t)b * M * N + rows[t] * N + cols[t]] = 1.0f;
    }
    free(rows); free(cols);
    return rc;
}
