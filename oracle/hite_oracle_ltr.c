/*
 * hite_oracle_ltr.c -- TEST INFRASTRUCTURE ONLY (see hite_oracle.c header).
 *
 * CPU restatement of the LTR flank-frame vote of the vendored FiLTR:
 *   judge_left_frame_LTR   /root/reference/bin/FiLTR-main/src/Util.py:9327-9462
 *   judge_right_frame_LTR  /root/reference/bin/FiLTR-main/src/Util.py:9175-9325
 * Pinned by tests/golden/ltr_frame.json.gz (generated from those two functions).
 */
#include <stdint.h>
#include <stdlib.h>

/* m: R rows x C columns (the frames of one side, one per copy).  side 0 = left frame (start column flank-1, walk to
 * the left, tolerance 5), side 1 = right frame (start column 0, walk to the right, tolerance 20).
 * returns is_ltr (1 / 0); *boundary = new boundary column or -1. */
int orc_ltr_frame(const uint8_t *m, int R, int C, int flank, int window, int side, int *boundary) {
    *boundary = -1;
    if (R <= 1) return 1;                                   /* :9191 / :9343 single copy: cannot be judged */
    const int pos = side == 0 ? flank - 1 : 0;
    const int vthr = R / 2;                                 /* int(row_num / 2) */
    const double thr = R <= 5 ? 0.95 : (R <= 10 ? 0.9 : 0.85);
    int *cols = (int *)malloc(sizeof(int) * (C + 1));
    double *ratio = (double *)malloc(sizeof(double) * (C + 1));
    int n = 0;
    for (int c = pos; n < flank && c >= 0 && c < C; c += side == 0 ? -1 : 1) {
        int cnt[256] = {0};
        for (int r = 0; r < R; r++) cnt[m[(size_t)r * C + c]]++;
        const int gap = cnt['-'];
        if (R - gap <= 1) continue;                         /* column with <= 1 base: not looked at */
        if (gap > vthr) continue;                           /* not a valid column */
        double mx = 0.0;
        for (int b = 0; b < 256; b++) {
            if (b == '-' || !cnt[b]) continue;
            double q = (double)cnt[b] / (double)R;          /* the ratio is over ALL rows (:9255) */
            if (q > mx) mx = q;
        }
        cols[n] = c; ratio[n] = mx; n++;
    }
    /* windows over the list in REVERSED visiting order (:9296 / :9437) */
    int b = -1;
    for (int i = 0; i + window <= n; i++) {
        double sum = 0.0;
        int first = -1;
        for (int k = 0; k < window; k++) {
            const int idx = n - 1 - (i + k);
            if (ratio[idx] >= thr - 0.1 && first == -1) first = cols[idx];
            sum += ratio[idx];
        }
        if (sum / (double)window >= thr) { b = first; break; }
    }
    free(cols); free(ratio);
    *boundary = b;
    const int tol = side == 0 ? 5 : 20;
    if (b != -1 && abs(b - pos) > tol) return 0;
    return 1;
}
