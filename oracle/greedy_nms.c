/* Oracle (TEST INFRASTRUCTURE ONLY): the greedy loop of non_maximum_suppression_3d
 * (reference cet_pick/models/decode.py:57-77 == cet_pick/utils/image.py:57-77).
 * The reference keeps suppressed flat indices in a Python set; only indices in [0,n) are ever
 * queried, so a byte mask over [0,n) is the same predicate. */
long greedy_nms3d_ref(const double *A, const long *order, long n, const long *deltas, long nd,
                      double threshold, unsigned char *supp, long *picks) {
    long j = 0;
    for (long t = 0; t < n; ++t) {
        long i = order[t];
        if (A[i] <= threshold) break;
        if (!supp[i]) {
            picks[j++] = i;
            for (long q = 0; q < nd; ++q) {
                long k = i + deltas[q];
                if (k >= 0 && k < n) supp[k] = 1;
            }
        }
    }
    return j;
}
