/* ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain-C restatement of the exact answer the reference index returns when its
 * graph search is exhaustive (reference src/indexes/hnsw.py, ef_search >= N;
 * SURVEY.md §8a "What 'same result' means for K-rows"):
 *
 *   stored row   x_i = v_i / ||v_i||                     hnsw.py:157  (add)
 *   query        q   = q  / ||q||                        hnsw.py:250/499
 *   distance     d_i = fp32(1.0 - fp32(dot(x_i, q)))     hnsw.py:59-66
 *   result       k smallest d_i, ordered by (d_i, id_i)  hnsw.py:269/518
 *   score        fp32(1.0 - d_i)                         hnsw.py:273/522
 *
 * numpy's fp32 dot uses a BLAS-dependent summation order that cannot be
 * reproduced on another machine; this oracle fixes the order instead:
 * dot = (float) sum_{i=0..D-1, in order} (double)x[i]*(double)q[i]
 * (each product is exact in fp64; one fp64 rounding per add).  The HIP path
 * uses the same chain, so ids AND distances are bit-exact between the two.
 * Against numpy the distances agree to ~1e-7 and the id lists agree wherever
 * neighbouring distances differ by more than that (checked on the golden
 * fixtures, tests/test_oracle_golden.py).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load the library built from this file.  Build: make -C oracle
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static float dot_f64chain(const float *a, const float *b, int dim) {
    double acc = 0.0;
    for (int i = 0; i < dim; ++i) acc += (double)a[i] * (double)b[i];
    return (float)acc;
}

/* In-place row normalisation with the same fixed-order definition the HIP
 * path uses for device-resident rows: n2 = fp64 chain of x*x,
 * x_i <- fp32(x_i / fp32(sqrt(n2))).  (Host-side callers of the product
 * normalise with numpy exactly like the reference instead.) */
void vq_oracle_normalize_rows(float *x, int64_t n, int dim) {
    for (int64_t r = 0; r < n; ++r) {
        float *row = x + r * (int64_t)dim;
        double acc = 0.0;
        for (int i = 0; i < dim; ++i) acc += (double)row[i] * (double)row[i];
        float nrm = (float)sqrt(acc);
        for (int i = 0; i < dim; ++i) row[i] = row[i] / nrm;
    }
}

/* distances of every row to one query: d[i] = 1.0f - dot */
void vq_oracle_distances(const float *rows, int64_t n, int dim, const float *q, float *d) {
    for (int64_t r = 0; r < n; ++r) d[r] = 1.0f - dot_f64chain(rows + r * (int64_t)dim, q, dim);
}

typedef struct { float d; int32_t id; } cand_t;

static int cand_less(cand_t a, cand_t b) { return a.d < b.d || (a.d == b.d && a.id < b.id); }

/* Exact top-k for nq queries.  ids/dist are [nq,k]; slots past min(k,n) get
 * id -1 / dist +inf.  Row numbers are the ids (the Python side maps them to
 * caller ids and re-applies the (distance,id) order when ids are not rows). */
void vq_oracle_topk(const float *rows, int64_t n, int dim, const float *queries, int nq, int k,
                    int32_t *ids, float *dist) {
    /* queries are independent: OpenMP over queries when built with -fopenmp
     * (the cpu_baseline leg reports the thread count it used) */
#pragma omp parallel for schedule(dynamic, 1)
    for (int qi = 0; qi < nq; ++qi) {
        cand_t *best = (cand_t *)malloc(sizeof(cand_t) * (size_t)(k > 0 ? k : 1));
        const float *q = queries + (int64_t)qi * dim;
        int have = 0;
        for (int64_t r = 0; r < n; ++r) {
            cand_t c = { 1.0f - dot_f64chain(rows + r * (int64_t)dim, q, dim), (int32_t)r };
            if (have == k && !cand_less(c, best[k - 1])) continue;
            int pos = have < k ? have++ : k - 1;          /* insertion into the sorted prefix */
            while (pos > 0 && cand_less(c, best[pos - 1])) { best[pos] = best[pos - 1]; --pos; }
            best[pos] = c;
        }
        for (int j = 0; j < k; ++j) {
            ids[(int64_t)qi * k + j] = j < have ? best[j].id : -1;
            dist[(int64_t)qi * k + j] = j < have ? best[j].d : INFINITY;
        }
        free(best);
    }
}
