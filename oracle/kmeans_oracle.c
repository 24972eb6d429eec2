/*
 * oracle/kmeans_oracle.c — TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement (plain C, FP64) of what the reference's from_msa path obtains from
 *     KMeans(n_clusters=k, random_state=2, algorithm="elkan").fit(X).predict(X)
 * (call site: /root/reference/make_prg/from_msa/cluster_sequences.py:262-266), i.e. of scikit-learn's
 * KMeans (third-party dependency of the reference; pinned scikit-learn 1.3.0 in /root/reference/poetry.lock:1115,
 * restated from the installed 1.7.2 sources: sklearn/cluster/_kmeans.py, _k_means_elkan.pyx, _k_means_common.pyx,
 * _k_means_lloyd.pyx, sklearn/metrics/pairwise.py, sklearn/utils/extmath.py) with the parity settings of
 * SURVEY.md §0.4/§0.6:
 *     n_init = 10, OMP_NUM_THREADS = 1, OPENBLAS_CORETYPE = Haswell.
 *
 * The reference's result depends on the floating-point summation ORDER of the BLAS / NumPy kernels underneath
 * scikit-learn (exact ties are routine on integer k-mer counts).  Every reduction below therefore
 * restates the order of the kernel the pinned configuration executes; each was established by black-box bit
 * comparison against NumPy 2.2.6 / OpenBLAS 0.3.29 (Haswell kernels) in the build container:
 *   - dgemm  (OpenBLAS Haswell): per C element one FMA chain over k inside a K-block; K-blocks of GEMM_Q=256 with the
 *            level-3 driver's split rule; blocks added into C in order; the 4x1 micro-kernel (4-row block x last odd
 *            column) keeps 4 accumulators round-robin over the 8-unrolled part.              -> gemm_dot()
 *   - dsyrk  (numpy turns A @ A.T into syrk): same chain, K split rule (min_l+1)/2.          -> syrk_dot()
 *   - dgemv_t (matrix @ vector): blocks of 2048 rows; 4-column groups: 4 FMA lanes (l0+l2)+(l1+l3); then a 2-column
 *            SSE2 kernel (2 non-fused lanes), then a 1-column kernel (4 non-fused lanes); m%4 tail.  -> gemv_t_col()
 *   - einsum("ij,ij->i") (row_norms): 2 SSE2 lanes, 8-element chunks accumulated vec3,vec2,vec1,vec0, non-fused.
 *   - np.add.reduce on a contiguous axis: pairwise summation (8 accumulators, blocks of 128).
 *   - reductions over axis 0 and the Cython loops: plain sequential order.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file's shared object.
 *
 * Build: gcc -O2 -mfma -ffp-contract=off -shared -fPIC oracle/kmeans_oracle.c -o oracle/_build/libkmeans_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ MT19937 (numpy.random.RandomState legacy) */
typedef struct { uint32_t mt[624]; int idx; } mt_t;
static void mt_seed(mt_t *s, uint32_t seed) {           /* numpy _legacy_seeding(int) -> init_genrand */
  s->mt[0] = seed;
  for (int i = 1; i < 624; i++) s->mt[i] = 1812433253u * (s->mt[i - 1] ^ (s->mt[i - 1] >> 30)) + (uint32_t)i;
  s->idx = 624;
}
static uint32_t mt_next(mt_t *s) {
  if (s->idx >= 624) {
    for (int i = 0; i < 624; i++) {
      uint32_t y = (s->mt[i] & 0x80000000u) | (s->mt[(i + 1) % 624] & 0x7fffffffu);
      s->mt[i] = s->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    s->idx = 0;
  }
  uint32_t y = s->mt[s->idx++];
  y ^= y >> 11; y ^= (y << 7) & 0x9d2c5680u; y ^= (y << 15) & 0xefc60000u; y ^= y >> 18;
  return y;
}
static double mt_double(mt_t *s) {                       /* random_sample(): 53-bit double */
  uint32_t a = mt_next(s) >> 5, b = mt_next(s) >> 6;
  return (a * 67108864.0 + b) / 9007199254740992.0;
}
/* exported for tests: fill out[n] with RandomState(seed).random_sample(n) */
void mprg_oracle_random_sample(uint32_t seed, int n, double *out) {
  mt_t s; mt_seed(&s, seed);
  for (int i = 0; i < n; i++) out[i] = mt_double(&s);
}

/* ------------------------------------------------------------------ NumPy reductions */
/* numpy/_core/src/umath/loops_utils.h.src DOUBLE_pairwise_sum (contiguous) */
static double pairwise_sum(const double *a, long n) {
  if (n < 8) {
    double res = 0.;
    for (long i = 0; i < n; i++) res += a[i];
    return res;
  } else if (n <= 128) {
    double r[8];
    for (int j = 0; j < 8; j++) r[j] = a[j];
    long i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int j = 0; j < 8; j++) r[j] += a[i + j];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
  } else {
    long n2 = n / 2;
    n2 -= n2 % 8;
    return pairwise_sum(a, n2) + pairwise_sum(a + n2, n - n2);
  }
}
double mprg_oracle_pairwise_sum(const double *a, long n) { return pairwise_sum(a, n); }

/* np.einsum("ij,ij->i") inner kernel double_sum_of_products_contig_contig_outstride0_two, SSE2 baseline (2 lanes) */
static double einsum_dot(const double *a, const double *b, int n) {
  double l0 = 0, l1 = 0;
  int i = 0;
  for (; n - i >= 8; i += 8)
    for (int v = 3; v >= 0; v--) {
      l0 = a[i + 2 * v] * b[i + 2 * v] + l0;
      l1 = a[i + 2 * v + 1] * b[i + 2 * v + 1] + l1;
    }
  for (; i < n; i += 2) {
    l0 = a[i] * b[i] + l0;
    if (i + 1 < n) l1 = a[i + 1] * b[i + 1] + l1;
  }
  return l0 + l1;
}
double mprg_oracle_einsum_dot(const double *a, const double *b, int n) { return einsum_dot(a, b, n); }

/* ------------------------------------------------------------------ OpenBLAS (Haswell) summation orders */
static double chain_fma(const double *a, const double *b, int k0, int k1) {
  double s = 0;
  for (int k = k0; k < k1; k++) s = fma(a[k], b[k], s);
  return s;
}
/* dgemm 4x1 micro-kernel: 4 accumulators over the 8-unrolled part, remainder into acc0, (a0+a1)+(a2+a3) */
static double chain_fma_4acc(const double *a, const double *b, int k0, int k1) {
  double acc[4] = {0, 0, 0, 0};
  int K = k1 - k0, main = (K / 8) * 8;
  for (int k = 0; k < main; k++) acc[k & 3] = fma(a[k0 + k], b[k0 + k], acc[k & 3]);
  for (int k = main; k < K; k++) acc[0] = fma(a[k0 + k], b[k0 + k], acc[0]);
  return (acc[0] + acc[1]) + (acc[2] + acc[3]);
}
/* One C element of dgemm: dot(a,b) over K. `m`/`M` index/extent along OpenBLAS' M (unroll 4), `n`/`N` along its N.
 * c0 is the value already in C (beta-scaled), alpha multiplies each block's accumulator (exact for +-1, +-2). */
static double gemm_dot(const double *a, const double *b, int K, int m, int M, int n, int N, double c0, double alpha) {
  int in_4row_block = m < (M & ~3);
  int last_odd_col = (N & 1) && n == N - 1;
  int use4 = in_4row_block && last_odd_col;
  double c = c0;
  for (int ls = 0; ls < K;) {
    int min_l = K - ls;
    if (min_l >= 512) min_l = 256;
    else if (min_l > 256) min_l = ((min_l / 2 + 3) / 4) * 4;
    double acc = use4 ? chain_fma_4acc(a, b, ls, ls + min_l) : chain_fma(a, b, ls, ls + min_l);
    c = c + alpha * acc;
    ls += min_l;
  }
  return c;
}
double mprg_oracle_gemm_dot(const double *a, const double *b, int K, int m, int M, int n, int N, double c0, double alpha) {
  return gemm_dot(a, b, K, m, M, n, N, c0, alpha);
}
static double syrk_dot(const double *a, const double *b, int K) {
  double c = 0;
  for (int ls = 0; ls < K;) {
    int min_l = K - ls;
    if (min_l >= 512) min_l = 256;
    else if (min_l > 256) min_l = (min_l + 1) / 2;
    c = c + chain_fma(a, b, ls, ls + min_l);
    ls += min_l;
  }
  return c;
}
double mprg_oracle_syrk_dot(const double *a, const double *b, int K) { return syrk_dot(a, b, K); }
/* y[j] of dgemv_t: y = A x, A row-major (n rows of length m); j = output index, n = number of outputs */
static double gemv_t_col(const double *a, const double *x, int m, int j, int n) {
  const int NBMAX = 2048;
  int kind;                                         /* 0: 4x4 AVX2+FMA kernel, 1: 4x2 SSE2 kernel, 2: 4x1 kernel */
  int n1 = n & ~3;
  if (j < n1) kind = 0;
  else if ((n & 2) && j < n1 + 2) kind = 1;
  else kind = 2;
  int m3 = m & 3, m1 = m & -4, m2 = (m & (NBMAX - 1)) - m3, NB = NBMAX;
  double y = 0;
  const double *ap = a, *xp = x;
  while (NB == NBMAX) {
    m1 -= NB;
    if (m1 < 0) { if (m2 == 0) break; NB = m2; }
    double yb;
    if (kind == 0) {
      double l[4] = {0, 0, 0, 0};
      for (int i = 0; i < NB; i++) l[i & 3] = fma(ap[i], xp[i], l[i & 3]);
      yb = (l[0] + l[2]) + (l[1] + l[3]);
    } else if (kind == 1) {
      double l[2] = {0, 0};
      for (int i = 0; i < NB; i++) l[i & 1] = l[i & 1] + ap[i] * xp[i];
      yb = l[0] + l[1];
    } else {
      double l[4] = {0, 0, 0, 0};
      for (int i = 0; i < NB; i++) l[i & 3] = l[i & 3] + ap[i] * xp[i];
      yb = (l[0] + l[2]) + (l[1] + l[3]);
    }
    y = y + yb;
    ap += NB; xp += NB;
  }
  if (m3 == 0) return y;
  const double *aj = a + (m - m3), *xj = x + (m - m3);
  if (m3 == 3) y = y + fma(aj[2], xj[2], fma(aj[0], xj[0], aj[1] * xj[1]));
  else if (m3 == 2) y = y + fma(aj[0], xj[0], aj[1] * xj[1]);
  else y = fma(aj[0], xj[0], y);
  return y;
}
double mprg_oracle_gemv_t_col(const double *a, const double *x, int m, int j, int n) { return gemv_t_col(a, x, m, j, n); }

/* ------------------------------------------------------------------ sklearn pieces */
/* _k_means_common.pyx:16-43 */
static double euclid(const double *a, const double *b, int nf, int squared) {
  int n = nf / 4, rem = nf % 4;
  double result = 0;
  for (int i = 0; i < n; i++) {
    result += ((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]) +
               (a[3] - b[3]) * (a[3] - b[3]));
    a += 4; b += 4;
  }
  for (int i = 0; i < rem; i++) result += (a[i] - b[i]) * (a[i] - b[i]);
  return squared ? result : sqrt(result);
}

typedef struct {
  int D, V, k;
  const double *X;        /* centred, D x V */
  const double *xsq;      /* row norms of X */
  double tol;
} km_t;

/* metrics/pairwise.py euclidean_distances(centers)/2 and np.partition(...,1,axis=0)[1] (_kmeans.py:521-524) */
static void center_half_dists(const km_t *p, const double *C, double *chd, double *dnext) {
  int k = p->k, V = p->V;
  double csq[16];
  for (int i = 0; i < k; i++) csq[i] = einsum_dot(C + (size_t)i * V, C + (size_t)i * V, V);
  for (int i = 0; i < k; i++)
    for (int j = 0; j < k; j++) {
      double d;
      if (i == j) d = 0.0;
      else {
        /* syrk computes one triangle; numpy mirrors it: use (min,max) ordering so both halves agree bitwise */
        int lo = i < j ? i : j, hi = i < j ? j : i;
        double dot = syrk_dot(C + (size_t)hi * V, C + (size_t)lo * V, V);
        d = -2 * dot;
        d += csq[i];
        d += csq[j];
        if (!(d > 0)) d = 0;  /* np.maximum(d, 0) */
        d = sqrt(d);
      }
      chd[i * k + j] = d / 2;
    }
  for (int j = 0; j < k; j++) {  /* second smallest of column j */
    double m1 = INFINITY, m2 = INFINITY;
    for (int i = 0; i < k; i++) {
      double v = chd[i * k + j];
      if (v < m1) { m2 = m1; m1 = v; } else if (v < m2) m2 = v;
    }
    dnext[j] = m2;
  }
}

/* _kmeans.py:174-272 _kmeans_plusplus; indices_out[k] */
static void kmeanspp(const km_t *p, mt_t *rs, double *centers, int *indices_out, double *work) {
  int D = p->D, V = p->V, k = p->k;
  const double *X = p->X;
  int n_local_trials = 2 + (int)log((double)k);
  double *closest = work;            /* D */
  double *cum = work + D;            /* D */
  double *dist = work + 2 * D;       /* 4*D */
  /* random_state.choice(n, p=w/w.sum()): cdf = cumsum(1/D); cdf /= cdf[-1]; searchsorted(u, 'right') */
  double pval = 1.0 / (double)D, c = 0;
  for (int i = 0; i < D; i++) { c += pval; cum[i] = c; }
  double last = cum[D - 1];
  for (int i = 0; i < D; i++) cum[i] /= last;
  double u = mt_double(rs);
  int center_id = 0;
  { int lo = 0, hi = D; while (lo < hi) { int mid = (lo + hi) / 2; if (cum[mid] <= u) lo = mid + 1; else hi = mid; } center_id = lo; }
  if (center_id > D - 1) center_id = D - 1;
  memcpy(centers, X + (size_t)center_id * V, sizeof(double) * V);
  indices_out[0] = center_id;
  /* closest_dist_sq = _euclidean_distances(centers[0:1], X): (1xV)@(VxD) -> dgemv_t over the D rows of X */
  for (int j = 0; j < D; j++) {
    double dot = gemv_t_col(X + (size_t)j * V, centers, V, j, D);
    double d = -2 * dot;
    d += p->xsq[center_id];
    d += p->xsq[j];
    closest[j] = d > 0 ? d : 0;
  }
  /* current_pot = closest_dist_sq @ sample_weight : (1,D)@(D,) */
  double current_pot;
  {
    double ones_stack[1];
    (void)ones_stack;
    double *ones = dist;  /* borrow */
    for (int j = 0; j < D; j++) ones[j] = 1.0;
    current_pot = gemv_t_col(closest, ones, D, 0, 1);
  }
  for (int cidx = 1; cidx < k; cidx++) {
    double rv[8];
    for (int t = 0; t < n_local_trials; t++) rv[t] = mt_double(rs) * current_pot;
    double s = 0;
    for (int j = 0; j < D; j++) { s += 1.0 * closest[j]; cum[j] = s; }
    int cand[8];
    for (int t = 0; t < n_local_trials; t++) {
      int lo = 0, hi = D;
      while (lo < hi) { int mid = (lo + hi) / 2; if (cum[mid] < rv[t]) lo = mid + 1; else hi = mid; }
      cand[t] = lo > D - 1 ? D - 1 : lo;
    }
    int T = n_local_trials;
    /* distance_to_candidates = _euclidean_distances(X[cand], X): (T x V) @ (V x D) -> dgemm, OpenBLAS M = D, N = T */
    for (int t = 0; t < T; t++) {
      const double *xc = X + (size_t)cand[t] * V;
      for (int j = 0; j < D; j++) {
        double dot = gemm_dot(X + (size_t)j * V, xc, V, j, D, t, T, 0.0, 1.0);
        double d = -2 * dot;
        d += p->xsq[cand[t]];
        d += p->xsq[j];
        d = d > 0 ? d : 0;
        dist[(size_t)t * D + j] = closest[j] < d ? closest[j] : d;  /* np.minimum(closest, d) */
      }
    }
    /* candidates_pot = dist @ ones(D,1): dgemv_t with n = T outputs, m = D */
    double *ones = cum;  /* cum no longer needed this round */
    for (int j = 0; j < D; j++) ones[j] = 1.0;
    int best = 0;
    double bestpot = 0;
    for (int t = 0; t < T; t++) {
      double pot = gemv_t_col(dist + (size_t)t * D, ones, D, t, T);
      if (t == 0 || pot < bestpot) { best = t; bestpot = pot; }
    }
    current_pot = bestpot;
    memcpy(closest, dist + (size_t)best * D, sizeof(double) * D);
    memcpy(centers + (size_t)cidx * V, X + (size_t)cand[best] * V, sizeof(double) * V);
    indices_out[cidx] = cand[best];
  }
}

/* NumPy arg-introselect (np.argpartition, float64, no NaNs): numpy/_core/src/npysort/selection.cpp.
 * perm is permuted so that perm[kth] is in sorted position, smaller before, not-smaller after. */
static void np_argpartition(const double *v, long *ts, long num, long kth) {
#define LT(a, b) ((a) < (b))
#define SWAPL(a, b) do { long t_ = (a); (a) = (b); (b) = t_; } while (0)
  long low = 0, high = num - 1;
  if (kth - low < 3) {                 /* dumb_select: O(n*kth) selection of the kth smallest */
    for (long i = 0; i <= kth - low; i++) {
      long minidx = i; double minval = v[ts[low + i]];
      for (long k = i + 1; k < high - low + 1; k++)
        if (LT(v[ts[low + k]], minval)) { minidx = k; minval = v[ts[low + k]]; }
      SWAPL(ts[low + i], ts[low + minidx]);
    }
    return;
  }
  if (kth == num - 1) {                /* maximum to the end; the LAST of equal maxima wins */
    long maxidx = low; double maxval = v[ts[low]];
    for (long k = low + 1; k < num; k++)
      if (!LT(v[ts[k]], maxval)) { maxidx = k; maxval = v[ts[k]]; }
    SWAPL(ts[kth], ts[maxidx]);
    return;
  }
  int depth_limit = 0;
  { unsigned long n = (unsigned long)num; while (n >>= 1) depth_limit++; depth_limit *= 2; }
  for (; low + 1 < high;) {
    long ll = low + 1, hh = high;
    if (depth_limit > 0 || hh - ll < 5) {
      const long mid = low + (high - low) / 2;           /* median of 3, pivot to low, 3-lowest to low+1 */
      if (LT(v[ts[high]], v[ts[mid]])) SWAPL(ts[high], ts[mid]);
      if (LT(v[ts[high]], v[ts[low]])) SWAPL(ts[high], ts[low]);
      if (LT(v[ts[low]], v[ts[mid]])) SWAPL(ts[low], ts[mid]);
      SWAPL(ts[mid], ts[low + 1]);
    } else {                                             /* median of medians of 5 (worst-case guard) */
      long nmed = (hh - ll) / 5;
      for (long i = 0, sub = ll; i < nmed; i++, sub += 5) {
        long *t5 = ts + sub; const double *vv = v;
        /* median5: sorting network on indices */
        if (LT(vv[t5[1]], vv[t5[0]])) SWAPL(t5[1], t5[0]);
        if (LT(vv[t5[4]], vv[t5[3]])) SWAPL(t5[4], t5[3]);
        if (LT(vv[t5[3]], vv[t5[0]])) SWAPL(t5[3], t5[0]);
        if (LT(vv[t5[4]], vv[t5[1]])) SWAPL(t5[4], t5[1]);
        if (LT(vv[t5[2]], vv[t5[1]])) SWAPL(t5[2], t5[1]);
        long m;
        if (LT(vv[t5[3]], vv[t5[2]])) m = LT(vv[t5[3]], vv[t5[1]]) ? 1 : 3; else m = 2;
        SWAPL(ts[sub + m], ts[ll + i]);
      }
      if (nmed > 2) np_argpartition(v, ts + ll, nmed, nmed / 2);
      SWAPL(ts[ll + nmed / 2], ts[low]);
      ll = low; hh = high + 1;
    }
    depth_limit--;
    const double pivot = v[ts[low]];
    for (;;) {                                           /* unguarded partition */
      do ll++; while (LT(v[ts[ll]], pivot));
      do hh--; while (LT(pivot, v[ts[hh]]));
      if (hh < ll) break;
      SWAPL(ts[hh], ts[ll]);
    }
    SWAPL(ts[low], ts[hh]);
    if (hh >= kth) high = hh - 1;
    if (hh <= kth) low = ll;
  }
  if (high == low + 1 && LT(v[ts[high]], v[ts[low]])) SWAPL(ts[high], ts[low]);
#undef LT
#undef SWAPL
}

/* _k_means_elkan.pyx:186-427 with n_threads = 1 */
static int elkan_iter(const km_t *p, const double *cold, double *cnew, double *wic, const double *chd,
                      const double *dnext, double *ub, double *lb, int *labels, double *cshift, int update) {
  int D = p->D, V = p->V, k = p->k;
  const double *X = p->X;
  int relocated = 0;
  if (update) { memset(cnew, 0, sizeof(double) * k * V); memset(wic, 0, sizeof(double) * k); }
  for (int i = 0; i < D; i++) {
    double upper = ub[i];
    int tight = 0, label = labels[i];
    if (!(dnext[label] >= upper)) {
      for (int j = 0; j < k; j++) {
        if (j != label && upper > lb[i * k + j] && upper > chd[label * k + j]) {
          if (!tight) {
            upper = euclid(X + (size_t)i * V, cold + (size_t)label * V, V, 0);
            lb[i * k + label] = upper;
            tight = 1;
          }
          if (upper > lb[i * k + j] || upper > chd[label * k + j]) {
            double d = euclid(X + (size_t)i * V, cold + (size_t)j * V, V, 0);
            lb[i * k + j] = d;
            if (d < upper) { label = j; upper = d; }
          }
        }
      }
      labels[i] = label;
      ub[i] = upper;
    }
    if (update) {
      wic[label] += 1.0;
      double *cn = cnew + (size_t)label * V;
      const double *xi = X + (size_t)i * V;
      for (int f = 0; f < V; f++) cn[f] += xi[f] * 1.0;
    }
  }
  if (update) {
    /* _relocate_empty_clusters_dense (_k_means_common.pyx:167-211) */
    int n_empty = 0, empty[16];
    for (int j = 0; j < k; j++) if (wic[j] == 0) empty[n_empty++] = j;
    if (n_empty > 0) {
      relocated = 1;
      /* distances = ((X - centers_old[labels])**2).sum(axis=1): pairwise sum per row */
      double *dist = (double *)malloc(sizeof(double) * D);
      double *row = (double *)malloc(sizeof(double) * V);
      double dmax = 0;
      for (int i = 0; i < D; i++) {
        const double *xi = X + (size_t)i * V, *cj = cold + (size_t)labels[i] * V;
        for (int f = 0; f < V; f++) { double t = xi[f] - cj[f]; row[f] = t * t; }
        dist[i] = pairwise_sum(row, V);
        if (i == 0 || dist[i] > dmax) dmax = dist[i];
      }
      if (dmax != 0) {
        /* far_from_centers = np.argpartition(dist, -n_empty)[:-n_empty-1:-1]: NumPy's arg-introselect
         * (numpy/_core/src/npysort/selection.cpp) restated below; the tail of its permutation, reversed. */
        long *perm = (long *)malloc(sizeof(long) * D);
        for (int i = 0; i < D; i++) perm[i] = i;
        np_argpartition(dist, perm, D, D - n_empty);
        char *used = (char *)calloc(D, 1);
        for (int e = 0; e < n_empty; e++) {
          int far = (int)perm[D - 1 - e];
          int newc = empty[e], oldc = labels[far];
          const double *xf = X + (size_t)far * V;
          for (int f = 0; f < V; f++) {
            cnew[(size_t)oldc * V + f] -= xf[f] * 1.0;
            cnew[(size_t)newc * V + f] = xf[f] * 1.0;
          }
          wic[newc] = 1.0;
          wic[oldc] -= 1.0;
        }
        free(perm);
        free(used);
      }
      free(dist); free(row);
    }
    /* _average_centers (:274-295) */
    int argmax = 0;
    for (int j = 1; j < k; j++) if (wic[j] > wic[argmax]) argmax = j;
    for (int j = 0; j < k; j++) {
      if (wic[j] > 0) {
        double alpha = 1.0 / wic[j];
        for (int f = 0; f < V; f++) cnew[(size_t)j * V + f] *= alpha;
      } else {
        for (int f = 0; f < V; f++) cnew[(size_t)j * V + f] = cnew[(size_t)argmax * V + f];
      }
    }
    /* _center_shift (:298-311) */
    for (int j = 0; j < k; j++) cshift[j] = euclid(cnew + (size_t)j * V, cold + (size_t)j * V, V, 0);
    for (int i = 0; i < D; i++) {
      ub[i] += cshift[labels[i]];
      for (int j = 0; j < k; j++) {
        lb[i * k + j] -= cshift[j];
        if (lb[i * k + j] < 0) lb[i * k + j] = 0;
      }
    }
  }
  return relocated;
}

/* _kmeans.py:456-618 _kmeans_single_elkan */
static double single_elkan(const km_t *p, double *centers /* in: init, out: final */, int *labels, int *n_iter,
                           int *flags) {
  int D = p->D, V = p->V, k = p->k;
  const double *X = p->X;
  double *cnew = (double *)calloc((size_t)k * V, sizeof(double));
  double *ub = (double *)calloc(D, sizeof(double));
  double *lb = (double *)calloc((size_t)D * k, sizeof(double));
  int *labels_old = (int *)malloc(sizeof(int) * D);
  double wic[16], chd[256], dnext[16], cshift[16];
  for (int i = 0; i < D; i++) { labels[i] = -1; labels_old[i] = -1; }
  memset(cshift, 0, sizeof cshift);
  double *cur = centers, *nxt = cnew;
  center_half_dists(p, cur, chd, dnext);
  /* init_bounds_dense (_k_means_elkan.pyx:24-97) */
  for (int i = 0; i < D; i++) {
    int best = 0;
    double mind = euclid(X + (size_t)i * V, cur, V, 0);
    lb[i * k] = mind;
    for (int j = 1; j < k; j++) {
      if (mind > chd[best * k + j]) {
        double d = euclid(X + (size_t)i * V, cur + (size_t)j * V, V, 0);
        lb[i * k + j] = d;
        if (d < mind) { mind = d; best = j; }
      }
    }
    labels[i] = best;
    ub[i] = mind;
  }
  int strict = 0, it = 0;
  for (it = 0; it < 300; it++) {
    if (elkan_iter(p, cur, nxt, wic, chd, dnext, ub, lb, labels, cshift, 1)) *flags |= 1;
    center_half_dists(p, nxt, chd, dnext);
    { double *t = cur; cur = nxt; nxt = t; }
    if (memcmp(labels, labels_old, sizeof(int) * D) == 0) { strict = 1; break; }
    double sq[16];
    for (int j = 0; j < k; j++) sq[j] = cshift[j] * cshift[j];
    double tot = pairwise_sum(sq, k);
    if (tot <= p->tol) break;
    memcpy(labels_old, labels, sizeof(int) * D);
  }
  int iters = it < 300 ? it + 1 : 300;
  if (!strict) elkan_iter(p, cur, cur, wic, chd, dnext, ub, lb, labels, cshift, 0);
  /* _inertia_dense (_k_means_common.pyx:95-125) */
  double inertia = 0;
  for (int i = 0; i < D; i++) inertia += euclid(X + (size_t)i * V, cur + (size_t)labels[i] * V, V, 1) * 1.0;
  if (cur != centers) memcpy(centers, cur, sizeof(double) * k * V);
  *n_iter = iters;
  free(cnew); free(ub); free(lb); free(labels_old);
  return inertia;
}

/* _k_means_common.pyx:314-328 */
/* exported for the pinning test of the selection alone (tests/test_argpartition.py): perm[0..num) = 0..num-1 on entry */
void mprg_oracle_argpartition(const double *v, long *perm, long num, long kth) { np_argpartition(v, perm, num, kth); }

static int same_clustering(const int *l1, const int *l2, int n, int k) {
  int map[16];
  for (int j = 0; j < k; j++) map[j] = -1;
  for (int i = 0; i < n; i++) {
    if (map[l1[i]] == -1) map[l1[i]] = l2[i];
    else if (map[l1[i]] != l2[i]) return 0;
  }
  return 1;
}

/*
 * Full fit(X).predict(X).  counts: D x V row-major (k-mer count matrix, reference cluster_sequences.py:41-56).
 * labels_out[D] = predict labels; fit_labels_out[D] (may be NULL) = labels_ of the best restart;
 * pp_indices_out[n_init*k] (may be NULL) = k-means++ centre indices of every restart;
 * info_out[4] (may be NULL) = {inertia, n_iter_best, best_restart, flags(bit0: empty-cluster relocation happened)}.
 * Returns 0, or -1 on bad arguments.
 */
int mprg_oracle_kmeans_fit_predict(const double *counts, int D, int V, int k, int n_init, uint32_t seed,
                                   int *labels_out, int *fit_labels_out, int *pp_indices_out, double *info_out) {
  if (k < 1 || k > 16 || D < k || V < 1) return -1;
  size_t DV = (size_t)D * V;
  double *X = (double *)malloc(sizeof(double) * DV);
  double *tmpV = (double *)malloc(sizeof(double) * V);
  double *mean = (double *)malloc(sizeof(double) * V);
  double *var = (double *)malloc(sizeof(double) * V);
  /* _tolerance (_kmeans.py:279-287): np.mean(np.var(X, axis=0)) * 1e-4 on the un-centred data */
  for (int f = 0; f < V; f++) { double s = 0; for (int i = 0; i < D; i++) s += counts[(size_t)i * V + f]; mean[f] = s / (double)D; }
  for (int f = 0; f < V; f++) {
    double s = 0;
    for (int i = 0; i < D; i++) { double t = counts[(size_t)i * V + f] - mean[f]; s += t * t; }
    var[f] = s / (double)D;
  }
  double tol = (pairwise_sum(var, V) / (double)V) * 1e-4;
  /* X -= X.mean(axis=0) (_kmeans.py:1477-1484) */
  for (size_t i = 0; i < (size_t)D; i++) for (int f = 0; f < V; f++) X[i * V + f] = counts[i * V + f] - mean[f];
  double *xsq = (double *)malloc(sizeof(double) * D);
  for (int i = 0; i < D; i++) xsq[i] = einsum_dot(X + (size_t)i * V, X + (size_t)i * V, V);
  km_t P = {D, V, k, X, xsq, tol};
  mt_t rs; mt_seed(&rs, seed);
  double *centers = (double *)malloc(sizeof(double) * k * V);
  double *best_centers = (double *)malloc(sizeof(double) * k * V);
  int *labels = (int *)malloc(sizeof(int) * D), *best_labels = (int *)malloc(sizeof(int) * D);
  double *work = (double *)malloc(sizeof(double) * (size_t)D * 8);
  double best_inertia = 0; int have_best = 0, best_iter = 0, best_restart = -1, flags = 0;
  int idx[16];
  for (int r = 0; r < n_init; r++) {
    kmeanspp(&P, &rs, centers, idx, work);
    if (pp_indices_out) for (int j = 0; j < k; j++) pp_indices_out[r * k + j] = idx[j];
    int n_iter = 0;
    double inertia = single_elkan(&P, centers, labels, &n_iter, &flags);
    if (!have_best || (inertia < best_inertia && !same_clustering(labels, best_labels, D, k))) {
      memcpy(best_labels, labels, sizeof(int) * D);
      memcpy(best_centers, centers, sizeof(double) * k * V);
      best_inertia = inertia; best_iter = n_iter; have_best = 1; best_restart = r;
    }
  }
  for (int j = 0; j < k; j++) for (int f = 0; f < V; f++) best_centers[(size_t)j * V + f] += mean[f];
  if (fit_labels_out) memcpy(fit_labels_out, best_labels, sizeof(int) * D);
  /* predict (_kmeans.py:1066-1098 -> _k_means_lloyd.pyx:168-218): chunks of 256 samples, scipy dgemm
   * C = -2 * X C^T + ||C||^2 ; row-major (n x k) => OpenBLAS M = k (clusters), N = chunk samples */
  double csq[16];
  for (int j = 0; j < k; j++) csq[j] = einsum_dot(best_centers + (size_t)j * V, best_centers + (size_t)j * V, V);
  int chunk = D > 256 ? 256 : D;
  for (int start = 0; start < D; start += chunk) {
    int end = start + chunk > D ? D : start + chunk;
    int ns = end - start;
    for (int i = start; i < end; i++) {
      int label = 0; double mind = 0;
      for (int j = 0; j < k; j++) {
        double d = gemm_dot(best_centers + (size_t)j * V, counts + (size_t)i * V, V, j, k, i - start, ns, csq[j], -2.0);
        if (j == 0 || d < mind) { mind = d; label = j; }
      }
      labels_out[i] = label;
    }
  }
  if (info_out) { info_out[0] = best_inertia; info_out[1] = best_iter; info_out[2] = best_restart; info_out[3] = flags; }
  free(X); free(tmpV); free(mean); free(var); free(xsq); free(centers); free(best_centers); free(labels);
  free(best_labels); free(work);
  return 0;
}
