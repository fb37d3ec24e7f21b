/*
 * saccot_oracle.c — CPU restatement of the SAC-COT compatibility-triangle hot path.
 *
 * *** TEST INFRASTRUCTURE ONLY. ***  Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline`
 * leg may load this.  The product library (libsaccot.so) never links, loads or calls it.
 *
 * *** PARITY UNPINNED. ***  The reference tree is /root/reference/README.md:1-2 (a title and one sentence
 * naming the paper); it holds no implementation, no golden vectors and no tests, so nothing of the
 * reference pins this restatement.  It follows, in order: README.md:2 (the algorithm's name),
 * BASELINE.json `north_star` (stage list, I/O shape, tolerances) and SURVEY.md §8(a) (canonical fp32
 * arithmetic and tie-breaks — build decisions, restated in DESIGN.md §3).  It is cross-checked by an
 * independent fp64 numpy restatement (oracle/saccot_fp64.py) and by synthetic ground truth.
 *
 * Arithmetic rules (they make GPU == CPU bit-for-bit): fp32 only; every fused multiply-add is an explicit
 * fmaf(); nothing else may be contracted (-ffp-contract=off); sqrtf and '/' are IEEE correctly rounded;
 * no libm transcendental is used (so_expf is a fixed polynomial).
 *
 * Build: see oracle/Makefile (gcc -O2 -mavx2 -mfma -ffp-contract=off -fopenmp).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define SO_OK 0
#define SO_EINVAL -1
#define SO_ENOMEM -2
#define SO_ENOHYP -5

static inline float as_f32(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t as_u32(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* ------------------------------------------------------------------------------------------------
 * derived constants ("d_thr = sigma*sqrt(-2 ln t_cmp) precomputed on host in fp64
 * -> fp32").  out[0]=d_thr out[1]=neg_inv2sig2 out[2]=tau2
 * ---------------------------------------------------------------------------------------------- */
int so_derive(float sigma, float t_cmp, float tau, float* out) {
  if (!(sigma > 0.f) || !(t_cmp > 0.f) || !(t_cmp < 1.f) || !(tau > 0.f)) return SO_EINVAL;
  out[0] = (float)((double)sigma * sqrt(-2.0 * log((double)t_cmp)));
  out[1] = (float)(-1.0 / (2.0 * (double)sigma * (double)sigma));
  out[2] = (float)((double)tau * (double)tau);
  return SO_OK;
}

/* exp(x) for x <= 0: k = round(x*log2 e) by the 1.5*2^23 trick, r = x - k ln2 (two-term), degree-6
 * Taylor/Horner in r, scale by 2^k through the exponent field.  Inputs below -87 are clamped. */
float so_expf(float x) {
  const float LOG2E = 0x1.715476p+0f, LN2_HI = 0x1.62e400p-1f, LN2_LO = 0x1.7f7d1cp-20f;
  const float MAGIC = 12582912.0f; /* 1.5 * 2^23 */
  x = fmaxf(x, -87.0f);
  float kf = fmaf(x, LOG2E, MAGIC);
  kf = kf - MAGIC;
  float r = fmaf(kf, -LN2_HI, x);
  r = fmaf(kf, -LN2_LO, r);
  float p = 0x1.6c16c2p-10f;           /* 1/720 */
  p = fmaf(p, r, 0x1.111112p-7f);      /* 1/120 */
  p = fmaf(p, r, 0x1.555556p-5f);      /* 1/24  */
  p = fmaf(p, r, 0x1.555556p-3f);      /* 1/6   */
  p = fmaf(p, r, 0.5f);
  p = fmaf(p, r, 1.0f);
  p = fmaf(p, r, 1.0f);
  int k = (int)kf;
  return p * as_f32((uint32_t)(k + 127) << 23);
}

static inline float dist3(const float* x, const float* y, const float* z, int64_t i, int64_t j) {
  float dx = x[i] - x[j], dy = y[i] - y[j], dz = z[i] - z[j];
  return sqrtf(fmaf(dz, dz, fmaf(dy, dy, dx * dx)));
}

/* ------------------------------------------------------------------------------------------------
 * Stage A — compat_graph.  src/tgt are SoA (x plane, y plane, z plane), n each.
 * S: n*n fp32 (may be NULL), bits: n*W u64 with W = ceil(n/64) (may be NULL), deg: n u32 (may be NULL).
 * ---------------------------------------------------------------------------------------------- */
int so_compat(const float* src, const float* tgt, int64_t n, float d_thr, float min_len,
              float neg_inv2sig2, float* S, uint64_t* bits, uint32_t* deg, int n_threads) {
  if (n < 1) return SO_EINVAL;
  const float *px = src, *py = src + n, *pz = src + 2 * n;
  const float *qx = tgt, *qy = tgt + n, *qz = tgt + 2 * n;
  const int64_t W = (n + 63) / 64;
  if (bits) memset(bits, 0, (size_t)n * W * 8);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
#endif
  for (int64_t i = 0; i < n; i++) {
    uint32_t d_i = 0;
    for (int64_t j = 0; j < n; j++) {
      float s = 0.f;
      if (j != i) {
        float dp = dist3(px, py, pz, i, j);
        float dq = dist3(qx, qy, qz, i, j);
        float d = fabsf(dp - dq);
        if (d <= d_thr && dp >= min_len && dq >= min_len) {
          s = so_expf((d * d) * neg_inv2sig2);
          d_i++;
          if (bits) bits[i * W + (j >> 6)] |= (uint64_t)1 << (j & 63);
        }
      }
      if (S) S[i * n + j] = s;
    }
    if (deg) deg[i] = d_i;
  }
  return SO_OK;
}

/* ------------------------------------------------------------------------------------------------
 * Stage B — triangles_topT.
 * Enumerates every i<j<k with the three edges present, key = fp32 bits of w = (s_ij + s_ik) + s_jk
 * (rank_mode 0; w > 0 so the bit pattern orders like the value) or deg_i + deg_j + deg_k (rank_mode 1),
 * and returns the top-T under the total order (key desc, then i asc, j asc, k asc).
 * Selection is done by value (two-level counting select on the key), then per-row survivor counts and one
 * re-enumeration that writes every row's survivors at its prefix, then a sort of the T survivors with the full
 * comparator — deliberately a different mechanism from the GPU's ordinal-indexed radix select.  Every
 * enumeration pass runs over the rows in parallel (OpenMP); results do not depend on the thread count.
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint32_t key, i, j, k; } so_tri;

static int tri_cmp(const void* a, const void* b) {
  const so_tri *x = (const so_tri*)a, *y = (const so_tri*)b;
  if (x->key != y->key) return x->key > y->key ? -1 : 1;
  if (x->i != y->i) return x->i < y->i ? -1 : 1;
  if (x->j != y->j) return x->j < y->j ? -1 : 1;
  if (x->k != y->k) return x->k < y->k ? -1 : 1;
  return 0;
}

static inline uint32_t tri_key(const float* S, const uint32_t* deg, int64_t n, int rank_mode,
                               int64_t i, int64_t j, int64_t k) {
  if (rank_mode == 1) return deg[i] + deg[j] + deg[k];
  float w = (S[i * n + j] + S[i * n + k]) + S[j * n + k];
  return as_u32(w);
}

/* calls f(ctx, key, i, j, k) for every triangle of ROW i (i < j < k) in lexicographic (j,k) order */
typedef void (*tri_fn)(void* ctx, uint32_t key, uint32_t i, uint32_t j, uint32_t k);
static void for_row_triangles(const float* S, const uint64_t* bits, const uint32_t* deg, int64_t n,
                              int rank_mode, int64_t i, tri_fn f, void* ctx) {
  const int64_t W = (n + 63) / 64;
  const uint64_t* bi = bits + i * W;
  for (int64_t wj = i >> 6; wj < W; wj++) {
    uint64_t mj = bi[wj];
    if (wj == (i >> 6)) mj &= ((i & 63) == 63) ? 0 : (~(uint64_t)0 << ((i & 63) + 1));
    while (mj) {
      int64_t j = wj * 64 + __builtin_ctzll(mj);
      mj &= mj - 1;
      const uint64_t* bj = bits + j * W;
      for (int64_t wk = j >> 6; wk < W; wk++) {
        uint64_t mk = bi[wk] & bj[wk];
        if (wk == (j >> 6)) mk &= ((j & 63) == 63) ? 0 : (~(uint64_t)0 << ((j & 63) + 1));
        while (mk) {
          int64_t k = wk * 64 + __builtin_ctzll(mk);
          mk &= mk - 1;
          f(ctx, tri_key(S, deg, n, rank_mode, i, j, k), (uint32_t)i, (uint32_t)j, (uint32_t)k);
        }
      }
    }
  }
}

/* Pass kinds.  Rows are independent, so every pass runs over the rows in parallel (OpenMP, dynamic schedule: low rows
 * hold more triangles); what is merged afterwards — histogram sums, per-row counts — does not depend on the thread
 * count or on which thread took which row, so the result is identical for any n_threads. */
typedef struct { uint64_t* hist; uint32_t prefix; int pass; uint64_t total; } hist_ctx;
static void hist_cb(void* c, uint32_t key, uint32_t i, uint32_t j, uint32_t k) {
  hist_ctx* h = (hist_ctx*)c; (void)i; (void)j; (void)k;
  if (h->pass == 0) { h->hist[key >> 16]++; h->total++; }
  else if ((key >> 16) == h->prefix) h->hist[key & 0xFFFF]++;
}
typedef struct { uint32_t kstar; uint64_t gt, eq; } count_ctx;
static void count_cb(void* c, uint32_t key, uint32_t i, uint32_t j, uint32_t k) {
  count_ctx* q = (count_ctx*)c; (void)i; (void)j; (void)k;
  q->gt += key > q->kstar; q->eq += key == q->kstar;
}
/* eq_left: how many triangles with key == kstar this row may still take (the ties go to the lowest (i,j,k)) */
typedef struct { so_tri* out; uint32_t kstar; uint64_t eq_left; } emit_ctx;
static void emit_cb(void* c, uint32_t key, uint32_t i, uint32_t j, uint32_t k) {
  emit_ctx* e = (emit_ctx*)c;
  int take = 0;
  if (key > e->kstar) take = 1;
  else if (key == e->kstar && e->eq_left > 0) { take = 1; e->eq_left--; }
  if (take) { so_tri t = {key, i, j, k}; *e->out++ = t; }
}

/* one histogram pass over all rows; returns the number of triangles seen (pass 0) */
static int hist_pass(const float* S, const uint64_t* bits, const uint32_t* deg, int64_t n, int rank_mode, int pass,
                     uint32_t prefix, uint64_t* hist, uint64_t* total, int n_threads) {
  int nt = n_threads > 0 ? n_threads : 1;
  uint64_t* th = (uint64_t*)calloc((size_t)nt * 65536, 8);
  uint64_t* tt = (uint64_t*)calloc((size_t)nt, 8);
  if (!th || !tt) { free(th); free(tt); return SO_ENOMEM; }
#ifdef _OPENMP
#pragma omp parallel num_threads(nt)
#endif
  {
#ifdef _OPENMP
    const int me = omp_get_thread_num();
#else
    const int me = 0;
#endif
    hist_ctx h = {th + (size_t)me * 65536, prefix, pass, 0};
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 4)
#endif
    for (int64_t i = 0; i < n; i++) for_row_triangles(S, bits, deg, n, rank_mode, i, hist_cb, &h);
    tt[me] = h.total;
  }
  memset(hist, 0, 65536 * 8);
  uint64_t tot = 0;
  for (int t = 0; t < nt; t++) {
    tot += tt[t];
    for (int b = 0; b < 65536; b++) hist[b] += th[(size_t)t * 65536 + b];
  }
  if (total) *total = tot;
  free(th); free(tt);
  return SO_OK;
}

/* tri: T*3 u32, key: T u32; *t_eff = number written; returns SO_OK; *tri_total = number of 3-cliques */
int so_triangles(const float* S, const uint64_t* bits, const uint32_t* deg, int64_t n, int rank_mode,
                 uint32_t T, uint32_t* tri, uint32_t* key, uint32_t* t_eff, uint64_t* tri_total, int n_threads) {
  if (n < 3 || (rank_mode == 0 && !S) || (rank_mode == 1 && !deg) || !bits) return SO_EINVAL;
  const int nt = n_threads > 0 ? n_threads : 1;
  uint64_t* hist = (uint64_t*)calloc(65536, 8);
  if (!hist) return SO_ENOMEM;
  uint64_t total = 0;
  int rc = hist_pass(S, bits, deg, n, rank_mode, 0, 0, hist, &total, nt);
  if (rc) { free(hist); return rc; }
  if (tri_total) *tri_total = total;
  uint64_t want = T < total ? T : total;
  *t_eff = (uint32_t)want;
  if (want == 0) { free(hist); return SO_OK; }
  /* level 1: high 16 bits */
  uint64_t above = 0; int32_t b = 65535;
  for (; b >= 0; b--) { if (above + hist[b] >= want) break; above += hist[b]; }
  uint32_t hi = (uint32_t)b;
  rc = hist_pass(S, bits, deg, n, rank_mode, 1, hi, hist, NULL, nt);
  if (rc) { free(hist); return rc; }
  for (b = 65535; b >= 0; b--) { if (above + hist[b] >= want) break; above += hist[b]; }
  uint32_t kstar = (hi << 16) | (uint32_t)b;
  free(hist);
  const uint64_t need_eq = want - above;
  /* per-row counts of keys above / equal to kstar, then their prefixes in row order: row i writes its survivors at
   * gt_before + min(eq_before, need_eq) and may take need_eq - eq_before ties (lowest (i,j,k) first) */
  uint64_t* rgt = (uint64_t*)malloc((size_t)(n + 1) * 8);
  uint64_t* req = (uint64_t*)malloc((size_t)(n + 1) * 8);
  so_tri* buf = (so_tri*)malloc((size_t)want * sizeof(so_tri));
  if (!rgt || !req || !buf) { free(rgt); free(req); free(buf); return SO_ENOMEM; }
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nt)
#endif
  for (int64_t i = 0; i < n; i++) {
    count_ctx q = {kstar, 0, 0};
    for_row_triangles(S, bits, deg, n, rank_mode, i, count_cb, &q);
    rgt[i] = q.gt; req[i] = q.eq;
  }
  uint64_t g = 0, q = 0;
  for (int64_t i = 0; i < n; i++) { uint64_t a = rgt[i], c = req[i]; rgt[i] = g; req[i] = q; g += a; q += c; }
  rgt[n] = g; req[n] = q;
  rc = (g == above && q >= need_eq) ? SO_OK : SO_EINVAL;
  if (rc == SO_OK) {
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(nt)
#endif
    for (int64_t i = 0; i < n; i++) {
      const uint64_t eqb = req[i] < need_eq ? req[i] : need_eq;
      if (rgt[i + 1] == rgt[i] && (req[i + 1] == req[i] || req[i] >= need_eq)) continue; /* nothing to emit */
      emit_ctx e = {buf + rgt[i] + eqb, kstar, need_eq - eqb};
      for_row_triangles(S, bits, deg, n, rank_mode, i, emit_cb, &e);
    }
    qsort(buf, (size_t)want, sizeof(so_tri), tri_cmp);
    for (uint64_t t = 0; t < want; t++) {
      tri[3 * t] = buf[t].i; tri[3 * t + 1] = buf[t].j; tri[3 * t + 2] = buf[t].k;
      if (key) key[t] = buf[t].key;
    }
  }
  free(rgt); free(req); free(buf);
  return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Stage C1 — kabsch3: rigid transform of one triangle by a fixed-sweep one-sided
 * Jacobi SVD of the 3x3 cross-covariance H = sum_m (p_m - pc)(q_m - qc)^T.
 * H is rank <= 2 for three points: the two dominant singular pairs give u1,u2 / v1,v2, the third
 * column of each frame is the cross product, so R = V U^T is a proper rotation without a det fix.
 * ---------------------------------------------------------------------------------------------- */
#define SO_JACOBI_SWEEPS 6

static inline float dot3(const float* a, const float* b) {
  return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0]));
}
static inline void cross3(const float* a, const float* b, float* c) {
  c[0] = fmaf(a[1], b[2], -(a[2] * b[1]));
  c[1] = fmaf(a[2], b[0], -(a[0] * b[2]));
  c[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}

void so_kabsch3_one(const float P[9], const float Q[9], float Rt[12]) {
  /* P, Q: three points, row m = point m (x,y,z) */
  const float THIRD = 0x1.555556p-2f;
  float pc[3], qc[3], a[3][3], b[3][3];
  for (int c = 0; c < 3; c++) {
    pc[c] = ((P[c] + P[3 + c]) + P[6 + c]) * THIRD;
    qc[c] = ((Q[c] + Q[3 + c]) + Q[6 + c]) * THIRD;
  }
  for (int m = 0; m < 3; m++)
    for (int c = 0; c < 3; c++) { a[m][c] = P[3 * m + c] - pc[c]; b[m][c] = Q[3 * m + c] - qc[c]; }
  /* columns of H: B[col][row] = H[row][col] */
  float B[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      B[c][r] = fmaf(a[2][r], b[2][c], fmaf(a[1][r], b[1][c], a[0][r] * b[0][c]));
  static const int PAIRS[3][2] = {{0, 1}, {0, 2}, {1, 2}};
  for (int sweep = 0; sweep < SO_JACOBI_SWEEPS; sweep++) {
    for (int pr = 0; pr < 3; pr++) {
      float *bp = B[PAIRS[pr][0]], *bq = B[PAIRS[pr][1]];
      float *vp = V[PAIRS[pr][0]], *vq = V[PAIRS[pr][1]];
      float alpha = dot3(bp, bp), beta = dot3(bq, bq), gamma = dot3(bp, bq);
      if (gamma == 0.0f) continue;
      float zeta = (beta - alpha) / (gamma + gamma);
      float den = fabsf(zeta) + sqrtf(fmaf(zeta, zeta, 1.0f));
      float tt = 1.0f / den;
      if (zeta < 0.0f) tt = -tt;
      float cs = 1.0f / sqrtf(fmaf(tt, tt, 1.0f));
      float sn = cs * tt;
      for (int r = 0; r < 3; r++) {
        float x = bp[r], y = bq[r];
        bp[r] = fmaf(-sn, y, cs * x);
        bq[r] = fmaf(sn, x, cs * y);
        x = vp[r]; y = vq[r];
        vp[r] = fmaf(-sn, y, cs * x);
        vq[r] = fmaf(sn, x, cs * y);
      }
    }
  }
  float nrm[3] = {dot3(B[0], B[0]), dot3(B[1], B[1]), dot3(B[2], B[2])};
  /* indices of the largest and second largest squared norms; ties resolved to the lower index */
  int i1 = 0;
  if (nrm[1] > nrm[i1]) i1 = 1;
  if (nrm[2] > nrm[i1]) i1 = 2;
  int i2 = (i1 == 0) ? 1 : 0;
  for (int c = 0; c < 3; c++)
    if (c != i1 && c != i2 && nrm[c] > nrm[i2]) i2 = c;
  float s1 = sqrtf(nrm[i1]), s2 = sqrtf(nrm[i2]);
  float u1[3], u2[3], u3[3], v3[3];
  for (int r = 0; r < 3; r++) { u1[r] = B[i1][r] / s1; u2[r] = B[i2][r] / s2; }
  const float *v1 = V[i1], *v2 = V[i2];
  cross3(u1, u2, u3);
  cross3(v1, v2, v3);
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++)
      Rt[3 * r + c] = fmaf(v3[r], u3[c], fmaf(v2[r], u2[c], v1[r] * u1[c]));
  for (int r = 0; r < 3; r++)
    Rt[9 + r] = qc[r] - fmaf(Rt[3 * r + 2], pc[2], fmaf(Rt[3 * r + 1], pc[1], Rt[3 * r] * pc[0]));
}

void so_kabsch3(const float* src, const float* tgt, int64_t n, const uint32_t* tri, uint32_t T, float* Rt,
                int n_threads) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
#endif
  for (int64_t h = 0; h < (int64_t)T; h++) {
    float P[9], Q[9];
    for (int m = 0; m < 3; m++) {
      int64_t v = tri[3 * h + m];
      for (int c = 0; c < 3; c++) { P[3 * m + c] = src[c * n + v]; Q[3 * m + c] = tgt[c * n + v]; }
    }
    so_kabsch3_one(P, Q, Rt + 12 * h);
  }
}

/* ------------------------------------------------------------------------------------------------
 * Stage C2 — score_hypotheses.  A hypothesis with a non-finite entry scores 0.
 * ---------------------------------------------------------------------------------------------- */
static inline int inlier(const float* Rt, float px, float py, float pz, float qx, float qy, float qz,
                         float tau2) {
  float ex = Rt[9]  + fmaf(Rt[2], pz, fmaf(Rt[1], py, fmaf(Rt[0], px, -qx)));
  float ey = Rt[10] + fmaf(Rt[5], pz, fmaf(Rt[4], py, fmaf(Rt[3], px, -qy)));
  float ez = Rt[11] + fmaf(Rt[8], pz, fmaf(Rt[7], py, fmaf(Rt[6], px, -qz)));
  float d2 = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
  return d2 < tau2;
}
static inline int finite12(const float* Rt) {
  for (int c = 0; c < 12; c++) if (!isfinite(Rt[c])) return 0;
  return 1;
}

/* score_mode (include/saccot.h): 0 inlier count; 1 sum of floor(1024 max(0, 1 - d2 / tau^2));
 * 2 sum of floor(1024 max(0, 1 - sqrt(d2) / tau)).  thr: tau^2, 1 / tau^2 or 1 / tau (so_derive2). */
static inline uint32_t score_term(const float* Rt, float px, float py, float pz, float qx, float qy, float qz,
                                  float thr, int mode) {
  float ex = Rt[9]  + fmaf(Rt[2], pz, fmaf(Rt[1], py, fmaf(Rt[0], px, -qx)));
  float ey = Rt[10] + fmaf(Rt[5], pz, fmaf(Rt[4], py, fmaf(Rt[3], px, -qy)));
  float ez = Rt[11] + fmaf(Rt[8], pz, fmaf(Rt[7], py, fmaf(Rt[6], px, -qz)));
  float d2 = fmaf(ez, ez, fmaf(ey, ey, ex * ex));
  if (mode == 0) return d2 < thr;
  float x = mode == 1 ? d2 : sqrtf(d2);
  return (uint32_t)(fmaxf(fmaf(-x, thr, 1.0f), 0.0f) * 1024.0f);
}

void so_derive2(float tau, float* out) { /* out[0] = 1 / tau^2, out[1] = 1 / tau, from fp64 */
  out[0] = (float)(1.0 / ((double)tau * (double)tau));
  out[1] = (float)(1.0 / (double)tau);
}

void so_score_mode(const float* src, const float* tgt, int64_t n, const float* Rt, uint32_t T, float thr, int mode,
                   uint32_t* cnt, int n_threads) {
  const float *px = src, *py = src + n, *pz = src + 2 * n;
  const float *qx = tgt, *qy = tgt + n, *qz = tgt + 2 * n;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
#endif
  for (int64_t h = 0; h < (int64_t)T; h++) {
    const float* M = Rt + 12 * h;
    uint32_t c = 0;
    if (finite12(M))
      for (int64_t m = 0; m < n; m++) c += score_term(M, px[m], py[m], pz[m], qx[m], qy[m], qz[m], thr, mode);
    cnt[h] = c;
  }
}

void so_score(const float* src, const float* tgt, int64_t n, const float* Rt, uint32_t T, float tau2,
              uint32_t* cnt, int n_threads) {
  const float *px = src, *py = src + n, *pz = src + 2 * n;
  const float *qx = tgt, *qy = tgt + n, *qz = tgt + 2 * n;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads > 0 ? n_threads : 1)
#endif
  for (int64_t h = 0; h < (int64_t)T; h++) {
    const float* M = Rt + 12 * h;
    uint32_t c = 0;
    if (finite12(M))
      for (int64_t m = 0; m < n; m++) c += (uint32_t)inlier(M, px[m], py[m], pz[m], qx[m], qy[m], qz[m], tau2);
    cnt[h] = c;
  }
}

/* winner key over hypotheses whose global rank index is rank_of[h] (NULL -> h):
 * K = (count << 32) | (0xFFFFFFFF - rank); hypotheses with count 0 do not compete; 0 = none. */
uint64_t so_best_key(const uint32_t* cnt, uint32_t T, const uint32_t* rank_of) {
  uint64_t best = 0;
  for (uint32_t h = 0; h < T; h++) {
    if (cnt[h] == 0) continue;
    uint32_t r = rank_of ? rank_of[h] : h;
    uint64_t k = ((uint64_t)cnt[h] << 32) | (uint64_t)(0xFFFFFFFFu - r);
    if (k > best) best = k;
  }
  return best;
}

/* Stage C3 — inlier_mask */
void so_mask(const float* src, const float* tgt, int64_t n, const float* Rt, float tau2, uint8_t* mask) {
  const float *px = src, *py = src + n, *pz = src + 2 * n;
  const float *qx = tgt, *qy = tgt + n, *qz = tgt + 2 * n;
  int ok = finite12(Rt);
  for (int64_t m = 0; m < n; m++)
    mask[m] = (uint8_t)(ok && inlier(Rt, px[m], py[m], pz[m], qx[m], qy[m], qz[m], tau2));
}

/* ------------------------------------------------------------------------------------------------
 * Winner refinement (optional): least-squares rigid refit over the inlier mask, in fp64.
 * Canonical order (so that the GPU reproduces it bit for bit): points are visited in chunks of 64 consecutive
 * indices; a chunk's masked sums are accumulated sequentially in index order; chunk sums are then added
 * sequentially in chunk order.  Pass 1: count, sum p, sum q -> centroids (division).  Pass 2:
 * H[r][c] = sum fma(a_r, b_c, H[r][c]) with a = p - pc, b = q - qc.  Then the same two-dominant-pairs +
 * cross-product construction as so_kabsch3_one, on H, in double, 10 Jacobi sweeps.  Fewer than 3 inliers or a
 * non-finite result leaves Rt untouched (returns 0); otherwise Rt is overwritten with the fp32-rounded refit.
 * ---------------------------------------------------------------------------------------------- */
#define SO_REFINE_SWEEPS 10
static inline double ddot3(const double* a, const double* b) { return fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0])); }
static inline void dcross3(const double* a, const double* b, double* c) {
  c[0] = fma(a[1], b[2], -(a[2] * b[1]));
  c[1] = fma(a[2], b[0], -(a[0] * b[2]));
  c[2] = fma(a[0], b[1], -(a[1] * b[0]));
}

int so_refine(const float* src, const float* tgt, int64_t n, const uint8_t* mask, float* Rt) {
  const int64_t nch = (n + 63) / 64;
  double S[7] = {0, 0, 0, 0, 0, 0, 0}; /* count, sum p (3), sum q (3) */
  for (int64_t ch = 0; ch < nch; ch++) {
    double c[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int64_t m = ch * 64; m < n && m < ch * 64 + 64; m++) {
      if (!mask[m]) continue;
      c[0] += 1.0;
      for (int k = 0; k < 3; k++) { c[1 + k] += (double)src[k * n + m]; c[4 + k] += (double)tgt[k * n + m]; }
    }
    for (int k = 0; k < 7; k++) S[k] += c[k];
  }
  if (S[0] < 3.0) return 0;
  double pc[3], qc[3];
  for (int k = 0; k < 3; k++) { pc[k] = S[1 + k] / S[0]; qc[k] = S[4 + k] / S[0]; }
  double H[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int64_t ch = 0; ch < nch; ch++) {
    double h[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int64_t m = ch * 64; m < n && m < ch * 64 + 64; m++) {
      if (!mask[m]) continue;
      double a[3], b[3];
      for (int k = 0; k < 3; k++) { a[k] = (double)src[k * n + m] - pc[k]; b[k] = (double)tgt[k * n + m] - qc[k]; }
      for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) h[r][c] = fma(a[r], b[c], h[r][c]);
    }
    for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) H[r][c] += h[r][c];
  }
  /* columns of H: B[col][row] */
  double B[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) B[c][r] = H[r][c];
  static const int PAIRS[3][2] = {{0, 1}, {0, 2}, {1, 2}};
  for (int sweep = 0; sweep < SO_REFINE_SWEEPS; sweep++) {
    for (int pr = 0; pr < 3; pr++) {
      double *bp = B[PAIRS[pr][0]], *bq = B[PAIRS[pr][1]], *vp = V[PAIRS[pr][0]], *vq = V[PAIRS[pr][1]];
      double alpha = ddot3(bp, bp), beta = ddot3(bq, bq), gamma = ddot3(bp, bq);
      if (gamma == 0.0) continue;
      double zeta = (beta - alpha) / (gamma + gamma);
      double tt = 1.0 / (fabs(zeta) + sqrt(fma(zeta, zeta, 1.0)));
      if (zeta < 0.0) tt = -tt;
      double cs = 1.0 / sqrt(fma(tt, tt, 1.0));
      double sn = cs * tt;
      for (int r = 0; r < 3; r++) {
        double x = bp[r], y = bq[r];
        bp[r] = fma(-sn, y, cs * x); bq[r] = fma(sn, x, cs * y);
        x = vp[r]; y = vq[r];
        vp[r] = fma(-sn, y, cs * x); vq[r] = fma(sn, x, cs * y);
      }
    }
  }
  double nrm[3] = {ddot3(B[0], B[0]), ddot3(B[1], B[1]), ddot3(B[2], B[2])};
  int i1 = 0;
  if (nrm[1] > nrm[i1]) i1 = 1;
  if (nrm[2] > nrm[i1]) i1 = 2;
  int i2 = (i1 == 0) ? 1 : 0;
  { const int c = 3 - i1 - i2; if (nrm[c] > nrm[i2]) i2 = c; }
  double s1 = sqrt(nrm[i1]), s2 = sqrt(nrm[i2]);
  double u1[3], u2[3], u3[3], v3[3];
  for (int r = 0; r < 3; r++) { u1[r] = B[i1][r] / s1; u2[r] = B[i2][r] / s2; }
  const double *v1 = V[i1], *v2 = V[i2];
  dcross3(u1, u2, u3);
  dcross3(v1, v2, v3);
  double R[9], t[3];
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 3; c++) R[3 * r + c] = fma(v3[r], u3[c], fma(v2[r], u2[c], v1[r] * u1[c]));
  for (int r = 0; r < 3; r++) t[r] = qc[r] - fma(R[3 * r + 2], pc[2], fma(R[3 * r + 1], pc[1], R[3 * r] * pc[0]));
  for (int k = 0; k < 9; k++) if (!isfinite(R[k])) return 0;
  for (int k = 0; k < 3; k++) if (!isfinite(t[k])) return 0;
  for (int k = 0; k < 9; k++) Rt[k] = (float)R[k];
  for (int k = 0; k < 3; k++) Rt[9 + k] = (float)t[k];
  return 1;
}

/* ------------------------------------------------------------------------------------------------
 * Whole path (SoA input).  stats[0]=edges stats[1]=tri_total stats[2]=t_eff stats[3]=best_rank
 * stats[4]=best_count.  stage_s (may be NULL): seconds for A, B, C1, C2, C3.
 * ---------------------------------------------------------------------------------------------- */
static double now_s(void) {
#ifdef _OPENMP
  return omp_get_wtime();
#else
  return 0.0;
#endif
}

int so_register_mode(const float* src, const float* tgt, int64_t n, float sigma, float t_cmp, float tau,
                     float min_len, uint32_t T, int rank_mode, int score_mode, int n_threads, float* R, float* t,
                     uint8_t* mask, uint64_t* stats, double* stage_s);

int so_register(const float* src, const float* tgt, int64_t n, float sigma, float t_cmp, float tau,
                float min_len, uint32_t T, int rank_mode, int n_threads, float* R, float* t,
                uint8_t* mask, uint64_t* stats, double* stage_s) {
  return so_register_mode(src, tgt, n, sigma, t_cmp, tau, min_len, T, rank_mode, 0, n_threads, R, t, mask, stats, stage_s);
}

int so_register_mode(const float* src, const float* tgt, int64_t n, float sigma, float t_cmp, float tau,
                     float min_len, uint32_t T, int rank_mode, int score_mode, int n_threads, float* R, float* t,
                     uint8_t* mask, uint64_t* stats, double* stage_s) {
  if (n < 3 || T == 0 || score_mode < 0 || score_mode > 2) return SO_EINVAL;
  float dv[3];
  if (so_derive(sigma, t_cmp, tau, dv)) return SO_EINVAL;
  for (int64_t m = 0; m < 3 * n; m++) if (!isfinite(src[m]) || !isfinite(tgt[m])) return SO_EINVAL;
  const int64_t W = (n + 63) / 64;
  float* S = (float*)malloc((size_t)n * n * 4);
  uint64_t* bits = (uint64_t*)malloc((size_t)n * W * 8);
  uint32_t* deg = (uint32_t*)malloc((size_t)n * 4);
  uint32_t* tri = (uint32_t*)malloc((size_t)T * 12);
  uint32_t* key = (uint32_t*)malloc((size_t)T * 4);
  float* Rt = (float*)malloc((size_t)T * 48);
  uint32_t* cnt = (uint32_t*)malloc((size_t)T * 4);
  int rc = SO_OK;
  if (!S || !bits || !deg || !tri || !key || !Rt || !cnt) { rc = SO_ENOMEM; goto done; }
  double t0 = now_s();
  so_compat(src, tgt, n, dv[0], min_len, dv[1], S, bits, deg, n_threads);
  double t1 = now_s();
  uint64_t edges = 0;
  for (int64_t i = 0; i < n; i++) edges += deg[i];
  uint32_t t_eff = 0; uint64_t tri_total = 0;
  rc = so_triangles(S, bits, deg, n, rank_mode, T, tri, key, &t_eff, &tri_total, n_threads);
  double t2 = now_s();
  if (rc) goto done;
  so_kabsch3(src, tgt, n, tri, t_eff, Rt, n_threads);
  double t3 = now_s();
  if (score_mode == 0) so_score(src, tgt, n, Rt, t_eff, dv[2], cnt, n_threads);
  else {
    float inv[2];
    so_derive2(tau, inv);
    so_score_mode(src, tgt, n, Rt, t_eff, inv[score_mode - 1], score_mode, cnt, n_threads);
  }
  uint64_t best = so_best_key(cnt, t_eff, NULL);
  double t4 = now_s();
  if (stats) { stats[0] = edges / 2; stats[1] = tri_total; stats[2] = t_eff; stats[3] = 0; stats[4] = 0; }
  if (best == 0) {
    static const float I9[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    memcpy(R, I9, 36); t[0] = t[1] = t[2] = 0.f; memset(mask, 0, (size_t)n);
    rc = SO_ENOHYP;
  } else {
    uint32_t r = 0xFFFFFFFFu - (uint32_t)(best & 0xFFFFFFFFu);
    memcpy(R, Rt + 12 * (size_t)r, 36); memcpy(t, Rt + 12 * (size_t)r + 9, 12);
    so_mask(src, tgt, n, Rt + 12 * (size_t)r, dv[2], mask);
    if (stats) { stats[3] = r; stats[4] = best >> 32; }
  }
  double t5 = now_s();
  if (stage_s) { stage_s[0] = t1 - t0; stage_s[1] = t2 - t1; stage_s[2] = t3 - t2; stage_s[3] = t4 - t3; stage_s[4] = t5 - t4; }
done:
  free(S); free(bits); free(deg); free(tri); free(key); free(Rt); free(cnt);
  return rc;
}

int so_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
