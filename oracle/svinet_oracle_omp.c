/* TEST / BENCH INFRASTRUCTURE -- never linked into the product (see svinet_oracle.h).
 *
 * An ALL-CORES CPU figure for bench.py's `cpu_baseline_allcores` ("optionally also an
 * all-cores OpenMP figure, labelled as such").  THIS IS NOT THE REFERENCE'S ALGORITHM ORDER: the reference's
 * link-sampling path is single-threaded (src/linksampling.cc:556-790 uses no thread), so the
 * contract's `cpu_baseline` stays the sequential restatement in svinet_oracle.c.  Here the same sweep is threaded --
 * the phi pass pull-style over a CSR of the training links (a thread owns a node's row: no atomics), the s3 link loop
 * and the node loops split across threads, the K-vectors as per-thread reductions -- so sums are taken in a different
 * order and results agree with the sequential oracle to rounding only (tests/test_oracle_omp.py: 1e-9 relative after 20 sweeps, equal link counts,
 * flags and tags).  It answers one question: what would every core of the GPU box's host do on this path.
 *
 * Built as its own library (libsvinet_oracle_omp.so, -fopenmp) from the oracle's translation unit, so that the pinned
 * sequential library is compiled exactly as before.
 */
#include "svinet_oracle.c"
#include <omp.h>

int orc_omp_max_threads(void) { return omp_get_max_threads(); }

/* CSR of the training links (both directions), rebuilt when the handle's link list changes (checked by a hash: handles
 * come and go at the same addresses) */
static struct { const void *owner; uint64_t nlinks, hash; uint64_t *ptr; uint32_t *col; } g_csr;

static void csr_for(const orc_ls *m, int nthreads) {
  const uint64_t L = m->nlinks;
  uint64_t h = 0;
#pragma omp parallel for num_threads(nthreads) reduction(+ : h) schedule(static)
  for (int64_t l = 0; l < (int64_t)L; ++l) h += (uint64_t)m->links[2 * l] * 2654435761u + m->links[2 * l + 1] + (uint64_t)l;
  if (g_csr.owner == (const void *)m && g_csr.nlinks == L && g_csr.hash == h && g_csr.ptr) return;
  free(g_csr.ptr); free(g_csr.col);
  g_csr.ptr = (uint64_t *)calloc((size_t)m->n + 1, sizeof(uint64_t));
  g_csr.col = (uint32_t *)malloc((size_t)(2 * L + 1) * sizeof(uint32_t));
  for (uint64_t l = 0; l < L; ++l) { g_csr.ptr[m->links[2 * l] + 1]++; g_csr.ptr[m->links[2 * l + 1] + 1]++; }
  for (uint32_t p = 0; p < m->n; ++p) g_csr.ptr[p + 1] += g_csr.ptr[p];
  uint64_t *fill = (uint64_t *)malloc((size_t)m->n * sizeof(uint64_t));
  memcpy(fill, g_csr.ptr, (size_t)m->n * sizeof(uint64_t));
  for (uint64_t l = 0; l < L; ++l) {
    uint32_t p = m->links[2 * l], q = m->links[2 * l + 1];
    g_csr.col[fill[p]++] = q;
    g_csr.col[fill[q]++] = p;
  }
  free(fill);
  g_csr.owner = m; g_csr.nlinks = L; g_csr.hash = h;
}

/* one sweep of orc_ls_sweep (src/linksampling.cc:556-790), threaded; same return values.
 * The phi pass is PULL-style: a thread owns a node, walks its training links and adds phi to the node's own row only
 * (every link is evaluated from both ends: twice the arithmetic, no atomics, no scatter); the K-vectors and the link
 * counters are taken from the p < q visit. */
int orc_ls_sweep_omp(orc_ls *m, int nthreads) {
  const uint32_t K = m->k, n = m->n;
  if (nthreads < 1) nthreads = 1;
  if (m->cfg.max_iterations && m->iter > m->cfg.max_iterations) return 1;
  if (m->cfg.max_iterations == 1) m->write_comm = 1;
  if (m->write_comm) {
    memset(m->member, 0, (size_t)n * K);
    memset(m->fmap, 0, (size_t)n * K * sizeof(double));
    m->member_valid = 1;
  }
  csr_for(m, nthreads);
  double *gnext = m->gammanext, *lnext = m->lambdanext;
  const double *elogpi = m->elogpi, *elogbeta = m->elogbeta;
  double *sum = m->sum, *s1 = m->s1, *s2 = m->s2, *s3 = m->s3;
  memset(s1, 0, K * sizeof(double)); memset(s2, 0, K * sizeof(double));
  memset(s3, 0, K * sizeof(double)); memset(sum, 0, K * sizeof(double));
  uint32_t c = 0, d = 0, sc = 0;
  const int sparse_ok = (int64_t)m->iter > (int64_t)m->cfg.sparse_after_iter;
  const int write_comm = m->write_comm;
  const int64_t L = (int64_t)m->nlinks;

  /* ---- phi pass, :605-725 ---- */
#pragma omp parallel num_threads(nthreads) reduction(+ : c, d, sc) reduction(+ : sum[:K])
  {
    double *phi = (double *)calloc(K, sizeof(double));
    double *acc = (double *)calloc(K, sizeof(double));
    uint16_t *uni = (uint16_t *)malloc(sizeof(uint16_t) * (2 * (size_t)m->k10 + 2));
#pragma omp for schedule(dynamic, 32)
    for (int64_t pi = 0; pi < (int64_t)n; ++pi) {
      const uint32_t p = (uint32_t)pi;
      const uint32_t pc = m->converged[p];
      memset(acc, 0, K * sizeof(double));
      for (uint64_t e = g_csr.ptr[p]; e < g_csr.ptr[p + 1]; ++e) {
        const uint32_t q = g_csr.col[e];
        const uint32_t qc = m->converged[q];
        const int once = p < q;
        if ((pc && !qc) || (!pc && qc)) {
          uint32_t k = (pc ? pc : qc) - 1;
          acc[k] += 1;
          if (once) { sum[k] += 2; sc++; }
          continue;
        }
        double r = .0;
        uint32_t max_k = 65535;
        double mx = .0;
        if (sparse_ok && m->active_comms[p] < m->k10 && m->active_comms[q] < m->k10) {
          uint32_t nu = 0;
          for (uint32_t j = 0; j < m->active_k_len[p]; ++j) uni[nu++] = m->active_k[(size_t)p * m->k10 + j];
          for (uint32_t j = 0; j < m->active_k_len[q]; ++j) uni[nu++] = m->active_k[(size_t)q * m->k10 + j];
          qsort(uni, nu, sizeof(uint16_t), cmp_u16);
          uint32_t w = 0;
          for (uint32_t j = 0; j < nu; ++j)
            if (w == 0 || uni[w - 1] != uni[j]) uni[w++] = uni[j];
          nu = w;
          for (uint32_t j = 0; j < nu; ++j) {
            uint32_t k = uni[j];
            phi[k] = elogpi[(size_t)p * K + k] + elogpi[(size_t)q * K + k] + elogbeta[2 * k];
            if (j == 0) r = phi[k];
            else if (phi[k] < r) r = r + log(1 + exp(phi[k] - r));
            else r = phi[k] + log(1 + exp(r - phi[k]));
          }
          for (uint32_t j = 0; j < nu; ++j) {
            uint32_t k = uni[j];
            double v = exp(phi[k] - r);
            acc[k] += v;
            if (once) sum[k] += 2 * v;
            if (v > mx) { mx = v; max_k = k; }       /* D1Array::max over the row: first strict maximum, columns ascending */
          }
          if (once) d++;
        } else {
          for (uint32_t k = 0; k < K; ++k) {
            phi[k] = elogpi[(size_t)p * K + k] + elogpi[(size_t)q * K + k] + elogbeta[2 * k];
            if (k == 0) r = phi[k];
            else if (phi[k] < r) r = r + log(1 + exp(phi[k] - r));
            else r = phi[k] + log(1 + exp(r - phi[k]));
          }
          for (uint32_t k = 0; k < K; ++k) {
            double v = exp(phi[k] - r);
            acc[k] += v;
            if (once) sum[k] += 2 * v;
            if (v > mx) { mx = v; max_k = k; }
          }
          if (once) c++;
        }
        if (write_comm && mx > m->cfg.link_thresh) {      /* :672-680,708-716, this end of the link */
          double f = ++m->fmap[(size_t)p * K + max_k];
          if (f > m->cfg.lt_min_deg) m->member[(size_t)p * K + max_k] = 1;
        }
      }
      for (uint32_t k = 0; k < K; ++k) gnext[(size_t)p * K + k] += acc[k];
    }
    free(phi); free(acc); free(uni);
  }
  for (uint32_t k = 0; k < K; ++k) lnext[2 * k] += sum[k];     /* lambdanext(k,0) took the same increments as sum[k] */
  m->c_dense = c; m->c_sparse = d; m->c_short = sc;

  /* ---- compute_mean_indicators, :526-545 ---- */
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(+ : s1[:K], s2[:K])
  for (int64_t p = 0; p < (int64_t)n; ++p) {
    double tl = m->training_links[p];
    if (tl == 0) continue;
    for (uint32_t k = 0; k < K; ++k) {
      size_t i = (size_t)p * K + k;
      m->mphi[i] = (gnext[i] - m->alpha) / tl;
      s1[k] += m->mphi[i];
      s2[k] += m->mphi[i] * m->mphi[i];
      gnext[i] += (n - tl - 1) * m->mphi[i];
      if (m->annealing) gnext[i] *= m->g->ones / sum[k];
    }
  }

  /* ---- s3 pass, :731-746 (Q2) ---- */
#pragma omp parallel for num_threads(nthreads) schedule(static) reduction(+ : s3[:K])
  for (int64_t l = 0; l < L; ++l) {
    uint32_t p = m->links[2 * (size_t)l], q = m->links[2 * (size_t)l + 1];
    uint32_t pc = m->converged[p], qc = m->converged[q];
    if (pc && !qc)
      s3[pc - 1] += (pc < K ? m->mphi[(size_t)q * K + pc] : 0.0);
    else if (!pc && qc)
      s3[qc - 1] += (qc < K ? m->mphi[(size_t)p * K + qc] : 0.0);
    else
      for (uint32_t k = 0; k < K; ++k) s3[k] += m->mphi[(size_t)p * K + k] * m->mphi[(size_t)q * K + k];
  }

  /* ---- :748-761 ---- */
  for (uint32_t k = 0; k < K; ++k) lnext[2 * k + 1] += s1[k] * s1[k] - s2[k] - s3[k];
  { double *t = m->gamma; m->gamma = m->gammanext; m->gammanext = t; }
  { double *t = m->lambda; m->lambda = m->lambdanext; m->lambdanext = t; }
  for (uint32_t k = 0; k < K; ++k) { m->lambdanext[2 * k] = m->eta0; m->lambdanext[2 * k + 1] = m->eta1; }
  {
    double *g = m->gamma, *e = m->elogpi, *gn = m->gammanext;
    const double alpha = m->alpha;
    const uint32_t k10 = m->k10;
#pragma omp parallel for num_threads(nthreads) schedule(static)
    for (int64_t p = 0; p < (int64_t)n; ++p) {
      double s = .0;
      for (uint32_t k = 0; k < K; ++k) { gn[(size_t)p * K + k] = alpha; s += g[(size_t)p * K + k]; }
      double psi_sum = orc_digamma(s);                                  /* set_dir_exp of the row */
      for (uint32_t k = 0; k < K; ++k) e[(size_t)p * K + k] = orc_digamma(g[(size_t)p * K + k]) - psi_sum;
      uint32_t active = 0, pk = 0;                                      /* prune of the row, :455-491 */
      m->active_k_len[p] = 0;
      for (uint32_t k = 0; k < K; ++k)
        if (g[(size_t)p * K + k] - alpha >= 1) {
          active++;
          if (active <= k10) m->active_k[(size_t)p * (k10 ? k10 : 1) + m->active_k_len[p]++] = (uint16_t)k;
          pk = k;
        }
      if (active > k10) m->active_k_len[p] = 0;
      if (active == 1) m->converged[p] = pk + 1;
      m->active_comms[p] = active;
    }
  }
  set_dir_exp(m->lambda, m->elogbeta, K, 2);

  /* ---- :768-787 ---- */
  m->write_comm = (m->iter % m->cfg.reportfreq == m->cfg.reportfreq - 1);
  int stopped = 0;
  if (m->iter % m->cfg.reportfreq == 0 && !m->skip_validation)
    stopped = validation_likelihood(m);
  if (stopped) return 2;
  m->iter++;
  return 0;
}
