/*
 * svinet_oracle.c -- CPU restatement of svinet's `-link-sampling` path.
 *
 * TEST INFRASTRUCTURE ONLY (see svinet_oracle.h).  Plain C99, single thread,
 * IEEE double, the reference's own loop order and accumulation order.
 *
 * The reference is C++ (src/linksampling.cc, src/network.cc, src/matrix.hh)
 * on top of GSL.  GSL is a third-party dependency that is NOT vendored in the
 * reference tree and NOT installed in this image (configure.ac:15-17 only
 * checks that -lgsl links; no version is pinned), so the reference cannot be
 * built here.  The GSL pieces on this path are restated from their published
 * algorithms:
 *   gsl_rng_default      = mt19937 (Matsumoto & Nishimura 1998, 2002 init)
 *   gsl_rng_set(r, 0)    = seed 4357
 *   gsl_rng_uniform      = get() / 2^32
 *   gsl_rng_uniform_int  = rejection with scale = 0xffffffff / n
 *   gsl_sf_psi           = digamma to double accuracy
 *   gsl_ran_bernoulli_pdf(k,p) = k ? p : 1-p
 * and pinned by the reference's shipped real-GSL outputs (tests/golden).
 */
#include "svinet_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ======================================================================= */
/* MT19937 as in GSL rng/mt.c                                               */
/* ======================================================================= */
#define MT_N 624
#define MT_M 397

struct orc_rng {
  uint32_t mt[MT_N];
  int mti;
};

static void mt_set(orc_rng *r, unsigned long s) {
  if (s == 0) s = 4357; /* GSL: the default seed */
  r->mt[0] = (uint32_t)(s & 0xffffffffUL);
  for (int i = 1; i < MT_N; i++)
    r->mt[i] = (uint32_t)(1812433253UL * (r->mt[i - 1] ^ (r->mt[i - 1] >> 30)) + (unsigned long)i);
  r->mti = MT_N;
}

orc_rng *orc_rng_new(unsigned long seed) {
  orc_rng *r = (orc_rng *)malloc(sizeof(orc_rng));
  mt_set(r, seed);
  return r;
}
void orc_rng_free(orc_rng *r) { free(r); }

uint32_t orc_rng_get(orc_rng *r) {
  static const uint32_t mag01[2] = {0x0u, 0x9908b0dfu};
  uint32_t *mt = r->mt;
  if (r->mti >= MT_N) {
    int kk;
    for (kk = 0; kk < MT_N - MT_M; kk++) {
      uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
      mt[kk] = mt[kk + MT_M] ^ (y >> 1) ^ mag01[y & 1u];
    }
    for (; kk < MT_N - 1; kk++) {
      uint32_t y = (mt[kk] & 0x80000000u) | (mt[kk + 1] & 0x7fffffffu);
      mt[kk] = mt[kk + (MT_M - MT_N)] ^ (y >> 1) ^ mag01[y & 1u];
    }
    uint32_t y = (mt[MT_N - 1] & 0x80000000u) | (mt[0] & 0x7fffffffu);
    mt[MT_N - 1] = mt[MT_M - 1] ^ (y >> 1) ^ mag01[y & 1u];
    r->mti = 0;
  }
  uint32_t k = mt[r->mti++];
  k ^= (k >> 11);
  k ^= (k << 7) & 0x9d2c5680u;
  k ^= (k << 15) & 0xefc60000u;
  k ^= (k >> 18);
  return k;
}

double orc_rng_uniform(orc_rng *r) { return orc_rng_get(r) / 4294967296.0; }

/* gsl_rng_uniform_int: offset=min=0, range=max-min=0xffffffff */
uint32_t orc_rng_uniform_int(orc_rng *r, uint32_t n) {
  uint32_t scale = 0xffffffffu / n;
  uint32_t k;
  do {
    k = orc_rng_get(r) / scale;
  } while (k >= n);
  return k;
}

/* ======================================================================= */
/* digamma, x > 0 : upward recurrence to x >= 10 then the asymptotic series */
/* (stands in for gsl_sf_psi at src/linksampling.hh:181,184)                */
/* ======================================================================= */
double orc_digamma(double x) {
  double acc = 0.0;
  while (x < 10.0) {
    acc -= 1.0 / x;
    x += 1.0;
  }
  double xi = 1.0 / x, xi2 = xi * xi;
  /* B2/2, B4/4, ... : 1/12, -1/120, 1/252, -1/240, 1/132, -691/32760, 1/12 */
  double ser = xi2 * (1.0 / 12.0 -
               xi2 * (1.0 / 120.0 -
               xi2 * (1.0 / 252.0 -
               xi2 * (1.0 / 240.0 -
               xi2 * (1.0 / 132.0 -
               xi2 * (691.0 / 32760.0 -
               xi2 * (1.0 / 12.0)))))));
  return acc + log(x) - 0.5 * xi - ser;
}

/* ======================================================================= */
/* Network: src/network.cc:10-116 (read), src/network.hh:134-193            */
/* ======================================================================= */
typedef struct {
  uint32_t *v;
  uint32_t n, cap;
} uvec;

static void uvec_push(uvec *a, uint32_t x) {
  if (a->n == a->cap) {
    a->cap = a->cap ? a->cap * 2 : 4;
    a->v = (uint32_t *)realloc(a->v, (size_t)a->cap * sizeof(uint32_t));
  }
  a->v[a->n++] = x;
}

struct orc_net {
  uint32_t n_declared;
  uint32_t curr_seq; /* distinct ids seen == nodes with >= 1 line */
  uint32_t ones;
  uvec *adj;         /* _sparse_y */
  uint32_t *deg;
  uvec edges;        /* flat pairs, file order, ordered (min,max) */
  uint32_t *seq2id;
  /* id -> seq open-addressing hash */
  uint32_t *hkey, *hval;
  uint8_t *hused;
  uint32_t hcap;
};

static uint32_t hash_u32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

static int net_find(const orc_net *g, uint32_t id, uint32_t *seq) {
  uint32_t h = hash_u32(id) & (g->hcap - 1);
  while (g->hused[h]) {
    if (g->hkey[h] == id) { *seq = g->hval[h]; return 1; }
    h = (h + 1) & (g->hcap - 1);
  }
  return 0;
}

/* Network::add, src/network.hh:134-148 */
static int net_add(orc_net *g, uint32_t id) {
  if (g->curr_seq >= g->n_declared) return 0;
  uint32_t h = hash_u32(id) & (g->hcap - 1);
  while (g->hused[h]) h = (h + 1) & (g->hcap - 1);
  g->hused[h] = 1; g->hkey[h] = id; g->hval[h] = g->curr_seq;
  g->seq2id[g->curr_seq] = id;
  g->curr_seq++;
  return 1;
}

/* Network::y, src/network.hh:158-175 : linear scan of the smaller endpoint */
int orc_net_y(const orc_net *g, uint32_t a, uint32_t b) {
  uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
  const uvec *v = &g->adj[lo];
  for (uint32_t j = 0; j < v->n; ++j)
    if (v->v[j] == hi) return 1;
  return 0;
}

static orc_net *net_alloc(uint32_t n_declared) {
  orc_net *g = (orc_net *)calloc(1, sizeof(orc_net));
  g->n_declared = n_declared;
  g->adj = (uvec *)calloc(n_declared ? n_declared : 1, sizeof(uvec));
  g->deg = (uint32_t *)calloc(n_declared ? n_declared : 1, sizeof(uint32_t));
  g->seq2id = (uint32_t *)calloc(n_declared ? n_declared : 1, sizeof(uint32_t));
  uint32_t cap = 16;
  while (cap < 2u * n_declared + 2u) cap <<= 1;
  g->hcap = cap;
  g->hkey = (uint32_t *)calloc(cap, sizeof(uint32_t));
  g->hval = (uint32_t *)calloc(cap, sizeof(uint32_t));
  g->hused = (uint8_t *)calloc(cap, 1);
  return g;
}

/* body of the while loop, src/network.cc:56-104 */
static void net_line(orc_net *g, uint32_t id1, uint32_t id2) {
  uint32_t p, q;
  if (!net_find(g, id1, &p)) {
    if (!net_add(g, id1)) return;
    p = g->curr_seq - 1;
  }
  if (!net_find(g, id2, &q)) {
    if (!net_add(g, id2)) return;
    q = g->curr_seq - 1;
  }
  if (p != q && orc_net_y(g, p, q) == 0) {
    uint32_t lo = p < q ? p : q, hi = p < q ? q : p;
    uvec_push(&g->edges, lo);
    uvec_push(&g->edges, hi);
    uvec_push(&g->adj[lo], hi);
    uvec_push(&g->adj[hi], lo);
    g->deg[p]++;
    g->deg[q]++;
    g->ones++;
  }
}

orc_net *orc_net_from_pairs(const int32_t *pairs, uint64_t nlines, uint32_t n_declared) {
  orc_net *g = net_alloc(n_declared);
  for (uint64_t i = 0; i < nlines; ++i)
    net_line(g, (uint32_t)pairs[2 * i], (uint32_t)pairs[2 * i + 1]);
  return g;
}

orc_net *orc_net_read(const char *path, uint32_t n_declared) {
  FILE *f = fopen(path, "r");
  if (!f) return NULL;
  orc_net *g = net_alloc(n_declared);
  int id1, id2;
  /* fscanf(f, "%d\t%d\n") : any whitespace separates, CRLF tolerated */
  while (fscanf(f, "%d %d", &id1, &id2) == 2)
    net_line(g, (uint32_t)id1, (uint32_t)id2);
  fclose(f);
  return g;
}

void orc_net_free(orc_net *g) {
  if (!g) return;
  for (uint32_t i = 0; i < g->n_declared; ++i) free(g->adj[i].v);
  free(g->adj); free(g->deg); free(g->seq2id); free(g->edges.v);
  free(g->hkey); free(g->hval); free(g->hused); free(g);
}
uint32_t orc_net_n(const orc_net *g) { return g->curr_seq; }
uint32_t orc_net_ones(const orc_net *g) { return g->ones; }
uint32_t orc_net_deg(const orc_net *g, uint32_t p) { return g->deg[p]; }
const uint32_t *orc_net_adj(const orc_net *g, uint32_t p) { return g->adj[p].v; }
const uint32_t *orc_net_edges(const orc_net *g) { return g->edges.v; }
const uint32_t *orc_net_seq2id(const orc_net *g) { return g->seq2id; }

/* ======================================================================= */
/* LinkSampling                                                             */
/* ======================================================================= */
struct orc_ls {
  const orc_net *g;
  orc_config cfg;
  uint32_t n, k;
  double alpha, eta0, eta1, epsilon;
  double total_pairs, ones_prob, zeros_prob;
  orc_rng *r;

  /* validation sample */
  uint32_t nval;
  uint32_t *val_accept; /* [V][3] */
  uint32_t *val_sorted; /* [V][3] */
  uint64_t *vkeys;      /* hash set of (a<<32|b) */
  uint8_t *vused;
  uint32_t vcap;

  double *gamma, *gammanext, *lambda, *lambdanext;
  double *elogpi, *elogbeta, *mphi, *fmap;
  double *s1, *s2, *s3, *sum, *phi;
  uint32_t *converged, *active_comms;
  uint16_t *active_k; /* [n][k/10] */
  uint32_t *active_k_len;
  uint32_t k10;
  double *training_links;
  uint32_t *links;
  uint32_t nlinks;
  uint8_t *member; /* [n][k]: pushed into _communities[k] in the last tagging sweep */
  int member_valid;

  uint32_t iter;
  int annealing, write_comm;
  double prev_h, max_h;
  int nh;
  int skip_validation;
  uint32_t c_dense, c_sparse, c_short;

  double *rows;
  uint32_t nrows, rows_cap;

  /* -load-test: the test map (distinct pairs, ordered, std::map order) and its likelihood rows */
  uint32_t ntest;
  uint32_t *test_sorted; /* [ntest][3] a,b,y */
  double *trows;
  uint32_t ntrows, trows_cap;
};

void orc_config_default(orc_config *c, uint32_t k) {
  memset(c, 0, sizeof(*c));
  c->k = k;
  c->seed = 0;
  c->heldout_ratio = 0.01;   /* src/main.cc: hol_ratio */
  c->link_thresh = 0.5;
  c->lt_min_deg = 0;
  c->eta_type = 0;
  c->reportfreq = 1;          /* src/main.cc:149-153 */
  c->max_iterations = 0;
  c->use_validation_stop = 1;
  c->skip_init = 0;
  c->sparse_after_iter = 1000;   /* src/linksampling.cc:634 */
}

static uint64_t hash_u64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
static int vset_has(const orc_ls *m, uint32_t a, uint32_t b) {
  if (!m->vcap) return 0;
  uint64_t key = ((uint64_t)a << 32) | b;
  uint32_t h = (uint32_t)hash_u64(key) & (m->vcap - 1);
  while (m->vused[h]) {
    if (m->vkeys[h] == key) return 1;
    h = (h + 1) & (m->vcap - 1);
  }
  return 0;
}
static void vset_add(orc_ls *m, uint32_t a, uint32_t b) {
  uint64_t key = ((uint64_t)a << 32) | b;
  uint32_t h = (uint32_t)hash_u64(key) & (m->vcap - 1);
  while (m->vused[h]) h = (h + 1) & (m->vcap - 1);
  m->vused[h] = 1; m->vkeys[h] = key;
}

/* LinkSampling::edge_ok, src/linksampling.hh:296-326 (validation and test map share the set) */
static int edge_ok(const orc_ls *m, uint32_t a, uint32_t b) {
  if (a == b) return 0;
  return !vset_has(m, a, b);
}

/* LinkSampling::get_random_edge, src/linksampling.hh:328-349 */
static void get_random_edge(orc_ls *m, int link, uint32_t *pa, uint32_t *pb) {
  uint32_t a, b;
  if (!link) {
    do {
      a = orc_rng_uniform_int(m->r, m->n);
      b = orc_rng_uniform_int(m->r, m->n);
      if (a > b) { uint32_t t = a; a = b; b = t; }
    } while (!edge_ok(m, a, b));
  } else {
    const uint32_t *edges = m->g->edges.v;
    do {
      uint32_t j = orc_rng_uniform_int(m->r, m->g->ones);
      a = edges[2 * j]; b = edges[2 * j + 1];
    } while (!edge_ok(m, a, b));
  }
  *pa = a; *pb = b;
}

static int cmp_triple(const void *x, const void *y) {
  const uint32_t *a = (const uint32_t *)x, *b = (const uint32_t *)y;
  if (a[0] != b[0]) return a[0] < b[0] ? -1 : 1;
  if (a[1] != b[1]) return a[1] < b[1] ? -1 : 1;
  return 0;
}

/* init_validation + set_validation_sample, src/linksampling.cc:164-188,281-309 */
static void init_validation(orc_ls *m) {
  int s1 = (int)(m->cfg.heldout_ratio * m->g->ones);
  int p = s1 / 2;
  int c0 = 0, c1 = 0;
  uint32_t cap = 16;
  while (cap < 4u * ((uint32_t)(2 * p + 1) + m->cfg.ntest)) cap <<= 1;   /* the test map shares the set (edge_ok looks in both) */
  m->vcap = cap;
  m->vkeys = (uint64_t *)calloc(cap, sizeof(uint64_t));
  m->vused = (uint8_t *)calloc(cap, 1);
  m->val_accept = (uint32_t *)malloc((size_t)(2 * p + 1) * 3 * sizeof(uint32_t));
  m->nval = 0;
  while (c0 < p || c1 < p) {
    uint32_t a, b;
    get_random_edge(m, c0 == p, &a, &b);
    int y = orc_net_y(m->g, a, b);
    if (y == 0 && c0 < p) {
      c0++;
      m->val_accept[3 * m->nval] = a; m->val_accept[3 * m->nval + 1] = b; m->val_accept[3 * m->nval + 2] = 0;
      m->nval++;
      vset_add(m, a, b);
    }
    if (y == 1 && c1 < p) {
      c1++;
      m->val_accept[3 * m->nval] = a; m->val_accept[3 * m->nval + 1] = b; m->val_accept[3 * m->nval + 2] = 1;
      m->nval++;
      vset_add(m, a, b);
    }
  }
  m->val_sorted = (uint32_t *)malloc((size_t)(m->nval + 1) * 3 * sizeof(uint32_t));
  memcpy(m->val_sorted, m->val_accept, (size_t)m->nval * 3 * sizeof(uint32_t));
  qsort(m->val_sorted, m->nval, 3 * sizeof(uint32_t), cmp_triple); /* std::map<Edge,bool> order */
}

/* set_dir_exp, src/linksampling.hh:170-187 */
static void set_dir_exp(const double *d, double *e, uint32_t rows, uint32_t cols) {
  for (uint32_t i = 0; i < rows; ++i) {
    double s = .0;
    for (uint32_t j = 0; j < cols; ++j) s += d[(size_t)i * cols + j];
    double psi_sum = orc_digamma(s);
    for (uint32_t j = 0; j < cols; ++j)
      e[(size_t)i * cols + j] = orc_digamma(d[(size_t)i * cols + j]) - psi_sum;
  }
}

/* init_gamma2, src/linksampling.cc:374-401 */
static void init_gamma2(orc_ls *m) {
  uint32_t K = m->k;
  double *phi = m->phi;
  for (uint32_t p = 0; p < m->n; ++p) {
    const uvec *e = &m->g->adj[p];
    for (uint32_t r = 0; r < e->n; ++r) {
      uint32_t q = e->v[r];
      if (p >= q) continue;
      for (uint32_t k = 0; k < K; ++k) phi[k] = orc_rng_uniform(m->r);
      double s = .0;
      for (uint32_t k = 0; k < K; ++k) s += phi[k];
      for (uint32_t k = 0; k < K; ++k) phi[k] = phi[k] / s;
      for (uint32_t k = 0; k < K; ++k) m->gamma[(size_t)p * K + k] += phi[k];
      for (uint32_t k = 0; k < K; ++k) m->gamma[(size_t)q * K + k] += phi[k];
    }
  }
}

/* check_and_set_converged + prune, src/linksampling.cc:455-491 */
static void prune(orc_ls *m) {
  uint32_t K = m->k;
  for (uint32_t p = 0; p < m->n; ++p) {
    uint32_t active = 0, pk = 0;
    m->active_k_len[p] = 0;
    for (uint32_t k = 0; k < K; ++k)
      if (m->gamma[(size_t)p * K + k] - m->alpha >= 1) {
        active++;
        if (active <= m->k10) m->active_k[(size_t)p * (m->k10 ? m->k10 : 1) + m->active_k_len[p]++] = (uint16_t)k;
        pk = k;
      }
    if (active > m->k10) m->active_k_len[p] = 0;
    if (active == 1) m->converged[p] = pk + 1; /* sticky */
    m->active_comms[p] = active;
  }
}

/* assign_training_links, src/linksampling.cc:493-523 */
static void assign_training_links(orc_ls *m) {
  m->nlinks = 0;
  m->links = (uint32_t *)malloc((size_t)(m->g->ones + 1) * 2 * sizeof(uint32_t));
  for (uint32_t p = 0; p < m->n; ++p) {
    const uvec *e = &m->g->adj[p];
    for (uint32_t r = 0; r < e->n; ++r) {
      uint32_t q = e->v[r];
      uint32_t lo = p < q ? p : q, hi = p < q ? q : p;
      if (!m->cfg.accuracy && !m->cfg.train_on_heldout && !edge_ok(m, lo, hi)) continue;   /* :503-510 */
      m->training_links[p]++;
      m->training_links[q]++;
      if (p >= q) continue;
      m->links[2 * (size_t)m->nlinks] = p;
      m->links[2 * (size_t)m->nlinks + 1] = q;
      m->nlinks++;
    }
  }
}

/* edge_likelihood, src/linksampling.hh:258-292 (K^2 loop for non-links kept) */
static double edge_likelihood(const orc_ls *m, uint32_t p, uint32_t q, int y, double *pi_p, double *pi_q) {
  uint32_t K = m->k;
  const double *gp = m->gamma + (size_t)p * K, *gq = m->gamma + (size_t)q * K;
  double sp = .0, sq = .0;
  for (uint32_t k = 0; k < K; ++k) sp += gp[k];
  for (uint32_t k = 0; k < K; ++k) pi_p[k] = gp[k] / sp;
  for (uint32_t k = 0; k < K; ++k) sq += gq[k];
  for (uint32_t k = 0; k < K; ++k) pi_q[k] = gq[k] / sq;
  double s = .0;
  if (y == 1) {
    for (uint32_t z = 0; z < K; ++z) {
      double brate = m->lambda[2 * z] / (m->lambda[2 * z] + m->lambda[2 * z + 1]);
      s += pi_p[z] * pi_q[z] * brate;
    }
  } else {
    double one_minus_eps = 1.0 - m->epsilon;
    for (uint32_t zp = 0; zp < K; ++zp) {
      double brate = m->lambda[2 * zp] / (m->lambda[2 * zp] + m->lambda[2 * zp + 1]);
      double omb = 1.0 - brate;
      for (uint32_t zq = 0; zq < K; ++zq)
        s += pi_p[zp] * pi_q[zq] * (zp == zq ? omb : one_minus_eps);
    }
  }
  if (s < 1e-30) s = 1e-30;
  return log(s);
}

static void push_row(orc_ls *m, const double *row) {
  if (m->nrows == m->rows_cap) {
    m->rows_cap = m->rows_cap ? m->rows_cap * 2 : 64;
    m->rows = (double *)realloc(m->rows, (size_t)m->rows_cap * 10 * sizeof(double));
  }
  memcpy(m->rows + (size_t)m->nrows * 10, row, 10 * sizeof(double));
  m->nrows++;
}

/* validation_likelihood, src/linksampling.cc:966-1050.  Returns 1 if the
 * reference would do_on_stop()+exit(0) here. */
static int validation_likelihood(orc_ls *m) {
  if (m->cfg.accuracy) return 0;   /* :969-970 */
  uint32_t K = m->k;
  double *pi_p = (double *)malloc(sizeof(double) * K), *pi_q = (double *)malloc(sizeof(double) * K);
  uint32_t k = 0, kzeros = 0, kones = 0;
  double s = .0, szeros = 0, sones = 0;
  for (uint32_t i = 0; i < m->nval; ++i) {
    uint32_t p = m->val_sorted[3 * i], q = m->val_sorted[3 * i + 1];
    int y = (int)m->val_sorted[3 * i + 2];
    double u = edge_likelihood(m, p, q, y, pi_p, pi_q);
    s += u; k += 1;
    if (y) { sones += u; kones++; } else { szeros += u; kzeros++; }
  }
  free(pi_p); free(pi_q);
  double nshol = (m->zeros_prob * (szeros / kzeros)) + (m->ones_prob * (sones / kones));
  double row[10] = {(double)m->iter, s / k, (double)k, szeros / kzeros, (double)kzeros,
                    sones / kones, (double)kones, m->zeros_prob * (szeros / kzeros),
                    m->ones_prob * (sones / kones), nshol};
  push_row(m, row);

  double a = nshol;
  int stop = 0;
  if (m->iter > 10) {
    if (a > m->prev_h && m->prev_h != 0 && fabs((a - m->prev_h) / m->prev_h) < 0.00001)
      stop = 1;
    else if (a < m->prev_h)
      m->nh++;
    else if (a > m->prev_h)
      m->nh = 0;
    if (a > m->max_h) m->max_h = a;
    if (m->nh > 2) stop = 1;
  }
  m->prev_h = nshol;
  if (m->annealing && stop) {
    m->annealing = 0;
    m->nh = 0;
    m->prev_h = 0;
  } else if (!m->annealing && stop) {
    if (m->cfg.use_validation_stop) return 1;
  }
  return 0;
}

/* LinkSampling::load_test, src/linksampling.cc:1417-1450: every pair of the file is ordered (Network::order_edge) and
 * entered into _test_map (a std::map: duplicates collapse, iteration in (first, second) order).  Called AFTER the
 * validation sample was drawn (src/linksampling.cc:97-108), so the sampler never saw these pairs. */
static void load_test(orc_ls *m) {
  uint32_t nt = m->cfg.ntest;
  uint32_t *t = (uint32_t *)malloc((size_t)(nt + 1) * 3 * sizeof(uint32_t));
  for (uint32_t i = 0; i < nt; ++i) {
    uint32_t a = m->cfg.test_pairs[2 * i], b = m->cfg.test_pairs[2 * i + 1];
    if (a > b) { uint32_t x = a; a = b; b = x; }
    t[3 * i] = a; t[3 * i + 1] = b; t[3 * i + 2] = 0;
  }
  qsort(t, nt, 3 * sizeof(uint32_t), cmp_triple);
  uint32_t u = 0;
  for (uint32_t i = 0; i < nt; ++i) {
    if (u && t[3 * (u - 1)] == t[3 * i] && t[3 * (u - 1) + 1] == t[3 * i + 1]) continue;
    t[3 * u] = t[3 * i]; t[3 * u + 1] = t[3 * i + 1];
    t[3 * u + 2] = (uint32_t)orc_net_y(m->g, t[3 * i], t[3 * i + 1]);   /* test_likelihood asks the network, :1160 */
    if (!vset_has(m, t[3 * u], t[3 * u + 1])) vset_add(m, t[3 * u], t[3 * u + 1]);
    u++;
  }
  m->test_sorted = t;
  m->ntest = u;
}

/* LinkSampling::init_gamma_external, src/linksampling.cc:405-453: gamma = alpha; for every adjacency ENTRY of p (both
 * directions of every link, held-out ones included) a vector phi = alpha everywhere, + n / |c(p)| on each of the
 * communities the file lists p in (once per listing), normalised, is added to gamma[p] -- deg(p) additions of the
 * same vector, kept as repeated additions.  No random draw. */
static void init_gamma_external(orc_ls *m) {
  uint32_t K = m->k, n = m->n;
  /* node -> its communities in the order of the file's lines (Network::_init_communities_seq) */
  uint32_t *cnt = (uint32_t *)calloc((size_t)n + 1, sizeof(uint32_t));
  uint32_t total = m->cfg.init_comm_ptr[m->cfg.ninit_comm];
  for (uint32_t i = 0; i < total; ++i) cnt[m->cfg.init_comm_nodes[i] + 1]++;
  for (uint32_t p = 0; p < n; ++p) cnt[p + 1] += cnt[p];
  uint32_t *fill = (uint32_t *)malloc((size_t)(n + 1) * sizeof(uint32_t));
  memcpy(fill, cnt, (size_t)(n + 1) * sizeof(uint32_t));
  uint32_t *comm = (uint32_t *)malloc((size_t)(total + 1) * sizeof(uint32_t));
  for (uint32_t c = 0; c < m->cfg.ninit_comm; ++c)
    for (uint32_t i = m->cfg.init_comm_ptr[c]; i < m->cfg.init_comm_ptr[c + 1]; ++i)
      comm[fill[m->cfg.init_comm_nodes[i]]++] = c;
  for (size_t i = 0; i < (size_t)n * K; ++i) m->gamma[i] = m->alpha;
  double *phi = m->phi;
  for (uint32_t p = 0; p < n; ++p) {
    const uvec *e = &m->g->adj[p];
    uint32_t nc = cnt[p + 1] - cnt[p];
    for (uint32_t r = 0; r < e->n; ++r) {
      for (uint32_t k = 0; k < K; ++k) phi[k] = m->alpha;
      for (uint32_t j = 0; j < nc; ++j) {
        uint32_t c = comm[cnt[p] + j];
        if (c < K) phi[c] += (double)n / nc;      /* (the reference indexes phi[r[j]] unchecked; a line beyond K is out of bounds there) */
      }
      double s = .0;
      for (uint32_t k = 0; k < K; ++k) s += phi[k];
      for (uint32_t k = 0; k < K; ++k) phi[k] = phi[k] / s;     /* Array::normalize, src/matrix.hh */
      for (uint32_t k = 0; k < K; ++k) m->gamma[(size_t)p * K + k] += phi[k];
    }
  }
  free(cnt); free(fill); free(comm);
}

/* test_likelihood, src/linksampling.cc:1147-1182: the columns of the validation row over the test map, no stop rule */
static void test_likelihood(orc_ls *m) {
  if (m->cfg.accuracy || !m->ntest) return;
  uint32_t K = m->k;
  double *pi_p = (double *)malloc(sizeof(double) * K), *pi_q = (double *)malloc(sizeof(double) * K);
  uint32_t k = 0, kzeros = 0, kones = 0;
  double s = .0, szeros = 0, sones = 0;
  for (uint32_t i = 0; i < m->ntest; ++i) {
    uint32_t p = m->test_sorted[3 * i], q = m->test_sorted[3 * i + 1];
    int y = (int)m->test_sorted[3 * i + 2];
    double u = edge_likelihood(m, p, q, y, pi_p, pi_q);
    s += u; k += 1;
    if (y) { sones += u; kones++; } else { szeros += u; kzeros++; }
  }
  free(pi_p); free(pi_q);
  double nshol = (m->zeros_prob * (szeros / kzeros)) + (m->ones_prob * (sones / kones));
  double row[10] = {(double)m->iter, s / k, (double)k, szeros / kzeros, (double)kzeros,
                    sones / kones, (double)kones, m->zeros_prob * (szeros / kzeros),
                    m->ones_prob * (sones / kones), nshol};
  if (m->ntrows == m->trows_cap) {
    m->trows_cap = m->trows_cap ? m->trows_cap * 2 : 64;
    m->trows = (double *)realloc(m->trows, (size_t)m->trows_cap * 10 * sizeof(double));
  }
  memcpy(m->trows + (size_t)m->ntrows * 10, row, 10 * sizeof(double));
  m->ntrows++;
}

orc_ls *orc_ls_create(const orc_net *g, const orc_config *cfg) {
  orc_ls *m = (orc_ls *)calloc(1, sizeof(orc_ls));
  m->g = g;
  m->cfg = *cfg;
  uint32_t n = g->curr_seq, K = cfg->k; /* env.n = network.n() - singles, src/main.cc:291 */
  m->n = n; m->k = K;
  m->alpha = (double)1 / K;              /* src/env.hh:344 */
  m->epsilon = 1e-30;                    /* src/env.hh:395 */
  /* uint32 arithmetic, src/linksampling.cc:36-37 and src/network.cc:225 (quirk Q5) */
  uint32_t tp32 = (uint32_t)(n * (n - 1u)) / 2u;
  m->total_pairs = (double)tp32;
  m->ones_prob = (double)g->ones / m->total_pairs;
  m->zeros_prob = 1 - m->ones_prob;
  /* Network::set_env_variables, src/network.cc:222-251 */
  switch (cfg->eta_type) {
    case 1: {
      uint64_t tp = tp32;
      double op = (double)g->ones / (double)tp;
      m->eta0 = tp * op / K;
      m->eta1 = tp * 1.0 / ((double)K * K) - m->eta0;
      if (m->eta1 <= 0) m->eta1 = 1.0;
    } break;
    case 2: m->eta0 = 0.97; m->eta1 = 6.33; break;       /* src/env.hh:376-377 */
    case 3: m->eta0 = 4700.59; m->eta1 = 0.77; break;    /* src/env.hh:371-372 */
    default: m->eta0 = 1; m->eta1 = 1; break;
  }
  if (cfg->eta_override0 > 0) m->eta0 = cfg->eta_override0;
  if (cfg->eta_override1 > 0) m->eta1 = cfg->eta_override1;
  size_t nk = (size_t)n * K;
  m->gamma = (double *)calloc(nk, sizeof(double));
  m->gammanext = (double *)calloc(nk, sizeof(double));
  m->elogpi = (double *)calloc(nk, sizeof(double));
  m->mphi = (double *)calloc(nk, sizeof(double));
  m->fmap = (double *)calloc(nk, sizeof(double));
  m->member = (uint8_t *)calloc(nk, 1);
  m->lambda = (double *)calloc(2 * (size_t)K, sizeof(double));
  m->lambdanext = (double *)calloc(2 * (size_t)K, sizeof(double));
  m->elogbeta = (double *)calloc(2 * (size_t)K, sizeof(double));
  m->s1 = (double *)calloc(K, sizeof(double));
  m->s2 = (double *)calloc(K, sizeof(double));
  m->s3 = (double *)calloc(K, sizeof(double));
  m->sum = (double *)calloc(K, sizeof(double));
  m->phi = (double *)calloc(K, sizeof(double));
  m->converged = (uint32_t *)calloc(n, sizeof(uint32_t));
  m->active_comms = (uint32_t *)calloc(n, sizeof(uint32_t));
  m->k10 = K / 10;
  m->active_k = (uint16_t *)calloc((size_t)n * (m->k10 ? m->k10 : 1), sizeof(uint16_t));
  m->active_k_len = (uint32_t *)calloc(n, sizeof(uint32_t));
  m->training_links = (double *)calloc(n, sizeof(double));
  m->annealing = 1;
  m->prev_h = -2147483647; m->max_h = -2147483647;
  m->iter = 0; /* quirk Q1: never initialised in the reference; 0 in every observed run */

  /* src/linksampling.cc:70-75 */
  m->r = orc_rng_new(cfg->seed ? (unsigned long)cfg->seed : 0ul);

  init_validation(m);
  if (cfg->test_pairs && cfg->ntest) load_test(m);            /* src/linksampling.cc:105-108 */
  if (cfg->skip_init) for (size_t i = 0; i < nk; ++i) m->gamma[i] = 1.0;
  else if (cfg->init_comm_ptr) init_gamma_external(m);        /* :112-115 (init_lambda follows: nolambda is false) */
  else init_gamma2(m);
  for (size_t i = 0; i < nk; ++i) m->gammanext[i] = m->alpha;
  for (uint32_t k = 0; k < K; ++k) {
    m->lambda[2 * k] = m->lambdanext[2 * k] = m->eta0;
    m->lambda[2 * k + 1] = m->lambdanext[2 * k + 1] = m->eta1;
  }
  set_dir_exp(m->gamma, m->elogpi, n, K);
  set_dir_exp(m->lambda, m->elogbeta, K, 2);
  if (!cfg->skip_init) validation_likelihood(m); /* ctor row, iter 0: src/linksampling.cc:149-150 */

  /* prologue of infer(), src/linksampling.cc:559-566 */
  memset(m->converged, 0, n * sizeof(uint32_t));
  assign_training_links(m);
  return m;
}

void orc_ls_refresh(orc_ls *m) {
  set_dir_exp(m->gamma, m->elogpi, m->n, m->k);
  set_dir_exp(m->lambda, m->elogbeta, m->k, 2);
}

void orc_ls_free(orc_ls *m) {
  if (!m) return;
  free(m->val_accept); free(m->val_sorted); free(m->vkeys); free(m->vused);
  free(m->gamma); free(m->gammanext); free(m->lambda); free(m->lambdanext);
  free(m->elogpi); free(m->elogbeta); free(m->mphi); free(m->fmap); free(m->member);
  free(m->s1); free(m->s2); free(m->s3); free(m->sum); free(m->phi);
  free(m->converged); free(m->active_comms); free(m->active_k); free(m->active_k_len);
  free(m->training_links); free(m->links); free(m->rows); free(m->test_sorted); free(m->trows);
  orc_rng_free(m->r);
  free(m);
}

/* D1Array<T>::max, src/matrix.hh:521-532 : first strict maximum above 0 */
static double phi_max(const double *phi, uint32_t K, uint32_t *idx) {
  double maxv = .0;
  for (uint32_t i = 0; i < K; ++i)
    if (phi[i] > maxv) { maxv = phi[i]; *idx = i; }
  return maxv;
}

static void tag_community(orc_ls *m, uint32_t p, uint32_t q) {
  uint32_t K = m->k, max_k = 65535;
  double mx = phi_max(m->phi, K, &max_k);
  if (mx > m->cfg.link_thresh) {              /* src/linksampling.cc:672-680,708-716 */
    m->fmap[(size_t)p * K + max_k]++;
    m->fmap[(size_t)q * K + max_k]++;
    if (m->fmap[(size_t)p * K + max_k] > m->cfg.lt_min_deg) m->member[(size_t)p * K + max_k] = 1;
    if (m->fmap[(size_t)q * K + max_k] > m->cfg.lt_min_deg) m->member[(size_t)q * K + max_k] = 1;
  }
}

static int cmp_u16(const void *a, const void *b) {
  return (int)*(const uint16_t *)a - (int)*(const uint16_t *)b;
}

int orc_ls_sweep(orc_ls *m) {
  const uint32_t K = m->k, n = m->n;
  /* src/linksampling.cc:573-579 */
  if (m->cfg.max_iterations && m->iter > m->cfg.max_iterations) return 1;
  if (m->cfg.max_iterations == 1) m->write_comm = 1;          /* :581-582 */
  if (m->write_comm) {                                         /* :584-587 */
    memset(m->member, 0, (size_t)n * K);
    memset(m->fmap, 0, (size_t)n * K * sizeof(double));
    m->member_valid = 1;
  }
  double *gnext = m->gammanext, *lnext = m->lambdanext;
  const double *elogpi = m->elogpi, *elogbeta = m->elogbeta;
  double *phi = m->phi;
  memset(m->s1, 0, K * sizeof(double)); memset(m->s2, 0, K * sizeof(double));   /* clear(), :547-554 */
  memset(m->s3, 0, K * sizeof(double)); memset(m->sum, 0, K * sizeof(double));
  uint32_t c = 0, d = 0, sc = 0;
  uint16_t *uni = (uint16_t *)malloc(sizeof(uint16_t) * (2 * (size_t)m->k10 + 2));

  /* ---- phi pass, src/linksampling.cc:605-725 ---- */
  for (uint32_t l = 0; l < m->nlinks; ++l) {
    uint32_t p = m->links[2 * (size_t)l], q = m->links[2 * (size_t)l + 1];
    memset(phi, 0, K * sizeof(double));
    uint32_t pc = m->converged[p], qc = m->converged[q];
    if (pc && !qc) {
      gnext[(size_t)p * K + pc - 1] += 1;
      gnext[(size_t)q * K + pc - 1] += 1;
      m->sum[pc - 1] += 2;
      lnext[2 * (pc - 1)] += 2;
      sc++;
    } else if (!pc && qc) {
      gnext[(size_t)q * K + qc - 1] += 1;
      gnext[(size_t)p * K + qc - 1] += 1;
      m->sum[qc - 1] += 2;
      lnext[2 * (qc - 1)] += 2;
      sc++;
    } else {
      double r = .0;
      if ((int64_t)m->iter > (int64_t)m->cfg.sparse_after_iter && m->active_comms[p] < m->k10 && m->active_comms[q] < m->k10) {
        /* sorted, unique union of the two active lists, :635-640 */
        uint32_t nu = 0;
        for (uint32_t j = 0; j < m->active_k_len[p]; ++j) uni[nu++] = m->active_k[(size_t)p * m->k10 + j];
        for (uint32_t j = 0; j < m->active_k_len[q]; ++j) uni[nu++] = m->active_k[(size_t)q * m->k10 + j];
        qsort(uni, nu, sizeof(uint16_t), cmp_u16);
        uint32_t w = 0;
        for (uint32_t j = 0; j < nu; ++j)
          if (w == 0 || uni[w - 1] != uni[j]) uni[w++] = uni[j];
        nu = w;
        int first = 0;
        for (uint32_t j = 0; j < nu; ++j) {
          uint32_t k = uni[j];
          phi[k] = elogpi[(size_t)p * K + k] + elogpi[(size_t)q * K + k] + elogbeta[2 * k];
          if (!first) { r = phi[k]; first = 1; }
          else if (phi[k] < r) r = r + log(1 + exp(phi[k] - r));
          else r = phi[k] + log(1 + exp(r - phi[k]));
        }
        for (uint32_t j = 0; j < nu; ++j) { uint32_t k = uni[j]; phi[k] = exp(phi[k] - r); }
        for (uint32_t j = 0; j < nu; ++j) {
          uint32_t k = uni[j];
          gnext[(size_t)p * K + k] += phi[k];
          gnext[(size_t)q * K + k] += phi[k];
          lnext[2 * k] += 2 * phi[k];
          m->sum[k] += 2 * phi[k];
        }
        d++;
        if (m->write_comm) tag_community(m, p, q);
      } else {
        for (uint32_t k = 0; k < K; ++k) {
          phi[k] = elogpi[(size_t)p * K + k] + elogpi[(size_t)q * K + k] + elogbeta[2 * k];
          if (k == 0) r = phi[k];
          else if (phi[k] < r) r = r + log(1 + exp(phi[k] - r));
          else r = phi[k] + log(1 + exp(r - phi[k]));
        }
        for (uint32_t k = 0; k < K; ++k) phi[k] = exp(phi[k] - r);   /* lognormalize, src/matrix.hh:320-325 */
        for (uint32_t k = 0; k < K; ++k) {
          gnext[(size_t)p * K + k] += phi[k];
          gnext[(size_t)q * K + k] += phi[k];
          lnext[2 * k] += 2 * phi[k];
          m->sum[k] += 2 * phi[k];
        }
        c++;
        if (m->write_comm) tag_community(m, p, q);
      }
    }
  }
  free(uni);
  m->c_dense = c; m->c_sparse = d; m->c_short = sc;

  /* ---- compute_mean_indicators, src/linksampling.cc:526-545 ---- */
  for (uint32_t p = 0; p < n; ++p) {
    double tl = m->training_links[p];
    if (tl == 0) continue;
    for (uint32_t k = 0; k < K; ++k) {
      size_t i = (size_t)p * K + k;
      m->mphi[i] = (gnext[i] - m->alpha) / tl;
      m->s1[k] += m->mphi[i];
      m->s2[k] += m->mphi[i] * m->mphi[i];
      gnext[i] += (n - tl - 1) * m->mphi[i];
      if (m->annealing) gnext[i] *= m->g->ones / m->sum[k];
    }
  }

  /* ---- s3 pass, src/linksampling.cc:731-746 (quirk Q2: index pc, not pc-1) ---- */
  for (uint32_t l = 0; l < m->nlinks; ++l) {
    uint32_t p = m->links[2 * (size_t)l], q = m->links[2 * (size_t)l + 1];
    uint32_t pc = m->converged[p], qc = m->converged[q];
    if (pc && !qc)
      m->s3[pc - 1] += (pc < K ? m->mphi[(size_t)q * K + pc] : 0.0);
    else if (!pc && qc)
      m->s3[qc - 1] += (qc < K ? m->mphi[(size_t)p * K + qc] : 0.0);
    else
      for (uint32_t k = 0; k < K; ++k) m->s3[k] += m->mphi[(size_t)p * K + k] * m->mphi[(size_t)q * K + k];
  }

  /* ---- :748-761 ---- */
  for (uint32_t k = 0; k < K; ++k) lnext[2 * k + 1] += m->s1[k] * m->s1[k] - m->s2[k] - m->s3[k];
  { double *t = m->gamma; m->gamma = m->gammanext; m->gammanext = t; }
  { double *t = m->lambda; m->lambda = m->lambdanext; m->lambdanext = t; }
  for (size_t i = 0; i < (size_t)n * K; ++i) m->gammanext[i] = m->alpha;
  for (uint32_t k = 0; k < K; ++k) { m->lambdanext[2 * k] = m->eta0; m->lambdanext[2 * k + 1] = m->eta1; }
  set_dir_exp(m->gamma, m->elogpi, n, K);
  set_dir_exp(m->lambda, m->elogbeta, K, 2);
  prune(m);

  /* ---- :768-787 ---- */
  m->write_comm = (m->iter % m->cfg.reportfreq == m->cfg.reportfreq - 1);
  int stopped = 0;
  if (m->iter % m->cfg.reportfreq == 0 && !m->skip_validation)
    stopped = validation_likelihood(m);
  if (stopped) return 2;
  if (m->iter % m->cfg.reportfreq == 0 && !m->skip_validation) test_likelihood(m);   /* :781, not reached on the stopping sweep */
  m->iter++;
  return 0;
}

void orc_ls_set_skip_validation(orc_ls *m, int skip) { m->skip_validation = skip; }

uint32_t orc_ls_n(const orc_ls *m) { return m->n; }
uint32_t orc_ls_k(const orc_ls *m) { return m->k; }
uint32_t orc_ls_nlinks(const orc_ls *m) { return m->nlinks; }
const uint32_t *orc_ls_links(const orc_ls *m) { return m->links; }
const double *orc_ls_training_links(const orc_ls *m) { return m->training_links; }
double *orc_ls_gamma(orc_ls *m) { return m->gamma; }
double *orc_ls_lambda(orc_ls *m) { return m->lambda; }
const double *orc_ls_elogpi(const orc_ls *m) { return m->elogpi; }
const double *orc_ls_elogbeta(const orc_ls *m) { return m->elogbeta; }
const double *orc_ls_mphi(const orc_ls *m) { return m->mphi; }
uint32_t *orc_ls_converged(orc_ls *m) { return m->converged; }
const uint32_t *orc_ls_active_comms(const orc_ls *m) { return m->active_comms; }
const double *orc_ls_fmap(const orc_ls *m) { return m->fmap; }
uint32_t orc_ls_nvalidation(const orc_ls *m) { return m->nval; }
const uint32_t *orc_ls_validation_accept(const orc_ls *m) { return m->val_accept; }
const uint32_t *orc_ls_validation_sorted(const orc_ls *m) { return m->val_sorted; }
uint32_t orc_ls_iter(const orc_ls *m) { return m->iter; }
void orc_ls_set_iter(orc_ls *m, uint32_t iter) { m->iter = iter; }
int orc_ls_annealing(const orc_ls *m) { return m->annealing; }
void orc_ls_set_annealing(orc_ls *m, int a) { m->annealing = a; }
int orc_ls_write_comm(const orc_ls *m) { return m->write_comm; }
double orc_ls_eta0(const orc_ls *m) { return m->eta0; }
double orc_ls_eta1(const orc_ls *m) { return m->eta1; }
double orc_ls_ones_prob(const orc_ls *m) { return m->ones_prob; }
double orc_ls_total_pairs(const orc_ls *m) { return m->total_pairs; }
uint32_t orc_ls_ntest(const orc_ls *m) { return m->ntest; }
const uint32_t *orc_ls_test_sorted(const orc_ls *m) { return m->test_sorted; }
uint32_t orc_ls_ntest_rows(const orc_ls *m) { return m->ntrows; }
const double *orc_ls_test_rows(const orc_ls *m) { return m->trows; }
uint32_t orc_ls_nrows(const orc_ls *m) { return m->nrows; }
const double *orc_ls_rows(const orc_ls *m) { return m->rows; }
void orc_ls_link_counts(const orc_ls *m, uint32_t *dense, uint32_t *sparse, uint32_t *shortcut) {
  *dense = m->c_dense; *sparse = m->c_sparse; *shortcut = m->c_short;
}

uint32_t orc_ls_communities(const orc_ls *m, uint8_t *out) {
  uint32_t K = m->k, lines = 0;
  memcpy(out, m->member, (size_t)m->n * K);
  for (uint32_t k = 0; k < K; ++k)
    for (uint32_t p = 0; p < m->n; ++p)
      if (m->member[(size_t)p * K + k]) { lines++; break; }
  return lines;
}

static int cmp_u32(const void *a, const void *b) {
  uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
  return x < y ? -1 : x > y;
}

/* save_model :804-837, write_communities :882-917, write_groups :1452-1476 */
int orc_ls_write_model(const orc_ls *m, const char *dir) {
  char path[4096];
  uint32_t K = m->k;
  snprintf(path, sizeof path, "%s/gamma.txt", dir);
  FILE *f = fopen(path, "w");
  if (!f) return -1;
  for (uint32_t i = 0; i < m->n; ++i) {
    fprintf(f, "%d\t", i);
    fprintf(f, "%d\t", m->g->seq2id[i]);
    for (uint32_t k = 0; k < K; ++k)
      fprintf(f, k == K - 1 ? "%.5f\n" : "%.5f\t", m->gamma[(size_t)i * K + k]);
  }
  fclose(f);
  snprintf(path, sizeof path, "%s/lambda.txt", dir);
  f = fopen(path, "w");
  if (!f) return -1;
  for (uint32_t k = 0; k < K; ++k)
    fprintf(f, "%d\t%.5f\t%.5f\n", k, m->lambda[2 * k], m->lambda[2 * k + 1]);
  fclose(f);
  snprintf(path, sizeof path, "%s/communities.txt", dir);
  f = fopen(path, "w");
  if (!f) return -1;
  uint32_t *ids = (uint32_t *)malloc(sizeof(uint32_t) * (m->n + 1));
  for (uint32_t k = 0; k < K; ++k) {
    uint32_t c = 0;
    for (uint32_t p = 0; p < m->n; ++p)
      if (m->member[(size_t)p * K + k]) ids[c++] = m->g->seq2id[p];
    if (!c) continue;
    qsort(ids, c, sizeof(uint32_t), cmp_u32);
    for (uint32_t j = 0; j < c; ++j) fprintf(f, "%d ", ids[j]);
    fprintf(f, "\n");
  }
  free(ids);
  fclose(f);
  snprintf(path, sizeof path, "%s/groups.txt", dir);
  f = fopen(path, "w");
  if (!f) return -1;
  for (uint32_t i = 0; i < m->n; ++i) {
    double s = .0;
    for (uint32_t k = 0; k < K; ++k) s += m->gamma[(size_t)i * K + k];
    fprintf(f, "%d\t%d\t", i, m->g->seq2id[i]);
    for (uint32_t k = 0; k < K; ++k)
      fprintf(f, k == K - 1 ? "%.3f\n" : "%.3f\t", m->gamma[(size_t)i * K + k] / s);
  }
  fclose(f);
  return 0;
}
