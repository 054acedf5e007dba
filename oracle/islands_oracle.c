/*
 * islands_oracle.c -- CPU restatement of the islands `core` search hot path.
 * TEST INFRASTRUCTURE ONLY (see islands_oracle.h for the rules and the pinning
 * status).  Build: gcc -O2 -ffp-contract=off -fno-fast-math -fPIC -shared.
 */
#include "islands_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------- */
/* distance.rs                                                               */
/* ------------------------------------------------------------------------- */

/* cosine_distance, src/core/distance.rs:71-88.  One pass, three sequential
 * f32 accumulators; Rust never contracts x*y+acc into an FMA. */
static float cosine_distance(const float* a, const float* b, size_t n) {
  float dot = 0.0f, norm_a = 0.0f, norm_b = 0.0f;
  for (size_t i = 0; i < n; i++) {
    float x = a[i], y = b[i];
    dot += x * y;
    norm_a += x * x;
    norm_b += y * y;
  }
  float norm = sqrtf(norm_a * norm_b);
  if (norm == 0.0f) return 1.0f;
  return 1.0f - (dot / norm);
}

/* euclidean_distance_squared, distance.rs:98-108 (Iterator::sum = left fold). */
static float euclidean_sq(const float* a, const float* b, size_t n) {
  float sum = 0.0f;
  for (size_t i = 0; i < n; i++) {
    float diff = a[i] - b[i];
    sum += diff * diff;
  }
  return sum;
}

/* dot_product_distance, distance.rs:112-115 */
static float dot_distance(const float* a, const float* b, size_t n) {
  float dot = 0.0f;
  for (size_t i = 0; i < n; i++) dot += a[i] * b[i];
  return -dot;
}

/* manhattan_distance, distance.rs:119-122 */
static float manhattan_distance(const float* a, const float* b, size_t n) {
  float sum = 0.0f;
  for (size_t i = 0; i < n; i++) sum += fabsf(a[i] - b[i]);
  return sum;
}

/* DistanceMetric::calculate, distance.rs:38-52 (lengths already equal). */
static float metric_calc(int metric, const float* a, const float* b, size_t n) {
  switch (metric) {
    case ORC_COSINE: return cosine_distance(a, b, n);
    case ORC_EUCLIDEAN: return sqrtf(euclidean_sq(a, b, n)); /* distance.rs:92-94 */
    case ORC_DOT: return dot_distance(a, b, n);
    default: return manhattan_distance(a, b, n);
  }
}

int orc_distance(int metric, const float* a, size_t na, const float* b, size_t nb, float* out) {
  if (na != nb) return ORC_DIMENSION_MISMATCH; /* distance.rs:39-44 */
  *out = metric_calc(metric, a, b, na);
  return ORC_OK;
}

/* calculate_squared, distance.rs:54-66 */
int orc_distance_squared(int metric, const float* a, size_t na, const float* b, size_t nb,
                         float* out) {
  if (na != nb) return ORC_DIMENSION_MISMATCH;
  if (metric == ORC_EUCLIDEAN) {
    *out = euclidean_sq(a, b, na);
  } else {
    float d = metric_calc(metric, a, b, na);
    *out = d * d;
  }
  return ORC_OK;
}

/* Distance::batch_calculate, distance.rs:32-34 */
int orc_batch_distance(int metric, const float* q, size_t d, const float* rows, size_t n,
                       float* out) {
  for (size_t i = 0; i < n; i++) out[i] = metric_calc(metric, q, rows + i * d, d);
  return ORC_OK;
}

/* normalize_vector, distance.rs:125-132 */
void orc_normalize(float* v, size_t d) {
  float s = 0.0f;
  for (size_t i = 0; i < d; i++) s += v[i] * v[i];
  float norm = sqrtf(s);
  if (norm > 0.0f)
    for (size_t i = 0; i < d; i++) v[i] /= norm;
}

/* ------------------------------------------------------------------------- */
/* [external] ordered_float total order + Rust BinaryHeap emulation          */
/* ------------------------------------------------------------------------- */

typedef struct {
  float d;
  uint64_t id;
} item_t;

/* OrderedFloat<f32>::cmp: NaN == NaN, NaN greater than everything, -0 == +0. */
static int of_cmp(float a, float b) {
  if (a < b) return -1;
  if (a > b) return 1;
  if (a == b) return 0;
  int an = isnan(a), bn = isnan(b);
  if (an && bn) return 0;
  return an ? 1 : -1;
}

/* (OrderedFloat<f32>, u64) tuple order: leann.rs:907-908 results heap. */
static int cmp_tuple(const item_t* a, const item_t* b) {
  int c = of_cmp(a->d, b->d);
  if (c) return c;
  return (a->id > b->id) - (a->id < b->id);
}
/* Reverse<(OrderedFloat<f32>, u64)>: leann.rs:907 candidates heap. */
static int cmp_tuple_rev(const item_t* a, const item_t* b) { return cmp_tuple(b, a); }
/* hnsw.rs:136-141 Candidate::cmp = other.distance.cmp(&self.distance) (id ignored). */
static int cmp_hnsw_cand(const item_t* a, const item_t* b) { return of_cmp(b->d, a->d); }
/* Reverse<Candidate>, hnsw.rs:349 results heap. */
static int cmp_hnsw_cand_rev(const item_t* a, const item_t* b) { return of_cmp(a->d, b->d); }

typedef int (*cmp_fn)(const item_t*, const item_t*);

typedef struct {
  item_t* data;
  size_t len, cap;
  cmp_fn cmp; /* max-heap w.r.t. cmp */
} heap_t;

static void heap_init(heap_t* h, cmp_fn cmp) {
  h->data = NULL;
  h->len = h->cap = 0;
  h->cmp = cmp;
}
static void heap_free(heap_t* h) { free(h->data); }

/* BinaryHeap::sift_up(start, pos) */
static size_t heap_sift_up(heap_t* h, size_t start, size_t pos) {
  item_t elt = h->data[pos];
  while (pos > start) {
    size_t parent = (pos - 1) / 2;
    if (h->cmp(&elt, &h->data[parent]) <= 0) break;
    h->data[pos] = h->data[parent];
    pos = parent;
  }
  h->data[pos] = elt;
  return pos;
}

/* BinaryHeap::sift_down_to_bottom(0) followed by sift_up (used by pop). */
static void heap_sift_down_to_bottom(heap_t* h, size_t pos) {
  size_t end = h->len, start = pos;
  item_t elt = h->data[pos];
  size_t child = 2 * pos + 1;
  size_t lim = end >= 2 ? end - 2 : 0; /* end.saturating_sub(2) */
  while (child <= lim && end >= 2) {
    /* child += (data[child] <= data[child+1]) */
    if (h->cmp(&h->data[child], &h->data[child + 1]) <= 0) child += 1;
    h->data[pos] = h->data[child];
    pos = child;
    child = 2 * pos + 1;
  }
  if (end >= 1 && child == end - 1) {
    h->data[pos] = h->data[child];
    pos = child;
  }
  h->data[pos] = elt;
  heap_sift_up(h, start, pos);
}

static void heap_push(heap_t* h, item_t it) {
  if (h->len == h->cap) {
    h->cap = h->cap ? h->cap * 2 : 64;
    h->data = (item_t*)realloc(h->data, h->cap * sizeof(item_t));
  }
  size_t old_len = h->len;
  h->data[h->len++] = it;
  heap_sift_up(h, 0, old_len);
}

/* BinaryHeap::pop: Vec::pop the last, swap with root if non-empty, sift. */
static int heap_pop(heap_t* h, item_t* out) {
  if (h->len == 0) return 0;
  item_t item = h->data[--h->len];
  if (h->len > 0) {
    item_t tmp = h->data[0];
    h->data[0] = item;
    item = tmp;
    heap_sift_down_to_bottom(h, 0);
  }
  *out = item;
  return 1;
}

/* Note on `end.saturating_sub(2)` above: with end < 2 the Rust loop condition
 * `child <= 0` is false because child >= 1, which the `end >= 2` guard mirrors. */

/* slice::sort_by(|a,b| a.1.partial_cmp(&b.1).unwrap_or(Equal)): stable; only
 * strict `<` moves an element, so equal (or unordered) keys keep array order. */
static void stable_sort_by_dist(item_t* v, size_t n) {
  if (n < 2) return;
  item_t* tmp = (item_t*)malloc(n * sizeof(item_t));
  for (size_t w = 1; w < n; w *= 2) {
    for (size_t lo = 0; lo < n; lo += 2 * w) {
      size_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
      size_t i = lo, j = mid, k = lo;
      while (i < mid && j < hi) {
        if (v[j].d < v[i].d) tmp[k++] = v[j++]; /* take right only if strictly less */
        else tmp[k++] = v[i++];
      }
      while (i < mid) tmp[k++] = v[i++];
      while (j < hi) tmp[k++] = v[j++];
    }
    memcpy(v, tmp, n * sizeof(item_t));
  }
  free(tmp);
}

/* HashSet<u64>: only insert()'s "was it new" result is observable. */
typedef struct {
  uint64_t* slots; /* value+1, 0 = empty */
  size_t cap, len;
} set_t;
static void set_init(set_t* s) {
  s->cap = 1024;
  s->len = 0;
  s->slots = (uint64_t*)calloc(s->cap, sizeof(uint64_t));
}
static void set_free(set_t* s) { free(s->slots); }
static uint64_t mix64(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33;
  x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return x;
}
static int set_insert(set_t* s, uint64_t v);
static void set_grow(set_t* s) {
  set_t n;
  n.cap = s->cap * 2;
  n.len = 0;
  n.slots = (uint64_t*)calloc(n.cap, sizeof(uint64_t));
  for (size_t i = 0; i < s->cap; i++)
    if (s->slots[i]) set_insert(&n, s->slots[i] - 1);
  free(s->slots);
  *s = n;
}
/* returns 1 if newly inserted (HashSet::insert -> true). ids == UINT64_MAX unsupported. */
static int set_insert(set_t* s, uint64_t v) {
  if ((s->len + 1) * 10 > s->cap * 7) set_grow(s);
  size_t mask = s->cap - 1, i = (size_t)mix64(v) & mask;
  while (s->slots[i]) {
    if (s->slots[i] == v + 1) return 0;
    i = (i + 1) & mask;
  }
  s->slots[i] = v + 1;
  s->len++;
  return 1;
}

/* ------------------------------------------------------------------------- */
/* leann.rs                                                                  */
/* ------------------------------------------------------------------------- */

int orc_csr_get_neighbors(const orc_csr* g, uint64_t node, const uint64_t** ptr, size_t* len) {
  if (node >= g->num_nodes) return -1; /* leann.rs:227-229 */
  uint64_t s = g->node_offsets[node], e = g->node_offsets[node + 1];
  *ptr = g->neighbors + s;
  *len = (size_t)(e - s);
  return 0;
}

/* InMemoryEmbeddingProvider::compute_embedding, leann.rs:145-150. */
static int provider_get(const float* vectors, uint64_t nvec, size_t d, int copy, uint64_t id,
                        const float** out, float** owned) {
  if (id >= nvec) return ORC_NODE_NOT_FOUND;
  const float* src = vectors + (size_t)id * d;
  if (copy) {
    float* c = (float*)malloc(d * sizeof(float) + 1);
    memcpy(c, src, d * sizeof(float));
    *owned = c;
    *out = c;
  } else {
    *owned = NULL;
    *out = src;
  }
  return ORC_OK;
}

/* apply_pruning_strategy, leann.rs:991-1056: returns how many of the leading
 * `n` unvisited candidates are kept (Global/Local are prefix rules).
 * Proportional draws from thread_rng (leann.rs:1043) and cannot be restated;
 * it is mapped to the deterministic fallback branch `take(num_to_keep)`. */
static size_t prune_keep(const orc_leann_params* p, size_t n, size_t results_len, size_t ef) {
  if (p->prune_ratio == 0.0f || n == 0) return n;
  float keepf = ceilf((float)n * (1.0f - p->prune_ratio));
  size_t num_to_keep = (size_t)keepf;
  if (num_to_keep < 1) num_to_keep = 1;
  if (p->pruning_strategy == ORC_PRUNE_GLOBAL) {
    float ratio = (float)results_len / (float)ef;
    float adj = ceilf((float)n * (1.0f - ratio * p->prune_ratio));
    /* `as usize` saturates: negative -> 0 */
    size_t adjusted = adj > 0.0f ? (size_t)adj : 0;
    if (adjusted < 1) adjusted = 1;
    return adjusted < n ? adjusted : n;
  }
  return num_to_keep < n ? num_to_keep : n;
}

/* search_layer_recompute, leann.rs:899-988 */
static int search_layer(const orc_csr* g, const orc_leann_params* p, const float* vectors,
                        uint64_t nvec, size_t d, int copy, const float* query, uint64_t entry,
                        size_t ef, item_t** out, size_t* out_n, orc_counters* ctr,
                        uint64_t* err_payload) {
  set_t visited;
  heap_t cand, res;
  set_init(&visited);
  heap_init(&cand, cmp_tuple_rev);
  heap_init(&res, cmp_tuple);
  int status = ORC_OK;
  uint64_t* unvisited = NULL;
  size_t unv_cap = 0;
  orc_counters c = {0, 0, 0, 0};

  const float* emb;
  float* owned;
  status = provider_get(vectors, nvec, d, copy, entry, &emb, &owned); /* :911 */
  if (status) {
    if (err_payload) *err_payload = entry;
    goto done;
  }
  float entry_dist = metric_calc(p->metric, query, emb, d); /* :912 */
  free(owned);
  c.evals++;

  set_insert(&visited, entry);
  item_t e0 = {entry_dist, entry};
  heap_push(&cand, e0);
  heap_push(&res, e0);
  c.pushes++;

  item_t cur;
  while (heap_pop(&cand, &cur)) { /* :922 */
    if (res.len > 0) {
      float worst = res.data[0].d; /* results.peek() */
      if (res.len >= ef && of_cmp(cur.d, worst) > 0) break; /* :924-928, OrderedFloat `>` */
    }
    const uint64_t* nb;
    size_t nn;
    if (orc_csr_get_neighbors(g, cur.id, &nb, &nn) != 0) continue; /* :931 None */
    c.expansions++;
    c.edges += nn;
    if (nn > unv_cap) {
      unv_cap = nn * 2;
      unvisited = (uint64_t*)realloc(unvisited, unv_cap * sizeof(uint64_t));
    }
    size_t nu = 0;
    for (size_t i = 0; i < nn; i++)
      if (set_insert(&visited, nb[i])) unvisited[nu++] = nb[i]; /* :933-937 */
    if (nu == 0) continue;                                      /* :939-941 */
    size_t keep = prune_keep(p, nu, res.len, ef);               /* :944 */
    /* compute_embeddings_batch runs over ALL kept ids before any distance (:947):
     * a missing id aborts the search before this hop changes the heaps. */
    for (size_t i = 0; i < keep; i++)
      if (unvisited[i] >= nvec) {
        status = ORC_NODE_NOT_FOUND;
        if (err_payload) *err_payload = unvisited[i];
        goto done;
      }
    for (size_t i = 0; i < keep; i++) { /* :953-970 */
      uint64_t nid = unvisited[i];
      provider_get(vectors, nvec, d, copy, nid, &emb, &owned);
      float nd = metric_calc(p->metric, query, emb, d);
      free(owned);
      c.evals++;
      int should_add = res.len < ef || (res.len == 0 || nd < res.data[0].d); /* raw f32 `<` */
      if (should_add) {
        item_t it = {nd, nid};
        heap_push(&cand, it);
        heap_push(&res, it);
        c.pushes++;
        if (res.len > ef) {
          item_t dropped;
          heap_pop(&res, &dropped);
        }
      }
    }
  }
  /* :984-986 results.into_iter() = backing-array order, then stable sort on d */
  *out = (item_t*)malloc((res.len ? res.len : 1) * sizeof(item_t));
  memcpy(*out, res.data, res.len * sizeof(item_t));
  *out_n = res.len;
  stable_sort_by_dist(*out, *out_n);
done:
  if (ctr) *ctr = c;
  free(unvisited);
  set_free(&visited);
  heap_free(&cand);
  heap_free(&res);
  return status;
}

int orc_leann_search_layer(const orc_csr* g, const orc_leann_params* p, const float* vectors,
                           uint64_t nvec, size_t d, const float* query, uint64_t entry, size_t ef,
                           uint64_t* out_ids, float* out_dist, size_t* out_count,
                           orc_counters* ctr, uint64_t* err_payload) {
  item_t* r = NULL;
  size_t n = 0;
  int st = search_layer(g, p, vectors, nvec, d, 0, query, entry, ef, &r, &n, ctr, err_payload);
  if (st) return st;
  for (size_t i = 0; i < n; i++) {
    out_ids[i] = r[i].id;
    out_dist[i] = r[i].d;
  }
  *out_count = n;
  free(r);
  return ORC_OK;
}

/* search_with_params, leann.rs:868-896 */
int orc_leann_search(const orc_csr* g, const orc_leann_params* p, const float* vectors,
                     uint64_t nvec, size_t d, int copy_per_node, const float* query, size_t qd,
                     size_t k, size_t ef, uint64_t* out_ids, float* out_dist, size_t* out_count,
                     orc_counters* ctr, uint64_t* err_payload) {
  *out_count = 0;
  if (ctr) memset(ctr, 0, sizeof(*ctr));
  if (g->num_nodes == 0) return ORC_OK; /* :875-877 */
  if (p->has_dimension && qd != p->dimension) { /* :880-887 */
    if (err_payload) *err_payload = qd;
    return ORC_DIMENSION_MISMATCH;
  }
  if (!g->has_entry) return ORC_INDEX_NOT_BUILT; /* :889 */
  if (ef < k) ef = k;                            /* :890 */
  /* metric.calculate length check (distance.rs:39-44) against the provider rows */
  if (qd != d) {
    if (err_payload) *err_payload = d;
    return ORC_DIMENSION_MISMATCH;
  }
  item_t* r = NULL;
  size_t n = 0;
  int st = search_layer(g, p, vectors, nvec, d, copy_per_node, query, g->entry_point, ef, &r, &n,
                        ctr, err_payload);
  if (st) return st;
  size_t m = n < k ? n : k; /* :895 take(k) */
  for (size_t i = 0; i < m; i++) {
    out_ids[i] = r[i].id;
    out_dist[i] = r[i].d;
  }
  *out_count = m;
  free(r);
  return ORC_OK;
}

/* ---- LeannIndex::build, leann.rs:560-833 ---- */

typedef struct {
  uint64_t* v;
  size_t len, cap;
} vec64;
static void v_push(vec64* a, uint64_t x) {
  if (a->len == a->cap) {
    a->cap = a->cap ? a->cap * 2 : 8;
    a->v = (uint64_t*)realloc(a->v, a->cap * sizeof(uint64_t));
  }
  a->v[a->len++] = x;
}
static int v_contains(const vec64* a, uint64_t x) {
  for (size_t i = 0; i < a->len; i++)
    if (a->v[i] == x) return 1;
  return 0;
}

/* search_layer_with_adjacency, leann.rs:692-749 */
static void build_search(const float* vectors, size_t d, int metric, const vec64* adj,
                         const float* query, uint64_t entry, size_t ef, item_t** out,
                         size_t* out_n) {
  set_t visited;
  heap_t cand, res;
  set_init(&visited);
  heap_init(&cand, cmp_tuple_rev);
  heap_init(&res, cmp_tuple);
  float ed = metric_calc(metric, query, vectors + (size_t)entry * d, d);
  set_insert(&visited, entry);
  item_t e0 = {ed, entry};
  heap_push(&cand, e0);
  heap_push(&res, e0);
  item_t cur;
  while (heap_pop(&cand, &cur)) {
    if (res.len > 0 && res.len >= ef && of_cmp(cur.d, res.data[0].d) > 0) break;
    const vec64* nb = &adj[cur.id];
    for (size_t i = 0; i < nb->len; i++) {
      uint64_t nid = nb->v[i];
      if (!set_insert(&visited, nid)) continue;
      float nd = metric_calc(metric, query, vectors + (size_t)nid * d, d);
      int should_add = res.len < ef || (res.len == 0 || nd < res.data[0].d);
      if (should_add) {
        item_t it = {nd, nid};
        heap_push(&cand, it);
        heap_push(&res, it);
        if (res.len > ef) {
          item_t dr;
          heap_pop(&res, &dr);
        }
      }
    }
  }
  *out = (item_t*)malloc((res.len ? res.len : 1) * sizeof(item_t));
  memcpy(*out, res.data, res.len * sizeof(item_t));
  *out_n = res.len;
  stable_sort_by_dist(*out, *out_n);
  set_free(&visited);
  heap_free(&cand);
  heap_free(&res);
}

typedef struct {
  uint64_t id;
  float dist;
  uint64_t degree;
} hub_t;

/* prune_with_degree_preservation_temp, leann.rs:761-833.  cands: (id,dist),
 * sorted ascending by dist.  Writes <= max_conn ids to out. */
static size_t prune_degree_preserving(const item_t* cands, size_t n, const vec64* adj,
                                      size_t nadj, size_t max_conn, float hub_percentile,
                                      uint64_t* out) {
  if (n <= max_conn) {
    for (size_t i = 0; i < n; i++) out[i] = cands[i].id;
    return n;
  }
  uint64_t* degrees = (uint64_t*)malloc(n * sizeof(uint64_t));
  for (size_t i = 0; i < n; i++)
    degrees[i] = cands[i].id < nadj ? adj[cands[i].id].len : 0;
  /* sort_unstable_by descending: values only, so any correct sort matches */
  for (size_t i = 1; i < n; i++) {
    uint64_t x = degrees[i];
    size_t j = i;
    while (j > 0 && degrees[j - 1] < x) {
      degrees[j] = degrees[j - 1];
      j--;
    }
    degrees[j] = x;
  }
  size_t hub_count = (size_t)ceilf((float)n * hub_percentile); /* :780 */
  uint64_t thr = UINT64_MAX;
  if (hub_count > 0 && hub_count < n) thr = degrees[hub_count - 1]; /* :781-785 */
  free(degrees);

  hub_t* hubs = (hub_t*)malloc(n * sizeof(hub_t));
  item_t* regs = (item_t*)malloc(n * sizeof(item_t));
  size_t nh = 0, nr = 0;
  for (size_t i = 0; i < n; i++) {
    uint64_t deg = cands[i].id < nadj ? adj[cands[i].id].len : 0;
    if (deg >= thr && thr < UINT64_MAX) {
      hubs[nh].id = cands[i].id;
      hubs[nh].dist = cands[i].d;
      hubs[nh].degree = deg;
      nh++;
    } else {
      regs[nr++] = cands[i];
    }
  }
  /* hub_nodes.sort_by(|a,b| b.2.cmp(&a.2)): stable, descending degree (:800) */
  for (size_t i = 1; i < nh; i++) {
    hub_t x = hubs[i];
    size_t j = i;
    while (j > 0 && hubs[j - 1].degree < x.degree) {
      hubs[j] = hubs[j - 1];
      j--;
    }
    hubs[j] = x;
  }
  stable_sort_by_dist(regs, nr); /* :802 */

  size_t sel = 0;
  size_t hub_slots = max_conn / 4;
  if (hub_slots < 1) hub_slots = 1; /* :807 */
  for (size_t i = 0; i < nh && i < hub_slots; i++) out[sel++] = hubs[i].id; /* :808-810 */
  for (size_t i = 0; i < nr; i++) { /* :813-820 */
    if (sel >= max_conn) break;
    int dup = 0;
    for (size_t j = 0; j < sel; j++)
      if (out[j] == regs[i].id) dup = 1;
    if (!dup) out[sel++] = regs[i].id;
  }
  for (size_t i = hub_slots; i < nh; i++) { /* :823-830 */
    if (sel >= max_conn) break;
    int dup = 0;
    for (size_t j = 0; j < sel; j++)
      if (out[j] == hubs[i].id) dup = 1;
    if (!dup) out[sel++] = hubs[i].id;
  }
  free(hubs);
  free(regs);
  return sel;
}

int orc_leann_build(const float* vectors, uint64_t n, size_t d, const orc_build_params* bp,
                    const uint64_t* levels, orc_csr_owned* out) {
  memset(out, 0, sizeof(*out));
  if (n == 0) return ORC_OK; /* :565-567 */
  vec64* adj = (vec64*)calloc(n, sizeof(vec64));
  uint64_t* sel = (uint64_t*)malloc((bp->ef_construction + bp->m0 + 8) * sizeof(uint64_t));
  out->levels = (uint64_t*)malloc(n * sizeof(uint64_t));
  size_t nadj = 0; /* adjacency.len() */
  for (uint64_t id = 0; id < n; id++) {
    uint64_t level = levels ? levels[id] : 0;
    size_t nsel = 0;
    if (nadj > 0) { /* :585-589 find_neighbors_for_insert_temp */
      uint64_t entry = out->has_entry ? out->entry_point : 0; /* :669 */
      item_t* cands;
      size_t nc;
      build_search(vectors, d, bp->metric, adj, vectors + (size_t)id * d, entry,
                   bp->ef_construction, &cands, &nc);
      if (bp->high_degree_pruning) { /* :681-683 (adjacency non-empty here) */
        nsel = prune_degree_preserving(cands, nc, adj, nadj, bp->m0, bp->hub_percentile, sel);
      } else {
        nsel = nc < bp->m0 ? nc : bp->m0; /* :685 truncate */
        for (size_t i = 0; i < nsel; i++) sel[i] = cands[i].id;
      }
      free(cands);
    }
    /* :592 adjacency.push(neighbors.clone()) */
    for (size_t i = 0; i < nsel; i++) v_push(&adj[id], sel[i]);
    nadj++;
    for (size_t i = 0; i < nsel; i++) { /* :593-607 */
      uint64_t nid = sel[i];
      if (!v_contains(&adj[nid], id)) {
        v_push(&adj[nid], id);
        if (adj[nid].len > bp->m0) { /* prune_neighbors_temp :634-658 */
          size_t cnt = adj[nid].len;
          item_t* sc = (item_t*)malloc(cnt * sizeof(item_t));
          for (size_t j = 0; j < cnt; j++) {
            sc[j].id = adj[nid].v[j];
            sc[j].d = metric_calc(bp->metric, vectors + (size_t)nid * d,
                                  vectors + (size_t)sc[j].id * d, d);
          }
          stable_sort_by_dist(sc, cnt);
          size_t keep = cnt < bp->m0 ? cnt : bp->m0;
          adj[nid].len = 0;
          for (size_t j = 0; j < keep; j++) v_push(&adj[nid], sc[j].id);
          free(sc);
        }
      }
    }
    if (!out->has_entry || level > out->max_level) { /* :610-613 */
      out->has_entry = 1;
      out->entry_point = id;
      out->max_level = level;
    }
    out->levels[id] = level;
  }
  /* :617-627 flatten */
  out->num_nodes = n;
  out->node_offsets = (uint64_t*)malloc((n + 1) * sizeof(uint64_t));
  out->degree_counts = (uint64_t*)malloc(n * sizeof(uint64_t));
  size_t total = 0;
  for (uint64_t i = 0; i < n; i++) total += adj[i].len;
  out->neighbors = (uint64_t*)malloc((total ? total : 1) * sizeof(uint64_t));
  size_t off = 0;
  out->node_offsets[0] = 0;
  for (uint64_t i = 0; i < n; i++) {
    memcpy(out->neighbors + off, adj[i].v, adj[i].len * sizeof(uint64_t));
    off += adj[i].len;
    out->node_offsets[i + 1] = off;
    out->degree_counts[i] = adj[i].len;
    free(adj[i].v);
  }
  free(adj);
  free(sel);
  return ORC_OK;
}

void orc_csr_free(orc_csr_owned* g) {
  free(g->node_offsets);
  free(g->neighbors);
  free(g->degree_counts);
  free(g->levels);
  memset(g, 0, sizeof(*g));
}
void orc_free(void* p) { free(p); }

/* ------------------------------------------------------------------------- */
/* hnsw.rs                                                                   */
/* ------------------------------------------------------------------------- */

typedef struct {
  uint64_t level;
  vec64* conn; /* level+1 lists */
  float* vec;
} hnode_t;

struct orc_hnsw {
  uint64_t m, m0, ef_construction;
  int metric;
  hnode_t* nodes; /* HashMap<u64,HnswNode> with ids 0..len-1 (next_id, hnsw.rs:227-228) */
  size_t len, cap;
  int has_entry;
  uint64_t entry, max_level;
  int has_dim;
  size_t dim;
};

orc_hnsw* orc_hnsw_new(uint64_t m, uint64_t m0, uint64_t ef_construction, int metric) {
  orc_hnsw* h = (orc_hnsw*)calloc(1, sizeof(orc_hnsw));
  h->m = m;
  h->m0 = m0;
  h->ef_construction = ef_construction;
  h->metric = metric;
  return h;
}
void orc_hnsw_free(orc_hnsw* h) {
  if (!h) return;
  for (size_t i = 0; i < h->len; i++) {
    for (uint64_t l = 0; l <= h->nodes[i].level; l++) free(h->nodes[i].conn[l].v);
    free(h->nodes[i].conn);
    free(h->nodes[i].vec);
  }
  free(h->nodes);
  free(h);
}
uint64_t orc_hnsw_len(const orc_hnsw* h) { return h->len; }
uint64_t orc_hnsw_max_level(const orc_hnsw* h) { return h->max_level; }
int orc_hnsw_entry(const orc_hnsw* h, uint64_t* e) {
  if (!h->has_entry) return -1;
  *e = h->entry;
  return 0;
}
uint64_t orc_hnsw_level(const orc_hnsw* h, uint64_t node) { return h->nodes[node].level; }
const float* orc_hnsw_vector(const orc_hnsw* h, uint64_t node) { return h->nodes[node].vec; }
int orc_hnsw_neighbors(const orc_hnsw* h, uint64_t node, uint64_t layer, const uint64_t** ptr,
                       size_t* len) {
  if (node >= h->len || layer > h->nodes[node].level) return -1;
  *ptr = h->nodes[node].conn[layer].v;
  *len = h->nodes[node].conn[layer].len;
  return 0;
}

/* HnswGraph::distance, hnsw.rs:449-455 (node always exists for ids < len). */
static float hdist(const orc_hnsw* h, const float* q, uint64_t id) {
  return metric_calc(h->metric, q, h->nodes[id].vec, h->dim);
}

/* search_layer, hnsw.rs:332-402.  `vis` = nodes.get(): a node missing from the
 * map (id >= visible) is skipped at :363; during insert the new node is not in
 * the map yet. */
static void hnsw_search_layer(const orc_hnsw* h, const float* q, uint64_t entry, size_t ef,
                              uint64_t layer, item_t** out, size_t* out_n, orc_counters* c) {
  set_t visited;
  heap_t cand, res;
  set_init(&visited);
  heap_init(&cand, cmp_hnsw_cand);
  heap_init(&res, cmp_hnsw_cand_rev);
  float ed = hdist(h, q, entry);
  if (c) c->evals++;
  set_insert(&visited, entry);
  item_t e0 = {ed, entry};
  heap_push(&cand, e0);
  heap_push(&res, e0);
  item_t cur;
  while (heap_pop(&cand, &cur)) {
    if (res.len > 0 && of_cmp(cur.d, res.data[0].d) > 0 && res.len >= ef) break; /* :356-360 */
    if (cur.id >= h->len) continue;
    const hnode_t* node = &h->nodes[cur.id];
    if (layer > node->level) continue; /* neighbors_at(layer) None */
    const vec64* nb = &node->conn[layer];
    if (c) {
      c->expansions++;
      c->edges += nb->len;
    }
    for (size_t i = 0; i < nb->len; i++) {
      uint64_t nid = nb->v[i];
      if (!set_insert(&visited, nid)) continue;
      float nd = hdist(h, q, nid);
      if (c) c->evals++;
      int should_add = res.len < ef || (res.len == 0 || nd < res.data[0].d);
      if (should_add) {
        item_t it = {nd, nid};
        heap_push(&cand, it);
        heap_push(&res, it);
        if (c) c->pushes++;
        if (res.len > ef) {
          item_t dr;
          heap_pop(&res, &dr);
        }
      }
    }
  }
  *out = (item_t*)malloc((res.len ? res.len : 1) * sizeof(item_t));
  memcpy(*out, res.data, res.len * sizeof(item_t));
  *out_n = res.len;
  stable_sort_by_dist(*out, *out_n); /* :400 partial_cmp().unwrap(): NaN would panic */
  set_free(&visited);
  heap_free(&cand);
  heap_free(&res);
}

/* greedy descent loop shared by insert (hnsw.rs:263-282) and search (:478-497) */
static void hnsw_greedy(const orc_hnsw* h, const float* q, uint64_t layer, uint64_t* current,
                        float* current_dist, orc_counters* c) {
  for (;;) {
    int changed = 0;
    const hnode_t* node = &h->nodes[*current];
    if (layer <= node->level) {
      const vec64* nb = &node->conn[layer];
      /* note: the loop keeps iterating the ORIGINAL node's list after `current` moves */
      for (size_t i = 0; i < nb->len; i++) {
        float dist = hdist(h, q, nb->v[i]);
        if (c) c->evals++;
        if (dist < *current_dist) {
          *current = nb->v[i];
          *current_dist = dist;
          changed = 1;
        }
      }
    }
    if (!changed) break;
  }
}

/* prune_connections, hnsw.rs:405-446 */
static void hnsw_prune(orc_hnsw* h, uint64_t node_id, uint64_t layer, size_t max_conn) {
  hnode_t* node = &h->nodes[node_id];
  vec64* conns = &node->conn[layer];
  size_t cnt = 0;
  item_t* sc = (item_t*)malloc((conns->len ? conns->len : 1) * sizeof(item_t));
  for (size_t i = 0; i < conns->len; i++) {
    uint64_t id = conns->v[i];
    if (id >= h->len) continue; /* filter_map: nodes.get(&id) None (the node being inserted) */
    sc[cnt].id = id;
    sc[cnt].d = metric_calc(h->metric, node->vec, h->nodes[id].vec, h->dim);
    cnt++;
  }
  stable_sort_by_dist(sc, cnt);
  size_t keep = cnt < max_conn ? cnt : max_conn;
  conns->len = 0;
  for (size_t i = 0; i < keep; i++) v_push(conns, sc[i].id);
  free(sc);
}

int orc_hnsw_insert(orc_hnsw* h, const float* v, size_t d, uint64_t level, uint64_t* out_id) {
  if (h->has_dim) {
    if (d != h->dim) return ORC_DIMENSION_MISMATCH; /* :216-222 */
  } else {
    h->has_dim = 1;
    h->dim = d;
  }
  uint64_t id = h->len; /* next_id */
  if (h->len == h->cap) {
    h->cap = h->cap ? h->cap * 2 : 64;
    h->nodes = (hnode_t*)realloc(h->nodes, h->cap * sizeof(hnode_t));
  }
  hnode_t node;
  node.level = level;
  node.conn = (vec64*)calloc(level + 1, sizeof(vec64));
  node.vec = (float*)malloc(d * sizeof(float));
  memcpy(node.vec, v, d * sizeof(float));
  if (out_id) *out_id = id;

  if (!h->has_entry) { /* :240-245 */
    h->has_entry = 1;
    h->entry = id;
    h->max_level = level;
    h->nodes[h->len++] = node;
    return ORC_OK;
  }
  /* insert_node :254-329; the new node is NOT yet in the map (h->len unchanged) */
  uint64_t current = h->entry;
  float current_dist = metric_calc(h->metric, node.vec, h->nodes[current].vec, h->dim);
  for (uint64_t layer = h->max_level; layer >= level + 1 && layer <= h->max_level; layer--) {
    hnsw_greedy(h, node.vec, layer, &current, &current_dist, NULL);
    if (layer == 0) break;
  }
  for (uint64_t layer = level;; layer--) { /* (0..=node.level).rev() */
    /* layers above max_level: search_layer starts at `current`, whose node has no such
     * layer -> only the entry itself is returned (hnsw.rs:363-364 None). */
    item_t* nbs;
    size_t nn;
    hnsw_search_layer(h, node.vec, current, h->ef_construction, layer, &nbs, &nn, NULL);
    size_t m = layer == 0 ? h->m0 : h->m;
    size_t ns = nn < m ? nn : m;
    node.conn[layer].len = 0;
    for (size_t i = 0; i < ns; i++) v_push(&node.conn[layer], nbs[i].id); /* :295-300 */
    for (size_t i = 0; i < ns; i++) {                                      /* :303-313 */
      uint64_t nid = nbs[i].id;
      hnode_t* nb = &h->nodes[nid];
      if (layer <= nb->level) {
        v_push(&nb->conn[layer], id);
        if (nb->conn[layer].len > m) hnsw_prune(h, nid, layer, m);
      }
    }
    if (ns > 0) current = nbs[0].id; /* :316-318 */
    free(nbs);
    if (layer == 0) break;
  }
  if (level > h->max_level) { /* :322-325 */
    h->max_level = level;
    h->entry = id;
  }
  h->nodes[h->len++] = node;
  return ORC_OK;
}

int orc_hnsw_search(const orc_hnsw* h, const float* q, size_t qd, size_t k, size_t ef,
                    uint64_t* out_ids, float* out_dist, size_t* out_count, orc_counters* ctr) {
  *out_count = 0;
  if (ctr) memset(ctr, 0, sizeof(*ctr));
  if (h->len == 0) return ORC_OK;                              /* :459-461 */
  if (h->has_dim && qd != h->dim) return ORC_DIMENSION_MISMATCH; /* :464-471 */
  if (!h->has_entry) return ORC_INDEX_NOT_BUILT;
  uint64_t current = h->entry;
  float current_dist = hdist(h, q, current);
  if (ctr) ctr->evals++;
  for (uint64_t layer = h->max_level; layer >= 1; layer--) /* :478-497 */
    hnsw_greedy(h, q, layer, &current, &current_dist, ctr);
  if (ef < k) ef = k; /* :500 */
  item_t* r;
  size_t n;
  hnsw_search_layer(h, q, current, ef, 0, &r, &n, ctr);
  size_t m = n < k ? n : k;
  for (size_t i = 0; i < m; i++) {
    out_ids[i] = r[i].id;
    out_dist[i] = r[i].d;
  }
  *out_count = m;
  free(r);
  return ORC_OK;
}

/* ------------------------------------------------------------------------- */
/* search.rs / indexer/service.rs merges                                     */
/* ------------------------------------------------------------------------- */

float orc_to_similarity(float score) { return 1.0f / (1.0f + score); }

typedef struct {
  float s;
  uint64_t id;
  uint32_t src;
} merged_t;

static int merge_common(size_t nlists, const uint64_t* const* list_ids,
                        const float* const* list_vals, const size_t* list_len, size_t top_k,
                        int service_mode, uint64_t* out_ids, float* out_scores, uint32_t* out_src,
                        size_t* out_count) {
  size_t total = 0;
  for (size_t i = 0; i < nlists; i++) total += list_len[i];
  merged_t* all = (merged_t*)malloc((total ? total : 1) * sizeof(merged_t));
  merged_t* tmp = (merged_t*)malloc((total ? total : 1) * sizeof(merged_t));
  size_t n = 0;
  for (size_t i = 0; i < nlists; i++)
    for (size_t j = 0; j < list_len[i]; j++) {
      float v = list_vals[i][j];
      if (!service_mode && isnan(v) && total > 1) { /* search.rs:231 unwrap() panics on NaN */
        free(all);
        free(tmp);
        return ORC_PANIC;
      }
      all[n].s = service_mode ? 1.0f - v : v; /* service.rs:791 */
      all[n].id = list_ids[i][j];
      all[n].src = (uint32_t)i;
      n++;
    }
  /* stable merge sort; ascending score (search.rs:231) or descending (service.rs:800) */
  for (size_t w = 1; w < n; w *= 2) {
    for (size_t lo = 0; lo < n; lo += 2 * w) {
      size_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
      size_t i = lo, j = mid, k = lo;
      while (i < mid && j < hi) {
        int take_right = service_mode ? (all[j].s > all[i].s) : (all[j].s < all[i].s);
        if (take_right) tmp[k++] = all[j++];
        else tmp[k++] = all[i++];
      }
      while (i < mid) tmp[k++] = all[i++];
      while (j < hi) tmp[k++] = all[j++];
    }
    memcpy(all, tmp, n * sizeof(merged_t));
  }
  size_t m = n < top_k ? n : top_k;
  for (size_t i = 0; i < m; i++) {
    out_ids[i] = all[i].id;
    out_scores[i] = all[i].s;
    if (out_src) out_src[i] = all[i].src;
  }
  *out_count = m;
  free(all);
  free(tmp);
  return ORC_OK;
}

int orc_multi_index_merge(size_t nlists, const uint64_t* const* list_ids,
                          const float* const* list_scores, const size_t* list_len, size_t top_k,
                          uint64_t* out_ids, float* out_scores, uint32_t* out_src,
                          size_t* out_count) {
  return merge_common(nlists, list_ids, list_scores, list_len, top_k, 0, out_ids, out_scores,
                      out_src, out_count);
}
int orc_service_merge(size_t nlists, const uint64_t* const* list_ids,
                      const float* const* list_dist, const size_t* list_len, size_t top_k,
                      uint64_t* out_ids, float* out_scores, uint32_t* out_src, size_t* out_count) {
  return merge_common(nlists, list_ids, list_dist, list_len, top_k, 1, out_ids, out_scores,
                      out_src, out_count);
}

/* ------------------------------------------------------------------------- */
/* pq.rs                                                                     */
/* ------------------------------------------------------------------------- */

int orc_pq_find_nearest(int metric, const float* centroids, size_t K, size_t dsub,
                        const float* sub, size_t sublen, uint64_t* out) {
  if (sublen != dsub) return ORC_DIMENSION_MISMATCH; /* pq.rs:87-92 */
  size_t best_idx = 0;
  float best = 3.40282347e+38f; /* f32::MAX */
  for (size_t i = 0; i < K; i++) {
    float dist = metric_calc(metric, sub, centroids + i * dsub, dsub);
    if (dist < best) {
      best = dist;
      best_idx = i;
    }
  }
  *out = best_idx;
  return ORC_OK;
}

int orc_pq_encode(int metric, const float* codebooks, size_t m, size_t K, size_t dsub,
                  const float* v, size_t d, uint16_t* codes) {
  if (d != m * dsub) return ORC_DIMENSION_MISMATCH; /* pq.rs:225-230 */
  for (size_t j = 0; j < m; j++) {
    uint64_t c;
    orc_pq_find_nearest(metric, codebooks + j * K * dsub, K, dsub, v + j * dsub, dsub, &c);
    codes[j] = (uint16_t)c; /* `as u16` */
  }
  return ORC_OK;
}

int orc_pq_decode(const float* codebooks, size_t m, size_t K, size_t dsub, const uint16_t* codes,
                  size_t ncodes, float* out) {
  if (ncodes != m) return ORC_PQ_ERROR; /* pq.rs:251-257 */
  for (size_t j = 0; j < m; j++) {
    if (codes[j] >= K) return ORC_PQ_ERROR; /* :262-266 */
    memcpy(out + j * dsub, codebooks + (j * K + codes[j]) * dsub, dsub * sizeof(float));
  }
  return ORC_OK;
}

/* (a - b).powi(2) == diff * diff exactly (llvm.powi with constant 2). */
static float sub_sqdist(const float* q, const float* c, size_t n) {
  float s = 0.0f;
  for (size_t i = 0; i < n; i++) {
    float df = q[i] - c[i];
    s += df * df;
  }
  return s;
}

int orc_pq_asymmetric_distance(const float* codebooks, size_t m, size_t K, size_t dsub,
                               const float* q, size_t d, const uint16_t* codes, size_t ncodes,
                               float* out) {
  if (d != m * dsub) return ORC_DIMENSION_MISMATCH; /* pq.rs:276-281 */
  float total = 0.0f;
  for (size_t j = 0; j < ncodes; j++) { /* zip over codes: extra codes would index OOB -> panic */
    if (j >= m) return ORC_PANIC;
    if (codes[j] >= K) return ORC_PQ_ERROR; /* :290-292 */
    float sd = sub_sqdist(q + j * dsub, codebooks + (j * K + codes[j]) * dsub, dsub);
    total += sd;
  }
  *out = sqrtf(total);
  return ORC_OK;
}

int orc_pq_build_tables(const float* codebooks, size_t m, size_t K, size_t dsub, const float* q,
                        size_t d, float* tables) {
  if (d != m * dsub) return ORC_DIMENSION_MISMATCH; /* pq.rs:308-313 */
  for (size_t j = 0; j < m; j++)
    for (size_t c = 0; c < K; c++)
      tables[j * K + c] = sub_sqdist(q + j * dsub, codebooks + (j * K + c) * dsub, dsub);
  return ORC_OK;
}

float orc_pq_table_distance(const float* tables, size_t m, size_t K, const uint16_t* codes) {
  float s = 0.0f; /* .sum::<f32>() left fold, pq.rs:342-347 */
  for (size_t j = 0; j < m; j++) s += tables[j * K + codes[j]];
  return sqrtf(s);
}

/* ------------------------------------------------------------------------- */
/* EXTENSION: two-level search with a PQ filter                              */
/* ------------------------------------------------------------------------- */
/* The reference promises this search (leann.rs:54-56, :855-857) and specifies it
 * as pseudo-code only (docs/leann-specification.md:223-275, "Algorithm 2"); no
 * Rust implementation exists, so there is nothing to be bit-identical to.  This
 * is the definition the device path is tested against.  Where the pseudo-code is
 * silent the rule is stated here:
 *   - visited, EQ, R start with the entry point and its exact distance (lines 1-4);
 *   - every key is the total order (OrderedFloat distance, id) of leann.rs:907-908;
 *     a distance is canonicalised before use: -0.0 -> +0.0, NaN -> bits 0x7FFFFFFF;
 *   - EQ holds exactly the members of R that were not expanded yet: an entry
 *     evicted from R (line 27) is farther than the worst result and would hit the
 *     break of lines 8-9 when popped; the loop ends when R has no unexpanded
 *     member (lines 5-9 under the total order);
 *   - AQ keeps every node ever given an approximate distance, promoted or not
 *     (that is what makes line 22's "if m not in EQ" meaningful);
 *     d_approx = ProductQuantizer::table_distance (pq.rs:341-348) on the tables of
 *     build_distance_tables (pq.rs:307-338);
 *   - line 19: M = the first ceil(a * |AQ|) entries of AQ in ascending
 *     (d_approx, id) order, the product taken in f32, at least one entry when AQ is
 *     not empty; the members of M not promoted before are promoted in that order;
 *   - exact distances come from DistanceMetric::calculate on the provider's row;
 *   - apply_pruning_strategy is not part of Algorithm 2 and is not applied;
 *   - results: R ascending in (distance, id), first k.
 * Counters: expansions, edges as in orc_leann_search; evals = exact distance
 * evaluations (entry included); pushes = approximate (table) evaluations.
 * Errors follow search_with_params (leann.rs:868-896); a neighbour without a
 * code row, or a promoted id without an embedding row, is NodeNotFound. */
static float tl_canon(float d) {
  if (d != d) {
    union { uint32_t u; float f; } c;
    c.u = 0x7FFFFFFFu;
    return c.f;
  }
  return d == 0.0f ? 0.0f : d;
}

typedef struct {
  item_t it;
  int flag; /* R: expanded; AQ: promoted */
} tl_item;

/* inserts into an ascending array (cmp_tuple order); returns the new length */
static size_t tl_insert(tl_item* a, size_t n, item_t it) {
  size_t pos = n;
  while (pos > 0 && cmp_tuple(&it, &a[pos - 1].it) < 0) {
    a[pos] = a[pos - 1];
    pos--;
  }
  a[pos].it = it;
  a[pos].flag = 0;
  return n + 1;
}

int orc_two_level_search(const orc_csr* g, const orc_leann_params* p, const float* vectors,
                         uint64_t nvec, size_t d, const float* codebooks, size_t m, size_t K,
                         size_t dsub, const uint16_t* codes, uint64_t ncodes, const float* query,
                         size_t qd, size_t k, size_t ef, float rerank_ratio, uint64_t* out_ids,
                         float* out_dist, size_t* out_count, orc_counters* ctr,
                         uint64_t* err_payload) {
  *out_count = 0;
  orc_counters c = {0, 0, 0, 0};
  if (ctr) *ctr = c;
  if (g->num_nodes == 0) return ORC_OK; /* leann.rs:875-877 */
  if (p->has_dimension && qd != p->dimension) { /* :880-887 */
    if (err_payload) *err_payload = qd;
    return ORC_DIMENSION_MISMATCH;
  }
  if (!g->has_entry) return ORC_INDEX_NOT_BUILT; /* :889 */
  if (ef < k) ef = k;                            /* :890 */
  if (qd != d || qd != m * dsub) { /* distance.rs:39-44, pq.rs:308-313 */
    if (err_payload) *err_payload = qd != d ? d : m * dsub;
    return ORC_DIMENSION_MISMATCH;
  }
  int status = ORC_OK;
  float* tables = (float*)malloc(m * K * sizeof(float));
  orc_pq_build_tables(codebooks, m, K, dsub, query, qd, tables);
  set_t visited;
  set_init(&visited);
  tl_item* R = (tl_item*)malloc((ef + 2) * sizeof(tl_item));
  size_t rlen = 0, aqlen = 0, aqcap = 1024;
  tl_item* AQ = (tl_item*)malloc(aqcap * sizeof(tl_item));

  uint64_t entry = g->entry_point;
  if (entry >= nvec) {
    if (err_payload) *err_payload = entry;
    status = ORC_NODE_NOT_FOUND;
    goto done;
  }
  {
    float ed;
    orc_distance(p->metric, query, d, vectors + (size_t)entry * d, d, &ed);
    c.evals = 1;
    set_insert(&visited, entry);
    item_t it;
    it.d = tl_canon(ed);
    it.id = entry;
    rlen = tl_insert(R, rlen, it);
  }
  for (;;) {
    size_t e = 0;
    while (e < rlen && R[e].flag) e++;
    if (e == rlen) break; /* lines 5-9 */
    R[e].flag = 1;
    uint64_t v = R[e].it.id;
    const uint64_t* nb;
    size_t deg;
    if (orc_csr_get_neighbors(g, v, &nb, &deg) != 0) continue; /* leann.rs:227-229 */
    c.expansions++;
    c.edges += deg;
    for (size_t i = 0; i < deg; i++) { /* lines 12-16 */
      uint64_t n = nb[i];
      if (!set_insert(&visited, n)) continue;
      if (n >= ncodes) {
        if (err_payload) *err_payload = n;
        status = ORC_NODE_NOT_FOUND;
        goto done;
      }
      item_t it;
      it.d = tl_canon(orc_pq_table_distance(tables, m, K, codes + (size_t)n * m));
      it.id = n;
      c.pushes++;
      if (aqlen + 1 >= aqcap) {
        aqcap *= 2;
        AQ = (tl_item*)realloc(AQ, aqcap * sizeof(tl_item));
      }
      aqlen = tl_insert(AQ, aqlen, it);
    }
    if (aqlen == 0) continue;
    float tf = ceilf(rerank_ratio * (float)aqlen); /* line 19 */
    size_t ntop = tf >= 1.0f ? (size_t)tf : 1; /* NaN and negatives -> 1 */
    if (tf >= (float)aqlen) ntop = aqlen;
    if (ntop > aqlen) ntop = aqlen;
    for (size_t i = 0; i < ntop; i++) { /* lines 21-27 */
      if (AQ[i].flag) continue;
      AQ[i].flag = 1;
      uint64_t mm = AQ[i].it.id;
      if (mm >= nvec) {
        if (err_payload) *err_payload = mm;
        status = ORC_NODE_NOT_FOUND;
        goto done;
      }
      float dd;
      orc_distance(p->metric, query, d, vectors + (size_t)mm * d, d, &dd);
      c.evals++;
      item_t it;
      it.d = tl_canon(dd);
      it.id = mm;
      rlen = tl_insert(R, rlen, it);
      if (rlen > ef) rlen = ef;
    }
  }
  {
    size_t n = rlen < k ? rlen : k;
    for (size_t i = 0; i < n; i++) {
      out_ids[i] = R[i].it.id;
      out_dist[i] = R[i].it.d;
    }
    *out_count = n;
  }
done:
  if (ctr) *ctr = c;
  free(tables);
  free(R);
  free(AQ);
  set_free(&visited);
  return status;
}

/* ------------------------------------------------------------------------- */
/* embedding/candle_provider.rs:434-488 (islands' own pooling code)          */
/* ------------------------------------------------------------------------- */
void orc_mean_pool_normalize(const float* hidden, const float* mask, size_t B, size_t L, size_t H,
                             int normalize, float* out) {
  for (size_t b = 0; b < B; b++) {
    float sum_mask = 0.0f;
    for (size_t t = 0; t < L; t++) sum_mask += mask[b * L + t];
    if (sum_mask < 1e-9f) sum_mask = 1e-9f; /* clamp(1e-9, MAX) :455-459 */
    for (size_t h = 0; h < H; h++) {
      float s = 0.0f;
      for (size_t t = 0; t < L; t++) s += hidden[(b * L + t) * H + h] * mask[b * L + t];
      out[b * H + h] = s / sum_mask; /* :462-463 */
    }
    if (normalize) { /* :466-481 */
      float ss = 0.0f;
      for (size_t h = 0; h < H; h++) ss += out[b * H + h] * out[b * H + h];
      float norm = sqrtf(ss);
      if (norm < 1e-12f) norm = 1e-12f;
      for (size_t h = 0; h < H; h++) out[b * H + h] /= norm;
    }
  }
}
