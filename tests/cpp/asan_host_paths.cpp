// Host-only paths of libislands_amd.so under AddressSanitizer (CPU build; GPU ASan is not
// available on this pool): the bincode readers of LeannIndex (leann.rs:1059-1066) and HnswGraph
// (hnsw.rs:511-514), the chunk framing of storage.rs:113-174, the host CSR accessors and the
// record-size arithmetic of the shard exchange.  Valid images must round-trip; every truncation,
// a few thousand seeded byte flips and every 8-byte field overwritten with wrapping lengths must
// come back as a status code -- never as a read or write outside a buffer (ASan aborts on those).
// Built by `make -C islands_amd/csrc asan` from the same sources as the library (host code
// instrumented, -fno-gpu-sanitize), linked without the device-side translation units: no entry
// point used here reaches them without a device.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "islands_amd.h"

static int failures = 0;
static int seen_status[128];
// any CoreError / ABI status is an answer; what must not happen is a crash or an ASan report
static bool is_status(isl_status st) {
  if (st >= 0 && st < 128) seen_status[st]++;
  return st >= 0 && (st <= ISL_ERR_EMBEDDING || (st >= 100 && st < 128));
}
#define EXPECT(cond)                                                                            \
  do {                                                                                          \
    if (!(cond)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); failures++; }     \
  } while (0)

static void put64(std::vector<uint8_t>& b, uint64_t v) { for (int i = 0; i < 8; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
static void put32(std::vector<uint8_t>& b, uint32_t v) { for (int i = 0; i < 4; ++i) b.push_back((uint8_t)(v >> (8 * i))); }
static void putf64(std::vector<uint8_t>& b, double d) { uint64_t v; memcpy(&v, &d, 8); put64(b, v); }

// bincode 1.x image of a small HnswGraph (hnsw.rs:15-28, 90-99, 150-164), nodes in `order`
static std::vector<uint8_t> hnsw_image(size_t n, size_t d, unsigned seed) {
  std::mt19937 rng(seed);
  std::uniform_real_distribution<float> u(-1.f, 1.f);
  std::vector<uint8_t> b;
  put64(b, 8); put64(b, 16); put64(b, 40); putf64(b, 0.48); put32(b, 1); put64(b, 16);
  put64(b, n);
  std::vector<size_t> order(n);
  for (size_t i = 0; i < n; ++i) order[i] = i;
  std::shuffle(order.begin(), order.end(), rng);
  for (size_t i : order) {
    put64(b, i); put64(b, i); put64(b, d);
    for (size_t j = 0; j < d; ++j) { float f = u(rng); uint32_t w; memcpy(&w, &f, 4); put32(b, w); }
    const uint64_t level = i == 0 ? 1 : 0;
    put64(b, level + 1);
    for (uint64_t L = 0; L <= level; ++L) {
      const uint64_t c = L == 0 ? (n > 1 ? 2 : 0) : 0;
      put64(b, c);
      for (uint64_t t = 0; t < c; ++t) put64(b, (i + 1 + t) % n);
    }
    put64(b, level);
  }
  b.push_back(1); put64(b, 0);   // entry_point: Some(0)
  put64(b, 1);                   // max_level
  b.push_back(1); put64(b, d);   // dimension: Some(d)
  put64(b, n);                   // next_id
  return b;
}

static std::vector<uint8_t> leann_image(size_t n, unsigned seed) {
  std::mt19937 rng(seed);
  std::vector<uint64_t> off(1, 0), nb, levels(n, 0), deg(n, 0);
  for (size_t i = 0; i < n; ++i) {
    const size_t c = rng() % 5;
    for (size_t t = 0; t < c; ++t) nb.push_back(rng() % n);
    off.push_back(nb.size());
    deg[i] = c;
  }
  isl_leann_config cfg;
  isl_leann_config_paper_default(&cfg);
  isl_index* idx = nullptr;
  EXPECT(isl_index_from_csr(&cfg, n, off.data(), nb.data(), levels.data(), deg.data(), 1, 0, 0, 1, 24, &idx) == ISL_OK);
  // CsrGraph::get_neighbors incl. out of range (leann.rs:225-233)
  const uint64_t* p = nullptr; size_t len = 0;
  EXPECT(isl_index_get_neighbors(idx, n - 1, &p, &len) == 1 && len == deg[n - 1]);
  EXPECT(isl_index_get_neighbors(idx, n, &p, &len) == 0);
  EXPECT(isl_index_get_neighbors(idx, ~0ull, &p, &len) == 0);
  uint8_t* bytes = nullptr; size_t bl = 0;
  EXPECT(isl_index_to_bytes(idx, &bytes, &bl) == ISL_OK);
  std::vector<uint8_t> out(bytes, bytes + bl);
  isl_free_bytes(bytes);
  isl_index_free(idx);
  return out;
}

template <class F>
static void mutate(const std::vector<uint8_t>& good, unsigned seed, F parse) {
  // every truncation (dense near the ends, strided in the middle of a long image)
  for (size_t cut = 0; cut < good.size(); cut += (cut < 256 || good.size() - cut < 256) ? 1 : 7) {
    std::vector<uint8_t> t(good.begin(), good.begin() + cut);  // exact-size heap block: any over-read is caught
    parse(t);
  }
  std::vector<uint8_t> longer(good);
  longer.push_back(0);
  parse(longer);
  // every aligned and unaligned 8-byte field replaced by lengths that wrap when scaled
  const uint64_t evil[] = {~0ull, 1ull << 63, (1ull << 62) + 1, (1ull << 61), 0x2000000000000001ull, 0xFFFFFFFFull,
                           good.size(), good.size() / 4 + 1};
  for (size_t off = 0; off + 8 <= good.size(); off += (good.size() > 4096 ? 4 : 1))
    for (uint64_t e : evil) {
      std::vector<uint8_t> t(good);
      memcpy(t.data() + off, &e, 8);
      parse(t);
      if (good.size() > 4096 && off > 512) break;  // one value per offset far into a long image
    }
  std::mt19937 rng(seed);
  for (int it = 0; it < 4000; ++it) {
    std::vector<uint8_t> t(good);
    const int flips = 1 + rng() % 4;
    for (int f = 0; f < flips; ++f) t[rng() % t.size()] ^= (uint8_t)(1u << (rng() % 8));
    parse(t);
  }
}

int main() {
  // ---- LeannIndex bytes ----
  for (size_t n : {1u, 7u, 60u}) {
    const std::vector<uint8_t> good = leann_image(n, (unsigned)n);
    isl_index* idx = nullptr;
    EXPECT(isl_index_from_bytes(good.data(), good.size(), &idx) == ISL_OK && isl_index_len(idx) == n);
    uint8_t* again = nullptr; size_t al = 0;
    EXPECT(isl_index_to_bytes(idx, &again, &al) == ISL_OK && al == good.size() && !memcmp(again, good.data(), al));
    isl_free_bytes(again);
    isl_index_free(idx);
    size_t ok = 0, bad = 0;
    mutate(good, 100 + (unsigned)n, [&](const std::vector<uint8_t>& t) {
      isl_index* x = nullptr;
      const isl_status st = isl_index_from_bytes(t.data(), t.size(), &x);
      if (st == ISL_OK) {
        // whatever parsed must be usable: accessors and a re-serialisation stay inside their buffers
        const uint64_t nn = isl_index_len(x);
        const uint64_t* p = nullptr; size_t l = 0;
        for (uint64_t i = 0; i < nn && i < 64; ++i) (void)isl_index_get_neighbors(x, i, &p, &l);
        uint8_t* b2 = nullptr; size_t l2 = 0;
        if (isl_index_to_bytes(x, &b2, &l2) == ISL_OK) isl_free_bytes(b2);
        isl_index_free(x);
        ok++;
      } else {
        EXPECT(is_status(st) && st != ISL_ERR_DEVICE);  // the LeannIndex reader never needs a device
        EXPECT(x == nullptr);
        bad++;
      }
    });
    std::printf("leann n=%zu: %zu mutants parsed, %zu rejected\n", n, ok, bad);
  }
  // ---- HnswGraph bytes: parsed completely before anything touches a device ----
  for (size_t n : {1u, 5u, 33u}) {
    const std::vector<uint8_t> good = hnsw_image(n, 6, (unsigned)n);
    isl_hnsw* h = nullptr;
    const isl_status st0 = isl_hnsw_from_bytes(good.data(), good.size(), 0, &h);
    EXPECT(st0 == ISL_OK || st0 == ISL_ERR_DEVICE);  // no card here: the upload is what fails
    if (h) isl_hnsw_free(h);
    size_t dev = 0, bad = 0;
    mutate(good, 200 + (unsigned)n, [&](const std::vector<uint8_t>& t) {
      isl_hnsw* x = nullptr;
      const isl_status st = isl_hnsw_from_bytes(t.data(), t.size(), 0, &x);
      if (x) isl_hnsw_free(x);
      if (st == ISL_OK || st == ISL_ERR_DEVICE) dev++; else bad++;
      EXPECT(is_status(st));
    });
    std::printf("hnsw n=%zu: %zu mutants got as far as the upload, %zu rejected\n", n, dev, bad);
  }
  // ---- storage.rs chunk framing ----
  {
    isl_index_metadata m;
    isl_index_metadata_new(1234, 768, 1700000000, &m);
    m.has_description = 1;
    snprintf(m.description, sizeof m.description, "a \"quoted\" description \\ with escapes");
    uint8_t* b = nullptr; size_t bl = 0;
    EXPECT(isl_storage_write_metadata(&m, &b, &bl) == ISL_OK);
    std::vector<uint8_t> good(b, b + bl);
    isl_free_bytes(b);
    isl_index_metadata r; size_t used = 0;
    EXPECT(isl_storage_read_metadata(good.data(), good.size(), &r, &used) == ISL_OK && used == good.size());
    EXPECT(r.num_vectors == 1234 && r.dimension == 768 && r.has_description && !strcmp(r.description, m.description));
    mutate(good, 300, [&](const std::vector<uint8_t>& t) {
      isl_index_metadata x; size_t u = 0;
      const isl_status st = t.empty() ? isl_storage_read_metadata((const uint8_t*)"", 0, &x, &u)
                                      : isl_storage_read_metadata(t.data(), t.size(), &x, &u);
      EXPECT(st == ISL_OK || st == ISL_ERR_IO || st == ISL_ERR_DESERIALIZATION);
      if (st == ISL_OK) EXPECT(u <= t.size());
    });
    // one-file persistence: save, load, and load of damaged files
    const std::vector<uint8_t> li = leann_image(20, 5);
    isl_index* idx = nullptr;
    EXPECT(isl_index_from_bytes(li.data(), li.size(), &idx) == ISL_OK);
    std::string dir = "/tmp/isl_asan_XXXXXX";
    EXPECT(mkdtemp(&dir[0]) != nullptr);
    const std::string path = dir + "/sub/dir/index.bin";
    EXPECT(isl_index_save(idx, path.c_str(), &m) == ISL_OK);
    isl_index_free(idx);
    isl_index* back = nullptr; isl_index_metadata mb;
    EXPECT(isl_index_load(path.c_str(), &back, &mb) == ISL_OK && isl_index_len(back) == 20 && mb.num_vectors == 1234);
    isl_index_free(back);
    FILE* f = fopen(path.c_str(), "rb");
    std::vector<uint8_t> file;
    uint8_t buf[4096]; size_t rd;
    while ((rd = fread(buf, 1, sizeof buf, f)) > 0) file.insert(file.end(), buf, buf + rd);
    fclose(f);
    const std::string p2 = dir + "/damaged.bin";
    mutate(file, 400, [&](const std::vector<uint8_t>& t) {
      static int every = 0;
      if (++every % 5) return;  // (files are slow: every fifth mutant)
      FILE* o = fopen(p2.c_str(), "wb");
      if (!t.empty()) fwrite(t.data(), 1, t.size(), o);
      fclose(o);
      isl_index* x = nullptr;
      const isl_status st = isl_index_load(p2.c_str(), &x, nullptr);
      if (x) isl_index_free(x);
      EXPECT(is_status(st) && st != ISL_ERR_DEVICE);
    });
    EXPECT(isl_index_load((dir + "/missing.bin").c_str(), &back, nullptr) == ISL_ERR_IO);
    remove(p2.c_str());
    remove(path.c_str());
  }
  // ---- staging arithmetic of the shard exchange ----
  for (uint64_t nq : {1ull, 3ull, 64ull, 1024ull, 4097ull})
    for (uint64_t k : {1ull, 7ull, 10ull, 100ull}) {
      const uint64_t B = isl_shard_record_bytes(nq, k);
      EXPECT(B % 16 == 0 && B >= nq * k * 12 + nq * 4 && B < nq * k * 12 + nq * 4 + 16);
    }
  // config validation never reads past the struct
  isl_leann_config c;
  isl_leann_config_fast(&c); EXPECT(isl_leann_config_validate(&c) == ISL_OK);
  isl_leann_config_accurate(&c); EXPECT(isl_leann_config_validate(&c) == ISL_OK);
  c.m = 0; EXPECT(isl_leann_config_validate(&c) == ISL_ERR_INVALID_CONFIG);
  std::printf("statuses seen:");
  for (int i = 0; i < 128; ++i)
    if (seen_status[i]) std::printf(" %s x%d", isl_status_name(i), seen_status[i]);
  std::printf("\n");
  std::printf(failures ? "asan host paths: %d FAILURES\n" : "asan host paths: ok\n", failures);
  return failures ? 1 : 0;
}
