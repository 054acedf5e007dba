// Index persistence, src/core/storage.rs: IndexMetadata (:16-47), the chunk framing of
// IndexWriter / IndexReader (tag[4] || u64 LE length || payload, :127-135 and :159-173) with the
// serde_json META chunk (:119-124, :149-156) and FileSystemStorage::save / load (:68-80).
// Host code only.  The reference writes nothing but the META chunk through this framing (its
// write_chunk is private and has one caller); the index itself travels as LeannIndex::to_bytes
// (leann.rs:1059).  isl_index_save puts both in one file: chunk "META", then chunk "LIDX" with
// those bytes -- the second tag is this library's choice, a reader that only wants the metadata
// (IndexReader::read_metadata) is served by the reference's own code.
#include "common.hpp"

#include <cerrno>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <sys/stat.h>
#include <vector>

namespace {

using isl::fail;

void put_chunk(std::vector<uint8_t>& out, const char tag[4], const uint8_t* data, uint64_t len) {
  out.insert(out.end(), tag, tag + 4);
  const uint8_t* lp = reinterpret_cast<const uint8_t*>(&len);  // little-endian host
  out.insert(out.end(), lp, lp + 8);
  out.insert(out.end(), data, data + len);
}

// serde_json string escaping (serde_json::ser::format_escaped_str): \" \\ \b \f \n \r \t,
// other control characters as \u00XX, everything else (UTF-8 included) verbatim.
void json_string(std::string& o, const char* s) {
  o.push_back('"');
  for (const unsigned char* p = (const unsigned char*)s; *p; ++p) {
    switch (*p) {
      case '"': o += "\\\""; break;
      case '\\': o += "\\\\"; break;
      case '\b': o += "\\b"; break;
      case '\f': o += "\\f"; break;
      case '\n': o += "\\n"; break;
      case '\r': o += "\\r"; break;
      case '\t': o += "\\t"; break;
      default:
        if (*p < 0x20) {
          char buf[8];
          snprintf(buf, sizeof buf, "\\u%04x", *p);
          o += buf;
        } else {
          o.push_back((char)*p);
        }
    }
  }
  o.push_back('"');
}

// serde_json::to_vec(&IndexMetadata): compact, fields in declaration order (storage.rs:16-29)
std::string metadata_json(const isl_index_metadata& m) {
  std::string o = "{\"version\":" + std::to_string(m.version) +
                  ",\"num_vectors\":" + std::to_string(m.num_vectors) +
                  ",\"dimension\":" + std::to_string(m.dimension) +
                  ",\"created_at\":" + std::to_string(m.created_at) +
                  ",\"updated_at\":" + std::to_string(m.updated_at) + ",\"description\":";
  if (m.has_description) json_string(o, m.description);
  else o += "null";
  o += "}";
  return o;
}

// Minimal JSON reader for the one object serde_json::from_slice::<IndexMetadata> accepts: any
// key order, whitespace, unknown keys ignored (serde's default), missing required key = error.
struct Json {
  const char* p;
  const char* e;
  bool ok = true;
  void ws() { while (p < e && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
  bool lit(const char* s) {
    size_t n = strlen(s);
    if ((size_t)(e - p) >= n && !memcmp(p, s, n)) { p += n; return true; }
    return false;
  }
  bool str(std::string& out) {
    ws();
    if (p >= e || *p != '"') return ok = false;
    ++p;
    out.clear();
    while (p < e && *p != '"') {
      if (*p == '\\') {
        if (++p >= e) return ok = false;
        switch (*p) {
          case '"': out.push_back('"'); break;
          case '\\': out.push_back('\\'); break;
          case '/': out.push_back('/'); break;
          case 'b': out.push_back('\b'); break;
          case 'f': out.push_back('\f'); break;
          case 'n': out.push_back('\n'); break;
          case 'r': out.push_back('\r'); break;
          case 't': out.push_back('\t'); break;
          case 'u': {
            if (e - p < 5) return ok = false;
            unsigned v = (unsigned)strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
            p += 4;
            if (v < 0x80) out.push_back((char)v);
            else if (v < 0x800) { out.push_back((char)(0xC0 | (v >> 6))); out.push_back((char)(0x80 | (v & 0x3F))); }
            else { out.push_back((char)(0xE0 | (v >> 12))); out.push_back((char)(0x80 | ((v >> 6) & 0x3F))); out.push_back((char)(0x80 | (v & 0x3F))); }
            break;
          }
          default: return ok = false;
        }
        ++p;
      } else {
        out.push_back(*p++);
      }
    }
    if (p >= e) return ok = false;
    ++p;
    return true;
  }
  bool integer(long long& v) {
    ws();
    const char* s = p;
    if (p < e && *p == '-') ++p;
    while (p < e && *p >= '0' && *p <= '9') ++p;
    if (p == s || (p < e && (*p == '.' || *p == 'e' || *p == 'E'))) return ok = false;
    v = strtoll(std::string(s, p).c_str(), nullptr, 10);
    return true;
  }
  bool skip_value() {  // unknown key: scalars, strings, flat arrays / objects
    ws();
    if (p >= e) return ok = false;
    if (*p == '"') { std::string t; return str(t); }
    if (*p == '{' || *p == '[') {
      int depth = 0;
      bool in_str = false;
      for (; p < e; ++p) {
        if (in_str) { if (*p == '\\') ++p; else if (*p == '"') in_str = false; continue; }
        if (*p == '"') in_str = true;
        else if (*p == '{' || *p == '[') ++depth;
        else if (*p == '}' || *p == ']') { if (--depth == 0) { ++p; return true; } }
      }
      return ok = false;
    }
    while (p < e && *p != ',' && *p != '}') ++p;
    return true;
  }
};

isl_status parse_metadata(const uint8_t* data, size_t len, isl_index_metadata* m) {
  Json j{(const char*)data, (const char*)data + len};
  memset(m, 0, sizeof *m);
  bool seen[6] = {false, false, false, false, false, false};
  j.ws();
  if (!j.lit("{")) return fail(ISL_ERR_DESERIALIZATION, "Deserialization error: expected a JSON object");
  j.ws();
  if (!j.lit("}")) {
    for (;;) {
      std::string key;
      if (!j.str(key)) break;
      j.ws();
      if (!j.lit(":")) { j.ok = false; break; }
      long long v = 0;
      if (key == "version") { if (j.integer(v)) { m->version = (uint32_t)v; seen[0] = true; } }
      else if (key == "num_vectors") { if (j.integer(v)) { m->num_vectors = (uint64_t)v; seen[1] = true; } }
      else if (key == "dimension") { if (j.integer(v)) { m->dimension = (uint64_t)v; seen[2] = true; } }
      else if (key == "created_at") { if (j.integer(v)) { m->created_at = v; seen[3] = true; } }
      else if (key == "updated_at") { if (j.integer(v)) { m->updated_at = v; seen[4] = true; } }
      else if (key == "description") {
        j.ws();
        if (j.lit("null")) { m->has_description = 0; seen[5] = true; }
        else {
          std::string d;
          if (j.str(d)) {
            if (d.size() >= sizeof m->description)
              return fail(ISL_ERR_UNSUPPORTED, "description longer than %zu bytes", sizeof m->description - 1);
            memcpy(m->description, d.data(), d.size());
            m->has_description = 1;
            seen[5] = true;
          }
        }
      } else {
        j.skip_value();
      }
      if (!j.ok) break;
      j.ws();
      if (j.lit(",")) continue;
      if (j.lit("}")) break;
      j.ok = false;
      break;
    }
  }
  if (!j.ok) return fail(ISL_ERR_DESERIALIZATION, "Deserialization error: malformed metadata JSON");
  static const char* names[5] = {"version", "num_vectors", "dimension", "created_at", "updated_at"};
  for (int i = 0; i < 5; i++)
    if (!seen[i]) return fail(ISL_ERR_DESERIALIZATION, "Deserialization error: missing field `%s`", names[i]);
  return ISL_OK;  // Option<String> may be absent (serde treats a missing Option as None)
}

// fs::create_dir_all(parent), storage.rs:69-71 / :107-109
void make_parents(const std::string& path) {
  for (size_t i = 1; i < path.size(); ++i)
    if (path[i] == '/') (void)mkdir(path.substr(0, i).c_str(), 0777);
}

}  // namespace

extern "C" {

void isl_index_metadata_new(uint64_t num_vectors, uint64_t dimension, int64_t now, isl_index_metadata* out) {
  if (!out) return;
  memset(out, 0, sizeof *out);
  out->version = 1;  // IndexMetadata::CURRENT_VERSION, storage.rs:33
  out->num_vectors = num_vectors;
  out->dimension = dimension;
  out->created_at = now;
  out->updated_at = now;
}

isl_status isl_storage_write_metadata(const isl_index_metadata* meta, uint8_t** out, size_t* len) {
  if (!meta || !out || !len) return fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  std::string js = metadata_json(*meta);
  std::vector<uint8_t> b;
  put_chunk(b, "META", (const uint8_t*)js.data(), js.size());
  uint8_t* p = (uint8_t*)malloc(b.size());
  if (!p) return fail(ISL_ERR_SERIALIZATION, "Serialization error: out of memory");
  memcpy(p, b.data(), b.size());
  *out = p;
  *len = b.size();
  return ISL_OK;
}

isl_status isl_storage_read_metadata(const uint8_t* bytes, size_t len, isl_index_metadata* meta,
                                     size_t* consumed) {
  if (!bytes || !meta) return fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  if (len < 12) return fail(ISL_ERR_IO, "IO error: failed to fill whole buffer");  // read_exact
  uint64_t n = 0;
  memcpy(&n, bytes + 4, 8);
  if (n > len - 12) return fail(ISL_ERR_IO, "IO error: failed to fill whole buffer");
  if (memcmp(bytes, "META", 4)) return fail(ISL_ERR_DESERIALIZATION, "Deserialization error: expected META chunk");
  ISL_TRY(parse_metadata(bytes + 12, (size_t)n, meta));
  if (consumed) *consumed = 12 + (size_t)n;
  return ISL_OK;
}

isl_status isl_index_save(const isl_index* idx, const char* path, const isl_index_metadata* meta) {
  if (!idx || !path) return fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  isl_index_metadata m;
  if (meta) m = *meta;
  else {
    uint64_t dim = 0;
    (void)isl_index_dimension(idx, &dim);
    isl_index_metadata_new(isl_index_len(idx), dim, (int64_t)time(nullptr), &m);
  }
  uint8_t* ib = nullptr;
  size_t il = 0;
  ISL_TRY(isl_index_to_bytes(idx, &ib, &il));
  std::string js = metadata_json(m);
  std::vector<uint8_t> b;
  put_chunk(b, "META", (const uint8_t*)js.data(), js.size());
  put_chunk(b, "LIDX", ib, il);
  isl_free_bytes(ib);
  make_parents(path);
  FILE* f = fopen(path, "wb");
  if (!f) return fail(ISL_ERR_IO, "IO error: %s: %s", path, strerror(errno));
  size_t w = fwrite(b.data(), 1, b.size(), f);
  int rc = fclose(f);
  if (w != b.size() || rc != 0) return fail(ISL_ERR_IO, "IO error: short write to %s", path);
  return ISL_OK;
}

isl_status isl_index_load(const char* path, isl_index** out, isl_index_metadata* meta) {
  if (!path || !out) return fail(ISL_ERR_INVALID_ARGUMENT, "NULL argument");
  FILE* f = fopen(path, "rb");
  if (!f) return fail(ISL_ERR_IO, "IO error: %s: %s", path, strerror(errno));
  std::vector<uint8_t> b;
  uint8_t buf[1 << 16];
  size_t r;
  while ((r = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + r);
  fclose(f);
  isl_index_metadata m;
  size_t used = 0;
  ISL_TRY(isl_storage_read_metadata(b.data(), b.size(), &m, &used));
  if (m.version != 1)
    return fail(ISL_ERR_DESERIALIZATION, "Deserialization error: index format version %u (this build reads 1)", m.version);
  if (b.size() - used < 12) return fail(ISL_ERR_IO, "IO error: failed to fill whole buffer");
  uint64_t n = 0;
  memcpy(&n, b.data() + used + 4, 8);
  if (memcmp(b.data() + used, "LIDX", 4))
    return fail(ISL_ERR_DESERIALIZATION, "Deserialization error: expected LIDX chunk");
  if (n > b.size() - used - 12) return fail(ISL_ERR_IO, "IO error: failed to fill whole buffer");
  ISL_TRY(isl_index_from_bytes(b.data() + used + 12, (size_t)n, out));
  if (meta) *meta = m;
  return ISL_OK;
}

}  // extern "C"
