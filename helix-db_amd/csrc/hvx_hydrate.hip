// hvx_hydrate.hip -- hydrating a device index from HelixDB's persisted vector rows.
//
// Host-only code: the value codecs of the rows the search path reads, and a hydrator that collects decoded
// rows and hands the dense image to hvx_index_import (the role VectorMemoryStore hydration plays for the
// reference's resident cache, memory_store.rs:97-105).  Formats restated from the reference
// (paths under crates/db/src/encoding/v1/):
//   layer-0 neighbours  values/vectors.rs:29-44,97-210   [0x12][count u32 BE][id u64 BE ...]  |
//                                                        [0x13][flags][count u32 BE][simhash u64 LE if flags&1][ids ...] | empty
//   upper neighbours    values/vectors/neighbors.rs:57-110  [count u32 BE][id u64 BE ...] (exact length)
//   vector item         values/vectors/item.rs:34-60 + distance/{cosine,euclidean}.rs headers:
//                       [header f32 native-endian][dim x f32 native-endian]
//   keys                keys/vectors.rs:23-50: [0xF1][index_id u64 BE][0x02][order_code u64 BE][node_id u64 BE] (item),
//                       [0xF0][index_id][0x16][node_id] (layer 0), [0xF0][index_id][0x11][layer u16 BE][node_id] (upper)
// The rkyv-archived VectorIndexMetadata row is not decoded here: the host passes entry point / max layer.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <vector>

#include "hvx_host.h"

using namespace hvx;

namespace {

inline uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
inline uint64_t be64(const uint8_t *p) { return ((uint64_t)be32(p) << 32) | be32(p + 4); }
inline uint64_t le64(const uint8_t *p) {
    uint64_t v;
    memcpy(&v, p, 8); // little-endian host
    return v;
}

int decode_ids(const uint8_t *p, size_t len, uint64_t count, std::vector<uint64_t> &out) {
    if (count > (SIZE_MAX - 16) / 8 || len != count * 8) return fail(HVX_ERR_INVARIANT, "neighbour row length %zu does not match its count %llu", len, (unsigned long long)count);
    out.resize(count);
    for (uint64_t i = 0; i < count; ++i) out[i] = be64(p + 8 * i);
    return HVX_OK;
}

// decode_layer0_neighbors_and_simhash (values/vectors.rs:187-210)
int decode_layer0(const uint8_t *v, size_t len, std::vector<uint64_t> &ids, bool &has_sh, uint64_t &sh) {
    ids.clear();
    has_sh = false;
    sh = 0;
    if (len == 0) return HVX_OK; // empty compatibility value
    if (v[0] == 0x12) {
        if (len < 5) return fail(HVX_ERR_INVARIANT, "layer-0 row shorter than its header");
        return decode_ids(v + 5, len - 5, be32(v + 1), ids);
    }
    if (v[0] == 0x13) {
        if (len < 6) return fail(HVX_ERR_INVARIANT, "layer-0 record shorter than its header");
        const uint8_t flags = v[1];
        if (flags & ~1u) return fail(HVX_ERR_INVARIANT, "invalid layer-0 record flags: 0x%02x", flags);
        size_t off = 6;
        if (flags & 1u) {
            if (len < off + 8) return fail(HVX_ERR_INVARIANT, "layer-0 record truncated inside its SimHash");
            sh = le64(v + off);
            has_sh = true;
            off += 8;
        }
        return decode_ids(v + off, len - off, be32(v + 2), ids);
    }
    return fail(HVX_ERR_INVARIANT, "invalid layer-0 encoding type 0x%02x", v[0]);
}

} // namespace

struct hvx_hydrator {
    uint32_t dim = 0, metric = 0;
    struct Node {
        std::vector<float> vec;
        float header = 0.f;
        bool has_vec = false, has_l0 = false;
        std::vector<uint64_t> l0;
        std::map<uint16_t, std::vector<uint64_t>> upper;
    };
    std::map<uint64_t, Node> nodes; // ordered by node id
    bool has_entry = false;
    uint64_t entry = 0;
    uint32_t max_layer = 0;
};

extern "C" int hvx_decode_layer0_row(const uint8_t *value, size_t len, uint64_t *out_ids, uint32_t cap, uint32_t *out_count,
                                     uint64_t *out_simhash, uint32_t *out_has_simhash) {
    std::vector<uint64_t> ids;
    bool has;
    uint64_t sh;
    int rc = decode_layer0(value, len, ids, has, sh);
    if (rc) return rc;
    if (out_count) *out_count = (uint32_t)ids.size();
    if (out_has_simhash) *out_has_simhash = has ? 1u : 0u;
    if (out_simhash) *out_simhash = sh;
    if (ids.size() > cap) return fail(HVX_ERR_INVARIANT, "layer-0 row holds %zu ids, buffer %u", ids.size(), cap);
    if (out_ids && !ids.empty()) memcpy(out_ids, ids.data(), ids.size() * 8);
    return HVX_OK;
}

// decode_upper_neighbors (values/vectors/neighbors.rs:79-110)
extern "C" int hvx_decode_upper_row(const uint8_t *value, size_t len, uint64_t *out_ids, uint32_t cap, uint32_t *out_count) {
    if (len < 4) return fail(HVX_ERR_INVARIANT, "upper-layer row shorter than its count");
    std::vector<uint64_t> ids;
    int rc = decode_ids(value + 4, len - 4, be32(value), ids);
    if (rc) return rc;
    if (out_count) *out_count = (uint32_t)ids.size();
    if (ids.size() > cap) return fail(HVX_ERR_INVARIANT, "upper row holds %zu ids, buffer %u", ids.size(), cap);
    if (out_ids && !ids.empty()) memcpy(out_ids, ids.data(), ids.size() * 8);
    return HVX_OK;
}

// values/vectors/simhash.rs:39-61: the standalone SimHash row is exactly eight little-endian bytes
extern "C" int hvx_decode_simhash_row(const uint8_t *value, size_t len, uint64_t *out_bits) {
    if (!value || len != 8) return fail(HVX_ERR_INVARIANT, "SimHash row must be exactly 8 bytes (got %zu)", len);
    uint64_t v = 0;
    for (int i = 7; i >= 0; --i) v = (v << 8) | value[i];
    if (out_bits) *out_bits = v;
    return HVX_OK;
}

// values/vectors/entry.rs:25-47: the entry-candidate node row stores its HNSW layer as exactly two big-endian bytes
extern "C" int hvx_decode_entry_candidate_layer(const uint8_t *value, size_t len, uint32_t *out_layer) {
    if (!value || len != 2) return fail(HVX_ERR_INVARIANT, "entry-candidate layer row must be exactly 2 bytes (got %zu)", len);
    if (out_layer) *out_layer = ((uint32_t)value[0] << 8) | value[1];
    return HVX_OK;
}

// keys/tenant.rs:13-15,69-95: a tenant-scoped key is [0xFD][tenant_id: u128 BE] ++ the logical key; the legacy namespace has
// no envelope.  Returns the envelope length (0 or 17) and the tenant id halves.
extern "C" uint32_t hvx_strip_tenant_envelope(const uint8_t *key, size_t len, uint64_t *tenant_hi, uint64_t *tenant_lo) {
    if (!key || len < 17 || key[0] != 0xFD) return 0;
    if (tenant_hi) *tenant_hi = be64(key + 1);
    if (tenant_lo) *tenant_lo = be64(key + 9);
    return 17;
}

// keys/vectors.rs: returns the key kind (0x02 item, 0x16 layer 0, 0x11 upper) or 0 when the key is none of them;
// a tenant envelope (keys/tenant.rs:69-95) in front of the logical key is skipped
extern "C" uint32_t hvx_parse_vector_key(const uint8_t *key, size_t len, uint64_t *index_id, uint64_t *node_id, uint64_t *order_code,
                                         uint32_t *layer) {
    if (!key) return 0;
    const uint32_t env = hvx_strip_tenant_envelope(key, len, nullptr, nullptr);
    key += env;
    len -= env;
    if (len < 10) return 0;
    const uint8_t ks = key[0], kind = key[9];
    if (ks == 0x03) { // index metadata row: [0x03][0x03][index_id: 8][0x01] (keys/vectors.rs:23-38)
        if (len == 11 && key[1] == 0x03 && key[10] == 0x01) {
            if (index_id) *index_id = be64(key + 2);
            return 0x01;
        }
        return 0;
    }
    if (index_id) *index_id = be64(key + 1);
    if (ks == 0xF1 && kind == 0x02 && len == 26) {
        if (order_code) *order_code = be64(key + 10);
        if (node_id) *node_id = be64(key + 18);
        return 0x02;
    }
    if (ks == 0xF0 && kind == 0x16 && len == 18) {
        if (node_id) *node_id = be64(key + 10);
        return 0x16;
    }
    if (ks == 0xF0 && kind == 0x11 && len == 20) {
        if (layer) *layer = ((uint32_t)key[10] << 8) | key[11];
        if (node_id) *node_id = be64(key + 12);
        return 0x11;
    }
    return 0;
}

// ---- the index metadata row: rkyv 0.8 archive of VectorIndexMetadata (values/vectors/metadata.rs:22-62) ----
// rkyv is a third-party dependency (Cargo.lock: rkyv 0.8.17, default format: little-endian, aligned, 32-bit relative
// pointers) that is not under /root/reference, and the reference holds no byte fixture of this row (its tests round-trip
// through rkyv itself): the layout below restates rkyv 0.8's published format and is PARITY UNPINNED until a real store's row
// is available.  Archived structs are repr(C) in field order; usize -> u32; bool -> u8; Option<u64> -> { tag: u8, [pad 7],
// value: u64 }; String -> 8 bytes: inline (<= 8 bytes, padded with 0xFF) or out of line { len: u32 with 0b10 in bits 7..6 of
// the first byte (len = (v & 0x3F) | ((v & ~0xFF) >> 2)), offset: i32 relative to the string's own position }.  The root
// object is the LAST 88 bytes of the value:
//   config  @0 : index_name 8 | property_name 8 | dimension, m, m0, ef_construction u32 | ml f32 | simhash_threshold u32 |
//                sampling_ratio f32 | adaptive_enabled u8 + 3 pad | adaptive_failure_prob f32                       (52 bytes)
//   @56 entry_point: tag u8, value u64 @64 | @72 max_layer u16 | @80 count u64
static uint32_t le32(const uint8_t *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
static uint64_t rk_le64(const uint8_t *p) { return (uint64_t)le32(p) | ((uint64_t)le32(p + 4) << 32); }
static float lef32(const uint8_t *p) { const uint32_t u = le32(p); float f; memcpy(&f, &u, 4); return f; }

static bool archived_string(const uint8_t *base, size_t len, size_t pos, char *out, size_t cap) {
    const uint8_t *r = base + pos;
    if ((r[0] & 0xC0) != 0x80) { // inline
        size_t n = 0;
        while (n < 8 && r[n] != 0xFF) ++n;
        if (n + 1 > cap) return false;
        memcpy(out, r, n);
        out[n] = 0;
        return true;
    }
    const uint32_t v = le32(r);
    const size_t n = (size_t)((v & 0x3Fu) | ((v & ~0xFFu) >> 2));
    const int64_t off = (int32_t)le32(r + 4);
    const int64_t at = (int64_t)pos + off;
    if (at < 0 || (uint64_t)at + n > len || n + 1 > cap) return false;
    memcpy(out, base + at, n);
    out[n] = 0;
    return true;
}

extern "C" int hvx_decode_index_metadata(const uint8_t *value, size_t len, hvx_index_metadata *out) {
    if (!out) return fail(HVX_ERR_INVARIANT, "null argument");
    memset(out, 0, sizeof(*out));
    if (!value || len == 0) return fail(HVX_ERR_INVARIANT, "Empty metadata data"); // metadata.rs:132-134
    constexpr size_t kRoot = 88;
    if (len < kRoot || (len - kRoot) % 8 != 0) return fail(HVX_ERR_INVARIANT, "Failed to access archived metadata: %zu bytes cannot hold an aligned root", len);
    const size_t root = len - kRoot;
    const uint8_t *c = value + root;
    if (!archived_string(value, len, root, out->index_name, sizeof(out->index_name)) ||
        !archived_string(value, len, root + 8, out->property_name, sizeof(out->property_name)))
        return fail(HVX_ERR_INVARIANT, "Failed to access archived metadata: string out of bounds");
    out->dimension = le32(c + 16);
    out->m = le32(c + 20);
    out->m0 = le32(c + 24);
    out->ef_construction = le32(c + 28);
    out->ml = lef32(c + 32);
    out->simhash_threshold = le32(c + 36);
    out->sampling_ratio = lef32(c + 40);
    if (c[44] > 1) return fail(HVX_ERR_INVARIANT, "Failed to access archived metadata: invalid bool");
    out->adaptive_enabled = c[44];
    out->adaptive_failure_prob = lef32(c + 48);
    if (c[56] > 1) return fail(HVX_ERR_INVARIANT, "Failed to access archived metadata: invalid Option tag");
    out->has_entry_point = c[56];
    out->entry_point = c[56] ? rk_le64(c + 64) : 0;
    out->max_layer = (uint32_t)c[72] | ((uint32_t)c[73] << 8);
    out->count = rk_le64(c + 80);
    return HVX_OK;
}

// VectorIndexMetadata::validated_state + the config checks the search path relies on (mod.rs:331-375): the row supplies entry
// point and top layer, and must agree with the hydrator's dimension
extern "C" int hvx_hydrator_set_metadata(hvx_hydrator *h, const uint8_t *value, size_t len) {
    if (!h) return fail(HVX_ERR_INVARIANT, "null argument");
    hvx_index_metadata md;
    const int rc = hvx_decode_index_metadata(value, len, &md);
    if (rc) return rc;
    if (md.dimension != h->dim) return fail(HVX_ERR_DIMENSION, "metadata dimension %llu differs from the hydrator's %u", (unsigned long long)md.dimension, h->dim);
    if (md.simhash_threshold > 64) return fail(HVX_ERR_INVARIANT, "metadata: SimHash threshold above 64");
    if (md.max_layer > 63) return fail(HVX_ERR_INVARIANT, "metadata: max_layer above 63");
    if (!md.has_entry_point && md.max_layer != 0) return fail(HVX_ERR_INVARIANT, "metadata: empty index with a non-zero top layer");
    if (md.has_entry_point) return hvx_hydrator_set_entry(h, md.entry_point, md.max_layer);
    return HVX_OK;
}

extern "C" int hvx_hydrator_new(uint32_t dim, uint32_t metric, hvx_hydrator **out) {
    if (!out) return fail(HVX_ERR_INVARIANT, "null argument");
    *out = nullptr;
    if (dim == 0) return fail(HVX_ERR_DIMENSION, "dimension must be non-zero");
    if (metric > HVX_MANHATTAN) return fail(HVX_ERR_UNSUPPORTED, "unknown metric %u", metric);
    hvx_hydrator *h = new hvx_hydrator();
    h->dim = dim;
    h->metric = metric;
    *out = h;
    return HVX_OK;
}

extern "C" void hvx_hydrator_free(hvx_hydrator *h) { delete h; }

// item row = [header: f32][dim x f32], native-endian (values/vectors/item.rs:34-60; mod.rs:873-877)
extern "C" int hvx_hydrator_add_item(hvx_hydrator *h, uint64_t node_id, const uint8_t *value, size_t len) {
    if (!h || !value) return fail(HVX_ERR_INVARIANT, "null argument");
    // decode_item_borrowed (mod.rs:889-949): payload length -> dimension, finiteness, then the header recomputed from the
    // payload must equal the persisted header byte for byte (VectorItemDecodeError::{DimensionMismatch, NonFiniteComponent,
    // HeaderMismatch}); the metric's domain rules (zero norm, magnitude) are applied by the import
    if (len < 4 || (len - 4) % 4 != 0 || (len - 4) / 4 != h->dim)
        return fail(HVX_ERR_DIMENSION, "vector row of node %llu: expected dimension %u, found %zu (%zu bytes)", (unsigned long long)node_id,
                    h->dim, len >= 4 ? (len - 4) / 4 : (size_t)0, len);
    std::vector<float> vec(h->dim);
    memcpy(vec.data(), value + 4, (size_t)h->dim * 4);
    for (uint32_t i = 0; i < h->dim; ++i)
        if (!std::isfinite(vec[i])) return fail(HVX_ERR_NONFINITE, "vector row of node %llu: non-finite component %u", (unsigned long long)node_id, i);
    float expect = 0.0f; // Euclidean / Manhattan headers are a zero bias (distance/euclidean.rs:42-44, manhattan.rs:41-43)
    if (h->metric == HVX_COSINE_HALF) { // NodeHeaderCosine { norm } (distance/cosine.rs:89-93): scaled serial f64 norm, saturated, as f32
        double scale = 0.0, scaled_sum = 1.0;
        for (uint32_t i = 0; i < h->dim; ++i) {
            const double mag = (double)std::fabs(vec[i]);
            if (mag == 0.0) continue;
            if (scale < mag) {
                const double ratio = scale / mag;
                scaled_sum = 1.0 + scaled_sum * ratio * ratio;
                scale = mag;
            } else {
                const double ratio = mag / scale;
                scaled_sum += ratio * ratio;
            }
        }
        double norm = scale == 0.0 ? 0.0 : scale * std::sqrt(scaled_sum);
        if (norm > 3.4028234663852886e+38) norm = 3.4028234663852886e+38;
        expect = (float)norm;
    }
    if (memcmp(&expect, value, 4) != 0)
        return fail(HVX_ERR_INVARIANT, "vector row of node %llu: persisted header does not match the payload (HeaderMismatch)", (unsigned long long)node_id);
    auto &n = h->nodes[node_id];
    n.header = expect;
    n.vec = std::move(vec);
    n.has_vec = true;
    return HVX_OK;
}

extern "C" int hvx_hydrator_add_layer0_row(hvx_hydrator *h, uint64_t node_id, const uint8_t *value, size_t len) {
    if (!h) return fail(HVX_ERR_INVARIANT, "null argument");
    std::vector<uint64_t> ids;
    bool has;
    uint64_t sh;
    int rc = decode_layer0(value, len, ids, has, sh);
    if (rc) return rc;
    // runtime canonicalisation (neighbor_set.rs:1-9): ascending, deduped, self-free
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    ids.erase(std::remove(ids.begin(), ids.end(), node_id), ids.end());
    auto &n = h->nodes[node_id];
    n.l0 = std::move(ids);
    n.has_l0 = true;
    return HVX_OK;
}

extern "C" int hvx_hydrator_add_upper_row(hvx_hydrator *h, uint64_t node_id, uint32_t layer, const uint8_t *value, size_t len) {
    if (!h) return fail(HVX_ERR_INVARIANT, "null argument");
    if (layer == 0 || layer > 63) return fail(HVX_ERR_INVARIANT, "upper row with layer %u", layer);
    if (len < 4) return fail(HVX_ERR_INVARIANT, "upper-layer row shorter than its count");
    std::vector<uint64_t> ids;
    int rc = decode_ids(value + 4, len - 4, be32(value), ids);
    if (rc) return rc;
    std::sort(ids.begin(), ids.end()); // historical rows may be distance-ordered; the runtime re-sorts
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    ids.erase(std::remove(ids.begin(), ids.end(), node_id), ids.end());
    h->nodes[node_id].upper[(uint16_t)layer] = std::move(ids);
    return HVX_OK;
}

extern "C" int hvx_hydrator_set_entry(hvx_hydrator *h, uint64_t entry_point, uint32_t max_layer) {
    if (!h) return fail(HVX_ERR_INVARIANT, "null argument");
    h->has_entry = true;
    h->entry = entry_point;
    h->max_layer = max_layer;
    return HVX_OK;
}

extern "C" int hvx_hydrator_finish(const hvx_hydrator *h, const hvx_index_desc *tmpl, hvx_index **out) { return hvx_hydrator_finish_reserve(h, tmpl, 0, 0, out); }

extern "C" int hvx_hydrator_finish_reserve(const hvx_hydrator *h, const hvx_index_desc *tmpl, uint64_t reserve_rows, uint64_t reserve_upper_rows, hvx_index **out) {
    if (!h || !tmpl || !out) return fail(HVX_ERR_INVARIANT, "null argument");
    // every node must have its canonical vector row; neighbour ids without a vector row are dangling and
    // dropped (the reference skips missing rows silently: search.rs:848-914)
    std::vector<uint64_t> ids;
    for (auto &kv : h->nodes)
        if (kv.second.has_vec) ids.push_back(kv.first);
    const uint64_t n = ids.size();
    std::vector<float> vecs((size_t)n * h->dim);
    std::vector<uint64_t> l0_off(n + 1, 0), l0_nb, up_off(1, 0), up_nb;
    std::vector<uint16_t> level(n, 0);
    auto known = [&](uint64_t id) { return std::binary_search(ids.begin(), ids.end(), id); };
    for (uint64_t i = 0; i < n; ++i) {
        const auto &nd = h->nodes.at(ids[i]);
        memcpy(&vecs[(size_t)i * h->dim], nd.vec.data(), (size_t)h->dim * 4);
        for (uint64_t x : nd.l0)
            if (known(x)) l0_nb.push_back(x);
        l0_off[i + 1] = l0_nb.size();
        uint16_t top = 0;
        for (auto &u : nd.upper) top = std::max(top, u.first);
        level[i] = top;
        for (uint16_t l = 1; l <= top; ++l) {
            auto it = nd.upper.find(l);
            if (it != nd.upper.end())
                for (uint64_t x : it->second)
                    if (known(x)) up_nb.push_back(x);
            up_off.push_back(up_nb.size());
        }
    }
    hvx_index_desc d = *tmpl;
    d.dim = h->dim;
    d.metric = h->metric;
    d.n = n;
    d.has_entry = (h->has_entry && n) ? 1u : 0u;
    d.entry_point = h->entry;
    d.max_layer = h->max_layer;
    if (n) { d.shard_id_lo = ids.front(); d.shard_id_hi = ids.back(); }
    if (l0_nb.empty()) l0_nb.push_back(0);
    if (up_nb.empty()) up_nb.push_back(0);
    if (reserve_rows || reserve_upper_rows)
        return hvx_index_import_reserve(&d, ids.data(), vecs.data(), l0_off.data(), l0_nb.data(), level.data(), up_off.data(), up_nb.data(), reserve_rows, reserve_upper_rows, out);
    return hvx_index_import(&d, ids.data(), vecs.data(), l0_off.data(), l0_nb.data(), level.data(), up_off.data(), up_nb.data(), out);
}
