// On-disk formats either side of the path, host side of libapd_hip.so (include/apd.h, "formats"):
//   * bincode 1.x `AutoEncoder` weight files written by the reference's AutoEncoder::save_file (src/neural.rs:13-19, 30-44),
//   * project/config/Discovery.toml (src/discovery.rs:7-36),
//   * the dendrogram bracket strings of Templates::dendrograms (src/reporting.rs:135-169).
// Pure host code: no device, no allocation visible to the caller.  PARITY UNPINNED: the reference ships no weight file, no
// rendered report and no test; the layouts follow the crates' documented defaults (bincode 1.x: little-endian, fixed-width
// integers, u64 lengths, fields in declaration order) and the format strings in the cited lines.
#include <cerrno>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/apd.h"

namespace {

bool read_u64(const unsigned char *p, uint64_t n, uint64_t *pos, uint64_t *v)
{
    if (*pos + 8 > n) return false;
    uint64_t r = 0;
    for (int k = 7; k >= 0; --k) r = (r << 8) | p[*pos + k];             // little-endian whatever the host is
    *v = r;
    *pos += 8;
    return true;
}

void write_u64(unsigned char *p, uint64_t v) { for (int k = 0; k < 8; ++k) p[k] = (unsigned char)(v >> (8 * k)); }

}  // namespace

// Mat { flat: Vec<f32>, cols: usize } (numerics.rs:171-174) x 4, in the order w_encode, w_decode, b_encode, b_decode
extern "C" int apd_autoencoder_parse(const void *bytes, uint64_t n_bytes, apd_autoencoder_view *view)
{
    if (!bytes || !view) return APD_ERR_INVALID_ARG;
    const unsigned char *p = (const unsigned char *)bytes;
    apd_mat_view *mats[4] = {&view->w_encode, &view->w_decode, &view->b_encode, &view->b_decode};
    uint64_t pos = 0;
    for (int k = 0; k < 4; ++k) {
        uint64_t len = 0, cols = 0;
        if (!read_u64(p, n_bytes, &pos, &len)) return APD_ERR_INVALID_ARG;       // Vec length
        if (len > (n_bytes - pos) / 4) return APD_ERR_INVALID_ARG;               // truncated
        const uint64_t off = pos;
        pos += 4 * len;
        if (!read_u64(p, n_bytes, &pos, &cols)) return APD_ERR_INVALID_ARG;      // usize as u64
        if (cols == 0 || len % cols != 0) return APD_ERR_INVALID_ARG;            // not a matrix
        mats[k]->offset = off; mats[k]->len = len; mats[k]->cols = cols;
    }
    if (pos != n_bytes) return APD_ERR_INVALID_ARG;                              // trailing bytes
    // shapes the forward pass relies on (neural.rs:46-53: w_encode D x L, b_encode 1 x L, w_decode L x D, b_decode 1 x D)
    const uint64_t L = view->w_encode.cols, D = view->w_encode.len / L;
    if (view->b_encode.len != L || view->b_encode.cols != L) return APD_ERR_INVALID_ARG;
    if (view->w_decode.len != L * D || view->w_decode.cols != D) return APD_ERR_INVALID_ARG;
    if (view->b_decode.len != D || view->b_decode.cols != D) return APD_ERR_INVALID_ARG;
    return APD_OK;
}

extern "C" int apd_autoencoder_serialize(const float *w_encode, const float *w_decode, const float *b_encode, const float *b_decode,
                                         uint32_t d_in, uint32_t latent, void *out, uint64_t capacity, uint64_t *n_bytes)
{
    if (!n_bytes || d_in == 0 || latent == 0) return APD_ERR_INVALID_ARG;
    const uint64_t lens[4] = {(uint64_t)d_in * latent, (uint64_t)d_in * latent, latent, d_in};
    const uint64_t cols[4] = {latent, d_in, latent, d_in};
    const float *src[4] = {w_encode, w_decode, b_encode, b_decode};
    uint64_t need = 0;
    for (int k = 0; k < 4; ++k) need += 16 + 4 * lens[k];
    *n_bytes = need;
    if (!out) return APD_OK;                                                     // size query
    if (capacity < need || !w_encode || !w_decode || !b_encode || !b_decode) return APD_ERR_INVALID_ARG;
    unsigned char *p = (unsigned char *)out;
    for (int k = 0; k < 4; ++k) {
        write_u64(p, lens[k]); p += 8;
        for (uint64_t i = 0; i < lens[k]; ++i) {                                 // f32 little-endian
            uint32_t b;
            std::memcpy(&b, &src[k][i], 4);
            p[0] = (unsigned char)b; p[1] = (unsigned char)(b >> 8); p[2] = (unsigned char)(b >> 16); p[3] = (unsigned char)(b >> 24);
            p += 4;
        }
        write_u64(p, cols[k]); p += 8;
    }
    return APD_OK;
}

extern "C" int apd_autoencoder_copy(const void *bytes, const apd_mat_view *mat, float *out)
{
    if (!bytes || !mat || !out) return APD_ERR_INVALID_ARG;
    const unsigned char *p = (const unsigned char *)bytes + mat->offset;
    for (uint64_t i = 0; i < mat->len; ++i) {
        const uint32_t b = (uint32_t)p[4 * i] | ((uint32_t)p[4 * i + 1] << 8) | ((uint32_t)p[4 * i + 2] << 16) | ((uint32_t)p[4 * i + 3] << 24);
        std::memcpy(&out[i], &b, 4);
    }
    return APD_OK;
}

// discovery.rs:7-36: `key = value  # comment` lines (the flat TOML the reference ships); every one of the 17 fields must
// be present exactly once (serde fails on a missing or duplicate field); integers for the usize fields.
extern "C" int apd_discovery_parse_toml(const char *text, apd_discovery *out)
{
    if (!text || !out) return APD_ERR_INVALID_ARG;
    struct Field { const char *name; int is_int; void *dst; };
    apd_discovery d{};
    const Field fields[17] = {
        {"dft_win", 1, &d.dft_win}, {"dft_step", 1, &d.dft_step}, {"ceps_filter", 1, &d.ceps_filter}, {"vat_moving", 1, &d.vat_moving},
        {"vat_percentile", 0, &d.vat_percentile}, {"vat_min_len", 1, &d.vat_min_len}, {"alignment_workers", 1, &d.alignment_workers},
        {"clustering_percentile", 0, &d.clustering_percentile}, {"warping_band_percentage", 0, &d.warping_band_percentage},
        {"insertion_penalty", 0, &d.insertion_penalty}, {"deletion_penalty", 0, &d.deletion_penalty}, {"match_penalty", 0, &d.match_penalty},
        {"auto_encoder", 1, &d.auto_encoder}, {"learning_rate", 0, &d.learning_rate}, {"epochs", 1, &d.epochs},
        {"epoch_drop", 0, &d.epoch_drop}, {"drop", 0, &d.drop}};
    bool seen[17] = {false};
    const char *p = text;
    while (*p) {
        const char *eol = std::strchr(p, '\n');
        std::string line(p, eol ? (size_t)(eol - p) : std::strlen(p));
        p = eol ? eol + 1 : p + line.size();
        const size_t hash = line.find('#');
        if (hash != std::string::npos) line.erase(hash);
        const size_t eq = line.find('=');
        auto trim = [](std::string s) {
            const size_t a = s.find_first_not_of(" \t\r"), b = s.find_last_not_of(" \t\r");
            return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
        };
        if (eq == std::string::npos) { if (!trim(line).empty()) return APD_ERR_INVALID_ARG; continue; }
        const std::string key = trim(line.substr(0, eq)), val = trim(line.substr(eq + 1));
        int f = -1;
        for (int k = 0; k < 17; ++k) if (key == fields[k].name) f = k;
        if (f < 0 || seen[f] || val.empty()) return APD_ERR_INVALID_ARG;         // unknown key (serde: deny? the struct ignores none), duplicate
        char *end = nullptr;
        errno = 0;
        if (fields[f].is_int) {
            if (val.find_first_of(".eE") != std::string::npos || val[0] == '-') return APD_ERR_INVALID_ARG;   // a float where usize is wanted
            const unsigned long long v = std::strtoull(val.c_str(), &end, 10);
            if (errno || *end) return APD_ERR_INVALID_ARG;
            *(uint64_t *)fields[f].dst = v;
        } else {
            const float v = std::strtof(val.c_str(), &end);
            if (errno || *end) return APD_ERR_INVALID_ARG;
            *(float *)fields[f].dst = v;
        }
        seen[f] = true;
    }
    for (int k = 0; k < 17; ++k) if (!seen[k]) return APD_ERR_INVALID_ARG;       // missing field
    *out = d;
    return APD_OK;
}

// reporting.rs:135-169: results[into] = "[.into [<left> <right> ] ]" with a leaf rendered as its label and an inner node as
// the string built for it earlier.  The reference keeps every intermediate string in a HashMap; here each root is
// expanded on demand (same characters).  An op that refers to a cluster no earlier op made is the reference's HashMap
// index panic: APD_ERR_INVALID_ARG.
extern "C" int apd_dendrograms(const apd_cluster_op *ops, uint32_t n_ops, const uint32_t *roots, uint32_t n_roots,
                               const char *const *labels, uint32_t n_labels, char *out, uint64_t capacity, uint64_t *n_bytes,
                               uint32_t *which_root, uint32_t *n_strings)
{
    if ((n_ops && !ops) || (n_roots && !roots) || !n_bytes || !n_strings || (n_labels && !labels)) return APD_ERR_INVALID_ARG;
    // the op that made a node id (later ops overwrite earlier ones, as HashMap::insert does)
    uint32_t max_id = 0;
    for (uint32_t t = 0; t < n_ops; ++t) max_id = std::max(max_id, ops[t].into);
    std::vector<int64_t> made((size_t)max_id + 1, -1);
    // validate in replay order: a cluster operand must exist when its op runs (reporting.rs:154,158,163-164)
    for (uint32_t t = 0; t < n_ops; ++t) {
        const apd_cluster_op &o = ops[t];
        const bool ci = o.operation == APD_CLUSTER2SEQUENCE || o.operation == APD_CLUSTER2CLUSTER;
        const bool cj = o.operation == APD_SEQUENCE2CLUSTER || o.operation == APD_CLUSTER2CLUSTER;
        if (o.operation > APD_CLUSTER2CLUSTER) return APD_ERR_INVALID_ARG;
        if (ci ? (o.merge_i > max_id || made[o.merge_i] < 0) : o.merge_i >= n_labels) return APD_ERR_INVALID_ARG;
        if (cj ? (o.merge_j > max_id || made[o.merge_j] < 0) : o.merge_j >= n_labels) return APD_ERR_INVALID_ARG;
        made[o.into] = t;
    }
    // The string of a node is a function of the op that made it AT THAT TIME; ids are unique in a clustering run
    // (into = n + t), so the final `made` table is the replay-time table.
    std::string all;
    uint32_t count = 0;
    for (uint32_t r = 0; r < n_roots; ++r) {
        const uint32_t id = roots[r];
        if (id > max_id || made[id] < 0) continue;                               // "Cluster not found ... Singular cluster" (:200)
        std::string s;
        // iterative expansion: a stack of (op index, stage)
        struct Frame { uint32_t op; int stage; };
        std::vector<Frame> stack{{(uint32_t)made[id], 0}};
        while (!stack.empty()) {
            Frame &f = stack.back();
            const apd_cluster_op &o = ops[f.op];
            const bool ci = o.operation == APD_CLUSTER2SEQUENCE || o.operation == APD_CLUSTER2CLUSTER;
            const bool cj = o.operation == APD_SEQUENCE2CLUSTER || o.operation == APD_CLUSTER2CLUSTER;
            if (f.stage == 0) {
                s += "[." + std::to_string(o.into) + " [";
                f.stage = 1;
                if (ci) { stack.push_back({(uint32_t)made[o.merge_i], 0}); continue; }
                s += labels[o.merge_i];
            }
            if (f.stage == 1) {
                s += " ";
                f.stage = 2;
                if (cj) { stack.push_back({(uint32_t)made[o.merge_j], 0}); continue; }
                s += labels[o.merge_j];
            }
            s += " ] ]";
            stack.pop_back();
        }
        if (which_root && count < n_roots) which_root[count] = r;
        all += s;
        all.push_back('\0');
        ++count;
    }
    *n_bytes = all.size();
    *n_strings = count;
    if (!out) return APD_OK;                                                     // size query
    if (capacity < all.size()) return APD_ERR_INVALID_ARG;
    std::memcpy(out, all.data(), all.size());
    return APD_OK;
}
