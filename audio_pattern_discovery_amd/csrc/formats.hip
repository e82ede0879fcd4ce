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
#include <new>
#include <unordered_map>
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
// be present exactly once (serde fails on a missing field, TOML on a duplicate key); integers for the usize fields; keys the
// struct does not have are ignored, as serde does without deny_unknown_fields.
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
    bool in_table = false;
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
        const std::string bare = trim(line);
        if (!bare.empty() && bare.front() == '[' && bare.back() == ']') { in_table = true; continue; }   // a [table]: what follows is not a field of Discovery
        if (eq == std::string::npos) { if (!bare.empty()) return APD_ERR_INVALID_ARG; continue; }
        const std::string key = trim(line.substr(0, eq)), val = trim(line.substr(eq + 1));
        if (key.empty() || val.empty()) return APD_ERR_INVALID_ARG;              // not `key = value`
        if (in_table) continue;
        int f = -1;
        for (int k = 0; k < 17; ++k) if (key == fields[k].name) f = k;
        if (f < 0) continue;                                                     // unknown key: #[derive(Deserialize)] without deny_unknown_fields ignores it (discovery.rs:7)
        if (seen[f]) return APD_ERR_INVALID_ARG;                                 // duplicate key: a TOML error
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
// the string results[operand] holds AT THAT MOMENT of the replay (HashMap::insert overwrites the entry of a repeated id, but a
// string built earlier has its operands embedded by value).  So every op records which earlier op had made each of its cluster
// operands when it ran; expansion follows those op indices -- strictly decreasing, hence finite whatever the ids are (an op
// whose `into` equals one of its own operands, a repeated `into`) -- and each root expands the op that made it last.  An op
// that refers to a cluster no earlier op made is the reference's HashMap index panic: APD_ERR_INVALID_ARG.  Nothing unwinds
// across the C boundary: allocation failure (a degenerate op list can describe an exponentially large string) is APD_ERR_OOM.
extern "C" int apd_dendrograms(const apd_cluster_op *ops, uint32_t n_ops, const uint32_t *roots, uint32_t n_roots,
                               const char *const *labels, uint32_t n_labels, char *out, uint64_t capacity, uint64_t *n_bytes,
                               uint32_t *which_root, uint32_t *n_strings)
{
    if ((n_ops && !ops) || (n_roots && !roots) || !n_bytes || !n_strings || (n_labels && !labels)) return APD_ERR_INVALID_ARG;
    constexpr size_t kMaxBytes = (size_t)1 << 30;                                // refuse to build more than 1 GiB of brackets
    try {
        std::unordered_map<uint32_t, uint32_t> made;                             // node id -> op that made it last (ids are any u32)
        std::vector<int64_t> src_i(n_ops, -1), src_j(n_ops, -1);                 // op that had made the cluster operand when op t ran
        for (uint32_t t = 0; t < n_ops; ++t) {                                   // replay order (reporting.rs:143)
            const apd_cluster_op &o = ops[t];
            if (o.operation > APD_CLUSTER2CLUSTER) return APD_ERR_INVALID_ARG;
            const bool ci = o.operation == APD_CLUSTER2SEQUENCE || o.operation == APD_CLUSTER2CLUSTER;
            const bool cj = o.operation == APD_SEQUENCE2CLUSTER || o.operation == APD_CLUSTER2CLUSTER;
            if (ci) {
                const auto it = made.find(o.merge_i);
                if (it == made.end()) return APD_ERR_INVALID_ARG;                // results[&i] panics (:158, :164)
                src_i[t] = it->second;
            } else if (o.merge_i >= n_labels) return APD_ERR_INVALID_ARG;        // images[i] panics (:149, :154)
            if (cj) {
                const auto it = made.find(o.merge_j);
                if (it == made.end()) return APD_ERR_INVALID_ARG;
                src_j[t] = it->second;
            } else if (o.merge_j >= n_labels) return APD_ERR_INVALID_ARG;
            made[o.into] = t;                                                    // HashMap::insert (:151, :156, :161, :166)
        }
        std::string all;
        uint32_t count = 0;
        for (uint32_t r = 0; r < n_roots; ++r) {
            const auto root = made.find(roots[r]);
            if (root == made.end()) continue;                                    // "Cluster not found ... Singular cluster" (:200)
            struct Frame { uint32_t op; int stage; };
            std::vector<Frame> stack{{root->second, 0}};
            while (!stack.empty()) {
                Frame &f = stack.back();
                const apd_cluster_op &o = ops[f.op];
                if (f.stage == 0) {
                    all += "[." + std::to_string(o.into) + " [";
                    f.stage = 1;
                    if (src_i[f.op] >= 0) { const uint32_t next = (uint32_t)src_i[f.op]; stack.push_back({next, 0}); continue; }
                    all += labels[o.merge_i];
                }
                if (f.stage == 1) {
                    all += " ";
                    f.stage = 2;
                    if (src_j[f.op] >= 0) { const uint32_t next = (uint32_t)src_j[f.op]; stack.push_back({next, 0}); continue; }
                    all += labels[o.merge_j];
                }
                all += " ] ]";
                stack.pop_back();
                if (all.size() > kMaxBytes) return APD_ERR_OOM;
            }
            if (which_root && count < n_roots) which_root[count] = r;
            all.push_back('\0');
            ++count;
        }
        *n_bytes = all.size();
        *n_strings = count;
        if (!out) return APD_OK;                                                 // size query
        if (capacity < all.size()) return APD_ERR_INVALID_ARG;
        std::memcpy(out, all.data(), all.size());
        return APD_OK;
    } catch (const std::bad_alloc &) {
        return APD_ERR_OOM;
    } catch (...) {
        return APD_ERR_INVALID_ARG;
    }
}
