// apd.hpp -- header-only C++ mirror of the reference's Rust surface for the alignment/clustering path, over the C ABI
// of apd.h.  Names and argument meaning follow the Rust items (file:line cited at each); the compute is libapd_hip.so.
// Errors that are panics in the reference (zero-length sequence, percentile index out of range) throw apd::Error.
#pragma once
#include <cstdint>
#include <fstream>
#include <iterator>
#include <limits>
#include <map>
#include <set>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "apd.h"

namespace apd {

struct Error : std::runtime_error {
    int status;
    Error(int s, const std::string &what) : std::runtime_error(what), status(s) {}
};

inline void check(int status, apd_context *ctx = nullptr)
{
    if (status == APD_OK) return;
    std::string msg = apd_status_string(status);
    if (ctx && status == APD_ERR_HIP) msg += std::string(": ") + apd_last_error(ctx);
    throw Error(status, msg);
}

class Context {
  public:
    explicit Context(int device = 0) { check(apd_create(device, &ctx_)); }
    ~Context() { if (ctx_) apd_destroy(ctx_); }
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;
    apd_context *get() const { return ctx_; }
    // 0 difference form, 1 hybrid (default), 2 strict: the Rust build's bits for every penalty set (apd.h, apd_set_distance_mode)
    void set_distance_mode(int mode, float tau = 0.0f) { check(apd_set_distance_mode(ctx_, mode, tau)); }
  private:
    apd_context *ctx_ = nullptr;
};

// spectrogram.rs:13-24 -- flat row-major frames [T][n_bins]
struct NDSequence {
    std::size_t n_bins = 0;
    std::vector<float> frames;
    std::size_t audio_id = 0;
    const float *vec(std::size_t t) const { return frames.data() + t * n_bins; }     // :99-101
    std::size_t len() const { return n_bins ? frames.size() / n_bins : 0; }         // :152-154
};

// numerics.rs:171-174
struct Mat {
    std::vector<float> flat;
    std::size_t cols = 0;
    std::size_t rows() const { return cols ? flat.size() / cols : 0; }
};

// neural.rs:12-71: the weight file and the forward pass (training is outside the accelerated path)
struct AutoEncoder {
    Mat w_encode, w_decode, b_encode, b_decode;
    std::size_t n_latent() const { return b_encode.cols; }                          // :22-24
    static AutoEncoder from_bytes(const std::vector<unsigned char> &buf)             // bincode::deserialize, :34
    {
        apd_autoencoder_view v;
        check(apd_autoencoder_parse(buf.data(), buf.size(), &v));
        AutoEncoder nn;
        const apd_mat_view *views[4] = {&v.w_encode, &v.w_decode, &v.b_encode, &v.b_decode};
        Mat *mats[4] = {&nn.w_encode, &nn.w_decode, &nn.b_encode, &nn.b_decode};
        for (int k = 0; k < 4; ++k) {
            mats[k]->flat.resize(views[k]->len);
            mats[k]->cols = views[k]->cols;
            check(apd_autoencoder_copy(buf.data(), views[k], mats[k]->flat.data()));
        }
        return nn;
    }
    static AutoEncoder from_file(const std::string &file)                            // :30-36
    {
        std::ifstream in(file, std::ios::binary);
        if (!in) throw Error(APD_ERR_INVALID_ARG, "cannot open " + file);            // File::open(file)? -> DiscoveryError::IO
        const std::vector<unsigned char> buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        return from_bytes(buf);
    }
    void save_file(const std::string &file) const                                    // :39-44
    {
        const uint32_t latent = (uint32_t)w_encode.cols, d_in = (uint32_t)w_encode.rows();
        uint64_t n = 0;
        check(apd_autoencoder_serialize(nullptr, nullptr, nullptr, nullptr, d_in, latent, nullptr, 0, &n));
        std::vector<unsigned char> buf(n);
        check(apd_autoencoder_serialize(w_encode.flat.data(), w_decode.flat.data(), b_encode.flat.data(), b_decode.flat.data(), d_in, latent,
                                        buf.data(), buf.size(), &n));
        std::ofstream out(file, std::ios::binary);
        out.write((const char *)buf.data(), (std::streamsize)buf.size());
        if (!out) throw Error(APD_ERR_INVALID_ARG, "cannot write " + file);
    }
    // AutoEncoder::predict on every frame of a sequence = NDSequence::encoded (neural.rs:55-71, spectrogram.rs:103-121)
    NDSequence encoded(Context &ctx, const NDSequence &x) const
    {
        if (x.n_bins != w_encode.rows()) throw Error(APD_ERR_INVALID_ARG, "self.cols == other.rows()");   // numerics.rs:306
        NDSequence out;
        out.n_bins = n_latent();
        out.audio_id = x.audio_id;
        out.frames.resize(x.len() * out.n_bins);
        check(apd_encode(ctx.get(), x.frames.data(), x.len(), (uint32_t)x.n_bins, w_encode.flat.data(), b_encode.flat.data(), (uint32_t)out.n_bins, 0,
                         out.frames.data()), ctx.get());
        return out;
    }
};

// alignments.rs:77-83
struct AlignmentParams {
    std::size_t warping_band;
    float insertion_penalty, deletion_penalty, match_penalty;
    static AlignmentParams default_for(std::size_t len) { return {len, 1.0f, 1.0f, 1.0f}; }   // :86-93
};

// discovery.rs:7-26 -- every field of project/config/Discovery.toml (defaults: the shipped file)
struct Discovery {
    std::size_t dft_win = 256, dft_step = 128, ceps_filter = 32, vat_moving = 15;
    float vat_percentile = 0.95f;
    std::size_t vat_min_len = 150;
    std::size_t alignment_workers = 4;      // accepted, unused: the GPU grid replaces the OS threads
    float clustering_percentile = 0.05f;
    float warping_band_percentage = 1.0f, insertion_penalty = 1.0f, deletion_penalty = 1.0f, match_penalty = 1.0f;
    std::size_t auto_encoder = 10;
    float learning_rate = 0.1f;
    std::size_t epochs = 25;
    float epoch_drop = 5.0f, drop = 0.5f;
    static Discovery from_toml(const std::string &file)                             // discovery.rs:28-36
    {
        std::ifstream in(file);
        if (!in) throw Error(APD_ERR_INVALID_ARG, "Template file not found");        // .expect() at :31
        const std::string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        apd_discovery d;
        check(apd_discovery_parse_toml(text.c_str(), &d));                            // toml::from_str(..).unwrap() at :34
        Discovery c;
        c.dft_win = d.dft_win; c.dft_step = d.dft_step; c.ceps_filter = d.ceps_filter; c.vat_moving = d.vat_moving;
        c.vat_percentile = d.vat_percentile; c.vat_min_len = d.vat_min_len; c.alignment_workers = d.alignment_workers;
        c.clustering_percentile = d.clustering_percentile; c.warping_band_percentage = d.warping_band_percentage;
        c.insertion_penalty = d.insertion_penalty; c.deletion_penalty = d.deletion_penalty; c.match_penalty = d.match_penalty;
        c.auto_encoder = d.auto_encoder; c.learning_rate = d.learning_rate; c.epochs = d.epochs; c.epoch_drop = d.epoch_drop; c.drop = d.drop;
        return c;
    }
    apd_align_config config() const { return {warping_band_percentage, insertion_penalty, deletion_penalty, match_penalty}; }
    AlignmentParams alignment_params(std::size_t n_size) const                       // discovery.rs:38-45
    {
        const apd_align_config c = config();
        apd_alignment_params p;
        check(apd_discovery_alignment_params(&c, n_size, &p));
        return {(std::size_t)p.warping_band, p.insertion_penalty, p.deletion_penalty, p.match_penalty};
    }
};

// alignments.rs:99-181.  `sparse` is never read outside the struct in the reference, so it is not materialised.
class Alignment {
  public:
    explicit Alignment(Context &ctx) : ctx_(ctx) {}                                  // Alignment::new, :107-111
    std::size_t n = 0, m = 0;
    void construct_alignment(const NDSequence &x, const NDSequence &y, const AlignmentParams &p)   // :165-180
    {
        n = x.len(); m = y.len();
        const apd_alignment_params cp{(uint64_t)p.warping_band, p.insertion_penalty, p.deletion_penalty, p.match_penalty};
        const uint32_t dim = (uint32_t)(x.n_bins ? x.n_bins : y.n_bins);
        check(apd_align_pair(ctx_.get(), x.frames.data(), n, y.frames.data(), m, dim, &cp, &score_), ctx_.get());
    }
    float score() const { return (n == 0 && m == 0) ? std::numeric_limits<float>::infinity() : score_; }   // :116-125
  private:
    Context &ctx_;
    float score_ = std::numeric_limits<float>::infinity();
};

// alignments.rs:11-68
class AlignmentWorkers {
  public:
    AlignmentWorkers(Context &ctx, std::vector<NDSequence> data) : data(std::move(data)), ctx_(ctx)   // ::new, :17-26
    {
        const std::size_t n = this->data.size();
        result.assign(n * n, 0.0f);
        offsets.assign(n + 1, 0);
        dim = n ? (uint32_t)this->data[0].n_bins : 1;
        for (std::size_t s = 0; s < n; ++s) {
            if (this->data[s].n_bins != dim) throw Error(APD_ERR_INVALID_ARG, "sequences must share n_bins");
            offsets[s + 1] = offsets[s] + this->data[s].len();
            frames.insert(frames.end(), this->data[s].frames.begin(), this->data[s].frames.end());
        }
        if (dim == 0) dim = 1;
        check(apd_batch_create(ctx_.get(), frames.data(), offsets.data(), (uint32_t)n, dim, 0, &batch_), ctx_.get());
    }
    ~AlignmentWorkers()
    {
        if (multi_) apd_multi_destroy(multi_);                                       // also destroys its batch
        if (batch_) apd_batch_destroy(batch_);
    }
    AlignmentWorkers(const AlignmentWorkers &) = delete;
    AlignmentWorkers &operator=(const AlignmentWorkers &) = delete;
    void align_all(const Discovery &params)                                          // :31-67, blocking
    {
        const apd_align_config c = params.config();
        check(apd_align_all(ctx_.get(), batch_, &c, result.data()), ctx_.get());
    }
    // The same over several GPUs of one node (the reference's `alignment_workers` threads, alignments.rs:33-41, become devices):
    // pair tiles sharded over `devices`, one RCCL all-gather inside the library.  The multi-device handle (contexts,
    // communicators, worker threads, the corpus resident on every device) is made on the first call and kept: a host that calls
    // align_all in a loop (main.rs:187-195) pays setup once.  Returns the ranks RCCL saw.
    uint32_t align_all(const Discovery &params, const std::vector<int> &devices)
    {
        const apd_align_config c = params.config();
        if (multi_ && devices != multi_devices_) { apd_multi_destroy(multi_); multi_ = nullptr; multi_batch_ = nullptr; }
        if (!multi_) {
            check(apd_multi_create(devices.data(), (uint32_t)devices.size(), &multi_));
            multi_devices_ = devices;
            const int rc = apd_multi_batch_create(multi_, frames.data(), nullptr, offsets.data(), (uint32_t)data.size(), dim, &multi_batch_);
            if (rc != APD_OK) throw Error(rc, std::string(apd_status_string(rc)) + ": " + apd_multi_last_error(multi_));
        }
        const int rc = apd_multi_align_all(multi_, multi_batch_, &c, result.data());
        if (rc != APD_OK) throw Error(rc, std::string(apd_status_string(rc)) + ": " + apd_multi_last_error(multi_));
        uint32_t seen = 0;
        check(apd_multi_ranks_seen(multi_, &seen));
        return seen;
    }
    const char *collective() const { return multi_ ? apd_multi_collective(multi_) : "none (one device)"; }
    std::vector<NDSequence> data;
    std::vector<float> result;                                                       // n*n row-major, diagonal 0.0
  private:
    Context &ctx_;
    apd_batch *batch_ = nullptr;
    apd_multi *multi_ = nullptr;                                                     // made by the first multi-device align_all
    apd_multi_batch *multi_batch_ = nullptr;
    std::vector<int> multi_devices_;
    std::vector<float> frames;                                                       // packed [sum len][dim], kept for the multi-device entry
    std::vector<uint64_t> offsets;
    uint32_t dim = 1;
};

enum class Merge { Sequence2Sequence = 0, Sequence2Cluster = 1, Cluster2Sequence = 2, Cluster2Cluster = 3 };   // clustering.rs:8-13
struct ClusteringOperation { std::size_t merge_i, merge_j, into; float distance; Merge operation; };              // :19-25

struct AgglomerativeClustering {
    // clustering.rs:81-110
    static std::pair<std::vector<ClusteringOperation>, std::set<std::size_t>> clustering(Context &ctx, const std::vector<float> &distances,
                                                                                            std::size_t n_instances, float perc)
    {
        std::vector<apd_cluster_op> ops(n_instances ? n_instances : 1);
        std::vector<uint32_t> roots(n_instances ? n_instances : 1);
        uint32_t n_ops = 0, n_roots = 0;
        float thr = 0.0f;
        check(apd_clustering(ctx.get(), distances.data(), 0, (uint32_t)n_instances, perc, ops.data(), &n_ops, roots.data(), &n_roots, &thr),
              ctx.get());
        std::vector<ClusteringOperation> out;
        for (uint32_t t = 0; t < n_ops; ++t)
            out.push_back({ops[t].merge_i, ops[t].merge_j, ops[t].into, ops[t].distance, (Merge)ops[t].operation});
        return {out, std::set<std::size_t>(roots.begin(), roots.begin() + n_roots)};
    }
    // clustering.rs:40-76
    static std::vector<std::vector<std::size_t>> cluster_sets(const std::vector<ClusteringOperation> &operations,
                                                              const std::set<std::size_t> &cluster_ids, std::size_t n_instances)
    {
        std::vector<apd_cluster_op> ops;
        for (const auto &o : operations) ops.push_back({(uint32_t)o.merge_i, (uint32_t)o.merge_j, (uint32_t)o.into, o.distance, (uint32_t)o.operation});
        std::vector<uint32_t> roots(cluster_ids.begin(), cluster_ids.end()), members(n_instances + ops.size() + 2), off(roots.size() + 2);
        uint32_t n_sets = 0;
        check(apd_cluster_sets(ops.data(), (uint32_t)ops.size(), roots.data(), (uint32_t)roots.size(), (uint32_t)n_instances, members.data(),
                               off.data(), &n_sets));
        std::vector<std::vector<std::size_t>> out;
        for (uint32_t s = 0; s < n_sets; ++s) out.emplace_back(members.begin() + off[s], members.begin() + off[s + 1]);
        return out;
    }
};

// Templates::dendrograms (reporting.rs:135-169), the strings: root id -> "[.k [<left> <right> ] ]" with leaf i rendered as
// labels[i]; roots no op made are absent (reporting.rs:200).
inline std::map<std::size_t, std::string> dendrograms(const std::vector<ClusteringOperation> &operations, const std::set<std::size_t> &clusters,
                                                      const std::vector<std::string> &labels)
{
    std::vector<apd_cluster_op> ops;
    for (const auto &o : operations) ops.push_back({(uint32_t)o.merge_i, (uint32_t)o.merge_j, (uint32_t)o.into, o.distance, (uint32_t)o.operation});
    const std::vector<uint32_t> roots(clusters.begin(), clusters.end());
    std::vector<const char *> lab;
    for (const auto &l : labels) lab.push_back(l.c_str());
    uint64_t bytes = 0;
    uint32_t n_strings = 0;
    std::vector<uint32_t> which(roots.size() + 1);
    check(apd_dendrograms(ops.data(), (uint32_t)ops.size(), roots.data(), (uint32_t)roots.size(), lab.data(), (uint32_t)lab.size(), nullptr, 0, &bytes,
                          which.data(), &n_strings));
    std::vector<char> buf(bytes + 1);
    check(apd_dendrograms(ops.data(), (uint32_t)ops.size(), roots.data(), (uint32_t)roots.size(), lab.data(), (uint32_t)lab.size(), buf.data(), bytes,
                          &bytes, which.data(), &n_strings));
    std::map<std::size_t, std::string> out;
    const char *p = buf.data();
    for (uint32_t i = 0; i < n_strings; ++i) { out[roots[which[i]]] = p; p += out[roots[which[i]]].size() + 1; }
    return out;
}

}  // namespace apd
