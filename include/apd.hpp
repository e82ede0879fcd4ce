// apd.hpp -- header-only C++ mirror of the reference's Rust surface for the alignment/clustering path, over the C ABI
// of apd.h.  Names and argument meaning follow the Rust items (file:line cited at each); the compute is libapd_hip.so.
// Errors that are panics in the reference (zero-length sequence, percentile index out of range) throw apd::Error.
#pragma once
#include <cstdint>
#include <limits>
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

// alignments.rs:77-83
struct AlignmentParams {
    std::size_t warping_band;
    float insertion_penalty, deletion_penalty, match_penalty;
    static AlignmentParams default_for(std::size_t len) { return {len, 1.0f, 1.0f, 1.0f}; }   // :86-93
};

// discovery.rs:7-26 -- the fields this path reads
struct Discovery {
    float warping_band_percentage = 1.0f, insertion_penalty = 1.0f, deletion_penalty = 1.0f, match_penalty = 1.0f;
    std::size_t alignment_workers = 4;      // accepted, unused: the GPU grid replaces the OS threads
    float clustering_percentile = 0.05f;
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
        std::vector<uint64_t> offsets(n + 1, 0);
        std::vector<float> frames;
        const uint32_t dim = n ? (uint32_t)this->data[0].n_bins : 1;
        for (std::size_t s = 0; s < n; ++s) {
            if (this->data[s].n_bins != dim) throw Error(APD_ERR_INVALID_ARG, "sequences must share n_bins");
            offsets[s + 1] = offsets[s] + this->data[s].len();
            frames.insert(frames.end(), this->data[s].frames.begin(), this->data[s].frames.end());
        }
        check(apd_batch_create(ctx_.get(), frames.data(), offsets.data(), (uint32_t)n, dim ? dim : 1, 0, &batch_), ctx_.get());
    }
    ~AlignmentWorkers() { if (batch_) apd_batch_destroy(batch_); }
    AlignmentWorkers(const AlignmentWorkers &) = delete;
    AlignmentWorkers &operator=(const AlignmentWorkers &) = delete;
    void align_all(const Discovery &params)                                          // :31-67, blocking
    {
        const apd_align_config c = params.config();
        check(apd_align_all(ctx_.get(), batch_, &c, result.data()), ctx_.get());
    }
    std::vector<NDSequence> data;
    std::vector<float> result;                                                       // n*n row-major, diagonal 0.0
  private:
    Context &ctx_;
    apd_batch *batch_ = nullptr;
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

}  // namespace apd
