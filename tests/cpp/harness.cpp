// C++ harness over include/apd.hpp: reads a batch (text), runs the reference's main.rs:187-203 sequence through the
// C++ mirror -- AlignmentWorkers::new -> align_all -> AgglomerativeClustering::clustering -> cluster_sets -- and prints
// the results for tests/test_gpu_cpp_mirror.py to compare with the oracle.
//   usage: harness <input.txt>    input: n dim pct ins del match perc, then per sequence: len, then len*dim floats
#include <cstdio>
#include <fstream>
#include <iostream>

#include "../../include/apd.hpp"

int main(int argc, char **argv)
{
    if (argc < 2) return 2;
    std::ifstream in(argv[1]);
    std::size_t n, dim;
    apd::Discovery cfg;
    in >> n >> dim >> cfg.warping_band_percentage >> cfg.insertion_penalty >> cfg.deletion_penalty >> cfg.match_penalty >>
        cfg.clustering_percentile;
    std::vector<apd::NDSequence> seqs(n);
    for (auto &s : seqs) {
        std::size_t len;
        in >> len;
        s.n_bins = dim;
        s.frames.resize(len * dim);
        for (auto &v : s.frames) in >> v;
    }
    try {
        apd::Context ctx(0);
        apd::AlignmentWorkers workers(ctx, seqs);
        workers.align_all(cfg);
        std::printf("dist");
        for (float v : workers.result) std::printf(" %.9g", v);
        std::printf("\n");
        auto res = apd::AgglomerativeClustering::clustering(ctx, workers.result, n, cfg.clustering_percentile);
        for (const auto &o : res.first) std::printf("op %zu %zu %zu %.9g %d\n", o.merge_i, o.merge_j, o.into, o.distance, (int)o.operation);
        std::printf("roots");
        for (auto r : res.second) std::printf(" %zu", r);
        std::printf("\n");
        for (const auto &s : apd::AgglomerativeClustering::cluster_sets(res.first, res.second, n)) {
            std::printf("set");
            for (auto m : s) std::printf(" %zu", m);
            std::printf("\n");
        }
        apd::Alignment a(ctx);
        a.construct_alignment(seqs[0], seqs[1], cfg.alignment_params(std::max(seqs[0].len(), seqs[1].len())));
        std::printf("pair01 %.9g\n", a.score());
    } catch (const apd::Error &e) {
        std::printf("error %d %s\n", e.status, e.what());
        return 1;
    }
    return 0;
}
