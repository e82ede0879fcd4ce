// C++ harness over include/apd.hpp: reads a batch (text), runs the reference's main.rs:187-203 sequence through the
// C++ mirror -- AlignmentWorkers::new -> align_all -> AgglomerativeClustering::clustering -> cluster_sets -- and prints
// the results for tests/test_gpu_cpp_mirror.py to compare with the oracle.
//   usage: harness <input.txt>    input: n dim pct ins del match perc, then per sequence: len, then len*dim floats
//          harness <input.txt> <auto_encoder.bin> <Discovery.toml>   the same with the reference's on-disk artefacts: the
//              parameters come from the TOML file, every sequence goes through AutoEncoder::from_file(..) -> encoded() first
//              (main.rs:142-161), and the weight file is written back with save_file and compared byte for byte
#include <cstdio>
#include <cstring>
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
        if (argc >= 4) {
            cfg = apd::Discovery::from_toml(argv[3]);
            std::printf("toml %zu %zu %zu %zu %.9g %zu %zu %.9g %.9g %.9g %.9g %.9g %zu %.9g %zu %.9g %.9g\n", cfg.dft_win, cfg.dft_step, cfg.ceps_filter,
                        cfg.vat_moving, cfg.vat_percentile, cfg.vat_min_len, cfg.alignment_workers, cfg.clustering_percentile,
                        cfg.warping_band_percentage, cfg.insertion_penalty, cfg.deletion_penalty, cfg.match_penalty, cfg.auto_encoder,
                        cfg.learning_rate, cfg.epochs, cfg.epoch_drop, cfg.drop);
            const apd::AutoEncoder nn = apd::AutoEncoder::from_file(argv[2]);
            std::printf("latent %zu\n", nn.n_latent());
            for (auto &s : seqs) s = nn.encoded(ctx, s);
            std::printf("enc0");
            for (float v : seqs[0].frames) std::printf(" %.9g", v);
            std::printf("\n");
            const std::string copy = std::string(argv[2]) + ".copy";
            nn.save_file(copy);
            std::ifstream a(argv[2], std::ios::binary), b(copy, std::ios::binary);
            const std::string sa((std::istreambuf_iterator<char>(a)), std::istreambuf_iterator<char>()), sb((std::istreambuf_iterator<char>(b)), std::istreambuf_iterator<char>());
            std::printf("roundtrip %d\n", (int)(sa == sb && !sa.empty()));
        }
        apd::AlignmentWorkers workers(ctx, seqs);
        workers.align_all(cfg);
        std::printf("dist");
        for (float v : workers.result) std::printf(" %.9g", v);
        std::printf("\n");
        {   // the multi-device entry with the one device a test box has: same bits, one rank
            const std::vector<float> single = workers.result;
            const uint32_t seen = workers.align_all(cfg, std::vector<int>{0});
            std::printf("multi %u %d\n", seen, (int)(std::memcmp(single.data(), workers.result.data(), single.size() * sizeof(float)) == 0));
            workers.align_all(cfg, std::vector<int>{0});                                  // the handle is kept: second call, same bits
            std::printf("multi_again %d %s\n", (int)(std::memcmp(single.data(), workers.result.data(), single.size() * sizeof(float)) == 0),
                        std::strncmp(workers.collective(), "rccl", 4) == 0 ? "rccl" : workers.collective());
        }
        auto res = apd::AgglomerativeClustering::clustering(ctx, workers.result, n, cfg.clustering_percentile);
        {   // SURVEY.md section 8(b)'s two one-call entry points, straight through the C ABI: same bits as the mirror's calls
            std::vector<float> flat;
            std::vector<uint64_t> off(1, 0);
            for (const auto &s : seqs) { flat.insert(flat.end(), s.frames.begin(), s.frames.end()); off.push_back(off.back() + s.len()); }
            std::vector<float> out(n * n, -1.0f);
            const int rc1 = apd_dtw_all_pairs(flat.data(), off.data(), (uint32_t)n, (uint32_t)seqs[0].n_bins, cfg.warping_band_percentage,
                                              cfg.insertion_penalty, cfg.deletion_penalty, cfg.match_penalty, 1, out.data());
            std::vector<apd_cluster_op> ops(n);
            std::vector<uint32_t> roots(n);
            uint32_t n_ops = 0, n_roots = 0;
            const int rc2 = apd_upgma(workers.result.data(), (uint32_t)n, cfg.clustering_percentile, ops.data(), &n_ops, roots.data(), &n_roots);
            bool same_ops = rc2 == 0 && n_ops == res.first.size() && n_roots == res.second.size();
            for (uint32_t t = 0; same_ops && t < n_ops; ++t)
                same_ops = ops[t].merge_i == res.first[t].merge_i && ops[t].merge_j == res.first[t].merge_j && ops[t].into == res.first[t].into &&
                           std::memcmp(&ops[t].distance, &res.first[t].distance, sizeof(float)) == 0;
            std::printf("onecall %d %d\n", (int)(rc1 == 0 && std::memcmp(out.data(), workers.result.data(), out.size() * sizeof(float)) == 0), (int)same_ops);
        }
        for (const auto &o : res.first) std::printf("op %zu %zu %zu %.9g %d\n", o.merge_i, o.merge_j, o.into, o.distance, (int)o.operation);
        std::printf("roots");
        for (auto r : res.second) std::printf(" %zu", r);
        std::printf("\n");
        for (const auto &s : apd::AgglomerativeClustering::cluster_sets(res.first, res.second, n)) {
            std::printf("set");
            for (auto m : s) std::printf(" %zu", m);
            std::printf("\n");
        }
        std::vector<std::string> labels;
        for (std::size_t i = 0; i < n; ++i) labels.push_back("{img" + std::to_string(i) + "}");
        for (const auto &kv : apd::dendrograms(res.first, res.second, labels)) std::printf("dendro %zu %s\n", kv.first, kv.second.c_str());
        apd::Alignment a(ctx);
        a.construct_alignment(seqs[0], seqs[1], cfg.alignment_params(std::max(seqs[0].len(), seqs[1].len())));
        std::printf("pair01 %.9g\n", a.score());
    } catch (const apd::Error &e) {
        std::printf("error %d %s\n", e.status, e.what());
        return 1;
    }
    return 0;
}
