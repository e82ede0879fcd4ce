// Sanitizer fuzz of the host-only format code (csrc/formats.hip compiled as plain C++ with -fsanitize=address,undefined; the
// pool offers no GPU sanitizer, and this code never touches the GPU).  Random and mutated inputs into every entry point of
// the "formats either side of the path" section of include/apd.h: nothing may crash, leak, overflow or unwind.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -x c++ tools/fuzz/fuzz_formats.cpp -o build/fuzz_formats
//   build/fuzz_formats [iterations] [seed]
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../audio_pattern_discovery_amd/csrc/formats.hip"

int main(int argc, char **argv)
{
    const long iters = argc > 1 ? std::atol(argv[1]) : 20000;
    std::mt19937_64 rng(argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 12345);
    auto rnd = [&](uint64_t n) { return (uint64_t)(rng() % (n ? n : 1)); };
    long ok_dendro = 0, ok_toml = 0, ok_ae = 0;
    const char *labels[6] = {"a", "b", "c", "{img3}", "", "f"};
    const std::string good_toml =
        "dft_win = 256\ndft_step = 128\nceps_filter = 32\nauto_encoder = 10\nlearning_rate = 0.1\nepochs = 25\nepoch_drop = 5.0\ndrop = 0.5\n"
        "vat_moving = 15\nvat_percentile = 0.95\nvat_min_len = 150\nwarping_band_percentage = 1.0\ninsertion_penalty = 1.0\n"
        "deletion_penalty = 1.0\nmatch_penalty = 1.0\nalignment_workers = 4\nclustering_percentile = 0.05\n";
    for (long it = 0; it < iters; ++it) {
        // --- dendrograms: op lists with repeated ids, self references, ids out of range, wrong kinds
        const uint32_t n_ops = (uint32_t)rnd(12), n_labels = (uint32_t)rnd(7);
        std::vector<apd_cluster_op> ops(n_ops);
        for (uint32_t t = 0; t < n_ops; ++t) {
            const uint32_t span = rnd(4) == 0 ? 0xFFFFFFFFu : 10;
            ops[t].merge_i = (uint32_t)rnd(span); ops[t].merge_j = (uint32_t)rnd(span); ops[t].into = (uint32_t)rnd(span);
            ops[t].distance = 0.5f; ops[t].operation = (uint32_t)rnd(rnd(10) == 0 ? 7 : 4);
            if (t && rnd(3) == 0) ops[t].merge_i = ops[rnd(t)].into;          // plausible cluster operands
            if (t && rnd(3) == 0) ops[t].merge_j = ops[rnd(t)].into;
            if (rnd(8) == 0) ops[t].into = ops[t].merge_i;                      // an op that overwrites its own operand
        }
        std::vector<uint32_t> roots(rnd(5));
        for (uint32_t &r : roots) r = n_ops && rnd(2) ? ops[rnd(n_ops)].into : (uint32_t)rnd(12);
        uint64_t n_bytes = 0;
        uint32_t n_strings = 0;
        std::vector<uint32_t> which(roots.size() + 1);
        int rc = apd_dendrograms(ops.data(), n_ops, roots.data(), (uint32_t)roots.size(), labels, n_labels, nullptr, 0, &n_bytes, which.data(), &n_strings);
        if (rc == APD_OK) {
            std::vector<char> out(n_bytes + 1);
            rc = apd_dendrograms(ops.data(), n_ops, roots.data(), (uint32_t)roots.size(), labels, n_labels, out.data(), n_bytes, &n_bytes, which.data(), &n_strings);
            if (rc != APD_OK) { std::fprintf(stderr, "dendrograms: size query ok, fill failed (%d)\n", rc); return 1; }
            ++ok_dendro;
        }
        // --- Discovery.toml: the shipped text with random byte edits, deleted lines, extra keys
        std::string text = good_toml;
        for (uint64_t e = rnd(4); e > 0; --e) {
            const uint64_t pos = rnd(text.size());
            switch (rnd(4)) {
                case 0: text[pos] = (char)rnd(256); break;
                case 1: text.erase(pos, rnd(20)); break;
                case 2: text.insert(pos, "extra_key = 7\n"); break;
                default: text.insert(pos, std::string(rnd(3), (char)('0' + rnd(70)))); break;
            }
        }
        apd_discovery d;
        if (apd_discovery_parse_toml(text.c_str(), &d) == APD_OK) ++ok_toml;
        // --- auto_encoder.bin: a valid image, truncated / extended / with corrupted length words
        const uint32_t d_in = 1 + (uint32_t)rnd(5), latent = 1 + (uint32_t)rnd(4);
        std::vector<float> we(d_in * latent, 0.25f), wd(d_in * latent, -0.5f), be(latent, 1.0f), bd(d_in, 2.0f);
        uint64_t need = 0;
        apd_autoencoder_serialize(we.data(), wd.data(), be.data(), bd.data(), d_in, latent, nullptr, 0, &need);
        std::vector<unsigned char> img(need);
        apd_autoencoder_serialize(we.data(), wd.data(), be.data(), bd.data(), d_in, latent, img.data(), need, &need);
        if (rnd(2)) img.resize(rnd(need + 8));
        for (uint64_t e = rnd(3); e > 0 && !img.empty(); --e) img[rnd(img.size())] = (unsigned char)rnd(256);
        apd_autoencoder_view view;
        if (apd_autoencoder_parse(img.data(), img.size(), &view) == APD_OK) {
            std::vector<float> m(view.w_encode.len);
            apd_autoencoder_copy(img.data(), &view.w_encode, m.data());
            ++ok_ae;
        }
    }
    std::printf("fuzz_formats: %ld iterations, accepted: %ld dendrogram lists, %ld toml texts, %ld weight images; no sanitizer report\n", iters,
                ok_dendro, ok_toml, ok_ae);
    return 0;
}
