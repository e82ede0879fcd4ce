// Percentile threshold + UPGMA (reference src/numerics.rs:125-133, src/clustering.rs:40-210).
#include <algorithm>
#include <cstring>
#include <vector>

#include "apd_internal.h"

using namespace apd;

extern "C" int apd_percentile(apd_context *ctx, const float *x, uint64_t len, float perc, int x_on_device, float *value)
{
    (void)ctx; (void)x; (void)len; (void)perc; (void)x_on_device; (void)value;
    return APD_ERR_UNSUPPORTED;
}

extern "C" int apd_clustering(apd_context *ctx, const float *distances, int distances_on_device, uint32_t n, float perc,
                              apd_cluster_op *ops, uint32_t *n_ops, uint32_t *roots, uint32_t *n_roots, float *threshold)
{
    (void)ctx; (void)distances; (void)distances_on_device; (void)n; (void)perc; (void)ops; (void)n_ops; (void)roots;
    (void)n_roots; (void)threshold;
    return APD_ERR_UNSUPPORTED;
}

// clustering.rs:40-76: replay the merge list into leaf lists.  Pure bookkeeping on <= 2n ids, host side.
extern "C" int apd_cluster_sets(const apd_cluster_op *ops, uint32_t n_ops, const uint32_t *roots, uint32_t n_roots,
                                uint32_t n, uint32_t *members, uint32_t *set_off, uint32_t *n_sets)
{
    if ((n_ops && !ops) || (n_roots && !roots) || !members || !set_off || !n_sets) return APD_ERR_INVALID_ARG;
    const uint64_t ids = (uint64_t)n + n_ops + 2;
    std::vector<std::vector<uint32_t>> results(ids);
    std::vector<char> present(ids, 0);
    for (uint32_t t = 0; t < n_ops; ++t) {
        const uint32_t i = ops[t].merge_i, j = ops[t].merge_j, k = ops[t].into;
        if (k >= ids) return APD_ERR_INVALID_ARG;
        std::vector<uint32_t> cluster;
        if (i < ids && present[i]) cluster.insert(cluster.end(), results[i].begin(), results[i].end());   // :48-49
        else cluster.push_back(i);                                                                         // :51
        if (j < ids && present[j]) cluster.insert(cluster.end(), results[j].begin(), results[j].end());   // :53-54
        else cluster.push_back(j);                                                                         // :56
        results[k] = std::move(cluster);                                                                   // :58
        present[k] = 1;
    }
    uint32_t ns = 0, pos = 0;
    set_off[0] = 0;
    for (uint32_t r = 0; r < n_roots; ++r) {                                                               // :61
        const uint32_t id = roots[r];
        if (id < ids && present[id]) {
            for (uint32_t v : results[id])
                if (v < n) { if (pos >= n) return APD_ERR_INVALID_ARG; members[pos++] = v; }               // :65-69
            set_off[++ns] = pos;
        }                                                                                                  // else: "Cluster not found", :71
    }
    *n_sets = ns;
    return APD_OK;
}
