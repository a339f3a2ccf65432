// tr1_order.h -- iteration order of libstdc++'s std::tr1::unordered_map after a sequence of distinct-key insertions,
// without building the container.
//
// Why: gridSampling / subSampleFrame (src/utility.cpp:167-201) emit the keypoints in the iteration order of a
// std::tr1::unordered_map<voxel, ...>, so the keypoint ORDER -- which decides what the ordered cut-off at max_num_residuals
// keeps (optimize.cpp:107) -- is defined by that container's layout: per-bucket singly linked lists, a new node goes to the
// HEAD of its bucket, a rehash walks the old buckets in order and again pushes every node to the head of its new bucket,
// iteration = bucket 0, 1, 2, ... each from head to tail (tr1/hashtable.h: _M_insert_bucket, _M_rehash).  Building the real
// container costs one heap node per voxel (~1 ms for the 14k voxels of a 24k-point frame); replaying the same moves on
// flat index arrays costs ~30 us.
//
// Nothing about the growth policy is restated from memory: the bucket-count schedule (initial count, the element counts
// at which _Prime_rehash_policy grows the table, the new counts) is RECORDED once from a real
// std::tr1::unordered_map<int, int> in this process and cached.  tests/test_tr1_order.py checks the replay against the
// real container (srl_grid_sampling) on random and adversarial key sets.
#pragma once
#include <tr1/unordered_map>

#include <cstddef>
#include <mutex>
#include <utility>
#include <vector>

namespace srl {

class Tr1Order {
public:
    // hashes[i] = std::hash<key> of the i-th inserted (distinct) key; out[r] = insertion index of the r-th element in
    // iteration order
    static void order(const std::size_t *hashes, int n, int *out) {
        const Schedule &S = schedule(n);
        std::vector<int> next((size_t)n, -1);
        std::size_t nb = S.initial;
        std::vector<int> head(nb, -1), head2;
        std::size_t step = 0;
        for (int i = 0; i < n; i++) {
            if (step < S.grow.size() && S.grow[step].first == (std::size_t)i + 1) {
                // the insertion that makes the element count reach grow[step].first rehashes FIRST (old nodes only) ...
                const std::size_t nb2 = S.grow[step].second;
                head2.assign(nb2, -1);
                for (std::size_t b = 0; b < nb; b++) {
                    int p = head[b];
                    while (p >= 0) {
                        const int nx = next[(size_t)p];
                        const std::size_t j = hashes[p] % nb2;
                        next[(size_t)p] = head2[j];
                        head2[j] = p;
                        p = nx;
                    }
                }
                head.swap(head2);
                nb = nb2;
                step++;
            }
            const std::size_t b = hashes[i] % nb;          // ... then links the new node at the head of its bucket
            next[(size_t)i] = head[b];
            head[b] = i;
        }
        int r = 0;
        for (std::size_t b = 0; b < nb; b++)
            for (int p = head[b]; p >= 0; p = next[(size_t)p]) out[r++] = p;
    }

private:
    struct Schedule {
        std::size_t initial = 0;
        std::size_t recorded = 0;                                     // element counts covered
        std::vector<std::pair<std::size_t, std::size_t>> grow;        // (element count reached by the rehashing insertion, new bucket count)
    };
    static const Schedule &schedule(int n) {
        static Schedule S;
        static std::mutex mu;
        std::lock_guard<std::mutex> lock(mu);
        if (S.recorded >= (std::size_t)n && S.initial) return S;
        // record from the real container (int keys: only the policy matters, not the hash)
        std::size_t upto = 1024;
        while (upto < (std::size_t)n) upto *= 2;
        std::tr1::unordered_map<int, int> probe;
        Schedule R;
        R.initial = probe.bucket_count();
        std::size_t nb = R.initial;
        for (std::size_t e = 1; e <= upto; e++) {
            probe[(int)e] = 0;
            if (probe.bucket_count() != nb) { nb = probe.bucket_count(); R.grow.push_back(std::make_pair(e, nb)); }
        }
        R.recorded = upto;
        S = R;
        return S;
    }
};

}  // namespace srl
