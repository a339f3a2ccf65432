// tr1_order.h -- iteration order of libstdc++'s std::tr1::unordered_map after a sequence of distinct-key insertions,
// without building the container.
//
// Why: gridSampling / subSampleFrame (src/utility.cpp:167-201) emit the keypoints in the iteration order of a
// std::tr1::unordered_map<voxel, ...>, so the keypoint ORDER -- which decides what the ordered cut-off at max_num_residuals
// keeps (optimize.cpp:107) -- is defined by that container's layout: per-bucket singly linked lists, a new node goes to the
// HEAD of its bucket, a rehash walks the old buckets in order and again pushes every node to the head of its new bucket,
// iteration = bucket 0, 1, 2, ... each from head to tail (tr1/hashtable.h: _M_insert_bucket, _M_rehash).  Building the real
// container costs one heap node per voxel (~1 ms for the 14k voxels of a 24k-point frame); the same moves restated as
// counting sorts over flat arrays cost ~0.1 ms.
//
// Nothing about the growth policy is restated from memory: the bucket-count schedule (initial count, the element counts
// at which _Prime_rehash_policy grows the table, the new counts) is RECORDED once from a real
// std::tr1::unordered_map<int, int> in this process and cached.  tests/test_tr1_order.py checks the replay against the
// real container (srl_grid_sampling) on random and adversarial key sets.
#pragma once
#include <tr1/unordered_map>

#include <cstddef>
#include <cstdint>
#include <mutex>
#include <utility>
#include <vector>

namespace srl {

class Tr1Order {
    // hash % buckets without a 64-bit division (Lemire, Kaser, Kurz: "Faster remainder by direct computation", fastmod_u64):
    // M = floor((2^128 - 1) / d) + 1;  a mod d = floor(((M * a) mod 2^128) * d / 2^128), exact for every 64-bit a and d.
    // The replay takes one remainder per insertion and one per node and rehash: ~3 n of them.
    struct FastMod {
        unsigned __int128 M;
        std::size_t d;
        explicit FastMod(std::size_t d_) : M(~(unsigned __int128)0 / d_ + 1), d(d_) {}
        std::size_t operator()(std::size_t a) const {
            const unsigned __int128 low = M * a;
            const unsigned __int128 bottom = (unsigned __int128)(std::uint64_t)low * d;
            const unsigned __int128 top = (low >> 64) * d;
            return (std::size_t)((top + (bottom >> 64)) >> 64);
        }
    };

public:
    // hashes[i] = std::hash<key> of the i-th inserted (distinct) key; out[r] = insertion index of the r-th element in
    // iteration order.
    //
    // The container's moves, restated without linked lists: while the table has nb buckets, every node that ARRIVES in a
    // bucket (a node carried over by the rehash that created this table, or a fresh insertion) goes to the bucket's head,
    // so a bucket read head-to-tail is its arrivals in reverse, and the table's iteration order is
    //     traversal = stable sort by bucket of the REVERSED arrival list.
    // A rehash walks the old table in iteration order and re-links every node (= the old traversal arrives first, in
    // that order), then the insertions of the new table's lifetime arrive in insertion order.  So each table level is
    // one counting sort over flat arrays -- sequential passes, no pointer chasing.
    static void order(const std::size_t *hashes, int n, int *out) {
        const Schedule &S = schedule(n);
        std::vector<int> arrivals, trav;
        arrivals.reserve((size_t)n); trav.reserve((size_t)n);
        std::vector<unsigned> bucket((size_t)n), start;
        std::size_t nb = S.initial;
        std::size_t step = 0;
        int inserted = 0;
        auto traverse = [&](std::size_t nbk) {           // trav = iteration order of the table holding `arrivals` in nbk buckets
            const FastMod mod(nbk);
            const int m = (int)arrivals.size();
            start.assign(nbk + 1, 0u);
            for (int r = 0; r < m; r++) { const unsigned b = (unsigned)mod(hashes[arrivals[(size_t)r]]); bucket[(size_t)r] = b; start[b + 1]++; }
            for (std::size_t b = 0; b < nbk; b++) start[b + 1] += start[b];
            trav.resize((size_t)m);
            for (int r = m - 1; r >= 0; r--) trav[start[bucket[(size_t)r]]++] = arrivals[(size_t)r];     // reversed arrivals, stable
        };
        while (inserted < n) {
            // insertions until the next rehash (the insertion that brings the element count to grow[step].first rehashes
            // FIRST, then links its own node into the new table)
            const int upto = (step < S.grow.size() && S.grow[step].first - 1 < (std::size_t)n) ? (int)(S.grow[step].first - 1) : n;
            for (; inserted < upto; inserted++) arrivals.push_back(inserted);
            if (inserted >= n) break;
            traverse(nb);
            arrivals.swap(trav);                         // the old traversal arrives first in the new table
            nb = S.grow[step].second;
            step++;
        }
        traverse(nb);
        for (int r = 0; r < n; r++) out[r] = trav[(size_t)r];
    }

    // The recorded growth schedule for up to n insertions, for a caller that replays the container elsewhere (the device ordering of
    // srl_frame_select_keypoints): nb[0] = initial bucket count; the insertion that brings the element count to first[s] rehashes into
    // nb[s + 1] buckets before it links its own node.  Returns the number of growth steps written (<= max_steps), -1 if they do not fit.
    static int export_schedule(int n, unsigned *first, unsigned *nb, int max_steps) {
        const Schedule S = schedule(n);
        nb[0] = (unsigned)S.initial;
        int steps = 0;
        for (const auto &g : S.grow) {
            if (g.first > (std::size_t)n) break;
            if (steps >= max_steps || g.second > 0xFFFFFFFFull) return -1;
            first[steps] = (unsigned)g.first;
            nb[steps + 1] = (unsigned)g.second;
            steps++;
        }
        return steps;
    }

private:
    struct Schedule {
        std::size_t initial = 0;
        std::size_t recorded = 0;                                     // element counts covered
        std::vector<std::pair<std::size_t, std::size_t>> grow;        // (element count reached by the rehashing insertion, new bucket count)
    };
    // Returned BY VALUE under the lock: another context's thread asking for a longer schedule replaces the shared one
    // (and reallocates its vector) while this caller is still walking its copy.
    static Schedule schedule(int n) {
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
