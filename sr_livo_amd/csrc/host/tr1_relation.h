// tr1_relation.h -- the iteration order of libstdc++'s std::tr1::unordered_map as a RELATION between two elements, for code that cannot
// replay the container's moves one after the other (the device ordering of srl_frame_select_keypoints; host/tr1_order.h is the replay).
//
// gridSampling emits the keypoints in the iteration order of a std::tr1::unordered_map<voxel, ...> (utility.cpp:167-201).  With
//   e        insertion index of a (distinct) key = its first-occurrence rank,
//   era(e)   number of rehashes done when e was linked (the insertion that triggers a rehash is linked into the NEW table),
//   n_L      bucket count at level L (level 0 = the initial table), G = level after all insertions,
// the container's moves (tr1/hashtable.h: _M_insert_bucket links a node at the HEAD of its bucket; _M_rehash walks the old table in
// iteration order and links every node at the head of its new bucket) give, for the iteration order T_L at level L of elements with
// era <= L:
//     different buckets (hash mod n_L):  the smaller bucket is met first;
//     same bucket: a chain is read head to tail = its arrivals REVERSED, and the arrivals of level L are the level L-1 table in its
//     iteration order followed by the insertions of level L in insertion order.  So if x or y was inserted AT level L the larger
//     insertion index is met first (elements of earlier levels have smaller indices), else T_L(x, y) = T_{L-1}(y, x).
// Final position of an element = (elements in smaller buckets at level G) + (elements of its bucket that precede it under T_G).
// tests/test_tr1_order.py checks this relation (srl_debug_tr1_order_by_relation: the same functions compiled for the host) against the
// real container; tests/test_gpu_frame_order.py checks the device result against the replay.
#pragma once

#if defined(__HIPCC__)
#define SRL_TR1_HD __host__ __device__
#else
#define SRL_TR1_HD
#endif

#define SRL_TR1_MAX_STEPS 24
#define SRL_TR1_BUCKET_SLOTS 16      // voxels of one bucket the device ranks in place (mean occupancy <= 1: max load factor 1)

// growth schedule (recorded from a real container: Tr1Order::export_schedule)
struct SrlTr1Sched {
    int steps;
    unsigned nb[SRL_TR1_MAX_STEPS + 1];      // bucket count at level L
    unsigned first[SRL_TR1_MAX_STEPS];       // the insertion that brings the element count to first[s] rehashes (level s -> s + 1) before it links its node
};

// rehashes done once `count` elements are linked; era(e) = srl_tr1_level(S, e + 1)
SRL_TR1_HD inline int srl_tr1_level(const SrlTr1Sched *S, unsigned count) {
    int g = 0;
    const int st = S->steps;
    while (g < st && S->first[g] <= count) ++g;
    return g;
}

// T_L(x, y): is x met before y?  x != y, both in one bucket at level L, era(x), era(y) <= L.
//   era:    functor e -> era(e)                (the schedule itself, or a table filled from it)
//   bucket: functor (e, level) -> hash(e) mod n_level   (computed, or looked up where a caller has cached a level)
// The device passes tables: a per-lane walk over the schedule and a 64-bit remainder per level are chains of dependent instructions
// that one lane pays for every pair of its bucket.
template <class EraOf, class BucketOf>
SRL_TR1_HD inline bool srl_tr1_before_t(int L, unsigned x, unsigned y, const EraOf &era, const BucketOf &bucket) {
    for (;;) {
        if (era(x) == L || era(y) == L) return y < x;
        const unsigned t = x; x = y; y = t;
        --L;
        const unsigned bx = bucket(x, L), by = bucket(y, L);
        if (bx != by) return bx < by;
    }
}
struct SrlTr1EraOfSchedule {
    const SrlTr1Sched *S;
    SRL_TR1_HD int operator()(unsigned e) const { return srl_tr1_level(S, e + 1); }
};
struct SrlTr1BucketOfHash {
    const SrlTr1Sched *S;
    const unsigned long long *hash;
    SRL_TR1_HD unsigned operator()(unsigned e, int L) const { return (unsigned)(hash[e] % S->nb[L]); }
};
SRL_TR1_HD inline bool srl_tr1_before(const SrlTr1Sched *S, const unsigned long long *hash, int L, unsigned x, unsigned y) {
    return srl_tr1_before_t(L, x, y, SrlTr1EraOfSchedule{S}, SrlTr1BucketOfHash{S, hash});
}
