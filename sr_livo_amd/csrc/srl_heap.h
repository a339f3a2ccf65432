// srl_heap.h -- the bounded max-heap of lioOptimization::searchNeighbors (optimize.cpp:355-363, 394-404, 411-422),
// i.e. std::priority_queue<tuple<double, ...>, vector, comparator> with comparator(left, right) = left.distance <
// right.distance, restated operation by operation from libstdc++'s <bits/stl_heap.h> (GCC 11: __push_heap,
// __adjust_heap, __pop_heap; priority_queue::emplace = push_back + push_heap, pop = pop_heap + pop_back).
//
// Why it exists: when candidate distances tie, which of the tied points survive and in which order they come out
// is decided by the heap's internal arrangement, not by any simple rule (with K = 4 and five
// candidates tied at the cut-off the survivor was the 3rd visited).  The device kernels rank by counting; whenever
// they detect a (near-)tie among the K+1 smallest distances they replay the reference's literal sequence with these
// routines, one lane, candidates in visit order.  Host-compilable: tests/test_heap_replay.py checks the routines
// against the real std::priority_queue (oracle side) without a GPU.
#pragma once
#include "srl_hash.h"   // SRL_HD

// heap arrays: hd[i] = distance, he[i] = candidate id (does not take part in comparisons)

// std::push_heap after push_back: size = number of elements INCLUDING the new one at [size - 1] = (v, e)
SRL_HD inline void srl_heap_push(double *hd, int *he, int size, double v, int e) {
    int hole = size - 1;
    int parent = (hole - 1) / 2;
    while (hole > 0 && hd[parent] < v) {              // __push_heap: comp(first + parent, value)
        hd[hole] = hd[parent]; he[hole] = he[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    hd[hole] = v; he[hole] = e;
}

// std::pop_heap + pop_back: size = number of elements BEFORE the pop; afterwards size - 1 remain
SRL_HD inline void srl_heap_pop(double *hd, int *he, int size) {
    if (size <= 1) return;                             // pop_heap does nothing for one element; pop_back drops it
    const int len = size - 1;                          // __pop_heap(first, last - 1, last - 1): value = *(last - 1)
    const double v = hd[len];
    const int e = he[len];
    // *result = *first is the element that pop_back discards: not stored
    int hole = 0, child = 0;                           // __adjust_heap(first, 0, len, value)
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (hd[child] < hd[child - 1]) child--;        // comp(first + secondChild, first + (secondChild - 1))
        hd[hole] = hd[child]; he[hole] = he[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        hd[hole] = hd[child - 1]; he[hole] = he[child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;                       // __push_heap(first, hole, 0, value)
    while (hole > 0 && hd[parent] < v) {
        hd[hole] = hd[parent]; he[hole] = he[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    hd[hole] = v; he[hole] = e;
}

// one candidate of the visiting loop (optimize.cpp:397-404); returns the new heap size
SRL_HD inline int srl_heap_offer(double *hd, int *he, int size, int K, double distance, int e) {
    if (size == K) {
        if (distance < hd[0]) {                        // distance < std::get<0>(priority_queue.top())
            srl_heap_pop(hd, he, size);
            hd[size - 1] = distance; he[size - 1] = e; // emplace: push_back ...
            srl_heap_push(hd, he, size, distance, e);  // ... + push_heap
        }
        return size;
    }
    hd[size] = distance; he[size] = e;
    srl_heap_push(hd, he, size + 1, distance, e);
    return size + 1;
}

// the read-out loop (optimize.cpp:411-422): closest_neighbors[size - 1 - i] = top(); pop()
SRL_HD inline void srl_heap_drain(double *hd, int *he, int size, int *out_ids) {
    for (int i = 0; i < size; ++i) {
        out_ids[size - 1 - i] = he[0];
        srl_heap_pop(hd, he, size - i);
    }
}
