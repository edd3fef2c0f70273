// ORACLE — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
// Restatement of the .NET 8 BCL algorithms the reference's results depend on at tie boundaries.
// The BCL is NOT under /root/reference (it is the runtime: Microsoft.NETCore.App 8.0.x, see
// src/Infidex/Infidex.csproj:4). These follow the published dotnet/runtime sources from memory:
//   System.Collections.Generic.ArraySortHelper<T>.IntrospectiveSort (List<T>.Sort / Array.Sort / Span.Sort)
//   System.Collections.Generic.PriorityQueue<TElement,TPriority> (array-backed 4-ary min-heap)
//   System.Math.Round(double) (banker's rounding, MidpointRounding.ToEven)
// PARITY UNPINNED at these boundaries: no reference test constructs a tie there.
#pragma once
#include <vector>
#include <cmath>
#include <cstdint>
#include <utility>

namespace orc { namespace dotnet {

// ---- ArraySortHelper<T>.IntrospectiveSort with a Comparison<T> ----------------------------
template <class T, class Cmp>
struct IntroSorter {
    T* k; Cmp cmp;   // cmp(a,b) -> int (<0, 0, >0)
    void swap_if_greater(int i, int j) { if (cmp(k[i], k[j]) > 0) std::swap(k[i], k[j]); }
    static int log2u(unsigned v) { int r = 0; while (v >>= 1) r++; return r; }
    void insertion(int lo, int n) {
        for (int i = 0; i < n - 1; i++) {
            T t = k[lo + i + 1];
            int j = i;
            while (j >= 0 && cmp(t, k[lo + j]) < 0) { k[lo + j + 1] = k[lo + j]; j--; }
            k[lo + j + 1] = t;
        }
    }
    void down_heap(int lo, int i, int n) {
        T d = k[lo + i - 1];
        while (i <= n / 2) {
            int child = 2 * i;
            if (child < n && cmp(k[lo + child - 1], k[lo + child]) < 0) child++;
            if (!(cmp(d, k[lo + child - 1]) < 0)) break;
            k[lo + i - 1] = k[lo + child - 1];
            i = child;
        }
        k[lo + i - 1] = d;
    }
    void heapsort(int lo, int n) {
        for (int i = n / 2; i >= 1; i--) down_heap(lo, i, n);
        for (int i = n; i > 1; i--) { std::swap(k[lo], k[lo + i - 1]); down_heap(lo, 1, i - 1); }
    }
    int partition(int lo, int n) {
        int hi = n - 1, mid = hi >> 1;
        T* a = k + lo;
        auto sig = [&](int i, int j) { if (cmp(a[i], a[j]) > 0) std::swap(a[i], a[j]); };
        sig(0, mid); sig(0, hi); sig(mid, hi);
        T pivot = a[mid];
        std::swap(a[mid], a[hi - 1]);
        int left = 0, right = hi - 1;
        while (left < right) {
            while (cmp(a[++left], pivot) < 0) {}
            while (cmp(pivot, a[--right]) < 0) {}
            if (left >= right) break;
            std::swap(a[left], a[right]);
        }
        if (left != hi - 1) std::swap(a[left], a[hi - 1]);
        return left;
    }
    void intro(int lo, int n, int depth) {
        int part = n;
        while (part > 1) {
            if (part <= 16) {
                if (part == 2) { swap_if_greater(lo, lo + 1); return; }
                if (part == 3) { swap_if_greater(lo, lo + 1); swap_if_greater(lo, lo + 2); swap_if_greater(lo + 1, lo + 2); return; }
                insertion(lo, part); return;
            }
            if (depth == 0) { heapsort(lo, part); return; }
            depth--;
            int p = partition(lo, part);
            intro(lo + p + 1, part - (p + 1), depth);
            part = p;
        }
    }
};
template <class T, class Cmp>
inline void sort(T* keys, int n, Cmp cmp) {
    if (n < 2) return;
    IntroSorter<T, Cmp> s{keys, cmp};
    s.intro(0, n, 2 * (IntroSorter<T, Cmp>::log2u((unsigned)n) + 1));
}
template <class T, class Cmp>
inline void sort(std::vector<T>& v, Cmp cmp) { sort(v.data(), (int)v.size(), cmp); }

// float.CompareTo semantics for non-NaN values
inline int cmp_float(float a, float b) { return a < b ? -1 : (a > b ? 1 : 0); }

// ---- PriorityQueue<TElement,TPriority>: 4-ary min-heap -------------------------------------
template <class E, class P, class Cmp>
struct PriorityQueue {
    struct Node { E e; P p; };
    std::vector<Node> nodes;
    Cmp cmp;   // cmp(p1,p2) -> int
    explicit PriorityQueue(Cmp c) : cmp(c) {}
    int count() const { return (int)nodes.size(); }
    void move_up(Node node, int idx) {
        while (idx > 0) {
            int parent = (idx - 1) >> 2;
            if (cmp(node.p, nodes[parent].p) < 0) { nodes[idx] = nodes[parent]; idx = parent; }
            else break;
        }
        nodes[idx] = node;
    }
    void move_down(Node node, int idx) {
        int size = (int)nodes.size();
        int i;
        while ((i = (idx << 2) + 1) < size) {
            Node minChild = nodes[i]; int minIdx = i;
            int ub = i + 4 < size ? i + 4 : size;
            while (++i < ub) {
                if (cmp(nodes[i].p, minChild.p) < 0) { minChild = nodes[i]; minIdx = i; }
            }
            if (cmp(node.p, minChild.p) <= 0) break;
            nodes[idx] = minChild; idx = minIdx;
        }
        nodes[idx] = node;
    }
    void enqueue(const E& e, const P& p) { nodes.push_back(Node{e, p}); move_up(Node{e, p}, (int)nodes.size() - 1); }
    const Node& peek() const { return nodes[0]; }
    Node dequeue() {
        Node root = nodes[0];
        Node last = nodes.back();
        nodes.pop_back();
        if (!nodes.empty()) move_down(last, 0);
        return root;
    }
    // returns the element that left the queue
    E enqueue_dequeue(const E& e, const P& p) {
        if (!nodes.empty()) {
            Node root = nodes[0];
            if (cmp(p, root.p) > 0) { move_down(Node{e, p}, 0); return root.e; }
        }
        return e;
    }
};

// Math.Round(double) — to even
inline double round_even(double x) { return std::nearbyint(x); }  // default FE_TONEAREST = ties-to-even

}} // namespace
