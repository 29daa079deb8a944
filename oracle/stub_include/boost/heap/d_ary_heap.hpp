// Stand-in for <boost/heap/d_ary_heap.hpp>: a mutable binary heap with
// handles, enough for the reference's state_space.h (push / top / pop /
// increase / update / erase / clear / iteration).  Our own code (test
// infrastructure); sift rules follow the documented behaviour of a 2-ary
// boost::heap::d_ary_heap: sift-up stops at equality, sift-down swaps at
// equality and takes the first maximal child.
#ifndef MPL_ORACLE_BOOST_HEAP_STUB
#define MPL_ORACLE_BOOST_HEAP_STUB
#include <cstddef>
#include <list>
#include <utility>
#include <vector>

namespace boost {
namespace heap {

template <bool B> struct mutable_ {};
template <unsigned N> struct arity {};
template <typename C> struct compare { typedef C type; };

template <typename T, typename A0, typename A1, typename A2>
class d_ary_heap;

template <typename T, bool M, unsigned N, typename Cmp>
class d_ary_heap<T, mutable_<M>, arity<N>, compare<Cmp>> {
  struct Node { T value; std::size_t pos; };
  typedef std::list<Node> Store;

 public:
  class handle_type {
   public:
    handle_type() {}
    T &operator*() const { return it_->value; }
   private:
    friend class d_ary_heap;
    explicit handle_type(typename Store::iterator it) : it_(it) {}
    typename Store::iterator it_;
  };

  class const_iterator {
   public:
    explicit const_iterator(typename std::vector<typename Store::iterator>::const_iterator p) : p_(p) {}
    const T &operator*() const { return (*p_)->value; }
    const_iterator &operator++() { ++p_; return *this; }
    bool operator!=(const const_iterator &o) const { return p_ != o.p_; }
   private:
    typename std::vector<typename Store::iterator>::const_iterator p_;
  };
  typedef const_iterator iterator;

  bool empty() const { return q_.empty(); }
  std::size_t size() const { return q_.size(); }
  const T &top() const { return q_.front()->value; }
  const_iterator begin() const { return const_iterator(q_.begin()); }
  const_iterator end() const { return const_iterator(q_.end()); }
  void clear() { q_.clear(); store_.clear(); }

  handle_type push(const T &v) {
    store_.push_back(Node{v, q_.size()});
    typename Store::iterator it = --store_.end();
    q_.push_back(it);
    up(q_.size() - 1);
    return handle_type(it);
  }
  void pop() { remove_at(0); }
  void erase(const handle_type &h) { remove_at(h.it_->pos); }
  void increase(const handle_type &h) { up(h.it_->pos); }
  void decrease(const handle_type &h) { down(h.it_->pos); }
  void update(const handle_type &h) {
    const std::size_t i = h.it_->pos;
    if (i > 0 && less(q_[(i - 1) / N], q_[i])) up(i); else down(i);
  }

 private:
  bool less(typename Store::iterator a, typename Store::iterator b) const { return cmp_(a->value, b->value); }
  void place(std::size_t i, typename Store::iterator it) { q_[i] = it; it->pos = i; }
  void swap_at(std::size_t i, std::size_t j) {
    typename Store::iterator a = q_[i], b = q_[j];
    place(i, b);
    place(j, a);
  }
  void up(std::size_t i) {
    while (i > 0) {
      const std::size_t p = (i - 1) / N;
      if (less(q_[p], q_[i])) { swap_at(p, i); i = p; } else return;
    }
  }
  void down(std::size_t i) {
    for (;;) {
      const std::size_t first = N * i + 1;
      if (first >= q_.size()) return;
      std::size_t best = first;
      for (std::size_t c = first + 1; c < first + N && c < q_.size(); c++)
        if (less(q_[best], q_[c])) best = c;
      if (!less(q_[best], q_[i])) { swap_at(best, i); i = best; } else return;
    }
  }
  void remove_at(std::size_t i) {
    typename Store::iterator victim = q_[i];
    const std::size_t last = q_.size() - 1;
    if (i != last) {
      place(i, q_[last]);
      q_.pop_back();
      if (i > 0 && less(q_[(i - 1) / N], q_[i])) up(i); else down(i);
    } else {
      q_.pop_back();
    }
    store_.erase(victim);
  }

  Store store_;
  std::vector<typename Store::iterator> q_;
  Cmp cmp_;
};

}  // namespace heap
}  // namespace boost
#endif
