// Stand-in for <boost/functional/hash.hpp>: only hash_combine / hash<T>, in the
// classic (Boost < 1.81) formulation.  The reference does not pin a Boost
// version; see oracle/mpl_oracle.cpp header.  Our own code.
#ifndef MPL_ORACLE_BOOST_HASH_STUB
#define MPL_ORACLE_BOOST_HASH_STUB
#include <cstddef>
namespace boost {
inline std::size_t hash_value(int v) { return static_cast<std::size_t>(v); }
template <class T>
inline void hash_combine(std::size_t &seed, const T &v) {
  seed ^= hash_value(v) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
}
template <class T>
struct hash {
  std::size_t operator()(const T &v) const { return hash_value(v); }  // ADL finds the user's overload
};
}  // namespace boost
#endif
