// Stand-in for <boost/unordered_map.hpp> (test infrastructure, our own code).
#ifndef MPL_ORACLE_BOOST_UNORDERED_MAP_STUB
#define MPL_ORACLE_BOOST_UNORDERED_MAP_STUB
#include <boost/functional/hash.hpp>
#include <functional>
#include <unordered_map>
namespace boost {
template <class K, class V, class H = boost::hash<K>, class E = std::equal_to<K>>
using unordered_map = std::unordered_map<K, V, H, E>;
}
#endif
