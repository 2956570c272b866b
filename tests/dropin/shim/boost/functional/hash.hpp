// boost/functional/hash.hpp -- stand-in (tests/dropin only): hash_combine / hash_value as the callers' matrix_hash uses them
#ifndef DROPIN_BOOST_HASH_HPP_
#define DROPIN_BOOST_HASH_HPP_
#include <cstddef>
#include <functional>
namespace boost {
template <typename T>
inline void hash_combine(std::size_t& seed, const T& v) {
  seed ^= std::hash<T>()(v) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
}
}  // namespace boost
#endif
