#ifndef BOOST_BIND_LITE_H_
#define BOOST_BIND_LITE_H_
namespace boost { template <typename... A> int bind(A...) { return 0; } }
struct PlaceholderLite {};
static const PlaceholderLite _1, _2;
#endif
