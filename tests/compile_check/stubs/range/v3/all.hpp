// COMPILE-CHECK INFRASTRUCTURE ONLY (tests/test_dropin_headers.py).
// Just enough of the range-v3 surface for g++ 11 to parse the reference headers and to instantiate the
// gfx950 overloads: the reference selects <range/v3/all.hpp> when the standard library has no
// std::views::zip (detail/ranges.hpp).  Everything is std::ranges except `view_` and a small random-access
// views::zip with a proxy reference (tuple of references, assignable through const) and an owning value type -- enough for the reference's
// matrix / vector concepts to be CHECKED (they instantiate the CPU row iterators' return types); no reference
// code path is executed through it.  This header pins nothing, is not a conforming zip_view, and is never
// part of the product or of the oracle.
#pragma once
#include <algorithm>
#include <cstddef>
#include <iterator>
#include <ranges>
#include <tuple>
#include <utility>

namespace ranges {
using namespace std::ranges;

template <typename T>
concept view_ = std::ranges::view<T>;

namespace stub_detail {
// Proxy reference / value pair of the zip iterator.  std::tuple<int&, ...> is not assignable through a const
// object before C++23, which std::ranges::sort needs from a proxy reference (indirectly_writable); range-v3 has its
// own common_tuple for the same reason.  The reference's matrix generators sort a zip of three vectors
// (backend/generate.hpp), and its tests call them.
template <typename... Rs>
struct ref_tuple;
template <typename... Rs>
struct val_tuple : std::tuple<std::remove_cvref_t<Rs>...> {
  using base = std::tuple<std::remove_cvref_t<Rs>...>;
  using base::base;
  val_tuple() = default;
  val_tuple(const ref_tuple<Rs...>& r) : base(static_cast<const std::tuple<Rs...>&>(r)) {}
  val_tuple& operator=(const ref_tuple<Rs...>& r) {
    static_cast<base&>(*this) = static_cast<const std::tuple<Rs...>&>(r);
    return *this;
  }
};
template <typename... Rs>
struct ref_tuple : std::tuple<Rs...> {
  using base = std::tuple<Rs...>;
  using base::base;
  ref_tuple(const ref_tuple&) = default;
  ref_tuple(val_tuple<Rs...>& v)
      : base(std::apply([](auto&... e) { return base(e...); }, static_cast<typename val_tuple<Rs...>::base&>(v))) {}
  template <typename Src>
  void assign_from(const Src& src) const {
    [&]<std::size_t... K>(std::index_sequence<K...>) { ((std::get<K>(static_cast<const base&>(*this)) = std::get<K>(src)), ...); }
    (std::index_sequence_for<Rs...>{});
  }
  const ref_tuple& operator=(const ref_tuple& o) const {
    assign_from(static_cast<const base&>(o));
    return *this;
  }
  const ref_tuple& operator=(const val_tuple<Rs...>& v) const {
    assign_from(static_cast<const typename val_tuple<Rs...>::base&>(v));
    return *this;
  }
  ref_tuple& operator=(const ref_tuple& o) {
    assign_from(static_cast<const base&>(o));
    return *this;
  }
  ref_tuple& operator=(const val_tuple<Rs...>& v) {
    assign_from(static_cast<const typename val_tuple<Rs...>::base&>(v));
    return *this;
  }
  friend void swap(const ref_tuple& a, const ref_tuple& b) {
    val_tuple<Rs...> t(a);
    a = b;
    b = t;
  }
};

template <typename... Its>
class zip_iterator {
public:
  using iterator_concept = std::random_access_iterator_tag;
  using iterator_category = std::random_access_iterator_tag;
  using difference_type = std::ptrdiff_t;
  using reference = ref_tuple<std::iter_reference_t<Its>...>;
  using value_type = val_tuple<std::iter_reference_t<Its>...>;
  zip_iterator() = default;
  explicit zip_iterator(Its... its) : its_(its...) {}
  reference operator*() const {
    return std::apply([](auto&... it) { return reference(*it...); }, its_);
  }
  reference operator[](difference_type n) const { return *(*this + n); }
  zip_iterator& operator++() { return *this += 1; }
  zip_iterator operator++(int) { auto t = *this; ++*this; return t; }
  zip_iterator& operator--() { return *this -= 1; }
  zip_iterator operator--(int) { auto t = *this; --*this; return t; }
  zip_iterator& operator+=(difference_type n) {
    std::apply([n](auto&... it) { ((it += n), ...); }, its_);
    return *this;
  }
  zip_iterator& operator-=(difference_type n) { return *this += -n; }
  friend zip_iterator operator+(zip_iterator a, difference_type n) { return a += n; }
  friend zip_iterator operator+(difference_type n, zip_iterator a) { return a += n; }
  friend zip_iterator operator-(zip_iterator a, difference_type n) { return a -= n; }
  friend difference_type operator-(const zip_iterator& a, const zip_iterator& b) {
    return std::get<0>(a.its_) - std::get<0>(b.its_);
  }
  friend bool operator==(const zip_iterator& a, const zip_iterator& b) { return std::get<0>(a.its_) == std::get<0>(b.its_); }
  friend auto operator<=>(const zip_iterator& a, const zip_iterator& b) {
    return (std::get<0>(a.its_) - std::get<0>(b.its_)) <=> difference_type(0);
  }

private:
  std::tuple<Its...> its_;
};

template <std::ranges::view... Vs>
class zip_view : public std::ranges::view_interface<zip_view<Vs...>> {
public:
  zip_view() = default;
  explicit zip_view(Vs... vs) : vs_(std::move(vs)...) {}
  auto begin() const {
    return std::apply([](const auto&... v) { return zip_iterator<std::ranges::iterator_t<const Vs>...>(std::ranges::begin(v)...); }, vs_);
  }
  auto end() const { return begin() + static_cast<std::ptrdiff_t>(size()); }
  std::size_t size() const {
    return std::apply([](const auto&... v) { return std::min({static_cast<std::size_t>(std::ranges::size(v))...}); }, vs_);
  }

private:
  std::tuple<Vs...> vs_;
};
} // namespace stub_detail

namespace views {
using namespace std::ranges::views;

struct zip_fn {
  template <std::ranges::viewable_range... Rs>
  auto operator()(Rs&&... rs) const {
    return stub_detail::zip_view<std::views::all_t<Rs>...>(std::views::all(std::forward<Rs>(rs))...);
  }
};
inline constexpr zip_fn zip{};
} // namespace views
} // namespace ranges

namespace std {
template <typename... Rs>
struct tuple_size<::ranges::stub_detail::ref_tuple<Rs...>> : integral_constant<size_t, sizeof...(Rs)> {};
template <size_t K, typename... Rs>
struct tuple_element<K, ::ranges::stub_detail::ref_tuple<Rs...>> : tuple_element<K, tuple<Rs...>> {};
template <typename... Rs>
struct tuple_size<::ranges::stub_detail::val_tuple<Rs...>> : integral_constant<size_t, sizeof...(Rs)> {};
template <size_t K, typename... Rs>
struct tuple_element<K, ::ranges::stub_detail::val_tuple<Rs...>> : tuple_element<K, tuple<remove_cvref_t<Rs>...>> {};
template <typename... Rs, template <typename> class TQ, template <typename> class UQ>
struct basic_common_reference<::ranges::stub_detail::ref_tuple<Rs...>, ::ranges::stub_detail::val_tuple<Rs...>, TQ, UQ> {
  using type = ::ranges::stub_detail::val_tuple<Rs...>;
};
template <typename... Rs, template <typename> class TQ, template <typename> class UQ>
struct basic_common_reference<::ranges::stub_detail::val_tuple<Rs...>, ::ranges::stub_detail::ref_tuple<Rs...>, TQ, UQ> {
  using type = ::ranges::stub_detail::val_tuple<Rs...>;
};
} // namespace std

template <typename... Vs>
inline constexpr bool std::ranges::enable_borrowed_range<ranges::stub_detail::zip_view<Vs...>> =
    (std::ranges::enable_borrowed_range<Vs> && ...);
