// COMPILE-CHECK INFRASTRUCTURE ONLY (tests/test_dropin_headers.py).
// Just enough of the range-v3 surface for g++ 11 to parse the reference headers and to instantiate the
// gfx950 overloads: the reference selects <range/v3/all.hpp> when the standard library has no
// std::views::zip (detail/ranges.hpp).  Everything is std::ranges except `view_` and a small random-access
// views::zip whose value_type is its reference type (a tuple of references) -- enough for the reference's
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
template <typename... Its>
class zip_iterator {
public:
  using iterator_concept = std::random_access_iterator_tag;
  using iterator_category = std::random_access_iterator_tag;
  using difference_type = std::ptrdiff_t;
  using reference = std::tuple<std::iter_reference_t<Its>...>;
  using value_type = reference;  // proxy: there is no tuple common_reference before C++23
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

template <typename... Vs>
inline constexpr bool std::ranges::enable_borrowed_range<ranges::stub_detail::zip_view<Vs...>> =
    (std::ranges::enable_borrowed_range<Vs> && ...);
