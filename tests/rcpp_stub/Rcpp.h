// Declaration-only stand-in for <Rcpp.h>: TEST INFRASTRUCTURE, not a substitute for Rcpp.
//
// R and Rcpp are not in the image this repository is developed in, so glue/obhip_glue.cpp (the
// replacement of the reference's src/interfaceR.cpp:661-793) cannot be built here.  This header
// declares just the names the glue uses, with the shapes Rcpp gives them, so that
//   g++ -std=c++11 -fsyntax-only -Itests/rcpp_stub -Iinclude glue/obhip_glue.cpp
// (tests/test_host_logic.py) catches what a parser and type checker catch: misspelt ABI calls,
// wrong argument counts / types against include/obhip.h, malformed module declarations.  It pins
// nothing about parity or about Rcpp's real semantics, and nothing links against it.
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

typedef std::ptrdiff_t R_xlen_t;

namespace R {
double unif_rand();
}

namespace Rcpp {

void stop(const char *msg);
void stop(const std::string &msg);

class Dimension {
public:
  Dimension(std::size_t, std::size_t);
  Dimension(std::size_t, std::size_t, std::size_t);
};

class CharacterVector;

class NamesProxy {
public:
  NamesProxy &operator=(const CharacterVector &);
};

class NumericVector {
public:
  NumericVector();
  // (Rcpp's Vector takes any arithmetic size type through one constructor template)
  template <typename T, typename = typename std::enable_if<std::is_integral<T>::value>::type>
  explicit NumericVector(T n);
  template <typename T, typename = typename std::enable_if<std::is_integral<T>::value>::type>
  NumericVector(T n, double fill);
  explicit NumericVector(const Dimension &);
  R_xlen_t size() const;
  double *begin();
  const double *begin() const;
  double *end();
  const double *end() const;
  double &operator[](R_xlen_t);
  const double &operator[](R_xlen_t) const;
  NamesProxy names();
};

class NumericMatrix {
public:
  NumericMatrix();
  template <typename T, typename U, typename = typename std::enable_if<std::is_integral<T>::value && std::is_integral<U>::value>::type>
  NumericMatrix(T rows, U cols);
  R_xlen_t size() const;
  int nrow() const;
  int ncol() const;
  double *begin();
  const double *begin() const;
  double &operator[](R_xlen_t);
  const double &operator[](R_xlen_t) const;
};

class IntegerVector {
public:
  IntegerVector();
  explicit IntegerVector(R_xlen_t n);
  R_xlen_t size() const;
  int *begin();
  int &operator[](R_xlen_t);
};

class StringProxy {
public:
  StringProxy &operator=(const std::string &);
  StringProxy &operator=(const char *);
  operator std::string() const;
};

class CharacterVector {
public:
  CharacterVector();
  template <typename T, typename = typename std::enable_if<std::is_integral<T>::value>::type>
  explicit CharacterVector(T n);
  R_xlen_t size() const;
  StringProxy operator[](R_xlen_t);
  const StringProxy operator[](R_xlen_t) const;
};
typedef CharacterVector StringVector;

class ListProxy {
public:
  operator NumericVector() const;
};

class List {
public:
  List();
  R_xlen_t size() const;
  ListProxy operator[](R_xlen_t) const;
};

template <typename T>
T clone(const T &);
template <typename T, typename U>
T as(const U &);

// RCPP_MODULE surface: every call returns the builder again
template <typename C>
class class_ {
public:
  explicit class_(const char *name);
  class_ &constructor();
  template <typename... A>
  class_ &constructor();
  template <typename F>
  class_ &method(const char *name, F f);
  template <typename F>
  class_ &field(const char *name, F f);
  template <typename F>
  class_ &field_readonly(const char *name, F f);
  template <typename G>
  class_ &property(const char *name, G getter);
  template <typename G, typename S>
  class_ &property(const char *name, G getter, S setter);
  template <typename B>
  class_ &derives(const char *name);
};

template <typename F>
void function(const char *name, F f, const char *doc = 0);

}  // namespace Rcpp

#define RCPP_EXPOSED_CLASS(cls)
#define RCPP_MODULE(name) void rcpp_stub_module_##name##_init()
